// Device-side band construction (band_device.hip): arguments and per-pair state.
#ifndef BG_BAND_DEVICE_H
#define BG_BAND_DEVICE_H
#include "banded_kernels.h"

namespace bgband_dev {

enum : uint32_t { BP_HOST_FALLBACK = 3 };  // continues BP_OK / BP_TOO_MANY_CELLS / BP_UNSUPPORTED

constexpr uint32_t kMaxChainMatches = 4095;    // matches per pair the LDS Fenwick tree of chain_kernel holds
constexpr uint32_t kSmallChainMatches = 2047;  // ... in its small LDS size class (two wavefronts per CU more)
constexpr uint32_t kChainGlobalMinPairs = 1024;  // from this many pairs per launch the chain tree goes to global scratch
constexpr uint32_t kMaxMatchesPerKmer = 32;   // matches of one x k-mer sorted in place by kmer_match_kernel

// what the builder leaves per pair (read back by the host: 48 bytes per pair)
struct BandDevPair {
    uint64_t cells;     // Band::num_cells
    uint64_t tb_bytes;  // traceback bytes K3 needs for this pair
    uint32_t n_matches, n_path;
    uint32_t start_0, end_0, start_n, end_n;
    uint32_t flags;     // BP_*
    uint32_t _pad;
};

struct BandDevArgs {
    const uint8_t* x;
    const uint64_t* x_off;
    const uint8_t* y;
    const uint64_t* y_off;
    uint64_t pair0;
    uint32_t n_pairs;
    uint32_t k, w;
    int32_t gap_open, gap_extend, xclip_prefix, xclip_suffix, yclip_prefix, yclip_suffix;
    uint32_t match_score;  // sparse-DP reward per base (banded.rs:105,1315-1318)
    uint32_t max_m, max_n;  // longest x / y of the sub-batch (scratch strides)
    uint32_t table_size, table_bits, cap_matches;
    uint32_t chain_min, chain_cap;
    int32_t join_global;   // 1: kmer_match_kernel (table in global memory) even where kmer_match_lds_kernel applies
    int32_t chain_rows;    // 1: the global-tree event loop runs four pairs per wavefront (chain_rows_kernel); 0: one (chain_kernel<false, 2>)
    int32_t chain_global;  // 1 / 0: force the global / LDS tree variant of chain_kernel, -1: by batch size  // chain_kernel: the range of match counts this launch handles
    // scratch, one slice per pair
    uint32_t* head;   // [table_size]
    uint32_t* next;   // [max_n]
    uint64_t* hy;     // [max_n]
    uint32_t *mx, *my, *path, *qpos, *upos;  // [cap_matches]
    int32_t* cont;                           // [cap_matches]
    void* g_tree;       // [cap_matches + 1] x 16 B: chain_kernel<false>
    uint32_t* g_score;  // [cap_matches]
    int16_t* g_back;    // [cap_matches]
    uint32_t *col_start, *col_end;           // [max_n + 1]
    BandDevPair* state;                      // [n_pairs]
    // outputs in the layout K3 / K4 read
    const uint64_t* row0;  // [n_pairs]: first row of the pair inside rowc / row_off
    int2* rowc;
    uint32_t* row_off;
};

int launch_band_match(const BandDevArgs& a, hipStream_t st);
int launch_band_chain(const BandDevArgs& a, hipStream_t st, int part = 0);   // B2: sdpkpp (part: see band_device.hip)
int launch_band_raster(const BandDevArgs& a, hipStream_t st);  // B3 + B4: Band::create_from_match_path, per-row ranges

}  // namespace bgband_dev
#endif
