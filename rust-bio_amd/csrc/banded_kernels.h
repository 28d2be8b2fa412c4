// Kernels K3 (banded_fill) and K4 (banded_traceback): batched `banded::Aligner::compute_alignment`
// (/root/reference/src/alignment/pairwise/banded.rs:406-869) on gfx950.
//
// The band arrives from the host (band_host.cpp) as per-ROW column ranges: for a band whose
// per-column row ranges never move up (what Band::create produces; checked on the host), the
// columns in which row i is inside the band form one interval [cf(i), cl(i)].  Everything the
// reference reads outside the band is MIN_SCORE / TB_START (its rolling arrays are reset just
// ahead of the band, banded.rs:556-561, 676-680, 689-691), except for the `S[curr][m]` slot and
// the borders — those are modelled explicitly below.
//
// K3 — one wavefront per pair, lane-owned rows in strips of 64*R rows, columns skewed by one step
//   per lane exactly like K1 (sw_kernels.h), but a strip only walks the columns its rows touch.
//   Cells outside the band are skipped and forward MIN_SCORE to their neighbours.  The traceback
//   is stored band-compact, ONE BYTE PER BAND CELL in row-major order (row i at
//   row_off[i] + j - cf(i)), i.e. 1.3 MB instead of the reference's 200 MB for a 10 kb pair.
//   Per column the running x-suffix-clip fold (`S[curr][m]`, Lx[j]) is published so that the
//   y-suffix-clip of row m (banded.rs:665-670) can be folded over all columns afterwards.
// K4 — one lane per pair: border passes (banded.rs:725-765), traceback (767-831) and the
//   "ended outside the band" fix-up (833-855).
#ifndef BG_BANDED_KERNELS_H
#define BG_BANDED_KERNELS_H
#include "sw_kernels.h"

namespace bgband_dev {

using namespace bgsw;

enum : uint32_t { BP_OK = 0, BP_TOO_MANY_CELLS = 1, BP_UNSUPPORTED = 2 };
// Traceback bytes of a pair (one per band cell, rows 1..m; row 0 is a closed form): a row's cells cf..cl in groups of
// 16 — K3v2 hands them over in complete 16-byte groups (banded_fill2.hip) — and the groups of eight consecutive rows
// interleaved: group g of row i sits at row_off[i] + g * 128, row_off[i] = (base of the rows (i-1)/8*8+1 ..) +
// ((i-1) % 8) * 16.  A traceback path that runs down a diagonal stays inside one 128-byte line for eight rows (with the
// rows' bytes back to back it touched a new line on every row: K4 fetched 1.4 MB per 10 kb pair, 23 GB per launch).
constexpr uint32_t kTbRowAlign = 16;   // cells per group
constexpr uint32_t kTbLineRows = 8;    // rows whose groups share a line
constexpr uint32_t kTbGroupStride = kTbRowAlign * kTbLineRows;
// byte offset of cell c (= j - cf) of a row behind its row_off
__host__ __device__ inline uint32_t tb_cell_off(uint32_t c) { return (c >> 4) * kTbGroupStride + (c & 15u); }

// One pair of a banded batch (device copy, built on the host)
struct BandPair {
    uint64_t rowc_off;  // index of row 0 in the rowc / row_off arrays
    uint64_t tb_off;    // byte offset of this pair's traceback bytes
    uint64_t aux_off;   // int32 offset of this pair's aux record
    uint32_t start_0, end_0;  // band of column 0 (banded.rs:443-444)
    uint32_t start_n, end_n;  // band of the last column (banded.rs:689, 705)
    uint32_t flags;           // BP_*
    uint32_t _pad;
};

// aux record (int32 words): [0] score  [1] S nibble of (m,n)  [2] Lx[n]  [3] Ly[m]  [5] K3p: "redo this pair"  [4,6,7] spare
//   Ly[m+1]  Lx[n+1]  V[n+1] (S[curr][m] after each column)  Sn[m+1]  bits[m+1 bytes]  bnd int4[n+1]
struct BandAux {
    uint32_t m, n;
    __host__ __device__ BandAux(uint32_t m_, uint32_t n_) : m(m_), n(n_) {}
    __host__ __device__ uint64_t off_Ly() const { return 8; }
    __host__ __device__ uint64_t off_Lx() const { return 8 + (uint64_t)(m + 1); }
    __host__ __device__ uint64_t off_V() const { return off_Lx() + (n + 1); }
    __host__ __device__ uint64_t off_Sn() const { return off_V() + (n + 1); }
    __host__ __device__ uint64_t off_bits() const { return off_Sn() + (m + 1); }
    __host__ __device__ uint64_t off_bnd() const { return (off_bits() + (m + 4) / 4 + 3) & ~3ull; }
    // K3v2 only: what the rows inside the band of column n looked like when the fill left them
    //   int2 {S(i,n), I(i,n)} [m+1], then bytes (cell | I case << 5) [m+1]
    __host__ __device__ uint64_t off_endv() const { return off_bnd() + 4ull * (n + 1); }
    __host__ __device__ uint64_t off_endc() const { return off_endv() + 2ull * (m + 1); }
    __host__ __device__ uint64_t words() const { return off_endc() + (m + 4) / 4; }
};

constexpr uint32_t kTbFlip = 24;  // bits 3 (I) and 4 (D) of a traceback byte

struct BandArgs {
    const uint8_t* x;
    const uint64_t* x_off;
    const uint8_t* y;
    const uint64_t* y_off;
    uint64_t pair0;
    uint32_t n_pairs;
    SwScoring sc;
    const int32_t* table;
    const uint8_t* code_map;
    int32_t alpha;
    const BandPair* pairs;   // [n_pairs]
    const int2* rowc;        // per row {cf, cl}; cl < cf: the row is never inside the band
    const uint32_t* row_off; // per row: offset of its traceback bytes inside the pair's block
    uint8_t* tb;
    int32_t* aux;
    bg_alignment_t* out;
    uint8_t* ops;
    uint64_t ops_stride;
    int32_t mode, filter_clips;
    uint32_t tb_flip;   // XORed onto every traceback byte K4 reads (kTbFlip after K3v2, 0 after K3)
    uint32_t* started;  // K3v2: every block counts itself in when it starts (nullptr: nobody is waiting for that)
    int32_t phase;  // K3v2: 0 all strips of every pair; 1 / 2: the strips before / behind the interior run (band_split)
    int32_t ring32; // K3i with 32-byte rings (33 KB of LDS per block instead of 65)
    int32_t split;  // the scoring admits interior runs (host decision, banded_api.hip): band_split may say yes
    int32_t packed;     // the interior runs go to K3p (banded_fill2p.hip) first; K3i redoes what it flags
    int32_t redo;       // K3v2 phase 1 / K3i: only the pairs K3p flagged (aux[5] != 0)
    int32_t pk_thresh;  // K3p: the threshold a band cell's key has to exceed (0: derived from the scoring; tests raise it)
    uint32_t* redo_count;  // K3p counts the pairs it flags here (nullptr: nobody asks)
    int32_t p_block512;    // K3p in blocks of eight wavefronts at 168 VGPRs (banded_fill2p.hip) instead of four at 187
};

// Interior run of a pair: the strips [s_a, s_b) of RS rows each that banded_fill2i_kernel takes with its reduced cell.
// Scoring-level condition (BandArgs::split, set by the host): scaled keys apply, xclip_prefix = xclip_suffix = MIN_SCORE and
// yclip_prefix is a real score (semiglobal-like) — then the x-suffix-clip fold S[curr][m] (banded.rs:648-653) only ever
// holds MIN_SCORE + something, every band cell holds a real score (the y-prefix-clip candidate, banded.rs:633-642), and
// the fold wins neither a cell nor Sn[m] once row m has had a band cell of its own — which is the pair-level condition
// below.  Strip-level: no row with column 0 in its band (closed forms, banded.rs:440-499) in or right above the strip,
// no row inside the band of column n (the j == n candidate and records, banded.rs:590-596, 683-723), not row m.
__device__ __forceinline__ bool band_split(const SwScoring& sc, const BandPair& bp, uint32_t m, const int2* rowc, uint32_t RS,
                                           uint32_t& s_a, uint32_t& s_b) {
    (void)sc;
    if (m < 2 || bp.start_n < 1) return false;
    const int2 rcm = rowc[m];
    if (!(rcm.y >= rcm.x && rcm.y >= 1)) return false;  // row m has a band cell in a column >= 1
    const uint32_t rows0 = bp.end_0 > bp.start_0 ? bp.end_0 : 0;  // rows below rows0 - 1 do not touch column 0
    s_a = max(1u, (rows0 + RS - 1) / RS);  // the row above strip s_a (s_a * RS) is >= end_0
    s_b = min((bp.start_n - 1) / RS, (m - 1) / RS);  // the strip of the first row inside column n's band / of row m
    return s_a < s_b;
}

constexpr uint32_t kSplitStripRows = 32;  // rows per strip of the kernels that split pairs (BF2_LP * BF2_R)
// Traceback byte of a row inside an interior run (banded_fill2i.hip: bit 0 = I opened, bits 1-3 = S move, bit 4 = D opened)
// in the layout of every other row (bits 0-2 = S move, bit 3 = I opened, bit 4 = D opened)
__device__ __forceinline__ uint32_t tb_cell_norm(uint32_t b) { return ((b >> 1) & 7u) | ((b & 1u) << 3) | (b & 16u); }

typedef void (*band_fill_fn)(const BandArgs);
band_fill_fn get_band_fill(int sm);
// K3v2 (banded_fill2.hip): LP lanes per pair, R rows per lane, MatchParams scoring; the last-column
// epilogue runs in its own kernel.  Returns false if the geometry is not instantiated.
bool launch_band_fill2(const BandArgs& a, bool narrow, hipStream_t st, hipEvent_t after_fill = nullptr, hipStream_t epi = nullptr,
                       hipStream_t pre = nullptr, hipEvent_t pre_done = nullptr);
void launch_fill2i(const BandArgs& a, dim3 grid, hipStream_t st);  // banded_fill2i.hip: the interior runs
void launch_fill2p(const BandArgs& a, hipStream_t st);             // banded_fill2p.hip: the same, two pairs per lane group
uint32_t band_fill2_blocks(uint32_t n_pairs);  // thread blocks launch_band_fill2 starts for n_pairs
// holds `st` until *counter >= target (or ~20 ms have passed): "the fill kernel's blocks are all resident"
void launch_band_wait_started(const uint32_t* counter, uint32_t target, hipStream_t st);
void launch_band_traceback(const BandArgs& a, hipStream_t st);

}  // namespace bgband_dev
#endif
