// Shared internals of libbiogpu (gfx950 only).
#ifndef BG_COMMON_H
#define BG_COMMON_H
#include <functional>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "biogpu.h"

extern thread_local std::string bg_tls_error;

#define BG_HIP(call)                                                                     \
    do {                                                                                 \
        hipError_t e__ = (call);                                                         \
        if (e__ != hipSuccess) {                                                         \
            bg_tls_error = std::string(#call) + ": " + hipGetErrorString(e__);           \
            return e__ == hipErrorOutOfMemory ? BG_ERR_OOM : BG_ERR_HIP;                 \
        }                                                                                \
    } while (0)

struct bg_band_scratch;  // banded_api.hip
void bg_band_scratch_free(bg_band_scratch*);
struct bg_seed_scratch;  // seed_extend.hip
void bg_seed_scratch_free(bg_seed_scratch*);
struct bg_host_pipe;  // sw_api.hip: staging sets of the pipelined host-buffer path
void bg_host_pipe_free(bg_host_pipe*);
struct bg_fm_pipe;  // fm_index.hip: staging sets of bg_fm_backward_search_batch
void bg_fm_pipe_free(bg_fm_pipe*);

struct bg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;  // internal stream of the host-buffer API
    // reusable device scratch of the SW pipeline (grown on demand)
    void* tb = nullptr;
    size_t tb_bytes = 0;
    void* aux = nullptr;
    size_t aux_bytes = 0;
    void* bnd = nullptr;
    size_t bnd_bytes = 0;
    void* io[6] = {};       // host-buffer API: device copies of x, y, x_off, y_off, out, ops
    size_t io_cap[6] = {};
    void* h_ops = nullptr;  // ... and the pinned landing zone of the operations
    size_t h_ops_cap = 0;
    void* unpk[2] = {};     // bg_align_batch_packed_dev: byte copies of the 2-bit streams for the kernels that take bytes
    size_t unpk_cap[2] = {};
    void* table = nullptr;  // compacted scoring table + code map
    size_t table_bytes = 0;
    // the full aligner's compacted matrix stays on the device between calls, in a buffer of its own (the banded path
    // writes `table`); a call recognises its matrix by hash AND by comparing the 256 KB kept on the host
    void* sw_table = nullptr;
    size_t sw_table_bytes = 0;
    uint64_t table_hash = 0;  // hash of the score matrix whose compacted form `sw_table` holds (0: none)
    int table_alpha = 0;      // ... its number of classes
    int32_t* table_matrix = nullptr;  // ... and the matrix itself (65 536 entries, host)
    bg_band_scratch* band = nullptr;  // persistent scratch of the banded pipeline
    bg_host_pipe* pipe = nullptr;     // persistent staging of bg_align_batch's pipelined path
    bg_seed_scratch* seed = nullptr;  // persistent scratch of the seed-and-extend pipeline
    bg_fm_pipe* fm_pipe = nullptr;    // persistent pinned / device staging of bg_fm_backward_search_batch
    uint64_t fm_wide_from = 0xFFFFFFFFull;  // texts of this many symbols or more get the 64-bit FM layout (tests: lower it)
    uint32_t fm_wide_sb_shift = 17;         // ... with superblocks of 2^this blocks (tests: a few blocks, so that bases matter)
    bool fm_host_bytes = false;       // tests, A/B: bg_fm_backward_search_batch stages the pattern bytes (no 2-bit packing on the host)
    int64_t host_chunk_pairs = 0;     // pairs per pipeline stage of bg_align_batch (0 = default)
    int64_t chunk_pairs = 0;  // 0 = default
    int64_t seed_chunk_reads = 0;  // reads per pass of bg_seed_extend_batch_dev (0 = equal passes of at most 2^21)
    bool force_wide = false;  // tests: disable the NARROW (28-bit key) kernels
    bool no_pk16 = false;     // tests: disable K1p (two pairs per lane in packed int16 halves)
    bool no_local_fast = false;  // tests: Aligner::local on the general K1p (no LF flavour)
    bool no_couples = false;  // tests: K1p without the (m, n) slot order on ragged batches
    int band_chain_global = -1;  // chain_kernel tree placement: -1 by batch size, 0 LDS, 1 global scratch
    bool band_host_sync = false;  // A/B: issue() waits on the host for K4 of two sub-batches ago before launching the chaining (rounds 2-4)
    bool fq_no_fused = false;  // tests, A/B: bg_fastq_parse_dev through F1 .. F6 only (no one-pass kernel in front)
    int64_t sa_chunk_symbols = 0;  // tests: suffixes per pass of round 0 of the device suffix-array builder (0: by free memory)
    bool band_chain_rows = true;  // global-tree chaining: four pairs per wavefront (chain_rows_kernel); false: one (A/B, tests)
    bool band_join_global = false;  // tests: k-mer join with its table in global memory even where the LDS flavour applies
    int band_fill_v1 = 0;  // 1: K3 (one pair per wavefront) even where K3v2 applies; -1: K3v2 even for small sub-batches; 0: by size
    bool band_interior_off = false;  // tests: K3v2 takes its general step in every strip (no reduced step in interior strips)
    bool band_packed_off = false;    // tests, A/B: interior runs on K3i (int32) only, no packed-int16 K3p in front of it
    int64_t band_packed_thresh = 0;  // tests: K3p's redo threshold in key units (0: derived from the scoring; 0xffff: every pair is redone)
    bool band_pre_serial = false;    // A/B: a fill's preparation (pair table, waits, K3v2 phase 1) on the fill stream, behind the previous fill
    bool band_join_serial = false;   // A/B: the k-mer join of a sub-batch on the builder's stream, behind the previous sub-batch's row ranges (one chain)
    bool band_tail_last = false;   // A/B: the remainder sub-batch of a large banded call runs last (round 3) instead of first
    bool band_window = false;      // A/B: K3i on 64-byte rings, and the fill waits for the next sub-batch's join + preparation (round 3's window between two fills)
    bool band_raster_late = false; // A/B: the raster of sub-batch c + 1 waits for fill c to leave the device (round 3)
    bool band_join_late = false;  // A/B: the k-mer join of sub-batch c + 2 waits for the chaining of c + 1 to leave the device (round 5: measured slower)
    bool band_p_block512 = false;  // A/B: K3p in blocks of eight wavefronts compiled for 168 VGPRs instead of four at 187 (round 5: measured 3 % slower)
    int64_t band_budget_gb = 0;  // traceback + aux bytes per scratch set of the banded pipeline, in GB (0: 40)
    bool band_on_host = false;  // build bands with the host builder (band_host.cpp) instead of band_device.hip
    // the scratch above is one set per ctx: a *_dev call arriving on another stream than the previous one first
    // waits (on the device) for that call's last kernel — see bg_scratch_guard
    hipEvent_t scratch_done = nullptr;
    hipStream_t scratch_stream = nullptr;
    bool scratch_used = false;
    // timing
    bool timing = false;
    hipEvent_t ev[2] = {nullptr, nullptr};
    bg_timing_t last = {};
};

// grow-only device scratch
int bg_reserve(void** p, size_t* cur, size_t need);
// exclusive scan of n uint32 counts into n + 1 uint64 offsets (fastq_ingest.hip); d_sums: 2 * (n / 2048 + 1) uint64 of scratch
int bg_scan_u32(const uint32_t* d_len, uint64_t n, uint64_t* d_off, uint64_t* d_sums, hipStream_t st);
// bg_align_batch_dev with what the caller knows about the lengths: 1 all pairs share (m, n), 0 they differ, -1 unknown
int bg_align_batch_dev_hint(bg_ctx* ctx, const bg_scoring_t* sc, int mode, uint64_t n_pairs, const uint8_t* d_x,
                            const uint64_t* d_x_off, const uint8_t* d_y, const uint64_t* d_y_off, uint32_t max_xlen,
                            uint32_t max_ylen, bg_alignment_t* d_out, uint8_t* d_ops, uint64_t ops_stride, void* stream,
                            int len_hint);

// Operations of a batch, compacted on the device (the host-buffer paths).  The fill / traceback kernels leave a pair's
// operations right-aligned in its own slot of the strided buffer (rec[p].ops_off says where); the caller's buffer wants
// them back to back.  The operation counts of the n records are scanned, 16 or 64 lanes per pair copy its bytes to
// d_compact + (global_offsets ? *d_cell : 0) + offset, the records get their FINAL ops_off = *d_cell + offset, then
// *d_cell += the batch's byte count (a running total over the stages / sub-batches of a call, zeroed by the caller) and
// *d_batch_total (if given) receives that count.  d_scratch: bg_compact_ops_scratch(n) bytes; long_ops: operation lists
// of thousands of bytes (a wavefront per pair copies) rather than a few hundred (16 lanes).  Asynchronous on st.
size_t bg_compact_ops_scratch(uint64_t n);
int bg_range_to_host(const uint8_t* d_src, uint8_t* h_dst, const uint64_t* d_slot /* {bytes, end offset} */, hipStream_t st);
int bg_compact_ops_dev(bg_alignment_t* d_rec, uint64_t n, const uint8_t* d_ops, uint8_t* d_compact, bool global_offsets, uint64_t* d_cell,
                       uint64_t* d_batch_total, void* d_scratch, bool long_ops, hipStream_t st);

// Serialises the users of a ctx's scratch across streams: constructed at the top of every *_dev entry point that
// touches ctx->tb / aux / bnd / table, it makes `st` wait for the event the previous user recorded (if that was
// another stream) and records its own when the entry point returns — two calls in flight on two streams with one
// ctx then run one after the other instead of overwriting each other's traceback words.
struct bg_scratch_guard {
    bg_ctx* ctx;
    hipStream_t st;
    bg_scratch_guard(bg_ctx* c, hipStream_t s) : ctx(c), st(s) {
        if (ctx->scratch_used && ctx->scratch_stream != st) hipStreamWaitEvent(st, ctx->scratch_done, 0);
    }
    ~bg_scratch_guard() {
        if (hipEventRecord(ctx->scratch_done, st) == hipSuccess) {
            ctx->scratch_stream = st;
            ctx->scratch_used = true;
        }
    }
};
// hipMemcpyAsync in pieces of at most BG_COPY_PIECE bytes (env, default 8 MB): transfers between pinned and device memory
// beyond ~20 MB were measured to stall behind running compute kernels on this stack (the drainer of bg_align_batch
// waited 1.7 ms for a 22 MB download against 0.2 ms for a 19.7 MB one), smaller ones ride the DMA engines
inline hipError_t bg_copy_pieces(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t st) {
    static const size_t piece = [] {
        const char* e = getenv("BG_COPY_PIECE");
        const long long v = e ? atoll(e) : 0;
        return v > 0 ? (size_t)v : (size_t)(8u << 20);
    }();
    for (size_t o = 0; o < bytes; o += piece) {
        const hipError_t rc = hipMemcpyAsync((uint8_t*)dst + o, (const uint8_t*)src + o, bytes - o < piece ? bytes - o : piece, kind, st);
        if (rc != hipSuccess) return rc;
    }
    return hipSuccess;
}
// host threads this process may really use (affinity mask and cgroup CPU quota)
unsigned bg_host_threads();
namespace bgpack {
// host_pack2.cpp: bytes -> 2-bit stream (16 symbols per dword); false if a byte is none of the four codes
bool pack2_host(const uint8_t* src, uint64_t n, const uint8_t codes[4], uint32_t* dst);
}
// fn(0) .. fn(nt - 1) on the process-wide worker threads (created once, bg_host_threads() - 1 of them) and the caller;
// returns when all have run.  Calls from several threads share the workers.  (Spawning 16 threads per parallel loop cost
// 0.4 ms a loop: a fifth of what bg_align_batch spends per stage.)
void bg_pool_run(unsigned nt, const std::function<void(unsigned)>& fn);

// ---- device helpers -------------------------------------------------------------------
// DPP cross-lane moves (gfx9 encodings): lane i <- lane i-1 / lane i+1 across the 64-lane wave.
// Lane 0 (resp. 63) keeps `old`.  Must be executed with all lanes active.
__device__ __forceinline__ int wave_shr1(int v, int old = 0) {
    return __builtin_amdgcn_update_dpp(old, v, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
}
__device__ __forceinline__ int wave_shl1(int v, int old = 0) {
    return __builtin_amdgcn_update_dpp(old, v, 0x130 /*wave_shl:1*/, 0xf, 0xf, false);
}
// ... with 0 in lane 0 (resp. 63): no register has to be initialised for the lane without a source
__device__ __forceinline__ int wave_shr1z(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ int wave_shl1z(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true); }
// sum over the 4 lanes of a quad, result in all 4
__device__ __forceinline__ unsigned quad_sum(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1 /*quad_perm:[1,0,3,2]*/, 0xf, 0xf, true);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E /*quad_perm:[2,3,0,1]*/, 0xf, 0xf, true);
    return v;
}

#endif
