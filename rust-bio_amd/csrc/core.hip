// Context management and shared utilities of libbiogpu.
#include <sched.h>

#include <algorithm>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#include <cstdlib>

#include "bg_common.h"

thread_local std::string bg_tls_error;

int bg_reserve(void** p, size_t* cur, size_t need) {
    if (need <= *cur) return BG_OK;
    if (*p) {
        hipFree(*p);
        *p = nullptr;
        *cur = 0;
    }
    need = (need + 255) & ~(size_t)255;
    BG_HIP(hipMalloc(p, need));
    *cur = need;
    return BG_OK;
}

unsigned bg_host_threads() {
    unsigned nt = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) nt = (unsigned)CPU_COUNT(&set);
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        long long quota = 0, period = 0;
        if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0)
            nt = std::min<unsigned>(nt, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
        fclose(f);
    }
    return nt;
}

namespace {
struct PoolJob {
    const std::function<void(unsigned)>* fn;
    unsigned next = 0, total = 0, done = 0;
};
struct Pool {
    std::mutex mu;
    std::condition_variable work, finished;
    std::deque<PoolJob*> jobs;  // jobs that still have indices to hand out
    bool started = false;
    void start() {
        const unsigned nw = std::max(1u, bg_host_threads()) - 1;
        for (unsigned w = 0; w < nw; w++) std::thread([this] { loop(); }).detach();
        started = true;
    }
    // runs one index of the front job; lk is held on entry and on return
    void run_one(std::unique_lock<std::mutex>& lk, PoolJob* job) {
        const unsigned idx = job->next++;
        if (job->next == job->total) jobs.erase(std::find(jobs.begin(), jobs.end(), job));
        lk.unlock();
        (*job->fn)(idx);
        lk.lock();
        if (++job->done == job->total) finished.notify_all();
    }
    void loop() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            work.wait(lk, [this] { return !jobs.empty(); });
            run_one(lk, jobs.front());
        }
    }
};
Pool& pool() {
    static Pool* p = new Pool();  // never destroyed: its detached workers outlive static destruction
    return *p;
}
}  // namespace

void bg_pool_run(unsigned nt, const std::function<void(unsigned)>& fn) {
    if (nt <= 1) {
        if (nt) fn(0);
        return;
    }
    Pool& P = pool();
    PoolJob job;
    job.fn = &fn;
    job.total = nt;
    std::unique_lock<std::mutex> lk(P.mu);
    if (!P.started) P.start();
    P.jobs.push_back(&job);
    P.work.notify_all();
    while (job.next < job.total) P.run_one(lk, &job);  // the caller works on its own job
    P.finished.wait(lk, [&] { return job.done == job.total; });
}

extern "C" int bg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int bg_init(int device, bg_ctx** out) {
    if (!out) return BG_ERR_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) {
        bg_tls_error = "no usable HIP device";
        return BG_ERR_NO_DEVICE;
    }
    BG_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    BG_HIP(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        // the code objects in this library are gfx950-only
        bg_tls_error = std::string("device is ") + prop.gcnArchName + ", library is built for gfx950";
        return BG_ERR_NO_DEVICE;
    }
    bg_ctx* ctx = new bg_ctx;
    ctx->device = device;
    // BG_BAND_FILL_V1: initial value of the option of that name (the test suite pins the eight-pairs-per-wavefront fill so
    // that its small batches keep exercising it)
    if (const char* e = getenv("BG_BAND_FILL_V1")) ctx->band_fill_v1 = atoi(e) > 0 ? 1 : atoi(e) < 0 ? -1 : 0;
    // The ctx's own stream carries the long kernels of the pipelines (the banded fills above all); helper streams (band
    // construction, traceback, copies) are created at the default priority.  BG_STREAM_PRIO=0: default priority here too (A/B)
    {
        int lo = 0, hi = 0;
        const char* e = getenv("BG_STREAM_PRIO");
        if ((!e || atoi(e) != 0) && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo)
            BG_HIP(hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, hi));
        else
            BG_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    }
    BG_HIP(hipEventCreate(&ctx->ev[0]));
    BG_HIP(hipEventCreate(&ctx->ev[1]));
    BG_HIP(hipEventCreateWithFlags(&ctx->scratch_done, hipEventDisableTiming));
    *out = ctx;
    return BG_OK;
}

extern "C" int bg_free(bg_ctx* ctx) {
    if (!ctx) return BG_OK;
    hipSetDevice(ctx->device);
    hipFree(ctx->tb);
    hipFree(ctx->aux);
    hipFree(ctx->bnd);
    hipFree(ctx->table);
    hipFree(ctx->sw_table);
    free(ctx->table_matrix);
    hipFree(ctx->unpk[0]);
    hipFree(ctx->unpk[1]);
    for (void* p : ctx->io) hipFree(p);
    if (ctx->h_ops) hipHostFree(ctx->h_ops);
    bg_band_scratch_free(ctx->band);
    bg_host_pipe_free(ctx->pipe);
    bg_seed_scratch_free(ctx->seed);
    bg_fm_pipe_free(ctx->fm_pipe);
    if (ctx->ev[0]) hipEventDestroy(ctx->ev[0]);
    if (ctx->ev[1]) hipEventDestroy(ctx->ev[1]);
    if (ctx->scratch_done) hipEventDestroy(ctx->scratch_done);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return BG_OK;
}

extern "C" const char* bg_strerror(int s) {
    switch (s) {
        case BG_OK: return "ok";
        case BG_ERR_INVALID_ARG: return "invalid argument";
        case BG_ERR_NO_DEVICE: return "no usable gfx950 device";
        case BG_ERR_HIP: return "HIP runtime error";
        case BG_ERR_OOM: return "out of device memory";
        case BG_ERR_SENTINEL:
            return "Expecting extra sentinel symbol being lexicographically smallest at the end of the text.";
        case BG_ERR_POSITIVE_PENALTY: return "gap/clip penalty can't be positive";
        case BG_ERR_OUT_OF_ALPHABET: return "symbol outside the alphabet (index out of bounds in the reference)";
        case BG_ERR_TOO_LARGE: return "input too large for the engine";
        case BG_ERR_OPS_CAP: return "operations buffer too small";
        case BG_ERR_TRACEBACK: return "traceback did not terminate";
        case BG_ERR_UNSUPPORTED: return "not supported by the device layout";
        case BG_ERR_IO: return "index file cannot be opened, is truncated or fails its checksum";
        default: return "unknown status";
    }
}

extern "C" const char* bg_last_error(void) { return bg_tls_error.c_str(); }

extern "C" int bg_set_option(bg_ctx* ctx, const char* key, int64_t value) {
    if (!ctx || !key) return BG_ERR_INVALID_ARG;
    if (!strcmp(key, "chunk_pairs")) {
        ctx->chunk_pairs = value;
        return BG_OK;
    }
    if (!strcmp(key, "seed_chunk_reads")) {
        ctx->seed_chunk_reads = value;
        return BG_OK;
    }
    if (!strcmp(key, "band_join_global")) {
        ctx->band_join_global = value != 0;
        return BG_OK;
    }
    if (!strcmp(key, "fm_wide_from")) {  // tests: indexes of this many symbols or more take the 64-bit layout (0: the default, 2^32 - 1)
        ctx->fm_wide_from = value > 0 ? (uint64_t)value : 0xFFFFFFFFull;
        return BG_OK;
    }
    if (!strcmp(key, "fm_wide_sb_shift")) {
        if (value < 0 || value > 24) return BG_ERR_INVALID_ARG;  // (2^24 blocks of 192 symbols: relative counts stay below 2^32)
        ctx->fm_wide_sb_shift = (uint32_t)value;
        return BG_OK;
    }
    if (!strcmp(key, "fq_no_fused")) {
        ctx->fq_no_fused = value != 0;
        return BG_OK;
    }
    if (!strcmp(key, "sa_chunk_symbols")) {
        if (value < 0) return BG_ERR_INVALID_ARG;
        ctx->sa_chunk_symbols = value;
        return BG_OK;
    }
    if (!strcmp(key, "band_host_sync")) {
        ctx->band_host_sync = value != 0;
        return BG_OK;
    }
    if (!strcmp(key, "band_chain_rows")) {
        ctx->band_chain_rows = value != 0;
        return BG_OK;
    }
    if (!strcmp(key, "band_chain_global")) {
        ctx->band_chain_global = (int)value;
        return BG_OK;
    }
    if (!strcmp(key, "band_fill_v1")) {
        ctx->band_fill_v1 = value > 0 ? 1 : value < 0 ? -1 : 0;
        return BG_OK;
    }
    if (!strcmp(key, "band_interior_off")) {
        ctx->band_interior_off = value != 0;
        return BG_OK;
    }
    if (!strcmp(key, "band_packed_off")) {
        ctx->band_packed_off = value != 0;
        return BG_OK;
    }
    if (!strcmp(key, "band_packed_thresh")) {
        if (value < 0 || value > 0xffff) return BG_ERR_INVALID_ARG;
        ctx->band_packed_thresh = value;
        return BG_OK;
    }
    if (!strcmp(key, "band_pre_serial")) {
        ctx->band_pre_serial = value != 0;
        return BG_OK;
    }
    if (!strcmp(key, "band_join_serial")) {
        ctx->band_join_serial = value != 0;
        return BG_OK;
    }
    if (!strcmp(key, "band_tail_last")) {
        ctx->band_tail_last = value != 0;
        return BG_OK;
    }
    if (!strcmp(key, "band_window")) {
        ctx->band_window = value != 0;
        return BG_OK;
    }
    if (!strcmp(key, "band_raster_late")) {
        ctx->band_raster_late = value != 0;
        return BG_OK;
    }
    if (!strcmp(key, "band_join_late")) {
        ctx->band_join_late = value != 0;
        return BG_OK;
    }
    if (!strcmp(key, "band_p_block512")) {
        ctx->band_p_block512 = value != 0;
        return BG_OK;
    }
    if (!strcmp(key, "band_budget_gb")) {
        if (value < 0 || value > 256) return BG_ERR_INVALID_ARG;
        ctx->band_budget_gb = value;
        return BG_OK;
    }
    if (!strcmp(key, "band_on_host")) {
        ctx->band_on_host = value != 0;
        return BG_OK;
    }
    if (!strcmp(key, "fm_host_bytes")) {
        ctx->fm_host_bytes = value != 0;
        return BG_OK;
    }
    if (!strcmp(key, "host_chunk_pairs")) {
        ctx->host_chunk_pairs = value;
        return BG_OK;
    }
    if (!strcmp(key, "no_couples")) {
        ctx->no_couples = value != 0;
        return BG_OK;
    }
    if (!strcmp(key, "no_local_fast")) {
        ctx->no_local_fast = value != 0;
        return BG_OK;
    }
    if (!strcmp(key, "no_pk16")) {
        ctx->no_pk16 = value != 0;
        return BG_OK;
    }
    if (!strcmp(key, "force_wide")) {
        ctx->force_wide = value != 0;
        return BG_OK;
    }
    return BG_ERR_INVALID_ARG;
}

extern "C" int bg_enable_timing(bg_ctx* ctx, int on) {
    if (!ctx) return BG_ERR_INVALID_ARG;
    ctx->timing = on != 0;
    ctx->last = {};
    return BG_OK;
}

extern "C" int bg_get_timing(bg_ctx* ctx, bg_timing_t* out) {
    if (!ctx || !out) return BG_ERR_INVALID_ARG;
    *out = ctx->last;
    ctx->last = {};
    return BG_OK;
}

// ---- operations compacted on the device (bg_common.h) ------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void ops_counts_kernel(const bg_alignment_t* __restrict__ rec, uint64_t n, uint32_t* __restrict__ cnt) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // a failed pair (BG_ERR_OPS_CAP: n_ops beyond its slot, ops_off before it) contributes no bytes, as the host
    // compaction this replaces did
    if (p < n) cnt[p] = rec[p].status == BG_OK ? rec[p].n_ops : 0u;
}
// LANES lanes per pair: bytes [src, src + n_ops) -> compact (+ *cell) + off[p]; ops_off becomes *cell + off[p]
template <int LANES>
__global__ __launch_bounds__(256) void ops_compact_kernel(bg_alignment_t* __restrict__ rec, uint64_t n, const uint8_t* __restrict__ ops,
                                                          const uint64_t* __restrict__ off, const uint64_t* __restrict__ cell,
                                                          uint8_t* __restrict__ compact, int global_offsets) {
    const uint64_t p = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES;
    const uint32_t l = threadIdx.x % LANES;
    if (p >= n) return;
    const uint32_t k = rec[p].status == BG_OK ? rec[p].n_ops : 0u;
    const uint64_t base = *cell;
    const uint8_t* src = ops + rec[p].ops_off;
    uint8_t* dst = compact + (global_offsets ? base : 0ull) + off[p];
    for (uint32_t i = l; i < k; i += LANES) dst[i] = src[i];
    // every lane of the group has read ops_off before lane 0 overwrites it (same wavefront, program order)
    if (l == 0) rec[p].ops_off = base + off[p];
}
__global__ void ops_advance_kernel(uint64_t* cell, const uint64_t* __restrict__ off, uint64_t n, uint64_t* __restrict__ batch_total) {
    if (batch_total) *batch_total = off[n];
    cell[0] += off[n];
}
inline size_t a256(size_t v) { return (v + 255) & ~(size_t)255; }
}  // namespace

// bytes [slot[1] - slot[0], slot[1]) of d_src to the same offsets of a pinned host buffer, by a few blocks on the caller's
// (high-priority) stream: a download whose extent only the device knows when it is enqueued (sw_api.hip's stage results
// are fetched the same way).  Whole 16-byte units around the range travel: the bytes around it belong to the neighbouring
// ranges, whose own downloads — earlier or later on the same stream — bring their final values.
typedef uint32_t bg_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void range_to_host_kernel(const uint8_t* __restrict__ d_src, uint8_t* __restrict__ h_dst,
                                                            const uint64_t* __restrict__ slot) {
    const uint64_t end = slot[1], start = end - slot[0];
    const uint64_t a = start >> 4, b = (end + 15) >> 4;
    const bg_u32x4* src = (const bg_u32x4*)d_src;
    bg_u32x4* dst = (bg_u32x4*)h_dst;
    for (uint64_t i = a + (uint64_t)blockIdx.x * 256 + threadIdx.x; i < b; i += (uint64_t)gridDim.x * 256)
        __builtin_nontemporal_store(src[i], &dst[i]);
}
int bg_range_to_host(const uint8_t* d_src, uint8_t* h_dst, const uint64_t* d_slot, hipStream_t st) {
    range_to_host_kernel<<<dim3(8), dim3(256), 0, st>>>(d_src, h_dst, d_slot);
    BG_HIP(hipGetLastError());
    return BG_OK;
}

size_t bg_compact_ops_scratch(uint64_t n) { return a256(n * 4) + a256((n + 1) * 8) + 2 * (n / 2048 + 2) * 8 + 256; }

int bg_compact_ops_dev(bg_alignment_t* d_rec, uint64_t n, const uint8_t* d_ops, uint8_t* d_compact, bool global_offsets, uint64_t* d_cell,
                       uint64_t* d_batch_total, void* d_scratch, bool long_ops, hipStream_t st) {
    if (n == 0) {
        if (d_batch_total) BG_HIP(hipMemsetAsync(d_batch_total, 0, 8, st));
        return BG_OK;
    }
    uint32_t* d_cnt = (uint32_t*)d_scratch;
    uint64_t* d_off = (uint64_t*)((uint8_t*)d_scratch + a256(n * 4));
    uint64_t* d_sums = (uint64_t*)((uint8_t*)d_off + a256((n + 1) * 8));
    ops_counts_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(d_rec, n, d_cnt);
    int rc = bg_scan_u32(d_cnt, n, d_off, d_sums, st);
    if (rc) return rc;
    // short-read batches: 16 lanes per pair; long operation lists (banded 10 kb pairs): a wavefront per pair
    if (long_ops)
        ops_compact_kernel<64><<<dim3((unsigned)((n * 64 + 255) / 256)), dim3(256), 0, st>>>(d_rec, n, d_ops, d_off, d_cell, d_compact, global_offsets);
    else
        ops_compact_kernel<16><<<dim3((unsigned)((n * 16 + 255) / 256)), dim3(256), 0, st>>>(d_rec, n, d_ops, d_off, d_cell, d_compact, global_offsets);
    ops_advance_kernel<<<dim3(1), dim3(1), 0, st>>>(d_cell, d_off, n, d_batch_total);
    BG_HIP(hipGetLastError());
    return BG_OK;
}

