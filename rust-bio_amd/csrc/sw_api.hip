// C-ABI entry points of the full-matrix aligner: bg_align_batch / bg_align_batch_dev.
// Host side of `Aligner::{custom,global,semiglobal,local}` (pairwise/mod.rs:591,925,954,986):
// validates the scoring like the reference's asserts, applies the clip-penalty overrides of
// the three wrappers, compacts a tabulated match function, and drives K1 + K2 over
// sub-batches that share one reusable scratch (traceback words, aux rows, strip buffer).
#include <algorithm>
#include <chrono>
#include <map>
#include <type_traits>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "sw_kernels.h"

namespace bgsw {
sw_fill_fn get_fill_params_narrow(int lp, int r, bool local);
sw_fill_fn get_fill_params_wide(int lp, int r, bool local);
sw_fill_fn get_fill_params_lf(int lp, int r);
sw_fill_fn get_fill_matrix(int lp, int r, int sm, bool narrow, bool local);
sw_fill_fn get_fill_pk16_local(int lp, int r, int which);
sw_fill_fn get_fill_pk16_localfast(int lp, int r, int which);
sw_fill_fn get_fill_pk16_semiglobal(int lp, int r, int which);
sw_fill_fn get_fill_pk16_global(int lp, int r, int which);
sw_fill_fn get_fill_pk16_custom(int lp, int r, int which);
void launch_traceback(const SwArgs& a, int nw, hipStream_t st);

struct Config {
    int lp, r;
};
// rows of x covered per strip = lp * r; short reads pack several pairs into one wavefront
static Config pick_config(uint32_t m_cap, int sm) {
    if (sm == SCORE_PARAMS) {
        if (m_cap <= 192) return {16, std::max(2, 2 * (int)((m_cap + 31) / 32))};
        if (m_cap <= 384) return {32, std::max(8, 2 * (int)((m_cap + 63) / 64))};
        return {64, 8};
    }
    if (m_cap <= 96) return {16, 6};
    if (m_cap <= 128) return {16, 8};
    if (m_cap <= 160) return {16, 10};
    if (m_cap <= 192) return {16, 12};
    if (m_cap <= 384) return {32, 12};
    return {64, 8};
}

// Bytes with identical rows and columns in the 256x256 table are interchangeable: compact the
// closure's table to one code per class so that it fits LDS (A <= 64) whatever bytes occur.
static int compact_matrix(const int32_t* matrix, std::vector<uint8_t>& code_map,
                          std::vector<int32_t>& table) {
    std::map<std::vector<int32_t>, int> classes;
    code_map.assign(256, 0);
    std::vector<int> rep;
    for (int b = 0; b < 256; b++) {
        std::vector<int32_t> sig(512);
        for (int c = 0; c < 256; c++) {
            sig[c] = matrix[b * 256 + c];
            sig[256 + c] = matrix[c * 256 + b];
        }
        auto it = classes.find(sig);
        if (it == classes.end()) {
            it = classes.emplace(std::move(sig), (int)rep.size()).first;
            rep.push_back(b);
        }
        code_map[b] = (uint8_t)it->second;
    }
    const int A = (int)rep.size();
    table.assign((size_t)A * A, 0);
    for (int p = 0; p < A; p++)
        for (int q = 0; q < A; q++) table[(size_t)p * A + q] = matrix[rep[p] * 256 + rep[q]];
    return A;
}
}  // namespace bgsw

// ---- couples for K1p: visit the pairs of a ragged sub-batch in (m, n) order ---------------------------------
// K1p aligns two pairs per lane group and needs them to have equal lengths (else each costs a pass of its own).
// Reads of one length usually exist in numbers, just not next to each other: a counting sort on the 20-bit key
// m << 11 | n (histogram with atomics, scan, scatter with atomic cursors) puts them into neighbouring slots.
// The order inside a key is whatever the atomics give; results do not depend on a pair's partner.
namespace bgsw {
constexpr uint32_t kLenKeys = 1u << 20;  // m <= 384 < 2^9, n <= 2038 < 2^11 (the 12-bit score bound)
__global__ __launch_bounds__(256) void sw_len_stats_kernel(const uint64_t* __restrict__ x_off, const uint64_t* __restrict__ y_off, uint64_t pair0,
                                                           uint32_t n, uint32_t* __restrict__ st /* min m, min n, max m, max n */) {
    uint32_t m = 0, nn = 0, m_lo = ~0u, n_lo = ~0u;
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {  // few blocks: few atomics
        const uint32_t a = (uint32_t)(x_off[pair0 + p + 1] - x_off[pair0 + p]), b = (uint32_t)(y_off[pair0 + p + 1] - y_off[pair0 + p]);
        m = max(m, a);
        m_lo = min(m_lo, a);
        nn = max(nn, b);
        n_lo = min(n_lo, b);
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        m = max(m, (uint32_t)__shfl_xor((int)m, o));
        nn = max(nn, (uint32_t)__shfl_xor((int)nn, o));
        m_lo = min(m_lo, (uint32_t)__shfl_xor((int)m_lo, o));
        n_lo = min(n_lo, (uint32_t)__shfl_xor((int)n_lo, o));
    }
    __shared__ uint32_t s[4][4];
    if ((threadIdx.x & 63) == 0) {
        s[threadIdx.x >> 6][0] = m_lo;
        s[threadIdx.x >> 6][1] = m;
        s[threadIdx.x >> 6][2] = n_lo;
        s[threadIdx.x >> 6][3] = nn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // four atomics per block
        atomicMin(&st[0], min(min(s[0][0], s[1][0]), min(s[2][0], s[3][0])));
        atomicMax(&st[2], max(max(s[0][1], s[1][1]), max(s[2][1], s[3][1])));
        atomicMin(&st[1], min(min(s[0][2], s[1][2]), min(s[2][2], s[3][2])));
        atomicMax(&st[3], max(max(s[0][3], s[1][3]), max(s[2][3], s[3][3])));
    }
}
__device__ __forceinline__ uint32_t len_key(const uint64_t* x_off, const uint64_t* y_off, uint64_t r) {
    const uint32_t m = (uint32_t)(x_off[r + 1] - x_off[r]), n = (uint32_t)(y_off[r + 1] - y_off[r]);
    return min(m, 511u) << 11 | min(n, 2047u);
}
// one atomic per distinct key of a wavefront (a batch that is nearly of one length would otherwise hammer a single
// counter a million times): returns this lane's rank among the lanes with its key and, in the key's first lane, adds
// their number to counter[key] — `base` is what the counter held before
__device__ __forceinline__ uint32_t wave_key_add(uint32_t* counter, uint32_t key, bool valid, uint32_t& base) {
    const int lane = threadIdx.x & 63;
    uint32_t rank = 0;
    base = 0;
    uint64_t todo = __ballot(valid);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t k = (uint32_t)__shfl((int)key, leader);
        const uint64_t same = __ballot(valid && key == k) & todo;
        uint32_t b = 0;
        if (lane == leader) b = atomicAdd(&counter[k], (uint32_t)__popcll(same));
        b = (uint32_t)__shfl((int)b, leader);
        if (valid && key == k) {
            base = b;
            rank = (uint32_t)__popcll(same & ((1ull << lane) - 1));
        }
        todo &= ~same;
    }
    return rank;
}
__global__ __launch_bounds__(256) void sw_key_hist_kernel(const uint64_t* __restrict__ x_off, const uint64_t* __restrict__ y_off, uint64_t pair0,
                                                          uint32_t n, uint32_t* __restrict__ cnt, const uint32_t* __restrict__ st) {
    if (st && st[0] == st[2] && st[1] == st[3]) return;  // one length: no order needed
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = p < n;
    uint32_t base;
    wave_key_add(cnt, valid ? len_key(x_off, y_off, pair0 + p) : 0u, valid, base);
}
// exclusive scan of the 2^20 counters in place: 1024 blocks x 1024 counters, block totals scanned by block 0 of
// the second launch
__global__ __launch_bounds__(256) void sw_key_block_sums_kernel(const uint32_t* __restrict__ cnt, uint32_t* __restrict__ sums,
                                                                const uint32_t* __restrict__ st) {
    if (st && st[0] == st[2] && st[1] == st[3]) return;
    uint32_t v = 0;
    for (int i = 0; i < 4; i++) v += cnt[blockIdx.x * 1024u + i * 256u + threadIdx.x];
    __shared__ uint32_t s[4];
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}
__global__ __launch_bounds__(1024) void sw_key_scan_kernel(uint32_t* __restrict__ cnt, const uint32_t* __restrict__ sums,
                                                           const uint32_t* __restrict__ st) {
    if (st && st[0] == st[2] && st[1] == st[3]) return;
    __shared__ uint32_t s[1024];
    // offset of this block = sum of the totals of the blocks before it (1024 totals: every block scans them itself)
    s[threadIdx.x] = sums[threadIdx.x];
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const uint32_t u = threadIdx.x >= (unsigned)o ? s[threadIdx.x - o] : 0;
        __syncthreads();
        s[threadIdx.x] += u;
        __syncthreads();
    }
    const uint32_t base = blockIdx.x ? s[blockIdx.x - 1] : 0;
    __syncthreads();
    const uint32_t v = cnt[blockIdx.x * 1024u + threadIdx.x];
    s[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const uint32_t u = threadIdx.x >= (unsigned)o ? s[threadIdx.x - o] : 0;
        __syncthreads();
        s[threadIdx.x] += u;
        __syncthreads();
    }
    cnt[blockIdx.x * 1024u + threadIdx.x] = base + s[threadIdx.x] - v;
}
__global__ __launch_bounds__(256) void sw_key_scatter_kernel(const uint64_t* __restrict__ x_off, const uint64_t* __restrict__ y_off, uint64_t pair0,
                                                             uint32_t n, uint32_t* __restrict__ cursor, uint32_t* __restrict__ perm,
                                                             const uint32_t* __restrict__ st) {
    if (st && st[0] == st[2] && st[1] == st[3]) return;
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = p < n;
    uint32_t base;
    const uint32_t rank = wave_key_add(cursor, valid ? len_key(x_off, y_off, pair0 + p) : 0u, valid, base);
    if (valid) perm[base + rank] = p;  // pairs of one wavefront and key stay in order: neighbours in memory stay neighbours
}
// n_eff[0] = pairs K2's identity flavour has to do, n_eff[1] = the permuted one's (sw_kernels.h); forced: 0 ragged, -1 ask st
__global__ void sw_decide_kernel(const uint32_t* __restrict__ st, uint32_t n, int forced, uint32_t* __restrict__ n_eff) {
    const bool uniform = forced < 0 && st[0] == st[2] && st[1] == st[3];
    n_eff[0] = uniform ? n : 0u;
    n_eff[1] = uniform ? 0u : n;
}
}  // namespace bgsw

using namespace bgsw;

// shared with banded_api.hip
int bg_compact_matrix(const int32_t* matrix, std::vector<uint8_t>& code_map, std::vector<int32_t>& table) {
    return compact_matrix(matrix, code_map, table);
}

static int check_scoring(const bg_scoring_t* sc) {
    // asserts of Scoring::new/from_scores (mod.rs:265-266,292-293) and
    // Aligner::with_capacity_and_scoring (mod.rs:554-571)
    if (sc->gap_open > 0 || sc->gap_extend > 0 || sc->xclip_prefix > 0 || sc->xclip_suffix > 0 ||
        sc->yclip_prefix > 0 || sc->yclip_suffix > 0)
        return BG_ERR_POSITIVE_PENALTY;
    return BG_OK;
}

// len_hint: what the caller knows about the lengths of the batch — 1 every pair has the same (m, n), 0 they differ,
// -1 unknown (a reduction on the device + one stream synchronisation per sub-batch finds out)
static int align_batch_dev_impl(bg_ctx* ctx, const bg_scoring_t* sc, int mode, uint64_t n_pairs,
                                const uint8_t* d_x, const uint64_t* d_x_off, const uint8_t* d_y,
                                const uint64_t* d_y_off, uint32_t max_xlen, uint32_t max_ylen,
                                bg_alignment_t* d_out, uint8_t* d_ops, uint64_t ops_stride,
                                void* stream, int len_hint, const uint8_t* packed_codes = nullptr) {
    if (!ctx || !sc || mode < BG_MODE_CUSTOM || mode > BG_MODE_LOCAL) return BG_ERR_INVALID_ARG;
    int rc = check_scoring(sc);
    if (rc) return rc;
    if (n_pairs == 0) return BG_OK;
    if (!d_x_off || !d_y_off || !d_out) return BG_ERR_INVALID_ARG;
    if (d_ops && ops_stride < (uint64_t)max_xlen + max_ylen + 4) return BG_ERR_OPS_CAP;
    if (max_xlen > (1u << 24) || max_ylen > (1u << 24)) return BG_ERR_TOO_LARGE;
    hipStream_t st = (hipStream_t)stream;
    BG_HIP(hipSetDevice(ctx->device));
    bg_scratch_guard guard(ctx, st);

    SwArgs a = {};
    a.x = d_x;
    a.x_off = d_x_off;
    a.y = d_y;
    a.y_off = d_y_off;
    a.sc = {sc->gap_open,     sc->gap_extend,   sc->xclip_prefix, sc->xclip_suffix,
            sc->yclip_prefix, sc->yclip_suffix, sc->match_score,  sc->mismatch_score};
    // the three wrappers overwrite the clip penalties (mod.rs:934-938, 963-967, 995-999)
    if (mode == BG_MODE_GLOBAL) a.sc.xp = a.sc.xs = a.sc.yp = a.sc.ys = BG_MIN_SCORE;
    if (mode == BG_MODE_SEMIGLOBAL) {
        a.sc.xp = a.sc.xs = BG_MIN_SCORE;
        a.sc.yp = a.sc.ys = 0;
    }
    if (mode == BG_MODE_LOCAL) a.sc.xp = a.sc.xs = a.sc.yp = a.sc.ys = 0;
    a.mode = mode;
    a.filter_clips = (mode == BG_MODE_SEMIGLOBAL || mode == BG_MODE_LOCAL);
    a.out = d_out;
    a.ops = d_ops;
    a.ops_stride = ops_stride;

    int sm = SCORE_PARAMS;
    if (sc->matrix) {
        // The compacted table stays on the device between calls: a caller aligns batch after batch under one matrix
        // (`Aligner::with_scoring` once, `local()` per pair), and compacting 65 536 entries, two copies and a
        // synchronisation per call were 3 ms next to a 31 ms step.  Recognised by a hash of the 256 KB, confirmed by comparing
        // the matrix with the copy kept on the host (a 64-bit non-cryptographic hash alone could, once in 2^64 calls, align
        // under the wrong matrix); the device copy has a buffer of its own that no other path writes.
        uint64_t h = 0xcbf29ce484222325ull;
        {
            const uint64_t* w = (const uint64_t*)sc->matrix;
            uint64_t h2 = 0x9e3779b97f4a7c15ull;
            for (size_t t = 0; t < 32768; t += 2) {
                h = (h ^ w[t]) * 0x100000001b3ull;
                h2 = (h2 + w[t + 1]) * 0xff51afd7ed558ccdull;
                h2 ^= h2 >> 32;
            }
            h ^= h2;
            if (h == 0) h = 1;
        }
        if (ctx->table_hash != h || !ctx->table_matrix || memcmp(ctx->table_matrix, sc->matrix, 65536 * sizeof(int32_t)) != 0) {
            std::vector<uint8_t> code_map;
            std::vector<int32_t> table;
            const int A = compact_matrix(sc->matrix, code_map, table);
            const size_t bytes = 256 + table.size() * 4;
            ctx->table_hash = 0;
            if (!ctx->table_matrix && !(ctx->table_matrix = (int32_t*)malloc(65536 * sizeof(int32_t)))) return BG_ERR_OOM;
            if ((rc = bg_reserve(&ctx->sw_table, &ctx->sw_table_bytes, bytes))) return rc;
            BG_HIP(hipMemcpyAsync((uint8_t*)ctx->sw_table + 256, table.data(), table.size() * 4,
                                  hipMemcpyHostToDevice, st));
            BG_HIP(hipMemcpyAsync(ctx->sw_table, code_map.data(), 256, hipMemcpyHostToDevice, st));
            BG_HIP(hipStreamSynchronize(st));  // the host vectors go out of scope
            memcpy(ctx->table_matrix, sc->matrix, 65536 * sizeof(int32_t));
            ctx->table_hash = h;
            ctx->table_alpha = A;
        }
        const int A = ctx->table_alpha;
        sm = A <= kMaxLdsAlphabet ? SCORE_LDS : SCORE_GLOBAL;
        a.code_map = (const uint8_t*)ctx->sw_table;
        a.table = (const int32_t*)((uint8_t*)ctx->sw_table + 256);
        a.alpha = A;
    }

    Config cfg = pick_config(max_xlen, sm);
    const bool all_zero_clips = a.sc.xp == 0 && a.sc.xs == 0 && a.sc.yp == 0 && a.sc.ys == 0;
    // NARROW kernels need every reachable score inside +-2^25 (sw_kernels.h): bound it by
    // (longest path) x (largest finite magnitude in the scoring)
    int64_t mag = std::max<int64_t>(std::abs((int64_t)a.sc.go), std::abs((int64_t)a.sc.ge));
    for (int32_t c : {a.sc.xp, a.sc.xs, a.sc.yp, a.sc.ys})
        if (c != BG_MIN_SCORE) mag = std::max<int64_t>(mag, std::abs((int64_t)c));
    if (sc->matrix) {
        for (size_t t = 0; t < 65536; t++) mag = std::max<int64_t>(mag, std::abs((int64_t)sc->matrix[t]));
    } else {
        mag = std::max<int64_t>(mag, std::max(std::abs((int64_t)a.sc.match), std::abs((int64_t)a.sc.mismatch)));
    }
    const bool narrow = !ctx->force_wide && mag * ((int64_t)max_xlen + max_ylen + 8) < (1 << 24);
    sw_fill_fn fill = sm == SCORE_PARAMS
                          ? (narrow ? get_fill_params_narrow(cfg.lp, cfg.r, all_zero_clips)
                                    : get_fill_params_wide(cfg.lp, cfg.r, all_zero_clips))
                          : get_fill_matrix(cfg.lp, cfg.r, sm, narrow, all_zero_clips);
    if (!fill && sm != SCORE_PARAMS && cfg.lp == 16 && cfg.r < 12) {
        // 8 and 10 rows per lane exist for the LDS table with narrow scores only (the protein case they were measured
        // on); a table beyond 64 classes or wide scores take the 12-row geometry that is instantiated for everything
        cfg.r = 12;
        fill = get_fill_matrix(cfg.lp, cfg.r, sm, narrow, all_zero_clips);
    }
    // K1's LF flavour (sw_fill.inc): Aligner::local under MatchParams whose gaps and mismatches cost something, scaled
    // keys, reads of one strip — the reference-width (int32) kernel without the clip machinery such alignments never use
    a.g.tb_fmt = 0;
    if (fill && sm == SCORE_PARAMS && narrow && all_zero_clips && a.sc.go < 0 && a.sc.mismatch < 0 && a.sc.match >= 0 &&
        !ctx->no_local_fast && max_xlen <= (uint32_t)(cfg.lp * cfg.r)) {
        if (sw_fill_fn lf = get_fill_params_lf(cfg.lp, cfg.r)) {
            fill = lf;
            a.g.tb_fmt = 3;
        }
    }
    // K1p: short reads whose scores fit 12 bits (sw_fill_pk16.inc) — two pairs per lane, one instantiation
    // unit per clip pattern.  Bound: no real DP value, nor the epilogue's go * i terms, may leave +-2040.
    sw_fill_fn fill_rest = nullptr, fill_second = nullptr;
    bool pk16 = !ctx->no_pk16 && sm == SCORE_PARAMS && cfg.lp <= 32 && max_xlen >= 1 &&
                      mag * ((int64_t)std::max(max_xlen, max_ylen) + 2) <= 2040;
    if (pk16) {
        const SwScoring& c = a.sc;
        const int32_t M = BG_MIN_SCORE;
        // local alignments whose gaps and mismatches cost something take the LF flavour (sw_fill_pk16.inc)
        // (match >= 0: the LF cell's unsigned mad needs matchkey - mismatchkey >= 0; MatchParams::new asserts it, mod.rs:199-200,
        //  but nothing in the C API does)
        const bool local_fast = all_zero_clips && c.go < 0 && c.mismatch < 0 && c.match >= 0 && !ctx->no_local_fast;
        auto getter = local_fast ? get_fill_pk16_localfast
                      : all_zero_clips ? get_fill_pk16_local
                      : (c.xp == M && c.xs == M && c.yp == 0 && c.ys == 0) ? get_fill_pk16_semiglobal
                      : (c.xp == M && c.xs == M && c.yp == M && c.ys == M) ? get_fill_pk16_global
                                                                           : get_fill_pk16_custom;
        // rows per lane: the fast launch wants row m on the last row of a lane (m % R == 0); reads of a
        // batch usually share one length, so prefer an instantiated R that divides the longest
        int r_pick = 0;
        for (int r = (int)((max_xlen + cfg.lp - 1) / cfg.lp); r <= 12 && !r_pick; r++)
            if (max_xlen % r == 0 && getter(cfg.lp, r, 0)) r_pick = r;
        for (int r = (int)((max_xlen + cfg.lp - 1) / cfg.lp); r <= 12 && !r_pick; r++)
            if (getter(cfg.lp, r, 0)) r_pick = r;
        if (!r_pick) {
            pk16 = false;  // no K1p instantiation for this shape: the general kernel K1 picked above runs
        } else {
            cfg.r = r_pick;
            fill = getter(cfg.lp, cfg.r, 0);
            fill_rest = local_fast ? nullptr : getter(cfg.lp, cfg.r, 1);  // LF: the first launch takes every wavefront
            fill_second = getter(cfg.lp, cfg.r, 2);
            a.g.tb_fmt = local_fast ? 2 : 1;
        }
    }
    if (!fill) return BG_ERR_UNSUPPORTED;
    if (packed_codes) {
        if (pk16) {
            a.packed = 1;  // K1p reads the 2-bit streams as they are
        } else {
            // every other kernel takes bytes: the streams are unpacked into ctx scratch (two totals come back to the host)
            uint64_t tot[2] = {0, 0};
            BG_HIP(hipMemcpyAsync(&tot[0], d_x_off + n_pairs, 8, hipMemcpyDeviceToHost, st));
            BG_HIP(hipMemcpyAsync(&tot[1], d_y_off + n_pairs, 8, hipMemcpyDeviceToHost, st));
            BG_HIP(hipStreamSynchronize(st));
            for (int k = 0; k < 2; k++)
                if ((rc = bg_reserve(&ctx->unpk[k], &ctx->unpk_cap[k], std::max<uint64_t>(tot[k], 16)))) return rc;
            if ((rc = bg_unpack2_dev(ctx, (const uint32_t*)d_x, tot[0], packed_codes, (uint8_t*)ctx->unpk[0], st))) return rc;
            if ((rc = bg_unpack2_dev(ctx, (const uint32_t*)d_y, tot[1], packed_codes, (uint8_t*)ctx->unpk[1], st))) return rc;
            a.x = (const uint8_t*)ctx->unpk[0];
            a.y = (const uint8_t*)ctx->unpk[1];
        }
    }
    const int nw = tb_words(cfg.r);
    const uint32_t pw = 64 / cfg.lp;
    SwGeom& g = a.g;
    g.lp = cfg.lp;
    g.r = cfg.r;
    g.r_inv = (uint32_t)(((1ull << 32) + cfg.r - 1) / cfg.r);
    g.lp_shift = cfg.lp == 16 ? 4 : cfg.lp == 32 ? 5 : 6;
    g.m_cap = max_xlen;
    g.n_cap = max_ylen;
    g.nsteps = max_ylen ? max_ylen + cfg.lp - 1 : 0;
    g.nstrips = std::max<uint32_t>(1, (max_xlen + cfg.lp * cfg.r - 1) / (cfg.lp * cfg.r));
    g.aux_stride = SwGeom::stride_for(max_xlen, max_ylen);

    // scratch per wavefront job / per pair
    const size_t tb_per_job = (size_t)tb_job_words(g.nstrips, g.nsteps, nw) * 4;
    const size_t aux_per_pair = (size_t)g.aux_stride * 4;
    const size_t bnd_per_pair = g.nstrips > 1 ? (size_t)(g.n_cap + 1) * 16 : 0;
    const size_t per_pair = tb_per_job / pw + aux_per_pair + bnd_per_pair + 1;
    uint64_t chunk = ctx->chunk_pairs > 0 ? (uint64_t)ctx->chunk_pairs : (1u << 20);
    const uint64_t budget = 48ull << 30;  // scratch budget (of 288 GB HBM)
    chunk = std::min<uint64_t>(chunk, std::max<uint64_t>(pw, budget / per_pair));
    chunk = std::min<uint64_t>(chunk, n_pairs);
    chunk = (chunk + pw - 1) / pw * pw;
    const uint64_t jobs = chunk / pw;
    if ((rc = bg_reserve(&ctx->tb, &ctx->tb_bytes, std::max<size_t>(jobs * tb_per_job, 64)))) return rc;
    if ((rc = bg_reserve(&ctx->aux, &ctx->aux_bytes, chunk * aux_per_pair))) return rc;
    if ((rc = bg_reserve(&ctx->bnd, &ctx->bnd_bytes, std::max<size_t>(chunk * bnd_per_pair, 64)))) return rc;
    a.tb = ctx->tb;
    a.aux = (int32_t*)ctx->aux;
    a.bnd = (int4*)ctx->bnd;
    // K1p on a ragged batch: slots in (m, n) order (the strip buffer is free: K1p has one strip)
    uint32_t *d_keycnt = nullptr, *d_keysum = nullptr, *d_perm = nullptr, *d_lenst = nullptr;
    if (pk16 && !ctx->no_couples) {
        const size_t need = (size_t)kLenKeys * 4 + 1024 * 4 + 64 + chunk * 4;
        if ((rc = bg_reserve(&ctx->bnd, &ctx->bnd_bytes, need))) return rc;
        a.bnd = (int4*)ctx->bnd;
        d_keycnt = (uint32_t*)ctx->bnd;
        d_keysum = d_keycnt + kLenKeys;
        d_lenst = d_keysum + 1024;
        d_perm = d_lenst + 16;
    }

    for (uint64_t p0 = 0; p0 < n_pairs; p0 += chunk) {
        a.pair0 = p0;
        a.n_pairs = (uint32_t)std::min<uint64_t>(chunk, n_pairs - p0);
        const uint32_t njobs = (a.n_pairs + pw - 1) / pw;
        a.perm = nullptr;
        a.len_stats = nullptr;
        a.n_eff = nullptr;
        if (d_perm && a.n_pairs >= 64 && len_hint != 1) {
            const uint32_t* d_st = nullptr;
            if (len_hint < 0) {  // unknown: the device finds out and every kernel below looks at its answer — no host round trip
                // minima start at ~0, maxima at 0 (memsets: a pageable upload would make the launching thread wait for the stream)
                BG_HIP(hipMemsetAsync(d_lenst, 0xff, 8, st));
                BG_HIP(hipMemsetAsync(d_lenst + 2, 0, 8, st));
                sw_len_stats_kernel<<<dim3(std::min<uint32_t>((a.n_pairs + 255) / 256, 256)), dim3(256), 0, st>>>(d_x_off, d_y_off, p0, a.n_pairs, d_lenst);
                d_st = d_lenst;
            }
            BG_HIP(hipMemsetAsync(d_keycnt, 0, (size_t)kLenKeys * 4, st));
            sw_key_hist_kernel<<<dim3((a.n_pairs + 255) / 256), dim3(256), 0, st>>>(d_x_off, d_y_off, p0, a.n_pairs, d_keycnt, d_st);
            sw_key_block_sums_kernel<<<dim3(1024), dim3(256), 0, st>>>(d_keycnt, d_keysum, d_st);
            sw_key_scan_kernel<<<dim3(1024), dim3(1024), 0, st>>>(d_keycnt, d_keysum, d_st);
            sw_key_scatter_kernel<<<dim3((a.n_pairs + 255) / 256), dim3(256), 0, st>>>(d_x_off, d_y_off, p0, a.n_pairs, d_keycnt, d_perm, d_st);
            BG_HIP(hipGetLastError());
            sw_decide_kernel<<<dim3(1), dim3(1), 0, st>>>(d_lenst, a.n_pairs, len_hint < 0 ? -1 : 0, d_lenst + 4);
            a.perm = d_perm;
            a.len_stats = d_st;
            a.n_eff = d_lenst + 4;
        }
        if (ctx->timing) BG_HIP(hipEventRecord(ctx->ev[0], st));
        const uint32_t nwaves = pk16 ? (njobs + 1) / 2 : njobs;  // a K1p wavefront takes two jobs
        fill<<<dim3((nwaves + 3) / 4), dim3(256), 0, st>>>(a);
        BG_HIP(hipGetLastError());
        if (ctx->timing) BG_HIP(hipEventRecord(ctx->ev[1], st));
        // what the fast launch skipped: wavefronts with other read lengths, then the second pairs of unequal couples
        if (fill_rest) fill_rest<<<dim3((nwaves + 3) / 4), dim3(256), 0, st>>>(a);
        if (fill_second && len_hint != 1) fill_second<<<dim3((nwaves + 3) / 4), dim3(256), 0, st>>>(a);  // (1: the caller knows the lengths agree)
        BG_HIP(hipGetLastError());
        if (ctx->timing) {
            BG_HIP(hipEventSynchronize(ctx->ev[1]));
            float ms = 0;
            BG_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
            ctx->last.fill_ms += ms;
            ctx->last.fill_launches += 1;
            BG_HIP(hipEventRecord(ctx->ev[0], st));
        }
        launch_traceback(a, nw, st);
        BG_HIP(hipGetLastError());
        if (ctx->timing) {
            BG_HIP(hipEventRecord(ctx->ev[1], st));
            BG_HIP(hipEventSynchronize(ctx->ev[1]));
            float ms = 0;
            BG_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
            ctx->last.traceback_ms += ms;
            ctx->last.traceback_launches += 1;
        }
    }
    return BG_OK;
}

int bg_align_batch_dev_hint(bg_ctx* ctx, const bg_scoring_t* sc, int mode, uint64_t n_pairs, const uint8_t* d_x,
                            const uint64_t* d_x_off, const uint8_t* d_y, const uint64_t* d_y_off, uint32_t max_xlen,
                            uint32_t max_ylen, bg_alignment_t* d_out, uint8_t* d_ops, uint64_t ops_stride, void* stream,
                            int len_hint) {
    return align_batch_dev_impl(ctx, sc, mode, n_pairs, d_x, d_x_off, d_y, d_y_off, max_xlen, max_ylen, d_out, d_ops, ops_stride,
                                stream, len_hint);
}

extern "C" int bg_align_batch_dev(bg_ctx* ctx, const bg_scoring_t* sc, int mode, uint64_t n_pairs,
                                  const uint8_t* d_x, const uint64_t* d_x_off, const uint8_t* d_y,
                                  const uint64_t* d_y_off, uint32_t max_xlen, uint32_t max_ylen,
                                  bg_alignment_t* d_out, uint8_t* d_ops, uint64_t ops_stride,
                                  void* stream) {
    return align_batch_dev_impl(ctx, sc, mode, n_pairs, d_x, d_x_off, d_y, d_y_off, max_xlen, max_ylen, d_out, d_ops, ops_stride, stream, -1);
}

// Aligner::{custom, global, semiglobal, local} on 2-bit streams (pack2.hip): x / y hold 16 symbols per dword, the offsets
// count symbols.  K1p loads the codes directly; whatever K1p does not take (long reads, wide scores, a tabulated match
// function — which sees the bytes `codes` stand for) is unpacked into ctx scratch first.
extern "C" int bg_align_batch_packed_dev(bg_ctx* ctx, const bg_scoring_t* sc, int mode, uint64_t n_pairs, const uint32_t* d_x,
                                         const uint64_t* d_x_off, const uint32_t* d_y, const uint64_t* d_y_off,
                                         const uint8_t codes[4], uint32_t max_xlen, uint32_t max_ylen, bg_alignment_t* d_out,
                                         uint8_t* d_ops, uint64_t ops_stride, void* stream) {
    if (!codes || (n_pairs && (!d_x || !d_y))) return BG_ERR_INVALID_ARG;
    for (int a = 0; a < 4; a++)
        for (int b = a + 1; b < 4; b++)
            if (codes[a] == codes[b]) return BG_ERR_INVALID_ARG;
    return align_batch_dev_impl(ctx, sc, mode, n_pairs, (const uint8_t*)d_x, d_x_off, (const uint8_t*)d_y, d_y_off, max_xlen, max_ylen,
                                d_out, d_ops, ops_stride, stream, -1, codes);
}

// ---- pipelined host-buffer path ------------------------------------------------------------------------------
// bg_align_batch on a large batch is PCIe + host memcpy + kernels; run as one serial sequence it spends three
// quarters of its time outside the kernels.  The batch is cut into stages of `host_chunk_pairs` pairs that flow
// through three staging sets: host threads copy a stage's sequences into pinned memory (and rebase its offsets),
// the copy-in stream uploads it, the kernel stream aligns it (bg_align_batch_dev), the copy-out stream downloads
// records + strided operations into pinned memory, and host threads compact those into the caller's buffers
// while the following stages are in flight.
struct bg_host_pipe {
    static constexpr int NSET = 3;
    struct Set {
        uint8_t *h_in = nullptr, *h_out = nullptr;  // pinned: x | y | x_off | y_off   and   records | stage total | compact ops
        uint8_t *d_in = nullptr, *d_out = nullptr;  // d_out: records | stage total | strided ops
        uint8_t *d_cmp = nullptr, *d_scan = nullptr;  // compact ops of the stage; n_ops counts, their offsets, scan partials
        size_t in_cap = 0, out_cap = 0, cmp_cap = 0, scan_cap = 0;
        hipEvent_t in_done = nullptr, k_done = nullptr, out_done = nullptr;
    } set[NSET];
    uint64_t* d_cell = nullptr;  // running total of operation bytes over the stages of a call
    hipStream_t s_in = nullptr, s_out = nullptr;  // uploads; downloads (stage_to_host_kernel)
};
void bg_host_pipe_free(bg_host_pipe* p) {
    if (!p) return;
    for (auto& s : p->set) {
        if (s.h_in) hipHostFree(s.h_in);
        if (s.h_out) hipHostFree(s.h_out);
        hipFree(s.d_in);
        hipFree(s.d_out);
        hipFree(s.d_cmp);
        hipFree(s.d_scan);
        for (hipEvent_t e : {s.in_done, s.k_done, s.out_done})
            if (e) hipEventDestroy(e);
    }
    hipFree(p->d_cell);
    if (p->s_in) hipStreamDestroy(p->s_in);
    if (p->s_out) hipStreamDestroy(p->s_out);
    delete p;
}
namespace {
// A stage's results go to the pinned host buffers by a kernel: records (size known) and the compact operations, whose
// size only the device knows when the copy is enqueued — as copy commands the two had to be issued one after the other
// by the host, the second once the first had arrived (1.2 ms per stage, more than the stage's kernels take).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void stage_to_host_kernel(const u32x4* __restrict__ rec, u32x4* __restrict__ h_rec, uint64_t rec_n16,
                                                            const u32x4* __restrict__ ops, u32x4* __restrict__ h_ops,
                                                            const uint64_t* __restrict__ d_total, uint64_t* __restrict__ h_total) {
    const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x, nth = (uint64_t)gridDim.x * 256;
    for (uint64_t i = tid; i < rec_n16; i += nth) {
        if (NT) __builtin_nontemporal_store(rec[i], &h_rec[i]);
        else h_rec[i] = rec[i];
    }
    const uint64_t T = d_total ? *d_total : 0, n16 = (T + 15) / 16;
    for (uint64_t i = tid; i < n16; i += nth) {
        if (NT) __builtin_nontemporal_store(ops[i], &h_ops[i]);
        else h_ops[i] = ops[i];
    }
    if (tid == 0) *h_total = T;
}

template <typename F>
void parallel_for(uint64_t n, uint64_t grain, F&& f) {  // f(begin, end) on the host threads this process may use
    const unsigned nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(bg_host_threads(), n / std::max<uint64_t>(grain, 1) + 1));
    if (nt == 1) {
        f((uint64_t)0, n);
        return;
    }
    bg_pool_run(nt, [&](unsigned t) { f(n * t / nt, n * (t + 1) / nt); });
}
void parallel_memcpy(uint8_t* dst, const uint8_t* src, uint64_t n) {
    parallel_for(n, 1 << 20, [&](uint64_t a, uint64_t b) { memcpy(dst + a, src + a, b - a); });
}

int align_batch_pipelined(bg_ctx* ctx, const bg_scoring_t* sc, int mode, uint64_t n_pairs, const uint8_t* x, const uint64_t* x_off,
                          const uint8_t* y, const uint64_t* y_off, uint32_t max_x, uint32_t max_y, uint64_t chunk, bg_alignment_t* out,
                          uint8_t* ops_buf, uint64_t ops_cap, uint64_t* ops_used) {
    if (!ctx->pipe) {
        ctx->pipe = new bg_host_pipe();
        BG_HIP(hipStreamCreateWithFlags(&ctx->pipe->s_in, hipStreamNonBlocking));
        {  // the download kernel of a finished stage goes ahead of the blocks of the next stage's fill
            int lo = 0, hi = 0;
            BG_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
            BG_HIP(hipStreamCreateWithPriority(&ctx->pipe->s_out, hipStreamNonBlocking, hi));
        }
        BG_HIP(hipMalloc((void**)&ctx->pipe->d_cell, 64));
        for (auto& s : ctx->pipe->set)
            for (hipEvent_t* e : {&s.in_done, &s.k_done, &s.out_done}) BG_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    bg_host_pipe& P = *ctx->pipe;
    const uint64_t stride = ops_buf ? (uint64_t)max_x + max_y + 4 : 0;
    // stage c covers pairs [cut[c], cut[c + 1]).  (Shorter first stages — kernels starting 1 ms earlier — were measured: no
    // gain, the device side of the stages is what the call waits for.)
    std::vector<uint64_t> cut{0};
    while (cut.back() < n_pairs) cut.push_back(std::min(n_pairs, cut.back() + chunk));
    const uint64_t nch = cut.size() - 1;
    // capacity of a staging set: the largest stage
    uint64_t max_xb = 0, max_yb = 0;
    for (uint64_t c = 0; c < nch; c++) {
        const uint64_t p0 = cut[c], p1 = cut[c + 1];
        max_xb = std::max(max_xb, x_off[p1] - x_off[p0]);
        max_yb = std::max(max_yb, y_off[p1] - y_off[p0]);
    }
    const uint64_t o_y = (max_xb + 255) & ~255ull, o_xo = o_y + ((max_yb + 255) & ~255ull), o_yo = o_xo + (chunk + 1) * 8;
    const size_t in_need = o_yo + (chunk + 1) * 8 + 256;
    // out: records | this stage's operation byte count | operations (device: strided slots; host: compact)
    const uint64_t o_tot = chunk * sizeof(bg_alignment_t), o_ops = o_tot + 256;
    const size_t out_need = o_ops + chunk * stride + 256;
    const size_t cmp_need = chunk * stride + 256;
    const size_t scan_need = bg_compact_ops_scratch(chunk);
    for (auto& s : P.set) {
        if (s.in_cap < in_need) {
            if (s.h_in) hipHostFree(s.h_in);
            hipFree(s.d_in);
            s.h_in = s.d_in = nullptr;
            s.in_cap = 0;
            BG_HIP(hipHostMalloc((void**)&s.h_in, in_need, hipHostMallocDefault));
            BG_HIP(hipMalloc((void**)&s.d_in, in_need));
            s.in_cap = in_need;
        }
        if (s.out_cap < out_need) {
            if (s.h_out) hipHostFree(s.h_out);
            hipFree(s.d_out);
            s.h_out = s.d_out = nullptr;
            s.out_cap = 0;
            BG_HIP(hipHostMalloc((void**)&s.h_out, out_need, hipHostMallocDefault));
            BG_HIP(hipMalloc((void**)&s.d_out, out_need));
            s.out_cap = out_need;
        }
        if (stride && s.cmp_cap < cmp_need) {
            hipFree(s.d_cmp);
            s.d_cmp = nullptr;
            s.cmp_cap = 0;
            BG_HIP(hipMalloc((void**)&s.d_cmp, cmp_need));
            s.cmp_cap = cmp_need;
        }
        if (stride && s.scan_cap < scan_need) {
            hipFree(s.d_scan);
            s.d_scan = nullptr;
            s.scan_cap = 0;
            BG_HIP(hipMalloc((void**)&s.d_scan, scan_need));
            s.scan_cap = scan_need;
        }
    }
    hipStream_t s_k = ctx->stream;
    BG_HIP(hipMemsetAsync(P.d_cell, 0, 8, s_k));
    uint64_t used = 0;
    int status = BG_OK;
    // BG_TRACE_HOST=1: where the host side of the stages spends its time (ms, summed over the call)
    const bool trace = getenv("BG_TRACE_HOST") != nullptr;
    double t_pack = 0, t_launch = 0, t_wait_set = 0, t_d_wait_rec = 0, t_d_rec = 0, t_d_ops = 0;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    // finish stage c: records (their ops_off already final) and the stage's compact operations into the caller's buffers —
    // two contiguous blocks; the operations are fetched with their exact size once the records (and the count) are here
    auto drain = [&](uint64_t c) -> int {
        bg_host_pipe::Set& S = P.set[c % bg_host_pipe::NSET];
        double t0 = now();
        BG_HIP(hipEventSynchronize(S.out_done));
        t_d_wait_rec += now() - t0;
        const uint64_t p0 = cut[c], np = cut[c + 1] - p0;
        const bg_alignment_t* h_rec = (const bg_alignment_t*)S.h_out;
        const uint64_t T = stride ? *(const uint64_t*)(S.h_out + o_tot) : 0;
        std::atomic<int> st_rec{BG_OK};
        t0 = now();
        parallel_for(np, 16384, [&](uint64_t a, uint64_t b) {
            memcpy(out + p0 + a, h_rec + a, (b - a) * sizeof(bg_alignment_t));
            int sr = BG_OK;
            for (uint64_t p = a; p < b; p++)
                if (h_rec[p].status) sr = h_rec[p].status;
            if (sr) st_rec = sr;
        });
        if (st_rec) status = st_rec;
        t_d_rec += now() - t0;
        if (T) {
            t0 = now();
            uint64_t fit = T;
            if (used + T > ops_cap) {  // the caller's buffer ends inside this stage: whole pairs only, as the serial path does
                if (status == BG_OK) status = BG_ERR_OPS_CAP;
                fit = 0;
                for (uint64_t p = 0; p < np; p++) {
                    if (h_rec[p].ops_off + h_rec[p].n_ops > ops_cap) break;
                    fit = h_rec[p].ops_off + h_rec[p].n_ops - used;
                }
            }
            if (fit) parallel_memcpy(ops_buf + used, S.h_out + o_ops, fit);
            t_d_ops += now() - t0;
        }
        used += T;
        return BG_OK;
    };
    // The stages are drained by a second host thread, in order, while this one packs and launches the following ones:
    // per stage the host's share is max(pack, drain) instead of their sum (the host side, not PCIe or the kernels, is
    // what bounds this entry point).  `used` / `status` belong to the drainer until it is joined.
    std::mutex mu;
    std::condition_variable cv;
    uint64_t submitted = 0, drained = 0;
    bool abort_drain = false;
    int drain_rc = BG_OK;
    std::string drain_err;  // the drainer's thread-local HIP error text, handed to the caller's bg_last_error()
    const int device = ctx->device;
    std::thread drainer([&] {
        if (hipSetDevice(device) != hipSuccess) {
            std::lock_guard<std::mutex> lk(mu);
            drain_rc = BG_ERR_HIP;
            drained = nch;
            cv.notify_all();
            return;
        }
        for (uint64_t c = 0; c < nch; c++) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return submitted > c || abort_drain; });
                if (submitted <= c) break;  // aborted before this stage was launched
            }
            const int r = drain(c);
            std::lock_guard<std::mutex> lk(mu);
            if (r && drain_rc == BG_OK) {
                drain_rc = r;
                drain_err = bg_tls_error;
            }
            drained = c + 1;
            cv.notify_all();
        }
        std::lock_guard<std::mutex> lk(mu);
        drained = nch;
        cv.notify_all();
    });
    auto stop_drainer = [&] {
        {
            std::lock_guard<std::mutex> lk(mu);
            abort_drain = true;
        }
        cv.notify_all();
        drainer.join();
    };
    int rc;
    for (uint64_t c = 0; c < nch; c++) {
        double t0 = now();
        if (c >= bg_host_pipe::NSET) {  // this stage's set is free once stage c - NSET has been drained
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return drained >= c - bg_host_pipe::NSET + 1; });
        }
        t_wait_set += now() - t0;
        t0 = now();
        bg_host_pipe::Set& S = P.set[c % bg_host_pipe::NSET];
        const uint64_t p0 = cut[c], np = cut[c + 1] - p0;
        const uint64_t xb = x_off[p0 + np] - x_off[p0], yb = y_off[p0 + np] - y_off[p0];
        parallel_memcpy(S.h_in, x + x_off[p0], xb);
        parallel_memcpy(S.h_in + o_y, y + y_off[p0], yb);
        uint64_t *hxo = (uint64_t*)(S.h_in + o_xo), *hyo = (uint64_t*)(S.h_in + o_yo);
        std::atomic<bool> ragged{false};  // the host knows the lengths: no reduction + synchronisation on the device
        const uint64_t len_x = np ? x_off[p0 + 1] - x_off[p0] : 0, len_y = np ? y_off[p0 + 1] - y_off[p0] : 0;
        parallel_for(np + 1, 16384, [&](uint64_t a, uint64_t b) {
            bool rag = false;
            for (uint64_t p = a; p < b; p++) {
                hxo[p] = x_off[p0 + p] - x_off[p0];
                hyo[p] = y_off[p0 + p] - y_off[p0];
                if (p < np) rag = rag || x_off[p0 + p + 1] - x_off[p0 + p] != len_x || y_off[p0 + p + 1] - y_off[p0 + p] != len_y;
            }
            if (rag) ragged = true;
        });
        t_pack += now() - t0;
        t0 = now();
        bool in_ok = true;
        if (xb) in_ok = in_ok && bg_copy_pieces(S.d_in, S.h_in, xb, hipMemcpyHostToDevice, P.s_in) == hipSuccess;
        if (yb) in_ok = in_ok && bg_copy_pieces(S.d_in + o_y, S.h_in + o_y, yb, hipMemcpyHostToDevice, P.s_in) == hipSuccess;
        in_ok = in_ok && hipMemcpyAsync(S.d_in + o_xo, S.h_in + o_xo, 2 * (chunk + 1) * 8, hipMemcpyHostToDevice, P.s_in) == hipSuccess;
        in_ok = in_ok && hipEventRecord(S.in_done, P.s_in) == hipSuccess && hipStreamWaitEvent(s_k, S.in_done, 0) == hipSuccess;
        const bool uniform = !ragged;
        bg_alignment_t* d_rec = (bg_alignment_t*)S.d_out;
        rc = !in_ok ? BG_ERR_HIP : align_batch_dev_impl(ctx, sc, mode, np, S.d_in, (const uint64_t*)(S.d_in + o_xo), S.d_in + o_y, (const uint64_t*)(S.d_in + o_yo),
                                  max_x, max_y, d_rec, stride ? S.d_out + o_ops : nullptr, stride, s_k, uniform ? 1 : 0);
        if (rc == BG_OK && stride)  // compact the stage's operations on the device, final ops_off into the records
            rc = bg_compact_ops_dev(d_rec, np, S.d_out + o_ops, S.d_cmp, false, P.d_cell, (uint64_t*)(S.d_out + o_tot), S.d_scan, false, s_k);
        if (rc == BG_OK && (hipEventRecord(S.k_done, s_k) != hipSuccess || hipStreamWaitEvent(P.s_out, S.k_done, 0) != hipSuccess))
            rc = BG_ERR_HIP;
        if (rc == BG_OK) {
            static const int d2h_blocks = getenv("BG_D2H_BLOCKS") ? std::max(1, atoi(getenv("BG_D2H_BLOCKS"))) : 8;
            static const bool d2h_nt = !(getenv("BG_D2H_NT") && atoi(getenv("BG_D2H_NT")) == 0);
            (d2h_nt ? stage_to_host_kernel<true> : stage_to_host_kernel<false>)<<<dim3(d2h_blocks), dim3(256), 0, P.s_out>>>((const u32x4*)S.d_out, (u32x4*)S.h_out, np * sizeof(bg_alignment_t) / 16,
                                                                     (const u32x4*)S.d_cmp, (u32x4*)(S.h_out + o_ops),
                                                                     stride ? (const uint64_t*)(S.d_out + o_tot) : nullptr, (uint64_t*)(S.h_out + o_tot));
            if (hipGetLastError() != hipSuccess || hipEventRecord(S.out_done, P.s_out) != hipSuccess) rc = BG_ERR_HIP;
        }
        if (rc) {
            hipDeviceSynchronize();
            stop_drainer();
            return rc;
        }
        t_launch += now() - t0;
        {
            std::lock_guard<std::mutex> lk(mu);
            submitted = c + 1;
        }
        cv.notify_all();
    }
    double t0j = now();
    drainer.join();
    if (trace)
        fprintf(stderr, "[bg host] %llu stages: pack %.2f launch %.2f wait-for-set %.2f join %.2f | drainer: wait-stage %.2f copy-records %.2f copy-ops %.2f ms\n",
                (unsigned long long)nch, t_pack, t_launch, t_wait_set, now() - t0j, t_d_wait_rec, t_d_rec, t_d_ops);
    if (drain_rc) {
        bg_tls_error = drain_err;
        return drain_rc;
    }
    if (ops_used) *ops_used = used;
    return status;
}
}  // namespace

extern "C" int bg_align_batch(bg_ctx* ctx, const bg_scoring_t* sc, int mode, uint64_t n_pairs,
                              const uint8_t* x, const uint64_t* x_off, const uint8_t* y,
                              const uint64_t* y_off, bg_alignment_t* out, uint8_t* ops_buf,
                              uint64_t ops_cap, uint64_t* ops_used) {
    if (!ctx || !sc) return BG_ERR_INVALID_ARG;
    int rc = check_scoring(sc);
    if (rc) return rc;
    if (ops_used) *ops_used = 0;
    if (n_pairs == 0) return BG_OK;
    if (!x_off || !y_off || !out) return BG_ERR_INVALID_ARG;
    BG_HIP(hipSetDevice(ctx->device));
    uint64_t max_x = 0, max_y = 0, max_sum = 0;
    bool uniform_len = true;
    {
        std::mutex mu;
        bool bad = false;
        parallel_for(n_pairs, 1 << 16, [&](uint64_t a, uint64_t b) {
            uint64_t mx = 0, my = 0, ms = 0;
            bool uni = true, neg = false;
            for (uint64_t p = a; p < b; p++) {
                neg = neg || x_off[p + 1] < x_off[p] || y_off[p + 1] < y_off[p];
                const uint64_t lx = x_off[p + 1] - x_off[p], ly = y_off[p + 1] - y_off[p];
                mx = std::max(mx, lx);
                my = std::max(my, ly);
                ms = std::max(ms, lx + ly);
                uni = uni && lx == x_off[1] - x_off[0] && ly == y_off[1] - y_off[0];
            }
            std::lock_guard<std::mutex> lk(mu);
            max_x = std::max(max_x, mx);
            max_y = std::max(max_y, my);
            max_sum = std::max(max_sum, ms);
            uniform_len = uniform_len && uni;
            bad = bad || neg;
        });
        if (bad) return BG_ERR_INVALID_ARG;
    }
    if (max_x > (1u << 24) || max_y > (1u << 24)) return BG_ERR_TOO_LARGE;
    {  // large batches flow through the staged pipeline
        const uint64_t chunk = ctx->host_chunk_pairs > 0 ? (uint64_t)ctx->host_chunk_pairs : 122880;  // five rounds of K1p blocks over the 256 CUs
        if (n_pairs >= 2 * chunk && !sc->matrix)
            return align_batch_pipelined(ctx, sc, mode, n_pairs, x, x_off, y, y_off, (uint32_t)max_x, (uint32_t)max_y, chunk, out, ops_buf, ops_cap,
                                         ops_used);
    }
    const uint64_t xb = x_off[n_pairs], yb = y_off[n_pairs];
    const uint64_t stride = ops_buf ? max_x + max_y + 4 : 0;
    (void)max_sum;
    // device copies of the batch and the pinned landing zone of the operations persist in the ctx
    const size_t need[6] = {std::max<uint64_t>(xb, 16), std::max<uint64_t>(yb, 16), (n_pairs + 1) * 8, (n_pairs + 1) * 8,
                            n_pairs * sizeof(bg_alignment_t), std::max<uint64_t>(n_pairs * stride, 16)};
    for (int i = 0; i < 6; i++)
        if ((rc = bg_reserve(&ctx->io[i], &ctx->io_cap[i], need[i]))) return rc;
    uint8_t *d_x = (uint8_t*)ctx->io[0], *d_y = (uint8_t*)ctx->io[1], *d_ops = stride ? (uint8_t*)ctx->io[5] : nullptr;
    uint64_t *d_xo = (uint64_t*)ctx->io[2], *d_yo = (uint64_t*)ctx->io[3];
    bg_alignment_t* d_out = (bg_alignment_t*)ctx->io[4];
    if (stride && n_pairs * stride > ctx->h_ops_cap) {
        if (ctx->h_ops) hipHostFree(ctx->h_ops);
        ctx->h_ops = nullptr;
        ctx->h_ops_cap = 0;
        BG_HIP(hipHostMalloc(&ctx->h_ops, n_pairs * stride, hipHostMallocDefault));
        ctx->h_ops_cap = n_pairs * stride;
    }
    hipStream_t st = ctx->stream;
    if (xb) BG_HIP(hipMemcpyAsync(d_x, x, xb, hipMemcpyHostToDevice, st));
    if (yb) BG_HIP(hipMemcpyAsync(d_y, y, yb, hipMemcpyHostToDevice, st));
    BG_HIP(hipMemcpyAsync(d_xo, x_off, (n_pairs + 1) * 8, hipMemcpyHostToDevice, st));
    BG_HIP(hipMemcpyAsync(d_yo, y_off, (n_pairs + 1) * 8, hipMemcpyHostToDevice, st));
    rc = align_batch_dev_impl(ctx, sc, mode, n_pairs, d_x, d_xo, d_y, d_yo, (uint32_t)max_x, (uint32_t)max_y, d_out, d_ops, stride, st,
                              uniform_len ? 1 : 0);
    if (rc) return rc;
    BG_HIP(hipMemcpyAsync(out, d_out, n_pairs * sizeof(bg_alignment_t), hipMemcpyDeviceToHost, st));
    if (stride) BG_HIP(hipMemcpyAsync(ctx->h_ops, d_ops, n_pairs * stride, hipMemcpyDeviceToHost, st));
    BG_HIP(hipStreamSynchronize(st));
    // compact the strided ops into the caller's buffer: offsets serially, bytes on all host threads
    const uint8_t* h_ops = (const uint8_t*)ctx->h_ops;
    uint64_t used = 0;
    int status = BG_OK;
    std::vector<uint64_t> src(n_pairs);
    for (uint64_t p = 0; p < n_pairs; p++) {
        if (out[p].status) status = out[p].status;
        src[p] = out[p].ops_off;
        out[p].ops_off = used;
        if (ops_buf && used + out[p].n_ops > ops_cap && status == BG_OK) status = BG_ERR_OPS_CAP;
        used += out[p].n_ops;
    }
    if (ops_buf) {
        unsigned nt = std::max(1u, std::min(bg_host_threads(), (unsigned)(n_pairs / 4096 + 1)));
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&, t] {
                for (uint64_t p = n_pairs * t / nt, pe = n_pairs * (t + 1) / nt; p < pe; p++)
                    if (out[p].ops_off + out[p].n_ops <= ops_cap) memcpy(ops_buf + out[p].ops_off, h_ops + src[p], out[p].n_ops);
            });
        for (auto& t : th) t.join();
    }
    if (ops_used) *ops_used = used;
    return status;
}
