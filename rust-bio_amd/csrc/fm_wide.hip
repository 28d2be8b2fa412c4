// FM index with 64-bit text positions (round 5): what lifts the engine's 2^32 - 2 symbol limit.
//
// The reference indexes with usize throughout — Interval { lower, upper } (/root/reference/src/data_structures/fmindex.rs:
// 70-71), Occ's counters (bwt.rs:94-125), less (bwt.rs:186-199), suffix-array entries (suffix_array.rs:264) — so a text of
// 4.3 G symbols and more is nothing special there; here the rank blocks, less[], l / r and the suffix-array samples were
// uint32 in five kernels and two builders (rounds 1-4: BG_ERR_TOO_LARGE).  This file is the same search on the layout
// fm_kernels.h describes under "64-bit positions": the 64-byte blocks of K5 with counters relative to a superblock, one
// absolute 64-bit base per code and superblock, everything a position can reach in 64 bits.
//   fm_wide_build_dev   the index from a BWT in HBM (bg_fm_build_dev, and bg_fm_build after an upload): the block kernels
//                       of fm_index.hip's device builder, a 64-bit scan of the per-block counts, heads relative to the
//                       superblock's first block
//   fmw_search_kernel   FMIndexable::backward_search (fmindex.rs:144-208) for byte patterns: a quad per query like K5's
//                       generic kernel, l and r 64-bit, Occ::get = base[superblock][code] + cnt[code] + popcount
//   fmw_sampled_get_kernel / fmw_raw_get_kernel   Interval::occ over a SampledSuffixArray / a raw one (suffix_array.rs:
//                       134-184) with 64-bit samples
// Narrow indexes (n < 2^32 - 1) never come here: their kernels, layouts and speed are those of rounds 1-4.  What is NOT
// offered on a wide index (BG_ERR_UNSUPPORTED, stated in biogpu.h): the 2-bit packed pattern entry points and the 2-step
// rank blocks (speed, not function: the byte entry points answer the same queries), seed-and-extend, the FMD-index
// kernels (their interval records are uint32 in the C ABI), and alphabets that need rank bit vectors.
#include <algorithm>
#include <cstring>
#include <numeric>
#include <string.h>
#include <vector>

#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "fm_kernels.h"

using namespace bgfm;

uint64_t fm_wide_threshold(const bg_ctx* ctx) { return ctx ? ctx->fm_wide_from : 0xFFFFFFFFull; }

namespace {

constexpr uint32_t kWideMaxExc = kMaxExcLds;

__global__ __launch_bounds__(256) void fmw_hist_kernel(const uint8_t* __restrict__ b, uint64_t n, unsigned long long* __restrict__ hist) {
    __shared__ uint32_t s[256];
    s[threadIdx.x] = 0;
    __syncthreads();
    // (a block's share of a 2^40-symbol text stays below 2^32: at most 2^40 / 8192 blocks' worth per block)
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) atomicAdd(&s[b[i]], 1u);
    __syncthreads();
    if (s[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)s[threadIdx.x]);
}

// one thread per block: 192 symbols -> 12 words of 2-bit codes + how many of each code the block holds
__global__ __launch_bounds__(256) void fmw_blocks_kernel(const uint8_t* __restrict__ b, uint64_t n, uint64_t nblk, const uint8_t* __restrict__ code_of,
                                                         uint32_t* __restrict__ blocks, uint32_t* __restrict__ cnt /* [4][nblk] */) {
    __shared__ uint8_t s_code[256];
    s_code[threadIdx.x] = code_of[threadIdx.x];
    __syncthreads();
    const uint64_t blk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blk >= nblk) return;
    const uint64_t lo = blk * kSymPerBlock;
    uint32_t c[4] = {0, 0, 0, 0};
    for (uint32_t w = 0; w < 12; w++) {
        uint32_t word = 0;
        for (uint32_t t = 0; t < 16; t++) {
            const uint64_t i = lo + 16 * w + t;
            if (i < n) {
                const uint32_t code = s_code[b[i]];
                word |= code << (2 * t);
                c[code]++;
            }
        }
        blocks[blk * 16 + 4 + w] = word;
    }
    for (int k = 0; k < 4; k++) cnt[(uint64_t)k * nblk + blk] = c[k];
}
// absolute counts (64-bit exclusive scan of cnt) -> the superblock's base and the block's counter relative to it
__global__ __launch_bounds__(256) void fmw_heads_kernel(uint64_t nblk, uint32_t sb_shift, const uint64_t* __restrict__ scanned /* [4][nblk] */,
                                                        uint32_t* __restrict__ blocks, uint64_t* __restrict__ sb) {
    const uint64_t blk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blk >= nblk) return;
    const uint64_t first = (blk >> sb_shift) << sb_shift;
    for (int k = 0; k < 4; k++) {
        const uint64_t base = scanned[(uint64_t)k * nblk + first];
        blocks[blk * 16 + k] = (uint32_t)(scanned[(uint64_t)k * nblk + blk] - base);
        if (blk == first) sb[(blk >> sb_shift) * 4 + k] = base;
    }
}
// sparse exceptions: (position, byte) appended in any order; the host sorts the few of them
__global__ __launch_bounds__(256) void fmw_sparse_kernel(const uint8_t* __restrict__ b, uint64_t n, const uint8_t* __restrict__ is_sparse,
                                                         uint32_t cap, uint32_t* __restrict__ n_out, ulonglong2* __restrict__ out) {
    // (grid-stride: a launch may not exceed 2^32 threads)
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint8_t ch = b[i];
        if (is_sparse[ch]) {
            const uint32_t k = atomicAdd(n_out, 1u);
            if (k < cap) out[k] = make_ulonglong2(i, ch);
        }
    }
}

struct U32ToU64 {
    __host__ __device__ uint64_t operator()(uint32_t v) const { return (uint64_t)v; }
};

// Occ::get(r, code) for the quad: the block's line is in `v` (lane t holds bytes [16t, 16t + 16)), `base` the superblock's
// absolute count of the code
__device__ __forceinline__ uint64_t wide_rank(const FmWideDev& fm, const uint4 v, uint32_t t, uint64_t blk, uint32_t o, uint32_t code) {
    return fm.sb[(blk >> fm.sb_shift) * 4 + code] + (uint64_t)quad_sum(block_part(v, t, o, code));
}

// K5 on 64-bit positions: FMIndexable::backward_search (fmindex.rs:144-208).  A quad of four lanes per query; both ranks of a
// step are issued together, one line when they fall into the same block.
__global__ __launch_bounds__(256) void fmw_search_kernel(FmWideDev fm, uint64_t n_q, const uint8_t* __restrict__ pat,
                                                         const uint64_t* __restrict__ pat_off, uint8_t* __restrict__ tag,
                                                         uint64_t* __restrict__ lower, uint64_t* __restrict__ upper,
                                                         uint32_t* __restrict__ matched_len) {
    __shared__ uint16_t s_class[256];
    __shared__ uint64_t s_less[256];
    __shared__ uint64_t s_exc[kWideMaxExc];
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) {
        s_class[i] = fm.sym_class[i];
        s_less[i] = fm.less[i];
    }
    for (uint32_t i = threadIdx.x; i < fm.n_exc; i += blockDim.x) s_exc[i] = fm.exc_pos[i];
    __syncthreads();
    const uint32_t t = threadIdx.x & 3;
    const uint64_t n_quads = (uint64_t)gridDim.x * (blockDim.x >> 2);
    uint64_t q = (uint64_t)blockIdx.x * (blockDim.x >> 2) + (threadIdx.x >> 2);
    bool active = false;
    uint64_t off = 0, l = 0, r = 0;
    uint32_t len = 0, pos = 0, matched = 0, a_next = 0;
    auto emit = [&](uint32_t tg, uint64_t lo, uint64_t hi, uint32_t ml) {
        if (t == 0) {
            tag[q] = (uint8_t)tg;
            lower[q] = lo;
            upper[q] = hi;
            matched_len[q] = ml;
        }
    };
    auto fetch = [&]() {  // the next non-empty query; empty patterns are Absent at once (fmindex.rs:185-207)
        active = false;
        while (q < n_q) {
            off = pat_off[q];
            len = (uint32_t)(pat_off[q + 1] - off);
            if (len) {
                pos = len;
                l = 0;
                r = fm.n - 1;  // fmindex.rs:148
                matched = 0;
                a_next = pat[off + pos - 1];
                active = true;
                return;
            }
            emit(BG_FM_ABSENT, 0, 0, 0);
            q += n_quads;
        }
    };
    fetch();
    while (__any(active)) {
        if (active) {
            // one iteration of the loop at fmindex.rs:160-182
            const uint32_t a = a_next;
            pos -= 1;
            if (pos) a_next = pat[off + pos - 1];  // (address independent of the ranks)
            const uint32_t cls = s_class[a];
            const uint64_t less_a = s_less[a];
            uint64_t occ_r = 0, occ_l = 0;
            bool stop = false;
            uint32_t stop_tag = BG_FM_PARTIAL;
            if (cls == kClsPanic) {
                stop = true;
                stop_tag = BG_FM_PANIC;
            } else if (cls < 4) {
                const uint64_t br = r / kSymPerBlock;
                const uint32_t orr = (uint32_t)(r - br * kSymPerBlock);
                const uint4 vr = fm.blocks[br * 4 + t];
                uint4 vl = vr;
                uint64_t bl = br;
                uint32_t ol = 0;
                if (l > 0) {
                    bl = (l - 1) / kSymPerBlock;
                    ol = (uint32_t)((l - 1) - bl * kSymPerBlock);
                    if (bl != br) vl = fm.blocks[bl * 4 + t];
                }
                occ_r = wide_rank(fm, vr, t, br, orr, cls);
                if (l > 0) occ_l = wide_rank(fm, vl, t, bl, ol, cls);
                if (cls == 0 && fm.n_exc) {  // sparse exceptions sit in the stream as code 0
                    occ_r -= count_le64(s_exc, 0u, fm.n_exc, r);
                    if (l > 0) occ_l -= count_le64(s_exc, 0u, fm.n_exc, l - 1);
                }
            } else if (cls >= kClsSparse) {  // (no dense symbols on a wide index)
                const uint32_t e = cls - kClsSparse;
                const uint32_t lo = fm.sparse_off[e], hi = fm.sparse_off[e + 1];
                occ_r = count_le64(fm.exc_sym_pos, lo, hi, r) - lo;
                if (l > 0) occ_l = count_le64(fm.exc_sym_pos, lo, hi, l - 1) - lo;
            }  // kClsZero: both stay 0
            const uint64_t pl = l, pr = r;
            if (!stop) {
                if (occ_r == 0) {  // fmindex.rs:167-170
                    stop = true;
                } else {
                    l = less_a + occ_l;  // fmindex.rs:171
                    r = less_a + occ_r - 1;
                    if (l > r)  // fmindex.rs:177-180
                        stop = true;
                    else
                        matched += 1;
                }
            }
            if (stop) {
                if (stop_tag == BG_FM_PANIC)
                    emit(BG_FM_PANIC, 0, 0, matched);
                else if (matched)
                    emit(BG_FM_PARTIAL, pl, pr + 1, matched);
                else
                    emit(BG_FM_ABSENT, 0, 0, 0);
                q += n_quads;
                fetch();
            } else if (pos == 0) {
                emit(BG_FM_COMPLETE, l, r + 1, matched);
                q += n_quads;
                fetch();
            }
        }
    }
}

// The same search with TWO queries per quad (round 5; what fm_search_fast2x_kernel is to the narrow index, fm_index.hip):
// a step is one dependent block access, eight wavefronts per SIMD are all the hardware holds, and the parallelism left to
// add is a second independent query inside the wavefront.  Phase A reads both streams' symbols and classes and issues
// their block loads and superblock bases — unconditional, in one basic block (a stream without a coded symbol reads
// block 0; the line of l - 1 is requested even where it is the line of r) — phase B ranks and updates both.  Stream
// (quad, u) takes queries (2 quad + u) + k * 2 quads.  bg_fm_set_option("ilp", 1): the kernel above.  On the 4.4 G-symbol
// index: 302 -> 404 M queries/s (profiles/r05_fm_wide_4g4.json), same arrays.
// Round 6: the flavours the narrow generic kernel has — SEEDS (the seed windows of a batch of reads: query q is seed q % S of
// read q / S, fm_kernels.h SeedSrc), PACKED (`pat` is a 2-bit stream in the index's codes, offsets in symbols: the class of a
// symbol is its code) and DEFER (only the queries the 2x fast kernel in front left tagged kTagDeferred).
template <bool SEEDS, bool PACKED, bool DEFER>
__global__ __launch_bounds__(256) void fmw_search2x_kernel(FmWideDev fm, uint64_t n_q, const uint8_t* __restrict__ pat,
                                                           const uint64_t* __restrict__ pat_off, uint8_t* __restrict__ tag,
                                                           uint64_t* __restrict__ lower, uint64_t* __restrict__ upper,
                                                           uint32_t* __restrict__ matched_len, const SeedSrc seeds) {
    constexpr int U = 2;
    __shared__ uint16_t s_class[256];
    __shared__ uint64_t s_less[256];
    __shared__ uint64_t s_exc[kWideMaxExc];
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) {
        s_class[i] = fm.sym_class[i];
        // PACKED: entry c (< 4) is less[] of the byte that code c stands for
        s_less[i] = (PACKED && i < 4) ? fm.less[(seeds.code_bytes >> (8 * i)) & 0xFFu] : fm.less[i];
    }
    for (uint32_t i = threadIdx.x; i < fm.n_exc; i += blockDim.x) s_exc[i] = fm.exc_pos[i];
    __syncthreads();
    const uint32_t* __restrict__ pk = (const uint32_t*)pat;
    const uint32_t t = threadIdx.x & 3;
    const uint64_t n_streams = (uint64_t)gridDim.x * (blockDim.x >> 2) * U;
    struct St {
        uint64_t q, off, l, r;
        uint32_t pos, matched, a_next;
        bool active;
    };
    St S[U];
    auto emit = [&](uint64_t q, uint32_t tg, uint64_t lo, uint64_t hi, uint32_t ml) {
        if (t == 0) {
            tag[q] = (uint8_t)tg;
            lower[q] = lo;
            upper[q] = hi;
            matched_len[q] = ml;
        }
    };
    auto symbol = [&](uint64_t at) -> uint32_t {  // the pattern symbol at stream position `at`: a byte, or a 2-bit code
        if (PACKED) return (pk[at >> 4] >> (2 * ((uint32_t)at & 15u))) & 3u;
        return pat[at];
    };
    auto fetch = [&](St& s) {  // the stream's next non-empty query; empty patterns are Absent at once (fmindex.rs:185-207)
        s.active = false;
        while (s.q < n_q) {
            if (DEFER && tag[s.q] != kTagDeferred) {
                s.q += n_streams;
                continue;
            }
            uint32_t len;
            if (SEEDS) {
                const uint64_t rd = s.q / seeds.S;
                const uint32_t k = (uint32_t)(s.q - rd * seeds.S);
                const uint64_t o = pat_off[rd];
                s.off = o + (uint64_t)k * seeds.stride;
                len = (uint64_t)k * seeds.stride + seeds.seed_len <= pat_off[rd + 1] - o ? seeds.seed_len : 0u;
            } else {
                s.off = pat_off[s.q];
                len = (uint32_t)(pat_off[s.q + 1] - s.off);
            }
            if (len) {
                s.pos = len;
                s.l = 0;
                s.r = fm.n - 1;  // fmindex.rs:148
                s.matched = 0;
                s.a_next = symbol(s.off + len - 1);
                s.active = true;
                return;
            }
            emit(s.q, BG_FM_ABSENT, 0, 0, 0);
            s.q += n_streams;
        }
    };
#pragma unroll
    for (int u = 0; u < U; u++) {
        S[u].q = ((uint64_t)blockIdx.x * (blockDim.x >> 2) + (threadIdx.x >> 2)) * U + u;
        S[u].off = S[u].l = S[u].r = 0;
        S[u].pos = S[u].matched = S[u].a_next = 0;
        fetch(S[u]);
    }
    for (;;) {
        bool any_active = false;
#pragma unroll
        for (int u = 0; u < U; u++) any_active |= S[u].active;
        if (!__any(any_active)) break;
        // ---- phase A
        uint4 vr[U], vl[U];
        uint64_t base_r[U], base_l[U], br[U], bl[U];
        uint32_t orr[U], ol[U], a[U], cls[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            St& s = S[u];
            a[u] = s.a_next;
            const uint32_t p1 = s.active ? s.pos - 1u : 0u;
            if (s.active && p1) s.a_next = symbol(s.off + p1 - 1);  // (address independent of the ranks)
            cls[u] = PACKED ? a[u] : (uint32_t)s_class[a[u]];
            const bool coded = s.active && cls[u] < 4;
            const uint64_t r_ = coded ? s.r : 0, l_ = coded && s.l ? s.l - 1 : 0;
            br[u] = r_ / kSymPerBlock;
            bl[u] = coded && s.l ? l_ / kSymPerBlock : br[u];
            orr[u] = (uint32_t)(r_ - br[u] * kSymPerBlock);
            ol[u] = (uint32_t)(l_ - bl[u] * kSymPerBlock);
            const uint32_t code = coded ? cls[u] : 0u;
            vr[u] = fm.blocks[br[u] * 4 + t];
            vl[u] = fm.blocks[bl[u] * 4 + t];
            base_r[u] = fm.sb[(br[u] >> fm.sb_shift) * 4 + code];
            base_l[u] = fm.sb[(bl[u] >> fm.sb_shift) * 4 + code];
        }
        // ---- phase B: one iteration of the loop at fmindex.rs:160-182 per stream
#pragma unroll
        for (int u = 0; u < U; u++) {
            St& s = S[u];
            if (!s.active) continue;
            s.pos -= 1;
            const uint64_t less_a = s_less[a[u]];
            uint64_t occ_r = 0, occ_l = 0;
            bool stop = false;
            uint32_t stop_tag = BG_FM_PARTIAL;
            if (!PACKED && cls[u] == kClsPanic) {
                stop = true;
                stop_tag = BG_FM_PANIC;
            } else if (PACKED || cls[u] < 4) {
                occ_r = base_r[u] + (uint64_t)quad_sum(block_part(vr[u], t, orr[u], cls[u]));
                if (s.l > 0) occ_l = base_l[u] + (uint64_t)quad_sum(block_part(vl[u], t, ol[u], cls[u]));
                if (cls[u] == 0 && fm.n_exc) {  // sparse exceptions sit in the stream as code 0
                    occ_r -= count_le64(s_exc, 0u, fm.n_exc, s.r);
                    if (s.l > 0) occ_l -= count_le64(s_exc, 0u, fm.n_exc, s.l - 1);
                }
            } else if (cls[u] >= kClsSparse) {  // (no dense symbols on a wide index)
                const uint32_t e = cls[u] - kClsSparse;
                const uint32_t lo = fm.sparse_off[e], hi = fm.sparse_off[e + 1];
                occ_r = count_le64(fm.exc_sym_pos, lo, hi, s.r) - lo;
                if (s.l > 0) occ_l = count_le64(fm.exc_sym_pos, lo, hi, s.l - 1) - lo;
            }  // kClsZero: both stay 0
            const uint64_t pl = s.l, pr = s.r;
            if (!stop) {
                if (occ_r == 0) {  // fmindex.rs:167-170
                    stop = true;
                } else {
                    s.l = less_a + occ_l;  // fmindex.rs:171
                    s.r = less_a + occ_r - 1;
                    if (s.l > s.r)  // fmindex.rs:177-180
                        stop = true;
                    else
                        s.matched += 1;
                }
            }
            if (stop) {
                if (stop_tag == BG_FM_PANIC)
                    emit(s.q, BG_FM_PANIC, 0, 0, s.matched);
                else if (s.matched)
                    emit(s.q, BG_FM_PARTIAL, pl, pr + 1, s.matched);
                else
                    emit(s.q, BG_FM_ABSENT, 0, 0, 0);
                s.q += n_streams;
                fetch(s);
            } else if (s.pos == 0) {
                emit(s.q, BG_FM_COMPLETE, s.l, s.r + 1, s.matched);
                s.q += n_streams;
                fetch(s);
            }
        }
    }
}

struct SaWideDev {
    const uint64_t* sa;         // raw SA or the samples
    const uint64_t* extra_row;  // sorted
    const uint64_t* extra_pos;
    const uint8_t* exc_byte;    // byte of exception e (parallel to FmWideDev::exc_pos)
    uint32_t n_extra;
    uint32_t rate;
    uint32_t sentinel;
    uint32_t code_byte;         // byte of code c in bits [8c, 8c+8)
};

__global__ __launch_bounds__(256) void fmw_raw_get_kernel(const uint64_t* __restrict__ sa, uint64_t n_text, uint64_t n, const uint64_t* index,
                                                          uint64_t* pos_out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t r = index[i];
    pos_out[i] = r < n_text ? sa[r] : BG_SA_NONE;
}

// SampledSuffixArray::get (suffix_array.rs:157-184): every row LF-walks until it reaches a sampled row or a row whose BWT
// byte is the sentinel; a quad per row, the rank block and the word that holds bwt[pos] loaded together
__global__ __launch_bounds__(256) void fmw_sampled_get_kernel(FmWideDev fm, SaWideDev sa, uint64_t n, const uint64_t* index, uint64_t* pos_out) {
    __shared__ uint16_t s_class[256];
    __shared__ uint64_t s_less[256];
    __shared__ uint64_t s_exc[kWideMaxExc];
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) {
        s_class[i] = fm.sym_class[i];
        s_less[i] = fm.less[i];
    }
    for (uint32_t i = threadIdx.x; i < fm.n_exc; i += blockDim.x) s_exc[i] = fm.exc_pos[i];
    __syncthreads();
    const uint32_t t = threadIdx.x & 3;
    const uint64_t n_quads = (uint64_t)gridDim.x * (blockDim.x >> 2);
    uint64_t q = (uint64_t)blockIdx.x * (blockDim.x >> 2) + (threadIdx.x >> 2);
    const uint32_t* blocks32 = (const uint32_t*)fm.blocks;
    bool active = false;
    uint64_t pos = 0, offset = 0;
    auto emit = [&](uint64_t v) {
        if (t == 0) pos_out[q] = v;
    };
    auto fetch = [&]() {
        active = false;
        while (q < n) {
            const uint64_t r = index[q];
            if (r < fm.n) {
                pos = r;
                offset = 0;
                active = true;
                return;
            }
            emit(BG_SA_NONE);  // SuffixArray::get -> None
            q += n_quads;
        }
    };
    fetch();
    while (__any(active)) {
        if (active) {
            if (pos % sa.rate == 0) {  // suffix_array.rs:162-164
                emit(sa.sa[pos / sa.rate] + offset);
                q += n_quads;
                fetch();
                continue;
            }
            const uint64_t pb = pos / kSymPerBlock, rb = (pos - 1) / kSymPerBlock;
            const uint32_t po = (uint32_t)(pos - pb * kSymPerBlock), ro = (uint32_t)((pos - 1) - rb * kSymPerBlock);
            const uint4 vr = fm.blocks[rb * 4 + t];
            const uint32_t word = blocks32[pb * 16 + 4 + (po >> 4)];
            const uint32_t code = (word >> (2 * (po & 15))) & 3u;
            uint32_t c = (sa.code_byte >> (8 * code)) & 255u;
            if (code == 0 && fm.n_exc) {  // sparse exceptions sit in the stream as code 0
                const uint32_t e = count_le64(s_exc, 0u, fm.n_exc, pos);
                if (e > 0 && s_exc[e - 1] == pos) c = sa.exc_byte[e - 1];
            }
            if (c == sa.sentinel) {  // suffix_array.rs:168-175
                uint32_t lo = 0, hi = sa.n_extra;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (sa.extra_row[mid] < pos)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                emit(lo < sa.n_extra && sa.extra_row[lo] == pos ? sa.extra_pos[lo] + offset : BG_SA_PANIC);
                q += n_quads;
                fetch();
                continue;
            }
            // pos = less[c] + occ.get(bwt, pos - 1, c)  (suffix_array.rs:177-178)
            const uint32_t cls = s_class[c];
            uint64_t occ = 0;
            if (cls < 4) {
                occ = wide_rank(fm, vr, t, rb, ro, cls);
                if (cls == 0 && fm.n_exc) occ -= count_le64(s_exc, 0u, fm.n_exc, pos - 1);
            } else if (cls != kClsPanic && cls >= kClsSparse) {
                const uint32_t e = cls - kClsSparse;
                const uint32_t lo = fm.sparse_off[e], hi = fm.sparse_off[e + 1];
                occ = count_le64(fm.exc_sym_pos, lo, hi, pos - 1) - lo;
            }
            pos = s_less[c] + occ;
            offset += 1;
        }
    }
}

}  // namespace

int fm_wide_build_dev(bg_ctx* ctx, const uint8_t* d_bwt, uint64_t n, const uint8_t* alphabet, uint32_t n_sym, const uint64_t* less_in,
                      uint32_t less_len_in, uint64_t* less_out, bg_fm** out, hipStream_t st) {
    if (n > (1ull << 40)) return BG_ERR_TOO_LARGE;
    BG_HIP(hipSetDevice(ctx->device));
    bool in_alpha[256] = {};
    uint32_t max_symbol = 0;
    for (uint32_t i = 0; i < n_sym; i++) {
        in_alpha[alphabet[i]] = true;
        max_symbol = std::max<uint32_t>(max_symbol, alphabet[i]);
    }
    const uint32_t m = max_symbol + 1;
    if ((uint32_t)'$' < m) in_alpha['$'] = true;  // bwt.rs:101-104: '$' is always tabulated
    const uint32_t less_len = max_symbol + 2;
    if (less_in && less_len_in != less_len) return BG_ERR_INVALID_ARG;

    std::vector<void*> tmp;  // device temporaries, freed on every exit
    auto dalloc = [&](void** p, size_t bytes) -> int {
        BG_HIP(hipMalloc(p, std::max<size_t>(bytes, 16)));
        tmp.push_back(*p);
        return BG_OK;
    };
    bg_fm* fm = nullptr;
    auto body = [&]() -> int {
        int rc;
        unsigned long long* d_hist = nullptr;
        if ((rc = dalloc((void**)&d_hist, 256 * 8))) return rc;
        BG_HIP(hipMemsetAsync(d_hist, 0, 256 * 8, st));
        fmw_hist_kernel<<<dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 8192)), dim3(256), 0, st>>>(d_bwt, n, d_hist);
        uint64_t hist[256];
        BG_HIP(hipMemcpyAsync(hist, d_hist, sizeof(hist), hipMemcpyDeviceToHost, st));
        BG_HIP(hipStreamSynchronize(st));
        for (uint32_t c = m; c < 256; c++)
            if (hist[c]) return BG_ERR_OUT_OF_ALPHABET;  // Occ::new: curr_occ[c] out of bounds
        // less(bwt, alphabet) (bwt.rs:186-199) falls out of the histogram; a caller's own is taken as it is
        uint64_t less[256] = {};
        {
            uint64_t acc = 0;
            for (uint32_t c = 0; c < less_len && c < 256; c++) {
                less[c] = less_in ? less_in[c] : acc;
                acc += hist[c];
            }
            if (less_out)
                for (uint32_t c = 0; c < less_len; c++) less_out[c] = c < 256 ? less[c] : acc;
        }
        // classes: the four most frequent bytes get the codes (ties: smaller byte first), every other byte that occurs is a
        // sparse exception — at most kWideMaxExc positions in all, or the text is not DNA-like
        int order[256];
        std::iota(order, order + 256, 0);
        std::stable_sort(order, order + 256, [&](int a, int b) { return hist[a] > hist[b]; });
        uint64_t beyond4 = 0;
        for (int i = 4; i < 256; i++) beyond4 += hist[order[i]];
        if (beyond4 > kWideMaxExc) return BG_ERR_UNSUPPORTED;  // would need rank bit vectors: not on 64-bit positions
        int code_of[256], sparse_of[256];
        std::fill(code_of, code_of + 256, -1);
        std::fill(sparse_of, sparse_of + 256, -1);
        int n_codes = 0;
        std::vector<int> sparse_syms;
        for (int i = 0; i < 4 && hist[order[i]] > 0; i++) code_of[order[i]] = n_codes++;
        for (int c = 0; c < 256; c++)
            if (hist[c] && code_of[c] < 0) {
                sparse_of[c] = (int)sparse_syms.size();
                sparse_syms.push_back(c);
            }
        uint16_t cls[256];
        uint8_t code_tab[256], sparse_tab[256];
        for (int c = 0; c < 256; c++) {
            cls[c] = !in_alpha[c] ? kClsPanic : code_of[c] >= 0 ? (uint16_t)code_of[c] : hist[c] == 0 ? kClsZero : (uint16_t)(kClsSparse + sparse_of[c]);
            code_tab[c] = code_of[c] >= 0 ? (uint8_t)code_of[c] : 0;
            sparse_tab[c] = sparse_of[c] >= 0 ? 1 : 0;
        }
        fm = new bg_fm;
        fm->ctx = ctx;
        fm->wide = true;
        fm->less_len = less_len;
        fm->fmd_ok = true;  // the BWT is a word over dna::n_alphabet() + '$' (FMDIndex::from, fmindex.rs:323-327)
        for (int c = 0; c < 256; c++)
            if (hist[c] && (c == 0 || !strchr("ACGTNacgtn$", c))) fm->fmd_ok = false;
        for (int c = 0; c < 256; c++)
            if (code_of[c] >= 0) fm->code_byte[code_of[c]] = (uint8_t)c;
        fm->n_codes = n_codes;
        auto keep = [&](void** p, size_t bytes) -> int {  // device memory the handle owns
            const size_t alloc = std::max<size_t>(bytes, 16);
            BG_HIP(hipMalloc(p, alloc));
            fm->bytes += alloc;
            return BG_OK;
        };
        const uint64_t nblk = (n + kSymPerBlock - 1) / kSymPerBlock;
        const uint32_t sb_shift = ctx->fm_wide_sb_shift;
        const uint64_t n_sb = ((nblk - 1) >> sb_shift) + 1;
        uint8_t *d_code = nullptr, *d_sparse = nullptr;
        uint32_t* d_cnt = nullptr;
        uint64_t* d_scan = nullptr;
        void* d_cub = nullptr;
        if ((rc = dalloc((void**)&d_code, 256)) || (rc = dalloc((void**)&d_sparse, 256))) return rc;
        BG_HIP(hipMemcpyAsync(d_code, code_tab, 256, hipMemcpyHostToDevice, st));
        BG_HIP(hipMemcpyAsync(d_sparse, sparse_tab, 256, hipMemcpyHostToDevice, st));
        if ((rc = keep(&fm->d_blocks, nblk * 64))) return rc;
        if ((rc = keep(&fm->d_sb, n_sb * 32))) return rc;
        if ((rc = dalloc((void**)&d_cnt, 4 * nblk * 4))) return rc;
        if ((rc = dalloc((void**)&d_scan, 4 * nblk * 8))) return rc;
        auto in64 = rocprim::make_transform_iterator(d_cnt, U32ToU64());
        size_t cub_bytes = 0;
        BG_HIP(rocprim::exclusive_scan(nullptr, cub_bytes, in64, d_scan, (uint64_t)0, nblk, rocprim::plus<uint64_t>(), st));
        if ((rc = dalloc(&d_cub, cub_bytes))) return rc;
        fmw_blocks_kernel<<<dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, st>>>(d_bwt, n, nblk, d_code, (uint32_t*)fm->d_blocks, d_cnt);
        BG_HIP(hipGetLastError());
        for (int k = 0; k < 4; k++) {
            auto ink = rocprim::make_transform_iterator(d_cnt + (uint64_t)k * nblk, U32ToU64());
            BG_HIP(rocprim::exclusive_scan(d_cub, cub_bytes, ink, d_scan + (uint64_t)k * nblk, (uint64_t)0, nblk, rocprim::plus<uint64_t>(), st));
        }
        fmw_heads_kernel<<<dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, st>>>(nblk, sb_shift, d_scan, (uint32_t*)fm->d_blocks, (uint64_t*)fm->d_sb);
        BG_HIP(hipGetLastError());
        // ---- sparse exceptions
        uint32_t* d_ns = nullptr;
        ulonglong2* d_sp = nullptr;
        if ((rc = dalloc((void**)&d_ns, 4))) return rc;
        if ((rc = dalloc((void**)&d_sp, (size_t)(kWideMaxExc + 8) * 16))) return rc;
        BG_HIP(hipMemsetAsync(d_ns, 0, 4, st));
        fmw_sparse_kernel<<<dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 1u << 22)), dim3(256), 0, st>>>(d_bwt, n, d_sparse, kWideMaxExc + 8, d_ns, d_sp);
        BG_HIP(hipGetLastError());
        uint32_t ns = 0;
        BG_HIP(hipMemcpyAsync(&ns, d_ns, 4, hipMemcpyDeviceToHost, st));
        BG_HIP(hipStreamSynchronize(st));
        if (ns > kWideMaxExc) return BG_ERR_HIP;  // cannot happen: counted above
        std::vector<ulonglong2> sp(ns);
        if (ns) BG_HIP(hipMemcpy(sp.data(), d_sp, (size_t)ns * 16, hipMemcpyDeviceToHost));
        std::sort(sp.begin(), sp.end(), [](const ulonglong2& a, const ulonglong2& b) { return a.x < b.x; });
        std::vector<uint64_t> exc_pos(ns), exc_sym_pos;
        std::vector<uint32_t> sparse_off(sparse_syms.size() + 1, 0);
        std::vector<uint8_t> exc_byte(ns);
        for (uint32_t e = 0; e < ns; e++) {
            exc_pos[e] = sp[e].x;
            exc_byte[e] = (uint8_t)sp[e].y;
        }
        for (size_t e = 0; e < sparse_syms.size(); e++) {
            for (uint32_t k = 0; k < ns; k++)
                if ((int)sp[k].y == sparse_syms[e]) exc_sym_pos.push_back(sp[k].x);
            sparse_off[e + 1] = (uint32_t)exc_sym_pos.size();
        }
        auto upload = [&](void** dptr, const void* src, size_t bytes) -> int {
            int r2 = keep(dptr, bytes);
            if (r2) return r2;
            if (bytes) BG_HIP(hipMemcpy(*dptr, src, bytes, hipMemcpyHostToDevice));
            return BG_OK;
        };
        if ((rc = upload(&fm->d_exc_pos, exc_pos.data(), exc_pos.size() * 8))) return rc;
        if ((rc = upload(&fm->d_exc_sym_pos, exc_sym_pos.data(), exc_sym_pos.size() * 8))) return rc;
        if ((rc = upload(&fm->d_sparse_off, sparse_off.data(), sparse_off.size() * 4))) return rc;
        if ((rc = upload(&fm->d_exc_byte, exc_byte.data(), exc_byte.size()))) return rc;
        if ((rc = upload(&fm->d_class, cls, 256 * sizeof(uint16_t)))) return rc;
        memcpy(fm->h_class, cls, sizeof(fm->h_class));
        if ((rc = upload(&fm->d_less, less, sizeof(less)))) return rc;
        BG_HIP(hipStreamSynchronize(st));
        fm->wdev.blocks = (const uint4*)fm->d_blocks;
        fm->wdev.sb = (const uint64_t*)fm->d_sb;
        fm->wdev.exc_pos = (const uint64_t*)fm->d_exc_pos;
        fm->wdev.exc_sym_pos = (const uint64_t*)fm->d_exc_sym_pos;
        fm->wdev.sparse_off = (const uint32_t*)fm->d_sparse_off;
        fm->wdev.sym_class = (const uint16_t*)fm->d_class;
        fm->wdev.less = (const uint64_t*)fm->d_less;
        fm->wdev.n = n;
        fm->wdev.n_exc = ns;
        fm->wdev.sb_shift = sb_shift;
        fm->n_text = 0;
        {
            // 2-step rank blocks (fm_step2.hip) lean on less[] being the BWT's own cumulative counts; a caller's less that
            // says otherwise keeps single steps (as on the 32-bit layout, fm_index.hip)
            bool consistent = true;
            uint64_t run = 0;
            for (uint32_t c = 0; c < m && c < 256 && consistent; c++) {
                if (hist[c] && less[c] != run) consistent = false;
                run += hist[c];
            }
            if (consistent) fm_build_step2_wide(fm, st);
        }
        return BG_OK;
    };
    const int rc = body();
    for (void* p : tmp) hipFree(p);
    if (rc) {
        bg_fm_free(fm);
        return rc;
    }
    *out = fm;
    return BG_OK;
}

namespace {
template <bool SEEDS, bool PACKED, bool DEFER>
int launch_2x(bg_fm* fm, uint64_t n_q, const uint8_t* d_pat, const uint64_t* d_pat_off, uint8_t* d_tag, uint64_t* d_lower, uint64_t* d_upper,
              uint32_t* d_matched_len, const SeedSrc& src, hipStream_t st) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fmw_search2x_kernel<SEEDS, PACKED, DEFER>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 4;
    const uint64_t blocks = std::min<uint64_t>((n_q + 127) / 128, 256ull * (uint64_t)per_cu);
    fmw_search2x_kernel<SEEDS, PACKED, DEFER><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(fm->wdev, n_q, d_pat, d_pat_off, d_tag, d_lower, d_upper,
                                                                                          d_matched_len, src);
    BG_HIP(hipGetLastError());
    return BG_OK;
}
}  // namespace

// Which kernel answers a search on 64-bit positions:
//   * the index has 2-step blocks (a DNA-like text with at most a handful of positions outside its four letters — T$R$ of a
//     genome without N) and neither "no_step2", "no_fast" nor "ilp" = 1 is set: fm_search_fast2x_kernel<WIDE> (fm_index.hip),
//     then this file's generic kernel for the queries it deferred (a byte outside the codes, more than 256 symbols);
//   * otherwise the generic kernels here: two queries per quad (every flavour), or one ("ilp" = 1, byte patterns).
int fm_wide_search_dev(bg_fm* fm, uint64_t n_q, const uint8_t* d_pat, const uint64_t* d_pat_off, uint8_t* d_tag, uint64_t* d_lower,
                       uint64_t* d_upper, uint32_t* d_matched_len, hipStream_t st, const SeedSrc* seeds, bool packed) {
    SeedSrc src{};
    if (seeds) src = *seeds;
    src.code_bytes = (uint32_t)fm->code_byte[0] | (uint32_t)fm->code_byte[1] << 8 | (uint32_t)fm->code_byte[2] << 16 | (uint32_t)fm->code_byte[3] << 24;
    const bool fast = fm->wdev2.blocks2 && !fm->no_step2 && !fm->no_fast && fm->ilp >= 2 && fm->n_codes == 4 && (!seeds || seeds->seed_len <= kFastSyms);
    int rc;
    if (fast) {
        if ((rc = fm_wide_fast2x_launch(fm, n_q, d_pat, d_pat_off, d_tag, d_lower, d_upper, d_matched_len, st, seeds, packed))) return rc;
        if (seeds) return launch_2x<true, false, true>(fm, n_q, d_pat, d_pat_off, d_tag, d_lower, d_upper, d_matched_len, src, st);
        if (packed) return launch_2x<false, true, true>(fm, n_q, d_pat, d_pat_off, d_tag, d_lower, d_upper, d_matched_len, src, st);
        return launch_2x<false, false, true>(fm, n_q, d_pat, d_pat_off, d_tag, d_lower, d_upper, d_matched_len, src, st);
    }
    if (seeds) return launch_2x<true, false, false>(fm, n_q, d_pat, d_pat_off, d_tag, d_lower, d_upper, d_matched_len, src, st);
    if (packed) return launch_2x<false, true, false>(fm, n_q, d_pat, d_pat_off, d_tag, d_lower, d_upper, d_matched_len, src, st);
    if (fm->ilp >= 2) return launch_2x<false, false, false>(fm, n_q, d_pat, d_pat_off, d_tag, d_lower, d_upper, d_matched_len, src, st);
    const uint64_t blocks = std::min<uint64_t>((n_q + 63) / 64, 256 * 8);
    fmw_search_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(fm->wdev, n_q, d_pat, d_pat_off, d_tag, d_lower, d_upper, d_matched_len);
    BG_HIP(hipGetLastError());
    return BG_OK;
}

int fm_wide_sa_get(bg_fm* fm, uint64_t n, const uint64_t* d_index, uint64_t* d_pos, hipStream_t st) {
    if (n == 0) return BG_OK;
    if (fm->sa_kind == 1) {
        fmw_raw_get_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>((const uint64_t*)fm->d_sa, fm->wdev.n, n, d_index, d_pos);
    } else {
        SaWideDev sa = {};
        sa.sa = (const uint64_t*)fm->d_sa;
        sa.extra_row = (const uint64_t*)fm->d_extra_row;
        sa.extra_pos = (const uint64_t*)fm->d_extra_pos;
        sa.exc_byte = (const uint8_t*)fm->d_exc_byte;
        sa.n_extra = (uint32_t)fm->n_extra;
        sa.rate = fm->sa_rate;
        sa.sentinel = fm->sa_sentinel;
        sa.code_byte = (uint32_t)fm->code_byte[0] | (uint32_t)fm->code_byte[1] << 8 | (uint32_t)fm->code_byte[2] << 16 |
                       (uint32_t)fm->code_byte[3] << 24;
        const uint64_t blocks = std::min<uint64_t>((n + 63) / 64, 256 * 8);
        fmw_sampled_get_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(fm->wdev, sa, n, d_index, d_pos);
    }
    BG_HIP(hipGetLastError());
    return BG_OK;
}
