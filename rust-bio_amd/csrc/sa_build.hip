// Suffix array + BWT construction on the device (bg_suffix_array_dev, bg_bwt_dev, bg_sa_sample_dev).
//
// Stands where rust-bio's host code runs `suffix_array` (/root/reference/src/data_structures/suffix_array.rs:264-284)
// and `bwt` (bwt.rs:39-49) when the text already lives in HBM: the host SA-IS (host_tables.cpp) needs ~100 s and 16 GB
// of host memory per Gbp, which is what kept BASELINE configs[4] (a 3 Gbp reference) out of reach.  The suffix array
// of the text as the reference transforms it is unique, so any correct sorter reproduces the reference's array.  Texts
// with several sentinels (several sequences, or T$R$ for an FMD index): transform_text (suffix_array.rs:444-466) turns
// them into distinct symbols ordered by position, the LAST occurrence smallest — here: keys stop at the first sentinel,
// the sentinel suffixes get their ranks (descending position) up front, and a doubling round looks at
// rank[min(i + h, next sentinel at or after i)], never across a sentinel.
//
// Prefix doubling with discarding (Larsson & Sadakane 2007, as usually run on GPUs):
//   round 0   64-bit key per suffix = its first K symbols (alphabet re-coded to b bits, K = 64 / b: 21 bases for
//             ACGT$), one radix sort of (key, suffix) pairs; rank[i] = first position of i's group;
//   round h   only suffixes in groups of more than one stay active; each gets the key (rank[i], rank[i + h]), the
//             active list is sorted, every group is rewritten in place in the order of the second rank and split;
//             h doubles until no group is left.
// A random genome is done after round 0 and one small round; repeats cost further (small) rounds.  Sorting,
// scans and compaction are rocPRIM's (rocprim::radix_sort_pairs / inclusive_scan / select): plain library primitives,
// like rust-bio's own use of a library suffix sorter; the kernels around them are below.  Memory: 29 bytes per symbol
// of scratch.
//
// Round 5: every kernel and the driver are templates over the position type P.  P = uint32_t is the builder of rounds
// 2-4 (texts below 2^32 - 1 symbols, bg_suffix_array_dev).  P = uint64_t (bg_suffix_array_dev64: the reference indexes with
// usize, suffix_array.rs:264) lifts that limit: positions, ranks and group heads are 64-bit, and a doubling round — whose
// key (rank[i], rank[i + h]) no longer fits one 64-bit radix key — is two stable passes, by the second rank and then by
// the first (LSD), over the bits a rank below n needs.  49 bytes of scratch per symbol: a 4.4 G-symbol text takes 216 GB
// of the 288 GB part next to its 35 GB suffix array.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/discard_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include <algorithm>
#include <vector>

#include "bg_common.h"

namespace {

struct CodeMap {
    uint8_t code[256];
};

__global__ __launch_bounds__(256) void sab_presence_kernel(const uint8_t* __restrict__ t, uint64_t n, uint32_t* __restrict__ cnt) {
    __shared__ uint32_t s[256];
    s[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) atomicAdd(&s[t[i]], 1u);
    __syncthreads();
    if (s[threadIdx.x]) atomicAdd(&cnt[threadIdx.x], min(s[threadIdx.x], 2u));  // 0, 1 or "more": saturating is enough
}

// key of suffix i: its first K symbols, b bits each, most significant first, up to and including its first sentinel
// (code 0): what follows a sentinel counts as 0 — a comparison never goes past one (transform_text makes every
// sentinel a symbol of its own) — and so do symbols past the end
// (the body of the key: shared by the kernels below)
__device__ __forceinline__ uint64_t sab_key_of(const uint8_t* __restrict__ t, uint64_t n, const CodeMap& cm, uint32_t b, uint32_t K, uint64_t i) {
    uint64_t k = 0;
    bool open = true;
    for (uint32_t u = 0; u < K; u++) {
        const uint64_t p = i + u;
        const uint64_t c = (open && p < n) ? (uint64_t)cm.code[t[p]] : 0ull;
        open = open && c != 0;
        k = (k << b) | c;
    }
    return k;
}
// Round 0 in several passes (round 6): a text whose (key, suffix) pairs do not fit the device twice over — 6.2 G symbols of
// T$R$ of a human genome: 32 bytes per symbol for the sort alone — is sorted bucket range by bucket range.  A bucket is the
// top `sbits` bits of the key (the suffix's first few symbols); the buckets' sizes say which rows each owns, contiguous
// bucket ranges of at most `cap` suffixes are collected (in position order: the sort stays stable), sorted and written to
// their rows.  Groups never span buckets, so ranks and the active flags come out as from one sort.
template <typename P>
__global__ __launch_bounds__(256) void sab_bucket_hist_kernel(const uint8_t* __restrict__ t, uint64_t n, CodeMap cm, uint32_t b, uint32_t S,
                                                              unsigned long long* __restrict__ hist) {
    __shared__ uint32_t s_h[4096];  // (at most 12 bits of bucket; a block's share of the text stays far below 2^32)
    const uint32_t nb = 1u << (b * S);
    for (uint32_t k = threadIdx.x; k < nb; k += blockDim.x) s_h[k] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        atomicAdd(&s_h[sab_key_of(t, n, cm, b, S, i)], 1u);
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < nb; k += blockDim.x)
        if (s_h[k]) atomicAdd(&hist[k], (unsigned long long)s_h[k]);
}
__global__ __launch_bounds__(256) void sab_bucket_flags_kernel(const uint8_t* __restrict__ t, uint64_t n, CodeMap cm, uint32_t b, uint32_t S,
                                                               uint32_t lo, uint32_t hi, uint8_t* __restrict__ flag) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t k = (uint32_t)sab_key_of(t, n, cm, b, S, i);
        flag[i] = k >= lo && k < hi;
    }
}
template <typename P>
__global__ __launch_bounds__(256) void sab_list_keys_kernel(const uint8_t* __restrict__ t, uint64_t n, CodeMap cm, uint32_t b, uint32_t K,
                                                            const P* __restrict__ val, uint64_t m, uint64_t* __restrict__ key) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (uint64_t)gridDim.x * blockDim.x)
        key[j] = sab_key_of(t, n, cm, b, K, (uint64_t)val[j]);
}
template <typename P>
struct CountFrom {  // positions 0, 1, 2, ... as the input of a stream compaction
    __host__ __device__ P operator()(uint64_t i) const { return (P)i; }
};

template <typename P>
__global__ __launch_bounds__(256) void sab_init_keys_kernel(const uint8_t* __restrict__ t, uint64_t n, CodeMap cm, uint32_t b, uint32_t K,
                                                            uint64_t* __restrict__ key, P* __restrict__ val) {
    // (grid-stride: a launch may not exceed 2^32 threads, and texts here do)
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t k = 0;
        bool open = true;
        for (uint32_t u = 0; u < K; u++) {
            const uint64_t p = i + u;
            const uint64_t c = (open && p < n) ? (uint64_t)cm.code[t[p]] : 0ull;
            open = open && c != 0;
            k = (k << b) | c;
        }
        key[i] = k;
        val[i] = (P)i;
    }
}

__global__ __launch_bounds__(256) void sab_count_byte_kernel(const uint8_t* __restrict__ t, uint64_t n, uint32_t byte,
                                                             unsigned long long* __restrict__ cnt) {
    uint32_t c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) c += t[i] == byte;
#pragma unroll
    for (int o = 32; o; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(cnt, (unsigned long long)c);
}

// The c suffixes that start with a sentinel are the c smallest of the text, the last occurrence first
// (suffix_array.rs:454-458): the sorted list holds them in rows 0 .. c - 1 in ascending position (stable sort of equal
// keys): `sent` keeps that list (next-sentinel lookups), rows / ranks are rewritten in descending position
template <typename P>
__global__ __launch_bounds__(256) void sab_sentinel_rows_kernel(const P* __restrict__ suf, uint64_t c, P* __restrict__ sent,
                                                                P* __restrict__ rank, P* __restrict__ sa,
                                                                uint8_t* __restrict__ active) {
    // (grid-stride: a launch may not exceed 2^32 threads, and texts here do)
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < c; j += (uint64_t)gridDim.x * blockDim.x) {
        sent[j] = suf[j];
        const P i = suf[c - 1 - j];
        rank[i] = (P)j;
        sa[j] = i;
        active[j] = 0;
    }
}

// hp[j] = j where a new group starts (else 0): an inclusive max-scan turns it into "start of my group"
// (row0: the row of the sorted list's first entry — a pass of round 0 owns rows [row0, row0 + n))
template <typename P>
__global__ __launch_bounds__(256) void sab_heads_kernel(const uint64_t* __restrict__ key, uint64_t n, uint64_t row0, P* __restrict__ hp) {
    // (grid-stride: a launch may not exceed 2^32 threads, and texts here do)
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x) {
        hp[j] = (j == 0 || key[j] != key[j - 1]) ? (P)(row0 + j) : (P)0;
    }
}

// round 0: rank of every suffix, the array itself, and which suffixes are still in a group of several
template <typename P>
__global__ __launch_bounds__(256) void sab_round0_kernel(const P* __restrict__ suf, const P* __restrict__ grp, uint64_t n, uint64_t row0,
                                                         P* __restrict__ rank, P* __restrict__ sa, uint8_t* __restrict__ active) {
    // (grid-stride: a launch may not exceed 2^32 threads, and texts here do)
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x) {
        const P i = suf[j], g = grp[j];
        rank[i] = g;
        sa[row0 + j] = i;
        const bool head = g == (P)(row0 + j), next_head = j + 1 == n || grp[j + 1] == (P)(row0 + j + 1);
        active[row0 + j] = !(head && next_head);
    }
}

// second key of an active suffix: the rank of the suffix h symbols on — or of its first sentinel, if that comes first:
// the members of a group are equal up to there, sentinel included, and the sentinels' own ranks (unique from the
// start) decide (sent: the c sentinel positions, ascending; c == 1: the text's last byte, never before i + h)
// rank of the suffix h symbols behind suffix i, or of i's first sentinel if that comes first (see above)
template <typename P>
__device__ __forceinline__ P sab_second_rank(uint64_t i, const P* __restrict__ rank, uint64_t n, uint64_t h, const P* __restrict__ sent, uint64_t c) {
    uint64_t at = i + h;
    if (c > 1) {  // first sentinel at or after i (there always is one: the text ends in one)
        uint64_t lo = 0, hi = c - 1;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if ((uint64_t)sent[mid] >= i)
                hi = mid;
            else
                lo = mid + 1;
        }
        at = min(at, (uint64_t)sent[lo]);
    }
    return at < n ? rank[at] : (P)0;  // at >= n cannot happen for an active suffix (it would hold the last sentinel)
}
// P = uint32_t: key = rank[i] << 32 | second rank (one sort).  P = uint64_t, FIRST == false: key = second rank (the first of
// two stable passes); FIRST == true: key = rank[i] (the second pass, over the list the first one left)
template <typename P, bool FIRST>
__global__ __launch_bounds__(256) void sab_round_keys_kernel(const P* __restrict__ act, uint64_t A, const P* __restrict__ rank,
                                                             uint64_t n, uint64_t h, const P* __restrict__ sent, uint64_t c,
                                                             uint64_t* __restrict__ key) {
    // (grid-stride: a launch may not exceed 2^32 threads, and texts here do)
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < A; p += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = act[p];
        if (sizeof(P) == 4)
            key[p] = (uint64_t)rank[i] << 32 | (uint64_t)sab_second_rank<P>(i, rank, n, h, sent, c);
        else
            key[p] = FIRST ? (uint64_t)rank[i] : (uint64_t)sab_second_rank<P>(i, rank, n, h, sent, c);
    }
}

// group heads of the sorted active list: hpH by the first rank, hpF by both.  P = uint64_t: `key` holds the first rank only
// (the second pass's key); the second one is looked up again for p and p - 1
template <typename P>
__global__ __launch_bounds__(256) void sab_round_heads_kernel(const uint64_t* __restrict__ key, const P* __restrict__ suf, uint64_t A,
                                                              const P* __restrict__ rank, uint64_t n, uint64_t h, const P* __restrict__ sent,
                                                              uint64_t c, P* __restrict__ hpH, P* __restrict__ hpF) {
    // (grid-stride: a launch may not exceed 2^32 threads, and texts here do)
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < A; p += (uint64_t)gridDim.x * blockDim.x) {
        bool newH, newF;
        if (sizeof(P) == 4) {
            const uint64_t k = key[p], kp = p ? key[p - 1] : ~k;
            newH = p == 0 || (k >> 32) != (kp >> 32);
            newF = p == 0 || k != kp;
        } else {
            newH = p == 0 || key[p] != key[p - 1];
            newF = newH || sab_second_rank<P>(suf[p], rank, n, h, sent, c) != sab_second_rank<P>(suf[p - 1], rank, n, h, sent, c);
        }
        hpH[p] = newH ? (P)p : (P)0;
        hpF[p] = newF ? (P)p : (P)0;
    }
}

// the refined rank of every member (rank writes must not race with the heads kernel's rank reads: a launch of its own)
template <typename P>
__global__ __launch_bounds__(256) void sab_round_apply_kernel(const uint64_t* __restrict__ key, const P* __restrict__ suf, uint64_t A,
                                                              const P* __restrict__ firstH, const P* __restrict__ firstF,
                                                              P* __restrict__ rank, P* __restrict__ sa,
                                                              uint8_t* __restrict__ active) {
    // (grid-stride: a launch may not exceed 2^32 threads, and texts here do)
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < A; p += (uint64_t)gridDim.x * blockDim.x) {
        const P g = sizeof(P) == 4 ? (P)(key[p] >> 32) : (P)key[p], i = suf[p];
        const P fH = firstH[p], fF = firstF[p];
        sa[(uint64_t)g + (p - fH)] = i;
        rank[i] = g + (fF - fH);
        const bool head = fF == (P)p, next_head = p + 1 == A || firstF[p + 1] == (P)(p + 1);
        active[p] = !(head && next_head);
    }
}

template <typename P>
__global__ __launch_bounds__(256) void sab_bwt_kernel(const uint8_t* __restrict__ t, const P* __restrict__ sa, uint64_t n,
                                                      uint8_t* __restrict__ bwt) {
    // (grid-stride: a launch may not exceed 2^32 threads, and texts here do)
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (uint64_t)gridDim.x * blockDim.x) {
        const P p = sa[r];
        bwt[r] = p > 0 ? t[p - 1] : t[n - 1];  // bwt.rs:43-47
    }
}

// RawSuffixArray::sample (suffix_array.rs:86-120): every rate-th entry, plus the rows whose BWT byte is the sentinel
template <typename P>
__global__ __launch_bounds__(256) void sab_sample_kernel(const P* __restrict__ sa, const uint8_t* __restrict__ bwt, uint64_t n,
                                                         uint32_t rate, uint32_t sentinel, uint64_t* __restrict__ sample,
                                                         uint64_t* __restrict__ extra, uint32_t extra_cap, uint32_t* __restrict__ n_extra) {
    // (grid-stride: a launch may not exceed 2^32 threads, and texts here do)
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (uint64_t)gridDim.x * blockDim.x) {
        if (r % rate == 0) {
            sample[r / rate] = sa[r];
        } else if (bwt[r] == sentinel) {
            const uint32_t k = atomicAdd(n_extra, 1u);
            if (k < extra_cap) {
                extra[2 * (uint64_t)k] = r;
                extra[2 * (uint64_t)k + 1] = sa[r];
            }
        }
    }
}

template <typename P>
struct MaxOf {
    __host__ __device__ P operator()(P a, P b) const { return a > b ? a : b; }
};

// blocks of 256 threads for n elements, capped: every kernel above strides over its elements (HIP refuses launches of
// 2^32 threads or more — a 4.4 G-symbol text has more elements than that)
inline unsigned nblk(uint64_t n) { return (unsigned)std::min<uint64_t>((n + 255) / 256, 1u << 22); }

}  // namespace

namespace {

template <typename P>
int sa_build_impl(bg_ctx* ctx, const uint8_t* d_text, uint64_t n, P* d_sa, hipStream_t st) {
    BG_HIP(hipSetDevice(ctx->device));
    // ---- alphabet: bytes that occur, the sentinel (last byte) must be the unique smallest one
    uint32_t* d_cnt = nullptr;
    BG_HIP(hipMalloc((void**)&d_cnt, 256 * 4));
    uint32_t cnt[256];
    uint8_t last = 0;
    auto probe = [&]() -> int {
        BG_HIP(hipMemsetAsync(d_cnt, 0, 256 * 4, st));
        sab_presence_kernel<<<dim3(std::min<uint64_t>(nblk(n), 4096)), dim3(256), 0, st>>>(d_text, n, d_cnt);
        BG_HIP(hipGetLastError());
        BG_HIP(hipMemcpyAsync(cnt, d_cnt, sizeof(cnt), hipMemcpyDeviceToHost, st));
        BG_HIP(hipMemcpyAsync(&last, d_text + (n - 1), 1, hipMemcpyDeviceToHost, st));
        BG_HIP(hipStreamSynchronize(st));
        return BG_OK;
    };
    int rc = probe();
    hipFree(d_cnt);
    if (rc) return rc;
    for (int c = 0; c < last; c++)
        if (cnt[c]) return BG_ERR_SENTINEL;  // suffix_array.rs:431-437: a byte below the sentinel
    // several sentinels are ranked by position (transform_text, suffix_array.rs:444-466): how many?  (The presence
    // counters saturate at 2 per block, which still tells "exactly once" from "more than once".)
    uint64_t n_sent = 1;
    if (cnt[last] != 1) {
        unsigned long long* d_c = nullptr;
        BG_HIP(hipMalloc((void**)&d_c, 8));
        auto count = [&]() -> int {
            BG_HIP(hipMemsetAsync(d_c, 0, 8, st));
            sab_count_byte_kernel<<<dim3(std::min<uint64_t>(nblk(n), 4096)), dim3(256), 0, st>>>(d_text, n, last, d_c);
            BG_HIP(hipGetLastError());
            BG_HIP(hipMemcpyAsync(&n_sent, d_c, 8, hipMemcpyDeviceToHost, st));
            BG_HIP(hipStreamSynchronize(st));
            return BG_OK;
        };
        rc = count();
        hipFree(d_c);
        if (rc) return rc;
    }
    CodeMap cm = {};
    uint32_t sigma = 0;
    for (int c = 0; c < 256; c++)
        if (cnt[c]) cm.code[c] = (uint8_t)sigma++;
    uint32_t b = 1;
    while ((1u << b) < sigma) b++;
    const uint32_t K = 64 / b;
    constexpr bool WIDE = sizeof(P) == 8;
    unsigned rank_bits = 1;  // bits of a rank below n (the radix passes of the 64-bit flavour's rounds)
    while (rank_bits < 64 && (n >> rank_bits)) rank_bits++;

    uint64_t *keyA = nullptr, *keyB = nullptr;
    P *valA = nullptr, *valB = nullptr, *rank = nullptr;
    uint8_t *active = nullptr, *flag = nullptr;
    void* tmp = nullptr;
    uint64_t* d_count = nullptr;
    unsigned long long* d_hist = nullptr;
    P* d_sent = nullptr;
    auto run = [&]() -> int {
        BG_HIP(hipMalloc((void**)&d_sent, n_sent * sizeof(P)));
        BG_HIP(hipMalloc((void**)&rank, n * sizeof(P)));
        BG_HIP(hipMalloc((void**)&active, n));
        BG_HIP(hipMalloc((void**)&d_count, 8));
        // ---- round 0, in passes of at most `cap` suffixes (one pass when everything fits: the round-2 .. 5 path)
        constexpr size_t kPairBytes = 16 + 2 * sizeof(P);  // (key, suffix), double-buffered
        uint64_t cap = n;
        if (ctx->sa_chunk_symbols > 0) {
            cap = (uint64_t)ctx->sa_chunk_symbols;
        } else {
            size_t free_b = 0, total_b = 0;
            BG_HIP(hipMemGetInfo(&free_b, &total_b));
            // the doubling rounds of a repetitive text want memory of their own later; a pass needs its pairs + the flags
            const uint64_t fit = (uint64_t)((double)free_b * 0.7) / kPairBytes;
            if (fit < n + (n >> 4)) cap = std::max<uint64_t>(fit > n / 64 ? fit - n / 64 : fit, 1u << 20);
        }
        uint32_t S = 1;  // symbols of a bucket: at most 12 bits of key
        while (S < K && b * (S + 1) <= 12) S++;
        const uint32_t n_bucket = 1u << (b * S);
        struct Pass {
            uint32_t lo, hi;       // buckets [lo, hi)
            uint64_t row0, m;      // rows [row0, row0 + m)
        };
        std::vector<Pass> passes;
        if (cap >= n) {
            passes.push_back({0, n_bucket, 0, n});
        } else {
            BG_HIP(hipMalloc((void**)&d_hist, (size_t)n_bucket * 8));
            BG_HIP(hipMemsetAsync(d_hist, 0, (size_t)n_bucket * 8, st));
            sab_bucket_hist_kernel<P><<<dim3(std::min<unsigned>(nblk(n), 8192)), dim3(256), 0, st>>>(d_text, n, cm, b, S, d_hist);
            BG_HIP(hipGetLastError());
            std::vector<unsigned long long> hist(n_bucket);
            BG_HIP(hipMemcpyAsync(hist.data(), d_hist, (size_t)n_bucket * 8, hipMemcpyDeviceToHost, st));
            BG_HIP(hipStreamSynchronize(st));
            uint64_t row = 0;
            for (uint32_t k = 0; k < n_bucket;) {
                Pass ps{k, k, row, 0};
                while (ps.hi < n_bucket && (ps.m == 0 || ps.m + hist[ps.hi] <= cap)) ps.m += hist[ps.hi++];
                passes.push_back(ps);
                row += ps.m;
                k = ps.hi;
            }
            BG_HIP(hipMalloc((void**)&flag, n));
        }
        uint64_t C = 0;  // the largest pass (one bucket may exceed cap: a text of few distinct words)
        for (const Pass& ps : passes) C = std::max(C, ps.m);
        C = std::max<uint64_t>(C, n_sent);
        BG_HIP(hipMalloc((void**)&keyA, C * 8));
        BG_HIP(hipMalloc((void**)&keyB, C * 8));
        BG_HIP(hipMalloc((void**)&valA, C * sizeof(P)));
        BG_HIP(hipMalloc((void**)&valB, C * sizeof(P)));
        rocprim::double_buffer<uint64_t> keys(keyA, keyB);
        rocprim::double_buffer<P> vals(valA, valB);
        size_t t_sort = 0, t_scan = 0, t_sel = 0, t_sel2 = 0;
        BG_HIP(rocprim::radix_sort_pairs(nullptr, t_sort, keys, vals, C, 0, 64, st));
        BG_HIP(rocprim::inclusive_scan(nullptr, t_scan, (P*)nullptr, (P*)nullptr, C, MaxOf<P>(), st));
        BG_HIP(rocprim::select(nullptr, t_sel, (P*)nullptr, (uint8_t*)nullptr, (P*)nullptr, d_count, n, st));
        auto positions = rocprim::make_transform_iterator(rocprim::counting_iterator<uint64_t>(0), CountFrom<P>());
        BG_HIP(rocprim::select(nullptr, t_sel2, positions, (uint8_t*)nullptr, (P*)nullptr, d_count, n, st));
        size_t tmp_bytes = std::max(std::max(t_sort, t_scan), std::max(t_sel, t_sel2));
        BG_HIP(hipMalloc(&tmp, std::max<size_t>(tmp_bytes, 256)));

        for (size_t pi = 0; pi < passes.size(); pi++) {
            const Pass& ps = passes[pi];
            if (ps.m == 0) continue;
            if (passes.size() == 1) {
                sab_init_keys_kernel<P><<<dim3(nblk(n)), dim3(256), 0, st>>>(d_text, n, cm, b, K, keys.current(), vals.current());
            } else {
                sab_bucket_flags_kernel<<<dim3(nblk(n)), dim3(256), 0, st>>>(d_text, n, cm, b, S, ps.lo, ps.hi, flag);
                BG_HIP(rocprim::select(tmp, tmp_bytes, positions, flag, vals.current(), d_count, n, st));
                sab_list_keys_kernel<P><<<dim3(nblk(ps.m)), dim3(256), 0, st>>>(d_text, n, cm, b, K, vals.current(), ps.m, keys.current());
            }
            BG_HIP(hipGetLastError());
            BG_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, vals, ps.m, 0, (unsigned)(b * K), st));
            // group heads and their scan: 32-bit positions — two arrays in the sort's other key buffer; 64-bit positions — one
            // there and one in the other value buffer
            P* hp = (P*)keys.alternate();
            P* grp = WIDE ? vals.alternate() : hp + ps.m;
            sab_heads_kernel<P><<<dim3(nblk(ps.m)), dim3(256), 0, st>>>(keys.current(), ps.m, ps.row0, hp);
            BG_HIP(rocprim::inclusive_scan(tmp, tmp_bytes, hp, grp, ps.m, MaxOf<P>(), st));
            sab_round0_kernel<P><<<dim3(nblk(ps.m)), dim3(256), 0, st>>>(vals.current(), grp, ps.m, ps.row0, rank, d_sa, active);
            // rows 0 .. n_sent - 1 (key 0, the first bucket of the first pass): the sentinel suffixes, final from here on
            if (ps.row0 == 0) sab_sentinel_rows_kernel<P><<<dim3(nblk(n_sent)), dim3(256), 0, st>>>(vals.current(), n_sent, d_sent, rank, d_sa, active);
            BG_HIP(hipGetLastError());
        }
        // active suffixes, in array order: counted first, so that the doubling rounds get buffers of THEIR size (a random
        // genome leaves a thousandth of its suffixes tied after 21-32 symbols; the pass buffers go back first)
        BG_HIP(rocprim::select(tmp, tmp_bytes, d_sa, active, rocprim::discard_iterator(), d_count, n, st));
        uint64_t A = 0;
        BG_HIP(hipMemcpyAsync(&A, d_count, 8, hipMemcpyDeviceToHost, st));
        BG_HIP(hipStreamSynchronize(st));
        if (A > C) {
            hipFree(keyA), hipFree(keyB), hipFree(valA), hipFree(valB);
            keyA = keyB = nullptr;
            valA = valB = nullptr;
            BG_HIP(hipMalloc((void**)&keyA, A * 8));
            BG_HIP(hipMalloc((void**)&keyB, A * 8));
            BG_HIP(hipMalloc((void**)&valA, A * sizeof(P)));
            BG_HIP(hipMalloc((void**)&valB, A * sizeof(P)));
            keys = rocprim::double_buffer<uint64_t>(keyA, keyB);
            vals = rocprim::double_buffer<P>(valA, valB);
            size_t t2 = 0;
            BG_HIP(rocprim::radix_sort_pairs(nullptr, t2, keys, vals, A, 0, 64, st));
            if (t2 > tmp_bytes) {
                hipFree(tmp);
                tmp = nullptr;
                tmp_bytes = t2;
                BG_HIP(hipMalloc(&tmp, tmp_bytes));
            }
        }
        if (flag) {
            hipFree(flag);
            flag = nullptr;
        }
        if (A) BG_HIP(rocprim::select(tmp, tmp_bytes, d_sa, active, vals.current(), d_count, n, st));

        // ---- doubling rounds over the active suffixes only
        for (uint64_t h = K; A > 0; h *= 2) {
            // every suffix is unique within n symbols (a group that is tied up to its sentinels needs one round whatever h is)
            if (h > 2 * n && h > 2 * (uint64_t)K) return BG_ERR_HIP;  // cannot happen
            if (!WIDE) {
                sab_round_keys_kernel<P, false><<<dim3(nblk(A)), dim3(256), 0, st>>>(vals.current(), A, rank, n, h, d_sent, n_sent, keys.current());
                BG_HIP(hipGetLastError());
                BG_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, vals, A, 0, 64, st));
            } else {  // (first rank, second rank) as two stable passes, least significant first
                sab_round_keys_kernel<P, false><<<dim3(nblk(A)), dim3(256), 0, st>>>(vals.current(), A, rank, n, h, d_sent, n_sent, keys.current());
                BG_HIP(hipGetLastError());
                BG_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, vals, A, 0, rank_bits, st));
                sab_round_keys_kernel<P, true><<<dim3(nblk(A)), dim3(256), 0, st>>>(vals.current(), A, rank, n, h, d_sent, n_sent, keys.current());
                BG_HIP(hipGetLastError());
                BG_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, vals, A, 0, rank_bits, st));
            }
            P *hpH = nullptr, *hpF = nullptr, *firstH = nullptr, *firstF = nullptr;
            if (!WIDE) {  // two 32-bit arrays fit the sort's other key buffer
                hpH = (P*)keys.alternate();
                hpF = hpH + A;
            } else {
                hpH = (P*)keys.alternate();
                BG_HIP(hipMalloc((void**)&hpF, A * sizeof(P)));
            }
            BG_HIP(hipMalloc((void**)&firstH, A * sizeof(P)));
            BG_HIP(hipMalloc((void**)&firstF, A * sizeof(P)));
            int rr = BG_OK;
            auto round = [&]() -> int {
                sab_round_heads_kernel<P><<<dim3(nblk(A)), dim3(256), 0, st>>>(keys.current(), vals.current(), A, rank, n, h, d_sent, n_sent, hpH, hpF);
                BG_HIP(rocprim::inclusive_scan(tmp, tmp_bytes, hpH, firstH, A, MaxOf<P>(), st));
                BG_HIP(rocprim::inclusive_scan(tmp, tmp_bytes, hpF, firstF, A, MaxOf<P>(), st));
                sab_round_apply_kernel<P><<<dim3(nblk(A)), dim3(256), 0, st>>>(keys.current(), vals.current(), A, firstH, firstF, rank, d_sa, active);
                BG_HIP(hipGetLastError());
                BG_HIP(rocprim::select(tmp, tmp_bytes, vals.current(), active, vals.alternate(), d_count, A, st));
                BG_HIP(hipMemcpyAsync(&A, d_count, 8, hipMemcpyDeviceToHost, st));
                BG_HIP(hipStreamSynchronize(st));
                return BG_OK;
            };
            rr = round();
            if (WIDE) hipFree(hpF);
            hipFree(firstH);
            hipFree(firstF);
            if (rr) return rr;
            vals.swap();
        }
        return BG_OK;
    };
    rc = run();
    hipFree(keyA);
    hipFree(keyB);
    hipFree(valA);
    hipFree(valB);
    hipFree(rank);
    hipFree(active);
    hipFree(flag);
    hipFree(d_hist);
    hipFree(d_count);
    hipFree(d_sent);
    hipFree(tmp);
    return rc;
}

}  // namespace

extern "C" int bg_suffix_array_dev(bg_ctx* ctx, const uint8_t* d_text, uint64_t n, uint32_t* d_sa, void* stream) {
    if (!ctx || !d_text || !d_sa || n == 0) return BG_ERR_INVALID_ARG;
    if (n >= 0xFFFFFFFFull) return BG_ERR_TOO_LARGE;  // 64-bit positions: bg_suffix_array_dev64
    return sa_build_impl<uint32_t>(ctx, d_text, n, d_sa, (hipStream_t)stream);
}

// the same with 64-bit positions (suffix_array.rs:264: the reference's usize): texts of 2^32 - 1 symbols and more, up to 2^40
extern "C" int bg_suffix_array_dev64(bg_ctx* ctx, const uint8_t* d_text, uint64_t n, uint64_t* d_sa, void* stream) {
    if (!ctx || !d_text || !d_sa || n == 0) return BG_ERR_INVALID_ARG;
    if (n > (1ull << 40)) return BG_ERR_TOO_LARGE;
    return sa_build_impl<uint64_t>(ctx, d_text, n, d_sa, (hipStream_t)stream);
}

extern "C" int bg_bwt_dev(bg_ctx* ctx, const uint8_t* d_text, const uint32_t* d_sa, uint64_t n, uint8_t* d_bwt, void* stream) {
    if (!ctx || !d_text || !d_sa || !d_bwt) return BG_ERR_INVALID_ARG;
    if (n == 0) return BG_OK;
    BG_HIP(hipSetDevice(ctx->device));
    sab_bwt_kernel<uint32_t><<<dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream>>>(d_text, d_sa, n, d_bwt);
    BG_HIP(hipGetLastError());
    return BG_OK;
}
extern "C" int bg_bwt_dev64(bg_ctx* ctx, const uint8_t* d_text, const uint64_t* d_sa, uint64_t n, uint8_t* d_bwt, void* stream) {
    if (!ctx || !d_text || !d_sa || !d_bwt) return BG_ERR_INVALID_ARG;
    if (n == 0) return BG_OK;
    BG_HIP(hipSetDevice(ctx->device));
    sab_bwt_kernel<uint64_t><<<dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream>>>(d_text, d_sa, n, d_bwt);
    BG_HIP(hipGetLastError());
    return BG_OK;
}

namespace {
template <typename P>
int sa_sample_impl(bg_ctx* ctx, const P* d_sa, const uint8_t* d_bwt, uint64_t n, uint32_t sampling_rate,
                   uint8_t sentinel, uint64_t* sample, uint64_t* extra_rows, uint64_t* extra_pos, uint64_t extra_cap,
                   uint64_t* n_extra, void* stream) {
    if (!ctx || !d_sa || !d_bwt || !sample || !n_extra || sampling_rate == 0 || n == 0) return BG_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    BG_HIP(hipSetDevice(ctx->device));
    const uint64_t ns = (n + sampling_rate - 1) / sampling_rate;
    const uint32_t cap = (uint32_t)std::min<uint64_t>(extra_cap, 1u << 24);
    uint64_t *d_sample = nullptr, *d_extra = nullptr;
    uint32_t* d_ne = nullptr;
    std::vector<uint64_t> h_extra;
    uint32_t ne = 0;
    auto run = [&]() -> int {
        BG_HIP(hipMalloc((void**)&d_sample, ns * 8));
        BG_HIP(hipMalloc((void**)&d_extra, std::max<uint64_t>(cap, 1) * 16));
        BG_HIP(hipMalloc((void**)&d_ne, 4));
        BG_HIP(hipMemsetAsync(d_ne, 0, 4, st));
        sab_sample_kernel<P><<<dim3(nblk(n)), dim3(256), 0, st>>>(d_sa, d_bwt, n, sampling_rate, sentinel, d_sample, d_extra, cap, d_ne);
        BG_HIP(hipGetLastError());
        BG_HIP(hipMemcpyAsync(sample, d_sample, ns * 8, hipMemcpyDeviceToHost, st));
        BG_HIP(hipMemcpyAsync(&ne, d_ne, 4, hipMemcpyDeviceToHost, st));
        BG_HIP(hipStreamSynchronize(st));
        if (ne > cap) return BG_ERR_OPS_CAP;
        h_extra.resize(2 * (size_t)ne);
        if (ne) BG_HIP(hipMemcpy(h_extra.data(), d_extra, (size_t)ne * 16, hipMemcpyDeviceToHost));
        return BG_OK;
    };
    int rc = run();
    hipFree(d_sample);
    hipFree(d_extra);
    hipFree(d_ne);
    *n_extra = ne;
    if (rc) return rc;
    // the reference keeps them in a hash map; the engine wants them sorted by row
    std::vector<std::pair<uint64_t, uint64_t>> v(ne);
    for (uint32_t k = 0; k < ne; k++) v[k] = {h_extra[2 * k], h_extra[2 * k + 1]};
    std::sort(v.begin(), v.end());
    for (uint32_t k = 0; k < ne; k++) {
        if (extra_rows) extra_rows[k] = v[k].first;
        if (extra_pos) extra_pos[k] = v[k].second;
    }
    return BG_OK;
}
}  // namespace

extern "C" int bg_sa_sample_dev(bg_ctx* ctx, const uint32_t* d_sa, const uint8_t* d_bwt, uint64_t n, uint32_t sampling_rate,
                                uint8_t sentinel, uint64_t* sample, uint64_t* extra_rows, uint64_t* extra_pos, uint64_t extra_cap,
                                uint64_t* n_extra, void* stream) {
    return sa_sample_impl<uint32_t>(ctx, d_sa, d_bwt, n, sampling_rate, sentinel, sample, extra_rows, extra_pos, extra_cap, n_extra, stream);
}
extern "C" int bg_sa_sample_dev64(bg_ctx* ctx, const uint64_t* d_sa, const uint8_t* d_bwt, uint64_t n, uint32_t sampling_rate,
                                  uint8_t sentinel, uint64_t* sample, uint64_t* extra_rows, uint64_t* extra_pos, uint64_t extra_cap,
                                  uint64_t* n_extra, void* stream) {
    return sa_sample_impl<uint64_t>(ctx, d_sa, d_bwt, n, sampling_rate, sentinel, sample, extra_rows, extra_pos, extra_cap, n_extra, stream);
}
