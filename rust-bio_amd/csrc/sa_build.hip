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
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>

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
__global__ __launch_bounds__(256) void sab_init_keys_kernel(const uint8_t* __restrict__ t, uint64_t n, CodeMap cm, uint32_t b, uint32_t K,
                                                            uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t k = 0;
    bool open = true;
    for (uint32_t u = 0; u < K; u++) {
        const uint64_t p = i + u;
        const uint64_t c = (open && p < n) ? (uint64_t)cm.code[t[p]] : 0ull;
        open = open && c != 0;
        k = (k << b) | c;
    }
    key[i] = k;
    val[i] = (uint32_t)i;
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
__global__ __launch_bounds__(256) void sab_sentinel_rows_kernel(const uint32_t* __restrict__ suf, uint64_t c, uint32_t* __restrict__ sent,
                                                                uint32_t* __restrict__ rank, uint32_t* __restrict__ sa,
                                                                uint8_t* __restrict__ active) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= c) return;
    sent[j] = suf[j];
    const uint32_t i = suf[c - 1 - j];
    rank[i] = (uint32_t)j;
    sa[j] = i;
    active[j] = 0;
}

// hp[j] = j where a new group starts (else 0): an inclusive max-scan turns it into "start of my group"
__global__ __launch_bounds__(256) void sab_heads_kernel(const uint64_t* __restrict__ key, uint64_t n, uint32_t* __restrict__ hp) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    hp[j] = (j == 0 || key[j] != key[j - 1]) ? (uint32_t)j : 0u;
}

// round 0: rank of every suffix, the array itself, and which suffixes are still in a group of several
__global__ __launch_bounds__(256) void sab_round0_kernel(const uint32_t* __restrict__ suf, const uint32_t* __restrict__ grp, uint64_t n,
                                                         uint32_t* __restrict__ rank, uint32_t* __restrict__ sa, uint8_t* __restrict__ active) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t i = suf[j], g = grp[j];
    rank[i] = g;
    sa[j] = i;
    const bool head = g == (uint32_t)j, next_head = j + 1 == n || grp[j + 1] == (uint32_t)(j + 1);
    active[j] = !(head && next_head);
}

// second key of an active suffix: the rank of the suffix h symbols on — or of its first sentinel, if that comes first:
// the members of a group are equal up to there, sentinel included, and the sentinels' own ranks (unique from the
// start) decide (sent: the c sentinel positions, ascending; c == 1: the text's last byte, never before i + h)
__global__ __launch_bounds__(256) void sab_round_keys_kernel(const uint32_t* __restrict__ act, uint64_t A, const uint32_t* __restrict__ rank,
                                                             uint64_t n, uint64_t h, const uint32_t* __restrict__ sent, uint32_t c,
                                                             uint64_t* __restrict__ key) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A) return;
    const uint64_t i = act[p];
    uint64_t at = i + h;
    if (c > 1) {  // first sentinel at or after i (there always is one: the text ends in one)
        uint32_t lo = 0, hi = c - 1;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (sent[mid] >= i)
                hi = mid;
            else
                lo = mid + 1;
        }
        at = min(at, (uint64_t)sent[lo]);
    }
    const uint32_t r2 = at < n ? rank[at] : 0u;  // at >= n cannot happen for an active suffix (it would hold the last sentinel)
    key[p] = (uint64_t)rank[i] << 32 | r2;
}

__global__ __launch_bounds__(256) void sab_round_heads_kernel(const uint64_t* __restrict__ key, uint64_t A, uint32_t* __restrict__ hpH,
                                                              uint32_t* __restrict__ hpF) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A) return;
    const uint64_t k = key[p], kp = p ? key[p - 1] : ~k;
    hpH[p] = (p == 0 || (k >> 32) != (kp >> 32)) ? (uint32_t)p : 0u;
    hpF[p] = (p == 0 || k != kp) ? (uint32_t)p : 0u;
}

// rewrite every group in the order of the second rank, give its members their refined ranks, mark what stays active
__global__ __launch_bounds__(256) void sab_round_apply_kernel(const uint64_t* __restrict__ key, const uint32_t* __restrict__ suf, uint64_t A,
                                                              const uint32_t* __restrict__ firstH, const uint32_t* __restrict__ firstF,
                                                              uint32_t* __restrict__ rank, uint32_t* __restrict__ sa,
                                                              uint8_t* __restrict__ active) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A) return;
    const uint32_t g = (uint32_t)(key[p] >> 32), i = suf[p];
    const uint32_t fH = firstH[p], fF = firstF[p];
    sa[(uint64_t)g + (p - fH)] = i;
    rank[i] = g + (fF - fH);
    const bool head = fF == (uint32_t)p, next_head = p + 1 == A || firstF[p + 1] == (uint32_t)(p + 1);
    active[p] = !(head && next_head);
}

__global__ __launch_bounds__(256) void sab_bwt_kernel(const uint8_t* __restrict__ t, const uint32_t* __restrict__ sa, uint64_t n,
                                                      uint8_t* __restrict__ bwt) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const uint32_t p = sa[r];
    bwt[r] = p > 0 ? t[p - 1] : t[n - 1];  // bwt.rs:43-47
}

// RawSuffixArray::sample (suffix_array.rs:86-120): every rate-th entry, plus the rows whose BWT byte is the sentinel
__global__ __launch_bounds__(256) void sab_sample_kernel(const uint32_t* __restrict__ sa, const uint8_t* __restrict__ bwt, uint64_t n,
                                                         uint32_t rate, uint32_t sentinel, uint64_t* __restrict__ sample,
                                                         uint64_t* __restrict__ extra, uint32_t extra_cap, uint32_t* __restrict__ n_extra) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
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

struct MaxU32 {
    __host__ __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; }
};

inline unsigned nblk(uint64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" int bg_suffix_array_dev(bg_ctx* ctx, const uint8_t* d_text, uint64_t n, uint32_t* d_sa, void* stream) {
    if (!ctx || !d_text || !d_sa || n == 0) return BG_ERR_INVALID_ARG;
    if (n >= 0xFFFFFFFFull) return BG_ERR_TOO_LARGE;
    hipStream_t st = (hipStream_t)stream;
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

    uint64_t *keyA = nullptr, *keyB = nullptr;
    uint32_t *valA = nullptr, *valB = nullptr, *rank = nullptr;
    uint8_t* active = nullptr;
    void* tmp = nullptr;
    uint64_t* d_count = nullptr;
    uint32_t* d_sent = nullptr;
    auto run = [&]() -> int {
        BG_HIP(hipMalloc((void**)&d_sent, n_sent * 4));
        BG_HIP(hipMalloc((void**)&keyA, n * 8));
        BG_HIP(hipMalloc((void**)&keyB, n * 8));
        BG_HIP(hipMalloc((void**)&valA, n * 4));
        BG_HIP(hipMalloc((void**)&valB, n * 4));
        BG_HIP(hipMalloc((void**)&rank, n * 4));
        BG_HIP(hipMalloc((void**)&active, n));
        BG_HIP(hipMalloc((void**)&d_count, 8));
        rocprim::double_buffer<uint64_t> keys(keyA, keyB);
        rocprim::double_buffer<uint32_t> vals(valA, valB);
        size_t t_sort = 0, t_scan = 0, t_sel = 0;
        BG_HIP(rocprim::radix_sort_pairs(nullptr, t_sort, keys, vals, n, 0, 64, st));
        BG_HIP(rocprim::inclusive_scan(nullptr, t_scan, (uint32_t*)nullptr, (uint32_t*)nullptr, n, MaxU32(), st));
        BG_HIP(rocprim::select(nullptr, t_sel, (uint32_t*)nullptr, (uint8_t*)nullptr, (uint32_t*)nullptr, d_count, n, st));
        size_t tmp_bytes = std::max(std::max(t_sort, t_scan), t_sel);
        BG_HIP(hipMalloc(&tmp, std::max<size_t>(tmp_bytes, 256)));

        // ---- round 0
        sab_init_keys_kernel<<<dim3(nblk(n)), dim3(256), 0, st>>>(d_text, n, cm, b, K, keys.current(), vals.current());
        BG_HIP(hipGetLastError());
        BG_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, vals, n, 0, (unsigned)(b * K), st));
        uint32_t* hp = (uint32_t*)keys.alternate();  // the sort's other key buffer is free now: two uint32 arrays fit
        uint32_t* grp = hp + n;
        sab_heads_kernel<<<dim3(nblk(n)), dim3(256), 0, st>>>(keys.current(), n, hp);
        BG_HIP(rocprim::inclusive_scan(tmp, tmp_bytes, hp, grp, n, MaxU32(), st));
        sab_round0_kernel<<<dim3(nblk(n)), dim3(256), 0, st>>>(vals.current(), grp, n, rank, d_sa, active);
        // rows 0 .. n_sent - 1 (key 0): the sentinel suffixes, final from here on
        sab_sentinel_rows_kernel<<<dim3(nblk(n_sent)), dim3(256), 0, st>>>(vals.current(), n_sent, d_sent, rank, d_sa, active);
        BG_HIP(hipGetLastError());
        // active suffixes, in array order
        uint32_t* act = vals.alternate();
        BG_HIP(rocprim::select(tmp, tmp_bytes, vals.current(), active, act, d_count, n, st));
        uint64_t A = 0;
        BG_HIP(hipMemcpyAsync(&A, d_count, 8, hipMemcpyDeviceToHost, st));
        BG_HIP(hipStreamSynchronize(st));
        vals.swap();  // the active list is the current value buffer from here on

        // ---- doubling rounds over the active suffixes only
        for (uint64_t h = K; A > 0; h *= 2) {
            // every suffix is unique within n symbols (a group that is tied up to its sentinels needs one round whatever h is)
            if (h > 2 * n && h > 2 * (uint64_t)K) return BG_ERR_HIP;  // cannot happen
            sab_round_keys_kernel<<<dim3(nblk(A)), dim3(256), 0, st>>>(vals.current(), A, rank, n, h, d_sent, (uint32_t)n_sent, keys.current());
            BG_HIP(hipGetLastError());
            BG_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, vals, A, 0, 64, st));
            uint32_t* hpH = (uint32_t*)keys.alternate();
            uint32_t* hpF = hpH + A;
            // firstH / firstF need their own storage: the other value buffer and the (idle) first half of ... `active`
            // is bytes; use two fresh slices of the alternate key buffer instead when A is small, else allocate
            uint32_t *firstH = nullptr, *firstF = nullptr;
            BG_HIP(hipMalloc((void**)&firstH, A * 4));
            BG_HIP(hipMalloc((void**)&firstF, A * 4));
            int rr = BG_OK;
            auto round = [&]() -> int {
                sab_round_heads_kernel<<<dim3(nblk(A)), dim3(256), 0, st>>>(keys.current(), A, hpH, hpF);
                BG_HIP(rocprim::inclusive_scan(tmp, tmp_bytes, hpH, firstH, A, MaxU32(), st));
                BG_HIP(rocprim::inclusive_scan(tmp, tmp_bytes, hpF, firstF, A, MaxU32(), st));
                sab_round_apply_kernel<<<dim3(nblk(A)), dim3(256), 0, st>>>(keys.current(), vals.current(), A, firstH, firstF, rank, d_sa, active);
                BG_HIP(hipGetLastError());
                BG_HIP(rocprim::select(tmp, tmp_bytes, vals.current(), active, vals.alternate(), d_count, A, st));
                BG_HIP(hipMemcpyAsync(&A, d_count, 8, hipMemcpyDeviceToHost, st));
                BG_HIP(hipStreamSynchronize(st));
                return BG_OK;
            };
            rr = round();
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
    hipFree(d_count);
    hipFree(d_sent);
    hipFree(tmp);
    return rc;
}

extern "C" int bg_bwt_dev(bg_ctx* ctx, const uint8_t* d_text, const uint32_t* d_sa, uint64_t n, uint8_t* d_bwt, void* stream) {
    if (!ctx || !d_text || !d_sa || !d_bwt) return BG_ERR_INVALID_ARG;
    if (n == 0) return BG_OK;
    BG_HIP(hipSetDevice(ctx->device));
    sab_bwt_kernel<<<dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream>>>(d_text, d_sa, n, d_bwt);
    BG_HIP(hipGetLastError());
    return BG_OK;
}

extern "C" int bg_sa_sample_dev(bg_ctx* ctx, const uint32_t* d_sa, const uint8_t* d_bwt, uint64_t n, uint32_t sampling_rate,
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
        sab_sample_kernel<<<dim3(nblk(n)), dim3(256), 0, st>>>(d_sa, d_bwt, n, sampling_rate, sentinel, d_sample, d_extra, cap, d_ne);
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
