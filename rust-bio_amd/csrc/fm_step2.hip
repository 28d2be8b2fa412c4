// 2-step rank blocks of the FM index (fm_kernels.h: Fm2Dev, block2_part; consumer: fm_search_fast_kernel<STEP2> in
// fm_index.hip), built on the device from the finished 1-step index — both builders (bg_fm_build: host BWT;
// bg_fm_build_dev: BWT in HBM) end here.  Reference semantics served: FMIndexable::backward_search
// (/root/reference/src/data_structures/fmindex.rs:144-208), two iterations of its loop per block access.
//
//   pass 1 (one thread per BWT position i): c = code of L[i]; j = LF(i) = less[c] + Occ(c, i) - 1 — a thread-serial rank
//           in the 1-step block of i; c2 = code of L[j] (the rows LF maps a symbol's occurrences to are consecutive, so the
//           four streams of j walk the BWT sequentially: cache-friendly) -> nibble c << 2 | c2 into a byte array; a
//           position whose L[i] or L[j] has no code goes to the exception list with 0 in that component;
//   pass 2 (one thread per 128-position block): 128 nibbles -> 64 bytes, bit-sliced per 32 positions, + how many of each of the 16 codes;
//   sixteen exclusive scans (rocPRIM) -> the counters; C2[a][b] = less[b] + Occ(b, less[a] - 1) by sixteen threads.
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include <algorithm>
#include <vector>

#include "fm_kernels.h"

using namespace bgfm;

namespace {

template <typename P>
struct Less4T {
    P v[4];
};
using Less4 = Less4T<uint32_t>;

// (the helpers below serve both layouts: FmDev with uint32 positions, FmWideDev with uint64 positions and counters relative
//  to a superblock — wide_base() adds the superblock's absolute count)
__device__ __forceinline__ uint64_t wide_base(const FmDev&, uint64_t, uint32_t) { return 0; }
__device__ __forceinline__ uint64_t wide_base(const FmWideDev& fm, uint64_t blk, uint32_t c) { return fm.sb[(blk >> fm.sb_shift) * 4 + c]; }
__device__ __forceinline__ uint32_t exc_count_le(const FmDev& fm, uint32_t p) { return count_le(fm.exc_pos, 0u, fm.n_exc, p); }
__device__ __forceinline__ uint32_t exc_count_le(const FmWideDev& fm, uint64_t p) { return count_le64(fm.exc_pos, 0u, fm.n_exc, p); }

template <typename Dev, typename P>
__device__ __forceinline__ uint32_t fm_code_at(const Dev& fm, P j) {
    const uint32_t* blk = (const uint32_t*)fm.blocks + (uint64_t)(j / kSymPerBlock) * 16;
    const uint32_t s = (uint32_t)(j % kSymPerBlock);
    return (blk[4 + (s >> 4)] >> (2 * (s & 15u))) & 3u;
}
template <typename Dev, typename P>
__device__ __forceinline__ bool fm_is_exc(const Dev& fm, P p) {
    if (!fm.n_exc) return false;
    const uint32_t k = exc_count_le(fm, p);
    return k > 0 && fm.exc_pos[k - 1] == p;
}
// Occ(code c, i) by ONE thread: counter + the matches among symbols 0 .. i % 192 of the block
template <typename Dev, typename P>
__device__ P fm_rank_thread(const Dev& fm, uint32_t c, P i) {
    const uint64_t b = (uint64_t)(i / kSymPerBlock);
    const uint32_t* blk = (const uint32_t*)fm.blocks + b * 16;
    const uint32_t o1 = (uint32_t)(i % kSymPerBlock) + 1;
    const uint32_t full = o1 >> 4, rem = o1 & 15u;
    const uint32_t pat = c * 0x55555555u;
    uint32_t n = blk[c];
    for (uint32_t w = 0; w < full; w++) {
        uint32_t e = ~(blk[4 + w] ^ pat);
        n += (uint32_t)__popc(e & (e >> 1) & 0x55555555u);
    }
    if (rem) {
        uint32_t e = ~(blk[4 + full] ^ pat);
        n += (uint32_t)__popc(e & (e >> 1) & 0x55555555u & ((1u << (2 * rem)) - 1u));
    }
    P r = (P)wide_base(fm, b, c) + n;
    if (c == 0 && fm.n_exc) r -= exc_count_le(fm, i);  // exceptions sit in the stream as code 0
    return r;
}

template <typename P>
struct ExcRec {  // a position whose first or second symbol has no code
    P pos;
    uint32_t nib;
    uint32_t pad;
};

template <typename Dev, typename P>
__global__ __launch_bounds__(256) void fm2_nibble_kernel(const Dev fm, const Less4T<P> less4, uint8_t* __restrict__ nib, uint32_t cap,
                                                         uint32_t* __restrict__ n_exc, ExcRec<P>* __restrict__ exc) {
    // (grid-stride: a launch may not exceed 2^32 threads, and wide texts do)
    for (uint64_t i64 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i64 < (uint64_t)fm.n; i64 += (uint64_t)gridDim.x * blockDim.x) {
        const P i = (P)i64;
        const uint32_t c = fm_code_at(fm, i);
        if (c == 0 && fm_is_exc(fm, i)) {  // L[i] has no code (the sentinel): first component 0, LF(i) is nobody's business
            nib[i64] = 0;
            const uint32_t k = atomicAdd(n_exc, 1u);
            if (k < cap) exc[k] = ExcRec<P>{i, 16u, 0u};
            continue;
        }
        const P j = less4.v[c] + fm_rank_thread(fm, c, i) - 1u;  // LF(i), suffix_array.rs:177-178
        uint32_t c2 = j < fm.n ? fm_code_at(fm, j) : 0u;
        const bool e2 = j >= fm.n || (c2 == 0 && fm_is_exc(fm, j));
        if (e2) c2 = 0;
        const uint32_t v = (c << 2) | c2;
        nib[i64] = (uint8_t)v;
        if (e2) {
            const uint32_t k = atomicAdd(n_exc, 1u);
            if (k < cap) exc[k] = ExcRec<P>{i, v, 0u};
        }
    }
}

__global__ __launch_bounds__(256) void fm2_pack_kernel(const uint8_t* __restrict__ nib, uint64_t n, uint64_t nblk, uint32_t* __restrict__ blocks2,
                                                       uint32_t* __restrict__ cnt /* [16][nblk] */) {
    const uint64_t blk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blk >= nblk) return;
    const uint64_t lo = blk * kSym2PerBlock;
    uint32_t c[16];
#pragma unroll
    for (int k = 0; k < 16; k++) c[k] = 0;
    for (uint32_t g = 0; g < 4; g++) {  // 32 positions: one dword per bit of the code (fm_kernels.h: block2_part)
        uint32_t plane[4] = {0, 0, 0, 0};
        for (uint32_t t = 0; t < 32; t++) {
            const uint64_t i = lo + 32 * g + t;
            if (i < n) {
                const uint32_t v = nib[i] & 15u;
#pragma unroll
                for (int b = 0; b < 4; b++) plane[b] |= ((v >> b) & 1u) << t;
#pragma unroll
                for (int k = 0; k < 16; k++) c[k] += (v == (uint32_t)k) ? 1u : 0u;
            }
        }
#pragma unroll
        for (int b = 0; b < 4; b++) blocks2[blk * 32 + 16 + 4 * g + b] = plane[b];
    }
#pragma unroll
    for (int k = 0; k < 16; k++) cnt[(uint64_t)k * nblk + blk] = c[k];
}
__global__ __launch_bounds__(256) void fm2_heads_kernel(uint64_t nblk, const uint32_t* __restrict__ scanned, uint32_t* __restrict__ blocks2) {
    const uint64_t blk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blk >= nblk) return;
#pragma unroll
    for (int k = 0; k < 16; k++) blocks2[blk * 32 + k] = scanned[(uint64_t)k * nblk + blk];
}
// 64-bit positions: counters relative to the superblock's first block; the superblock's twenty bases (sixteen codes, then
// the four sums a single step adds up: first component == a)
__global__ __launch_bounds__(256) void fm2_heads_wide_kernel(uint64_t nblk, uint32_t sb_shift, const uint64_t* __restrict__ scanned /* [16][nblk] */,
                                                             uint32_t* __restrict__ blocks2, uint64_t* __restrict__ sb2) {
    const uint64_t blk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blk >= nblk) return;
    const uint64_t first = (blk >> sb_shift) << sb_shift;
    uint64_t sum4[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint64_t base = scanned[(uint64_t)k * nblk + first];
        blocks2[blk * 32 + k] = (uint32_t)(scanned[(uint64_t)k * nblk + blk] - base);
        sum4[k >> 2] += base;
        if (blk == first) sb2[(blk >> sb_shift) * 20 + k] = base;
    }
    if (blk == first)
        for (int a = 0; a < 4; a++) sb2[(blk >> sb_shift) * 20 + 16 + a] = sum4[a];
}
template <typename Dev, typename P>
__global__ void fm2_c2_kernel(const Dev fm, const Less4T<P> less4, P* __restrict__ c2) {
    const uint32_t c = threadIdx.x;
    if (c >= 16) return;
    const uint32_t a = c >> 2, b = c & 3u;
    const P la = less4.v[a];
    c2[c] = less4.v[b] + (la ? fm_rank_thread(fm, b, (P)(la - 1u)) : (P)0);
}

struct U32ToU64 {
    __host__ __device__ uint64_t operator()(uint32_t v) const { return (uint64_t)v; }
};

}  // namespace

// Builds fm->dev2 (best effort: on any failure the index simply keeps single steps).  Synchronises `st`.
void fm_build_step2(bg_fm* fm, hipStream_t st) {
    fm->dev2 = Fm2Dev{};
    if (getenv("BG_FM_NO_STEP2")) return;
    const uint32_t n = fm->dev.n;
    if (fm->dev.n_dense || fm->n_codes != 4 || n < 2 || fm->dev.n_exc > kMaxExc2 / 2) return;
    const uint64_t nblk = ((uint64_t)n + kSym2PerBlock - 1) / kSym2PerBlock;
    Less4 l4;
    uint32_t less32[256];
    if (hipMemcpy(less32, fm->d_less, sizeof(less32), hipMemcpyDeviceToHost) != hipSuccess) return;
    for (int c = 0; c < 4; c++) l4.v[c] = less32[fm->code_byte[c]];
    std::vector<void*> tmp;
    auto dalloc = [&](void** p, size_t bytes) {
        if (hipMalloc(p, std::max<size_t>(bytes, 16)) != hipSuccess) return false;
        tmp.push_back(*p);
        return true;
    };
    void* d_b2 = nullptr;
    uint8_t* d_nib = nullptr;
    uint32_t *d_cnt = nullptr, *d_scan = nullptr, *d_ne = nullptr, *d_c2 = nullptr;
    ExcRec<uint32_t>* d_exc = nullptr;
    void* d_cub = nullptr;
    size_t cub_bytes = 0;
    const uint32_t cap = 4 * kMaxExc2;
    bool ok = hipMalloc(&d_b2, nblk * 128) == hipSuccess;
    ok = ok && dalloc((void**)&d_nib, n) && dalloc((void**)&d_cnt, 16 * nblk * 4) && dalloc((void**)&d_scan, 16 * nblk * 4) &&
         dalloc((void**)&d_ne, 4) && dalloc((void**)&d_c2, 64) && dalloc((void**)&d_exc, cap * sizeof(ExcRec<uint32_t>));
    ok = ok && rocprim::exclusive_scan(nullptr, cub_bytes, d_cnt, d_scan, 0u, nblk, rocprim::plus<uint32_t>(), st) == hipSuccess;
    ok = ok && dalloc(&d_cub, cub_bytes);
    uint32_t ne = 0, c2[16];
    ExcRec<uint32_t> exc[4 * kMaxExc2];
    if (ok) {
        (void)hipMemsetAsync(d_ne, 0, 4, st);
        fm2_nibble_kernel<FmDev, uint32_t><<<dim3((unsigned)(((uint64_t)n + 255) / 256)), dim3(256), 0, st>>>(fm->dev, l4, d_nib, cap, d_ne, d_exc);
        fm2_pack_kernel<<<dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, st>>>(d_nib, n, nblk, (uint32_t*)d_b2, d_cnt);
        for (int k = 0; k < 16 && ok; k++)
            ok = rocprim::exclusive_scan(d_cub, cub_bytes, d_cnt + (uint64_t)k * nblk, d_scan + (uint64_t)k * nblk, 0u, nblk,
                                         rocprim::plus<uint32_t>(), st) == hipSuccess;
        fm2_heads_kernel<<<dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, st>>>(nblk, d_scan, (uint32_t*)d_b2);
        fm2_c2_kernel<FmDev, uint32_t><<<dim3(1), dim3(64), 0, st>>>(fm->dev, l4, d_c2);
        ok = ok && hipGetLastError() == hipSuccess;
        ok = ok && hipMemcpyAsync(&ne, d_ne, 4, hipMemcpyDeviceToHost, st) == hipSuccess;
        ok = ok && hipMemcpyAsync(c2, d_c2, 64, hipMemcpyDeviceToHost, st) == hipSuccess;
        ok = ok && hipStreamSynchronize(st) == hipSuccess;
        ok = ok && ne <= kMaxExc2;
        if (ok && ne) ok = hipMemcpy(exc, d_exc, (size_t)ne * sizeof(ExcRec<uint32_t>), hipMemcpyDeviceToHost) == hipSuccess;
    }
    for (void* p : tmp) hipFree(p);
    if (!ok) {
        hipFree(d_b2);
        return;
    }
    std::sort(exc, exc + ne, [](const ExcRec<uint32_t>& a, const ExcRec<uint32_t>& b) { return a.pos < b.pos; });
    fm->d_blocks2 = d_b2;
    fm->bytes += nblk * 128;
    fm->dev2.blocks2 = (const uint4*)d_b2;
    for (int k = 0; k < 16; k++) fm->dev2.c2[k] = c2[k];
    for (uint32_t e = 0; e < kMaxExc2; e++) {  // unused entries: a position no rank reaches (the kernel reads two of them unconditionally)
        fm->dev2.exc_pos[e] = 0xFFFFFFFFu;
        fm->dev2.exc_nib[e] = 0xFF;
    }
    for (uint32_t e = 0; e < ne; e++) {
        fm->dev2.exc_pos[e] = exc[e].pos;
        fm->dev2.exc_nib[e] = (uint8_t)exc[e].nib;
    }
    fm->dev2.n_exc = ne;
}

// The same behind an index on 64-bit positions (fm_wide.hip): counters relative to a superblock + 64-bit bases, C2 and the
// exception positions 64-bit; the nibble pass walks the 1-step wide blocks.  Best effort, synchronises `st`.
void fm_build_step2_wide(bg_fm* fm, hipStream_t st) {
    fm->wdev2 = Fm2WideDev{};
    if (getenv("BG_FM_NO_STEP2")) return;
    const uint64_t n = fm->wdev.n;
    if (fm->n_codes != 4 || n < 2 || fm->wdev.n_exc > kMaxExc2 / 2) return;
    const uint64_t nblk = (n + kSym2PerBlock - 1) / kSym2PerBlock;
    const uint32_t sb_shift = fm->wdev.sb_shift;  // (blocks of 128 instead of 192 positions: relative counts only get smaller)
    const uint64_t n_sb = ((nblk - 1) >> sb_shift) + 1;
    Less4T<uint64_t> l4;
    uint64_t less64[256];
    if (hipMemcpy(less64, fm->d_less, sizeof(less64), hipMemcpyDeviceToHost) != hipSuccess) return;
    for (int c = 0; c < 4; c++) l4.v[c] = less64[fm->code_byte[c]];
    std::vector<void*> tmp;
    auto dalloc = [&](void** p, size_t bytes) {
        if (hipMalloc(p, std::max<size_t>(bytes, 16)) != hipSuccess) return false;
        tmp.push_back(*p);
        return true;
    };
    void *d_b2 = nullptr, *d_sb2 = nullptr;
    uint8_t* d_nib = nullptr;
    uint32_t *d_cnt = nullptr, *d_ne = nullptr;
    uint64_t *d_scan = nullptr, *d_c2 = nullptr;
    ExcRec<uint64_t>* d_exc = nullptr;
    void* d_cub = nullptr;
    size_t cub_bytes = 0;
    const uint32_t cap = 4 * kMaxExc2;
    bool ok = hipMalloc(&d_b2, nblk * 128) == hipSuccess && hipMalloc(&d_sb2, n_sb * 160) == hipSuccess;
    ok = ok && dalloc((void**)&d_nib, n) && dalloc((void**)&d_cnt, 16 * nblk * 4) && dalloc((void**)&d_scan, 16 * nblk * 8) &&
         dalloc((void**)&d_ne, 4) && dalloc((void**)&d_c2, 128) && dalloc((void**)&d_exc, cap * sizeof(ExcRec<uint64_t>));
    if (ok) {
        auto in64 = rocprim::make_transform_iterator(d_cnt, U32ToU64());
        ok = rocprim::exclusive_scan(nullptr, cub_bytes, in64, d_scan, (uint64_t)0, nblk, rocprim::plus<uint64_t>(), st) == hipSuccess;
    }
    ok = ok && dalloc(&d_cub, cub_bytes);
    uint32_t ne = 0;
    uint64_t c2[16];
    ExcRec<uint64_t> exc[4 * kMaxExc2];
    if (ok) {
        (void)hipMemsetAsync(d_ne, 0, 4, st);
        fm2_nibble_kernel<FmWideDev, uint64_t><<<dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 1u << 22)), dim3(256), 0, st>>>(fm->wdev, l4, d_nib, cap, d_ne, d_exc);
        fm2_pack_kernel<<<dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, st>>>(d_nib, n, nblk, (uint32_t*)d_b2, d_cnt);
        for (int k = 0; k < 16 && ok; k++) {
            auto ink = rocprim::make_transform_iterator(d_cnt + (uint64_t)k * nblk, U32ToU64());
            ok = rocprim::exclusive_scan(d_cub, cub_bytes, ink, d_scan + (uint64_t)k * nblk, (uint64_t)0, nblk, rocprim::plus<uint64_t>(), st) == hipSuccess;
        }
        fm2_heads_wide_kernel<<<dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, st>>>(nblk, sb_shift, d_scan, (uint32_t*)d_b2, (uint64_t*)d_sb2);
        fm2_c2_kernel<FmWideDev, uint64_t><<<dim3(1), dim3(64), 0, st>>>(fm->wdev, l4, d_c2);
        ok = ok && hipGetLastError() == hipSuccess;
        ok = ok && hipMemcpyAsync(&ne, d_ne, 4, hipMemcpyDeviceToHost, st) == hipSuccess;
        ok = ok && hipMemcpyAsync(c2, d_c2, 128, hipMemcpyDeviceToHost, st) == hipSuccess;
        ok = ok && hipStreamSynchronize(st) == hipSuccess;
        ok = ok && ne <= kMaxExc2;
        if (ok && ne) ok = hipMemcpy(exc, d_exc, (size_t)ne * sizeof(ExcRec<uint64_t>), hipMemcpyDeviceToHost) == hipSuccess;
    }
    for (void* p : tmp) hipFree(p);
    if (!ok) {
        hipFree(d_b2);
        hipFree(d_sb2);
        return;
    }
    std::sort(exc, exc + ne, [](const ExcRec<uint64_t>& a, const ExcRec<uint64_t>& b) { return a.pos < b.pos; });
    fm->d_blocks2 = d_b2;
    fm->d_sb2 = d_sb2;
    fm->bytes += nblk * 128 + n_sb * 160;
    fm->wdev2.blocks2 = (const uint4*)d_b2;
    fm->wdev2.sb2 = (const uint64_t*)d_sb2;
    fm->wdev2.sb_shift = sb_shift;
    for (int k = 0; k < 16; k++) fm->wdev2.c2[k] = c2[k];
    for (uint32_t e = 0; e < kMaxExc2; e++) {  // unused entries: a position no rank reaches
        fm->wdev2.exc_pos[e] = ~0ull;
        fm->wdev2.exc_nib[e] = 0xFF;
    }
    for (uint32_t e = 0; e < ne; e++) {
        fm->wdev2.exc_pos[e] = exc[e].pos;
        fm->wdev2.exc_nib[e] = (uint8_t)exc[e].nib;
    }
    fm->wdev2.n_exc = ne;
}
