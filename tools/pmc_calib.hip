// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE against known byte counts in the access shapes the engine uses.
// Every kernel moves an exactly known number of bytes over tables far beyond the 256 MiB Infinity Cache; the EXPECT lines
// are parsed by tools/pmc_summary.py, which divides what rocprofv3 reports by what was moved and applies 1 / ratio to the
// engine kernels of the same shape (MI355X_MICROARCH.md "HBM": on gfx950 FETCH_SIZE tallies 128-byte requests at 64 B).
//   k_store8       8 B/lane coalesced stores                              (K1p's traceback tiles, records)
//   k_store16      16 B/lane coalesced stores                             (K3p's traceback lines, ingest gather)
//   k_store4       4 B/lane coalesced stores
//   k_load4/8/16   coalesced streaming loads of 4 / 8 / 16 B per lane     (ingest text passes, K3p chunk loads)
//   k_quadload     16 B/lane, each quad reads one random 64-byte block    (K5 1-step blocks, K2's slots, bit vectors)
//   k_gather128x4  4 lanes per query, 2 x 16 B per lane of a random 128-byte line   (K5 2-step blocks)
//   k_gather128x8  8 lanes per query, 16 B per lane of a random 128-byte line
//   k_bytestreams  1 B/lane stores to 64 per-lane streams                 (the known outlier)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_store8(uint2* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = make_uint2((unsigned)i, 7u);
}
__global__ void k_store16(uint4* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = make_uint4((unsigned)i, 7u, 9u, 11u);
}
__global__ void k_store4(unsigned* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (unsigned)i;
}
template <typename T>
__device__ __forceinline__ unsigned fold(const T& v);
template <>
__device__ __forceinline__ unsigned fold<unsigned>(const unsigned& v) { return v; }
template <>
__device__ __forceinline__ unsigned fold<uint2>(const uint2& v) { return v.x ^ v.y; }
template <>
__device__ __forceinline__ unsigned fold<uint4>(const uint4& v) { return v.x ^ v.y ^ v.z ^ v.w; }
template <typename T>
__device__ __forceinline__ void load_body(const T* __restrict__ in, size_t n, unsigned* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= fold<T>(in[i]);
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// plain names: pmc_summary.py matches the EXPECT names as substrings of rocprofv3's kernel names
__global__ void k_load4(const unsigned* in, size_t n, unsigned* out) { load_body(in, n, out); }
__global__ void k_load8(const uint2* in, size_t n, unsigned* out) { load_body(in, n, out); }
__global__ void k_load16(const uint4* in, size_t n, unsigned* out) { load_body(in, n, out); }
__global__ void k_quadload(const uint4* tab, size_t nblk, unsigned* out, int iters) {
    size_t q = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    unsigned t = threadIdx.x & 3, acc = 0;
    unsigned long long s = q * 0x9E3779B97F4A7C15ull + 12345;
    for (int it = 0; it < iters; it++) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        size_t b = (s >> 20) % nblk;
        uint4 v = tab[b * 4 + t];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// G lanes per query read one random 128-byte line: 128 / G bytes per lane in 16-byte loads (G = 4: K5's 2-step blocks)
template <int G>
__device__ __forceinline__ void gather128_body(const uint4* tab, size_t nline, unsigned* out, int iters) {
    constexpr int PER = 8 / G;
    size_t q = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    unsigned t = threadIdx.x % G, acc = 0;
    unsigned long long s = q * 0x9E3779B97F4A7C15ull + 777;
    for (int it = 0; it < iters; it++) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        size_t b = (s >> 20) % nline;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            uint4 v = tab[b * 8 + t * PER + k];
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void k_gather128x4(const uint4* tab, size_t nline, unsigned* out, int iters) { gather128_body<4>(tab, nline, out, iters); }
__global__ void k_gather128x8(const uint4* tab, size_t nline, unsigned* out, int iters) { gather128_body<8>(tab, nline, out, iters); }
__global__ void k_bytestreams(unsigned char* out, size_t per_lane, int steps) {
    size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned char* p = out + lane * per_lane;
    for (int s = 0; s < steps; s++) p[s] = (unsigned char)s;
}
int main() {
    const size_t GiB2 = (size_t)2 << 30;
    unsigned char* buf;   // 3 GiB: streaming tables (first 2 GiB) and the gather tables (all of it)
    if (hipMalloc(&buf, (size_t)3 << 30) != hipSuccess) return 1;
    unsigned* o;
    if (hipMemset(buf, 1, (size_t)3 << 30) != hipSuccess || hipMalloc(&o, 2048 * 256 * 4) != hipSuccess) return 1;
    k_store8<<<2048, 256>>>((uint2*)buf, GiB2 / 8);
    k_store16<<<2048, 256>>>((uint4*)buf, GiB2 / 16);
    k_store4<<<2048, 256>>>((unsigned*)buf, GiB2 / 4);
    k_load4<<<2048, 256>>>((const unsigned*)buf, GiB2 / 4, o);
    k_load8<<<2048, 256>>>((const uint2*)buf, GiB2 / 8, o);
    k_load16<<<2048, 256>>>((const uint4*)buf, GiB2 / 16, o);
    size_t nblk = (size_t)1 << 24;  // 1 GiB table of 64-B blocks
    k_quadload<<<2048, 256>>>((const uint4*)buf, nblk, o, 256);  // 131072 quads x 256 lines x 64 B = 2 GiB
    size_t nline = ((size_t)3 << 30) / 128;  // 3 GiB table of 128-B lines (the 2-step blocks of a 3 Gbp index)
    k_gather128x4<<<2048, 256>>>((const uint4*)buf, nline, o, 128);  // 131072 queries x 128 lines x 128 B = 2 GiB
    k_gather128x8<<<2048, 256>>>((const uint4*)buf, nline, o, 256);  //  65536 queries x 256 lines x 128 B = 2 GiB
    size_t lanes = 2048 * 256, per = 4096;
    k_bytestreams<<<2048, 256>>>(buf, per, 4096);  // 2 GiB of byte stores
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    // lines parsed by tools/pmc_summary.py
    printf("EXPECT k_store8 WRITE_SIZE %zu\n", GiB2);
    printf("EXPECT k_store16 WRITE_SIZE %zu\n", GiB2);
    printf("EXPECT k_store4 WRITE_SIZE %zu\n", GiB2);
    printf("EXPECT k_load4 FETCH_SIZE %zu\n", GiB2);
    printf("EXPECT k_load8 FETCH_SIZE %zu\n", GiB2);
    printf("EXPECT k_load16 FETCH_SIZE %zu\n", GiB2);
    printf("EXPECT k_quadload FETCH_SIZE %zu\n", (size_t)131072 * 256 * 64);
    printf("EXPECT k_gather128x4 FETCH_SIZE %zu\n", (size_t)131072 * 128 * 128);
    printf("EXPECT k_gather128x8 FETCH_SIZE %zu\n", (size_t)65536 * 256 * 128);
    printf("EXPECT k_bytestreams WRITE_SIZE %zu\n", lanes * per);
    return 0;
}
