// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE against known byte counts in the access patterns
// the engine uses: (a) 8 B/lane coalesced stores, (b) 16 B/lane loads where each quad reads one
// random 64-B line of a large table, (c) 1 B/lane stores to 64 per-lane streams.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_store8(uint2* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = make_uint2((unsigned)i, 7u);
}
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
__global__ void k_bytestreams(unsigned char* out, size_t per_lane, int steps) {
    size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned char* p = out + lane * per_lane;
    for (int s = 0; s < steps; s++) p[s] = (unsigned char)s;
}
int main() {
    size_t n8 = (size_t)1 << 28;  // 2 GiB of 8-B stores
    uint2* a; hipMalloc(&a, n8 * 8);
    k_store8<<<2048, 256>>>(a, n8);
    size_t nblk = (size_t)1 << 24;  // 1 GiB table of 64-B blocks
    uint4* tab; hipMalloc(&tab, nblk * 64); hipMemset(tab, 1, nblk * 64);
    unsigned* o; hipMalloc(&o, 2048 * 256 * 4);
    k_quadload<<<2048, 256>>>(tab, nblk, o, 256);  // 131072 quads x 256 lines x 64 B = 2 GiB
    size_t lanes = 2048 * 256, per = 4096;
    unsigned char* bs; hipMalloc(&bs, lanes * per);
    k_bytestreams<<<2048, 256>>>(bs, per, 4096);  // 2 GiB of byte stores
    hipDeviceSynchronize();
    // lines parsed by tools/pmc_summary.py
    printf("EXPECT k_store8 WRITE_SIZE %zu\n", n8 * 8);
    printf("EXPECT k_quadload FETCH_SIZE %zu\n", (size_t)131072 * 256 * 64);
    printf("EXPECT k_bytestreams WRITE_SIZE %zu\n", lanes * per);
    return 0;
}
