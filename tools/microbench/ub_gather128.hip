// ub_gather128 — would a 2-step rank block pay?    K5 sits at 0.88-0.95 of the chip's rate for random
// 64-byte line gathers (ub_gather64); halving the REQUESTS per query needs rank blocks that answer two pattern symbols
// at once — 16 pair counters + 4-bit pair symbols, i.e. 128-byte lines (index: n bytes instead of n / 3).  This measures
// what the memory system delivers for such lines, with K5's grid and dependence structure:
//   w128x8 : 8 lanes per query, lane t reads bytes [16t, 16t+16) of a 128-byte line  (8 queries per wavefront)
//   w128x4 : 4 lanes per query, lane t reads bytes [32t, 32t+32): two 16-byte loads   (16 queries per wavefront, like K5)
//   w64x4  : ub_gather64's pattern (the baseline, same binary, same run)
// over table footprints 3 x those of ub_gather64 (the 2-step index is 3 x the 1-step one): 100 MB, 1 GB, 3 GB, 9 GB.
// One JSON line per configuration: lines/s; "queries_step_rate" = steps/s a search would see (lines / 2 per step).
//   hipcc --offload-arch=gfx950 -O3 -o ub_gather128 ub_gather128.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                      \
    do {                                                           \
        hipError_t e = (x);                                        \
        if (e != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); \
            return 1;                                              \
        }                                                          \
    } while (0)

template <int G>
__device__ __forceinline__ unsigned group_xor(unsigned v) {  // xor over the G lanes of a query
    v ^= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
    v ^= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);  // quad_perm [2,3,0,1]
    if (G == 8) v ^= (unsigned)__shfl_xor((int)v, 4);
    return v;
}

// LINE: bytes per line (64 / 128); G: lanes per query; each lane reads LINE / G bytes in 16-byte loads
template <int LINE, int G>
__global__ __launch_bounds__(256) void gather_kernel(const uint4* __restrict__ tab, unsigned nblk, unsigned* __restrict__ out, int steps) {
    constexpr int PER = LINE / G / 16;  // 16-byte loads per lane per line
    const unsigned t = threadIdx.x % G;
    const unsigned q = (blockIdx.x * blockDim.x + threadIdx.x) / G;
    unsigned s = q * 0x9E3779B9u + 12345u;
    unsigned acc = 0;
    for (int it = 0; it < steps; it++) {
        s = s * 1664525u + 1013904223u;
        const unsigned b0 = __umulhi(s ^ acc, nblk), b1 = __umulhi((s * 0x85EBCA6Bu) ^ acc, nblk);
        unsigned x = 0;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const uint4 v0 = tab[(size_t)b0 * (LINE / 16) + t * PER + k];
            const uint4 v1 = tab[(size_t)b1 * (LINE / 16) + t * PER + k];
            x ^= (v0.x ^ v0.y ^ v0.z ^ v0.w) ^ (v1.x + v1.y + v1.z + v1.w);
        }
        acc = group_xor<G>(x);  // every lane's next address depends on all bytes of both lines
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ void fill_kernel(uint4* tab, size_t n16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u;
        tab[i] = make_uint4(h, h ^ 0x5bd1e995u, h * 31u, h + 7u);
    }
}

template <int LINE, int G>
static int run(const char* name, const uint4* tab, size_t bytes, unsigned* out, int steps, double mb) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int blocks = 256 * 8;
    const unsigned nblk = (unsigned)(bytes / LINE);
    gather_kernel<LINE, G><<<blocks, 256>>>(tab, nblk, out, steps / 4 + 1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        gather_kernel<LINE, G><<<blocks, 256>>>(tab, nblk, out, steps);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double lines = (double)blocks * (256 / G) * steps * 2;
    printf("{\"variant\": \"%s\", \"line_bytes\": %d, \"lanes_per_query\": %d, \"footprint_mb\": %.0f, \"ms\": %.3f, \"glines_per_s\": %.3f, "
           "\"gb_per_s\": %.1f, \"gsteps_per_s\": %.3f}\n",
           name, LINE, G, mb, best, lines / best / 1e6, lines * LINE / best / 1e6, lines / 2 / best / 1e6);
    fflush(stdout);
    return 0;
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 2000;
    std::vector<double> sizes_mb = {33.3, 100.0, 333.3, 1000.0, 3000.0, 9000.0};
    if (argc > 2) {
        sizes_mb.clear();
        for (int i = 2; i < argc; i++) sizes_mb.push_back(atof(argv[i]));
    }
    unsigned* out;
    CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    for (double mb : sizes_mb) {
        const size_t bytes = (size_t)(mb * 1e6) / 128 * 128;
        uint4* tab;
        CK(hipMalloc(&tab, bytes));
        fill_kernel<<<2048, 256>>>(tab, bytes / 16);
        CK(hipDeviceSynchronize());
        if (run<64, 4>("w64x4", tab, bytes, out, steps, mb)) return 1;
        if (run<128, 8>("w128x8", tab, bytes, out, steps, mb)) return 1;
        if (run<128, 4>("w128x4", tab, bytes, out, steps, mb)) return 1;
        CK(hipFree(tab));
    }
    return 0;
}
