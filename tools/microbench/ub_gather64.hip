// ub_gather64 — what the memory system of this chip delivers for K5's access pattern, without K5's arithmetic:
// quads of 4 lanes, each step two random 64-byte lines (lane t reads bytes [16t, 16t+16): one coalesced 64-B request
// per line, exactly fm_backward_search_kernel's rank(l-1) / rank(r) loads), full occupancy (256-thread blocks,
// 8 waves per SIMD, 2048 blocks like K5's persistent grid), over table footprints of 33 MB (100 Mbp index: inside the
// 256 MiB Infinity Cache), 333 MB (1 Gbp), 1 GB (3 Gbp) and 3 GB.
//   dep = 1: the addresses of step s+1 are a function of the data loaded at step s (an LF step: l, r come out of the
//            ranks) — the chain K5 has;
//   dep = 0: addresses from a per-quad generator, independent of the data — as many lines in flight as the
//            compiler unrolls (upper bound of what more memory-level parallelism could buy).
//   lines = 2 (K5: two ranks per step) or 1 (both ranks in one block: the narrow-interval steps).
// Prints one JSON line per configuration: lines/s and GB/s of 64-byte lines.  bench.py --gather-ceiling runs it and
// reports K5's fetched bytes / s against the dep = 1 figure at the same footprint (`frac_of_gather_ceiling`).
//   hipcc --offload-arch=gfx950 -O3 -o ub_gather64 ub_gather64.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e = (x);                                                                \
        if (e != hipSuccess) {                                                             \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                         \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

__device__ __forceinline__ unsigned quad_xor(unsigned v) {
    v ^= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);
    v ^= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);
    return v;
}

template <int LINES, bool DEP>
__global__ __launch_bounds__(256) void gather_kernel(const uint4* __restrict__ tab, unsigned nblk, unsigned* __restrict__ out, int steps) {
    const unsigned t = threadIdx.x & 3;
    const unsigned quad = (blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    unsigned s = quad * 0x9E3779B9u + 12345u;  // uniform inside the quad
    unsigned acc = 0;
    for (int it = 0; it < steps; it++) {
        s = s * 1664525u + 1013904223u;
        // multiply-high maps 32 random bits to [0, nblk)
        const unsigned b0 = __umulhi(s ^ (DEP ? acc : 0u), nblk);
        const uint4 v0 = tab[(size_t)b0 * 4 + t];
        unsigned x = v0.x ^ v0.y ^ v0.z ^ v0.w;
        if (LINES == 2) {
            const unsigned b1 = __umulhi((s * 0x85EBCA6Bu) ^ (DEP ? acc : 0u), nblk);
            const uint4 v1 = tab[(size_t)b1 * 4 + t];
            x ^= v1.x + v1.y + v1.z + v1.w;
        }
        // the quad's combined word: every lane's next address depends on all 64 bytes of both lines (K5: quad_sum of the parts)
        acc = DEP ? quad_xor(x) : (acc ^ x);
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

template <int LINES, bool DEP>
static int run(const uint4* tab, unsigned nblk, unsigned* out, int steps, double mb) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int blocks = 256 * 8;
    gather_kernel<LINES, DEP><<<blocks, 256>>>(tab, nblk, out, steps / 4 + 1);  // warm-up (and the cache state of a repeated launch)
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        gather_kernel<LINES, DEP><<<blocks, 256>>>(tab, nblk, out, steps);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double lines = (double)blocks * 64 * steps * LINES;
    printf("{\"footprint_mb\": %.0f, \"dep\": %d, \"lines_per_step\": %d, \"ms\": %.3f, \"glines_per_s\": %.3f, \"gb_per_s\": %.1f}\n", mb,
           (int)DEP, LINES, best, lines / best / 1e6, lines * 64 / best / 1e6);
    fflush(stdout);
    return 0;
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 2000;
    std::vector<double> sizes_mb = {33.3, 333.3, 1000.0, 3000.0};
    if (argc > 2) {
        sizes_mb.clear();
        for (int i = 2; i < argc; i++) sizes_mb.push_back(atof(argv[i]));
    }
    unsigned* out;
    CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    for (double mb : sizes_mb) {
        const size_t bytes = (size_t)(mb * 1e6) / 64 * 64;
        const unsigned nblk = (unsigned)(bytes / 64);
        uint4* tab;
        CK(hipMalloc(&tab, bytes));
        fill_kernel<<<2048, 256>>>(tab, bytes / 16);
        CK(hipDeviceSynchronize());
        if (run<2, true>(tab, nblk, out, steps, mb)) return 1;
        if (run<1, true>(tab, nblk, out, steps, mb)) return 1;
        if (run<2, false>(tab, nblk, out, steps, mb)) return 1;
        if (run<1, false>(tab, nblk, out, steps, mb)) return 1;
        CK(hipFree(tab));
    }
    return 0;
}
