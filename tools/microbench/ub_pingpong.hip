// Round trip of a 64-bit word between two blocks through global memory: block A writes k, block B waits for k and writes k
// back, N times.  (Round 6: what does one hop of fq_fused_kernel's decoupled look-back cost, across XCDs and inside one?)
//   pair (a, b) of block indices out of a grid of 256 blocks (one per CU; blocks go round-robin over the 8 XCDs: a % 8 is the XCD)
//   mode 0: relaxed agent-scope atomics (what the kernel uses: sc1 loads / stores);  mode 1: stores agent-scope, loads `sc0` only
//           (served by the XCD's own L2: only sees what was written through this L2)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ unsigned long long ld_sc0(const unsigned long long* p) {
    unsigned long long v;
    asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__global__ __launch_bounds__(64) void k_pp(unsigned long long* w, int a, int b, int n, int mode, unsigned long long* out) {
    if (threadIdx.x) return;
    const int me = blockIdx.x;
    if (me != a && me != b) return;
    unsigned long long* mine = w + (me == a ? 0 : 32);    // (different lines)
    unsigned long long* theirs = w + (me == a ? 32 : 0);
    const unsigned long long t0 = wall_clock64();
    for (int k = 1; k <= n; k++) {
        if (me == a) __hip_atomic_store(mine, (unsigned long long)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long v;
        unsigned spins = 0;
        do {
            v = mode == 0 ? __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ld_sc0(theirs);
        } while (v < (unsigned long long)k && ++spins < (1u << 12));
        if (v < (unsigned long long)k) { out[1] = (unsigned long long)k; break; }  // never seen: give up
        if (me == b) __hip_atomic_store(mine, (unsigned long long)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (me == a) out[0] = wall_clock64() - t0;
}
int main() {
    unsigned long long *w, *out, h[2];
    hipMalloc(&w, 4096);
    hipMalloc(&out, 16);
    const int n = 500;
    const int pairs[][2] = {{0, 1}, {0, 8}, {0, 16}, {3, 11}, {0, 4}, {5, 6}, {0, 128}, {0, 129}};
    for (int mode = 0; mode < 2; mode++)
        for (auto& p : pairs) {
            hipMemset(w, 0, 4096);
            hipMemset(out, 0, 16);
            k_pp<<<256, 64>>>(w, p[0], p[1], n, mode, out);
            hipDeviceSynchronize();
            hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
            printf("mode %d (%s) blocks %3d <-> %3d (XCD %d, %d): %.0f ns per round trip\n", mode, mode ? "loads sc0" : "agent-scope atomics", p[0], p[1], p[0] % 8,
                   p[1] % 8, h[0] * 10.0 / n);
            if (h[1]) printf("      (gave up at round %llu: the word never arrived)\n", h[1]);
            fflush(stdout);
        }
    return 0;
}
