// How fast can N blocks each take a ticket (one atomicAdd on ONE address by thread 0, then a barrier) — the first thing every
// block of a single-pass (decoupled look-back) kernel does?  Round 6, fq_fused_kernel: 19 715 tiles.
//   ub_ticket [blocks=19715]   -> us per launch with the ticket, with blockIdx instead, and with one ticket per 4 / 16 blocks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void k_ticket(unsigned* ticket, unsigned* sink, int mode) {
    __shared__ unsigned s;
    if (threadIdx.x == 0) {
        if (mode == 0) s = atomicAdd(ticket, 1u);
        else if (mode == 1) s = blockIdx.x;
        else s = (blockIdx.x % (unsigned)mode == 0) ? atomicAdd(ticket, 1u) : blockIdx.x;
    }
    __syncthreads();
    if (s == 0xFFFFFFFFu) sink[threadIdx.x] = s;
}
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 19715;
    unsigned *t, *sink;
    hipMalloc(&t, 4);
    hipMalloc(&sink, 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int mode : {0, 1, 4, 16}) {
        for (int rep = 0; rep < 3; rep++) {
            hipMemset(t, 0, 4);
            hipEventRecord(e0);
            for (int k = 0; k < 10; k++) k_ticket<<<n, 256>>>(t, sink, mode);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("mode %2d (%s): %.1f us per launch of %d blocks\n", mode, mode == 0 ? "ticket per block" : mode == 1 ? "blockIdx" : "ticket per k blocks", ms * 100, n);
        }
    }
    return 0;
}
