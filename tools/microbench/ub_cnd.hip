#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE> __global__ __launch_bounds__(256) void k(uint32_t* o, uint32_t s, int iters) {
    uint32_t a[8]; for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 7 + i + s; uint32_t b = s | 0x10001;
    uint64_t m = (threadIdx.x & 1) ? 0x5555555555555555ull : 0x3333333333333333ull;
    m = __builtin_amdgcn_readfirstlane((int)m) | ((uint64_t)__builtin_amdgcn_readfirstlane((int)(m >> 32)) << 32);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(b), "s"(m));
                if (MODE == 1) asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(a[i]), "v"(b) : );
                if (MODE == 2) asm volatile("v_cmp_lt_u32_e64 %1, %2, %3\n v_cndmask_b32_e64 %0, %2, %3, %1" : "=v"(a[i]), "=s"(m) : "v"(a[i]), "v"(b));
                if (MODE == 3) asm volatile("v_cmp_lt_u32_e32 vcc, %1, %2" : "+v"(a[i]) : "v"(a[(i+1)&7]), "v"(b) : "vcc");
                if (MODE == 4) asm volatile("v_cmp_lt_u32_e64 %0, %1, %2" : "=s"(m) : "v"(a[i]), "v"(b));
                if (MODE == 6) asm volatile("v_cmp_lt_u32_e32 vcc, %1, %2\n v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(a[i]), "v"(b) : "vcc");
                if (MODE == 7) asm volatile("v_cmp_lt_u32_e32 vcc, %1, %2\n v_cndmask_b32_e32 %0, %1, %2, vcc\n v_cndmask_b32_e32 %3, %3, %2, vcc" : "=v"(a[i]), "+v"(a[(i+4)&7]) : "v"(a[i]), "v"(b) : "vcc");
                if (MODE == 8) asm volatile("v_max_i32 %0, %1, %2\n v_and_b32 %0, %0, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
                if (MODE == 9) asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(b), "v"(a[(i+4)&7]));
                if (MODE == 10) asm volatile("v_sub_u32 %0, %1, %2\n v_ashrrev_i32 %0, 31, %0" : "=v"(a[i]) : "v"(a[i]), "v"(b));
                if (MODE == 5) asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(b), "v"(a[i]), "v"(a[(i+1)&7]));
            }
        }
    }
    uint32_t r = (uint32_t)m; for (int i = 0; i < 8; i++) r ^= a[i]; o[blockIdx.x * 256 + threadIdx.x] = r;
}
template <typename F> void run(const char* nm, F f, uint32_t* d, int w, int per) {
    const int iters = 2000; const int blocks = 256 * w;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f<<<blocks, 256>>>(d, 1, 10); hipDeviceSynchronize();
    hipEventRecord(e0); f<<<blocks, 256>>>(d, 1, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %.3f ms -> %.2f cycles per wave-instruction per SIMD\n", nm, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 64 * per * w));
}
int main() {
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    int w = 4;
    run("v_cndmask_b32_e64 (sgpr pair)", k<0>, d, w, 1);
    run("v_cndmask_b32_e32 (vcc)", k<1>, d, w, 1);
    run("v_cmp_e64 + v_cndmask_e64 (dependent pair)", k<2>, d, w, 2);
    run("v_cmp_lt_u32_e32 -> vcc", k<3>, d, w, 1);
    run("v_cmp_lt_u32_e64 -> sgpr", k<4>, d, w, 1);
    run("v_bfi_b32", k<5>, d, w, 1);
    run("v_cmp_e32 vcc + v_cndmask_e32 (pair)", k<6>, d, w, 2);
    run("v_cmp_e32 vcc + 2 x v_cndmask_e32 (triple)", k<7>, d, w, 3);
    run("v_max_i32 + v_and_b32 dependent (pair)", k<8>, d, w, 2);
    run("v_max3_i32 (3 vgprs)", k<9>, d, w, 1);
    run("v_sub_u32 + v_ashrrev_i32 dependent (pair)", k<10>, d, w, 2);
    return 0;
}
