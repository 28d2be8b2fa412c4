// micro-benchmark: latency of DEPENDENT VALU chains on gfx950 (one wave, then 2 and 4 per SIMD): what a kernel whose step is
// one long dependency chain (the banded fills) pays per instruction
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define OPD(name, body) \
__global__ __launch_bounds__(256) void name(uint32_t* o, uint32_t s, int iters) { \
    uint32_t a = threadIdx.x * 7 + s, b = s | 0x10001, c = s * 3 + 5; (void)c; \
    for (int it = 0; it < iters; it++) { \
        _Pragma("unroll") for (int u = 0; u < 64; u++) { body; } } \
    o[blockIdx.x * 256 + threadIdx.x] = a; }
OPD(d_and, asm volatile("v_and_b32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b)))
OPD(d_add, asm volatile("v_add_u32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b)))
OPD(d_max, asm volatile("v_max_i32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b)))
OPD(d_pkmax, asm volatile("v_pk_max_u16 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b)))
OPD(d_pksub, asm volatile("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(a) : "v"(a), "v"(b)))
OPD(d_pkmad, asm volatile("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(a) : "v"(a), "v"(b), "v"(c)))
OPD(d_bfi, asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(a) : "v"(b), "v"(a), "v"(c)))
OPD(d_dpp, asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(a) : "v"(a)))
OPD(d_dpp_row, asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(a) : "v"(a)))
OPD(d_mix, asm volatile("v_pk_sub_u16 %0, %1, %2 clamp\n v_and_b32 %0, %0, %3" : "=v"(a) : "v"(a), "v"(b), "v"(c)))
OPD(d_dppmax, asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_pk_max_u16 %0, %0, %2" : "=v"(a) : "v"(a), "v"(b)))
template <typename F> void run(const char* nm, F f, uint32_t* d, int w, int per) {
    const int iters = 2000; const int blocks = 256 * w;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f<<<blocks, 256>>>(d, 1, 10); hipDeviceSynchronize();
    hipEventRecord(e0); f<<<blocks, 256>>>(d, 1, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)iters * 64 * per;
    printf("%-26s waves/SIMD %d: %.3f ms  -> %.2f cycles per dependent instruction per wave, %.2f per SIMD\n", nm, w, ms,
           ms * 1e-3 * 2.4e9 / insts, ms * 1e-3 * 2.4e9 / (insts * w));
}
int main() {
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 4}) {
        run("v_and_b32", d_and, d, w, 1);
        run("v_add_u32", d_add, d, w, 1);
        run("v_max_i32", d_max, d, w, 1);
        run("v_pk_max_u16", d_pkmax, d, w, 1);
        run("v_pk_sub_u16 clamp", d_pksub, d, w, 1);
        run("v_pk_mad_u16", d_pkmad, d, w, 1);
        run("v_bfi_b32", d_bfi, d, w, 1);
        run("v_mov_dpp wave_shr", d_dpp, d, w, 1);
        run("v_mov_dpp row_shr", d_dpp_row, d, w, 1);
        run("pk_sub + and", d_mix, d, w, 2);
        run("dpp wave_shr + pk_max", d_dppmax, d, w, 2);
    }
    return 0;
}
