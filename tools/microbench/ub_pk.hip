// micro-benchmark: issue rate of packed 16-bit integer VALU ops vs 32-bit ones on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define OPK(name, ins) \
template <int DEP> __global__ __launch_bounds__(256) void name(uint32_t* o, uint32_t s, int iters) { \
    uint32_t a[8]; for (int k = 0; k < 8; k++) a[k] = threadIdx.x * 7 + k + s; uint32_t b = s | 0x10001; \
    for (int it = 0; it < iters; it++) { \
        _Pragma("unroll") for (int u = 0; u < 8; u++) { \
            _Pragma("unroll") for (int k = 0; k < 8; k++) { \
                if (DEP) asm volatile(ins " %0, %1, %2" : "=v"(a[0]) : "v"(a[0]), "v"(b)); \
                else asm volatile(ins " %0, %1, %2" : "=v"(a[k]) : "v"(a[k]), "v"(b)); } } } \
    uint32_t r = 0; for (int k = 0; k < 8; k++) r ^= a[k]; o[blockIdx.x * 256 + threadIdx.x] = r; }
OPK(k_max32, "v_max_i32")
OPK(k_and32, "v_and_b32")
OPK(k_add32, "v_add_u32")
OPK(k_pkmax, "v_pk_max_i16")
OPK(k_pkadd, "v_pk_add_i16")
OPK(k_pkaddu, "v_pk_add_u16")
OPK(k_pkmul, "v_pk_mul_lo_u16")
OPK(k_pklshl, "v_pk_lshlrev_b16")
OPK(k_pkaddf16, "v_pk_add_f16")
OPK(k_pkmaxf16, "v_pk_max_f16")
OPK(k_pkminu, "v_pk_min_u16")
template <typename F> void run(const char* nm, F f, uint32_t* d, int waves_per_simd) {
    const int iters = 2000; const int blocks = 256 * waves_per_simd;  // 4 waves per block = 1 per SIMD per CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f<<<blocks, 256>>>(d, 1, 10); hipDeviceSynchronize();
    hipEventRecord(e0); f<<<blocks, 256>>>(d, 1, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)iters * 64;  // per wave
    double cyc = ms * 1e-3 * 2.4e9 / (insts * waves_per_simd);
    printf("%-22s waves/SIMD %d: %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", nm, waves_per_simd, ms, cyc);
}
int main() {
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 4}) {
        run("v_max_i32 indep", k_max32<0>, d, w); run("v_max_i32 dep", k_max32<1>, d, w);
        run("v_and_b32 indep", k_and32<0>, d, w);
        run("v_add_u32 indep", k_add32<0>, d, w);
        run("v_pk_max_i16 indep", k_pkmax<0>, d, w); run("v_pk_max_i16 dep", k_pkmax<1>, d, w);
        run("v_pk_add_i16 indep", k_pkadd<0>, d, w);
        run("v_pk_add_u16 indep", k_pkaddu<0>, d, w);
        run("v_pk_min_u16 indep", k_pkminu<0>, d, w);
        run("v_pk_mul_lo_u16 indep", k_pkmul<0>, d, w);
        run("v_pk_lshlrev_b16 indep", k_pklshl<0>, d, w);
        run("v_pk_add_f16 indep", k_pkaddf16<0>, d, w);
        run("v_pk_max_f16 indep", k_pkmaxf16<0>, d, w); run("v_pk_max_f16 dep", k_pkmaxf16<1>, d, w);
    }
    return 0;
}
