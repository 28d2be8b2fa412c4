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
#define OPK3(name, ins) \
template <int DEP> __global__ __launch_bounds__(256) void name(uint32_t* o, uint32_t s, int iters) { \
    uint32_t a[8]; for (int k = 0; k < 8; k++) a[k] = threadIdx.x * 7 + k + s; uint32_t b = s | 0x10001, c = s * 3 + 5; \
    for (int it = 0; it < iters; it++) { \
        _Pragma("unroll") for (int u = 0; u < 8; u++) { \
            _Pragma("unroll") for (int k = 0; k < 8; k++) { \
                if (DEP) asm volatile(ins " %0, %1, %2, %3" : "=v"(a[0]) : "v"(a[0]), "v"(b), "v"(c)); \
                else asm volatile(ins " %0, %1, %2, %3" : "=v"(a[k]) : "v"(a[k]), "v"(b), "v"(c)); } } } \
    uint32_t r = 0; for (int k = 0; k < 8; k++) r ^= a[k]; o[blockIdx.x * 256 + threadIdx.x] = r; }
OPK(k2_0, "v_max_i32")
OPK(k2_1, "v_max_u32")
OPK(k2_2, "v_min_i32")
OPK(k2_3, "v_and_b32")
OPK(k2_4, "v_or_b32")
OPK(k2_5, "v_xor_b32")
OPK(k2_6, "v_add_u32")
OPK(k2_7, "v_sub_u32")
OPK(k2_8, "v_lshlrev_b32")
OPK(k2_9, "v_ashrrev_i32")
OPK(k2_10, "v_max_f32")
OPK(k2_11, "v_add_f32")
OPK(k2_12, "v_mul_f32")
OPK(k2_13, "v_max_i16")
OPK(k2_14, "v_max_u16")
OPK(k2_15, "v_add_u16")
OPK(k2_16, "v_pk_max_i16")
OPK(k2_17, "v_pk_add_i16")
OPK(k2_18, "v_pk_max_u16")
OPK(k2_19, "v_mul_lo_u32")
OPK(k2_20, "v_mul_u32_u24")
OPK3(k3_0, "v_max3_i32")
OPK3(k3_1, "v_max3_u32")
OPK3(k3_2, "v_max3_f32")
OPK3(k3_3, "v_med3_i32")
OPK3(k3_4, "v_bfi_b32")
OPK3(k3_5, "v_and_or_b32")
OPK3(k3_6, "v_or3_b32")
OPK3(k3_7, "v_lshl_or_b32")
OPK3(k3_8, "v_add3_u32")
OPK3(k3_9, "v_lshl_add_u32")
OPK3(k3_10, "v_xad_u32")
OPK3(k3_11, "v_perm_b32")
OPK3(k3_12, "v_mad_u32_u24")
OPK3(k3_13, "v_mad_i32_i24")
OPK3(k3_14, "v_pk_mad_u16")
OPK3(k3_15, "v_pk_mad_i16")
OPK3(k3_16, "v_fma_f32")
OPK3(k3_17, "v_max3_i16")
OPK3(k3_18, "v_add_lshl_u32")
OPK3(k3_19, "v_alignbit_b32")
OPK3(k3_20, "v_bfe_u32")
OPK3(k3_21, "v_bfe_i32")
OPK3(k3_22, "v_sad_u32")

template <int DEP> __global__ __launch_bounds__(256) void k_dpp(uint32_t* o, uint32_t s, int iters) {
    uint32_t a[8]; for (int k = 0; k < 8; k++) a[k] = threadIdx.x * 7 + k + s;
    for (int it = 0; it < iters; it++) {
        _Pragma("unroll") for (int u = 0; u < 8; u++) {
            _Pragma("unroll") for (int k = 0; k < 8; k++) {
                asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(a[k]) : "v"(a[(k+1)&7])); } } }
    uint32_t r = 0; for (int k = 0; k < 8; k++) r ^= a[k]; o[blockIdx.x * 256 + threadIdx.x] = r; }
template <int DEP> __global__ __launch_bounds__(256) void k_cnd(uint32_t* o, uint32_t s, int iters) {
    uint32_t a[8]; for (int k = 0; k < 8; k++) a[k] = threadIdx.x * 7 + k + s; uint32_t b = s | 0x10001;
    for (int it = 0; it < iters; it++) {
        _Pragma("unroll") for (int u = 0; u < 8; u++) {
            _Pragma("unroll") for (int k = 0; k < 8; k++) {
                asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[k]) : "v"(a[k]), "v"(b)); } } }
    uint32_t r = 0; for (int k = 0; k < 8; k++) r ^= a[k]; o[blockIdx.x * 256 + threadIdx.x] = r; }
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
    for (int w : {2, 4}) {
        run("v_max_i32", k2_0<0>, d, w);
        run("v_max_u32", k2_1<0>, d, w);
        run("v_min_i32", k2_2<0>, d, w);
        run("v_and_b32", k2_3<0>, d, w);
        run("v_or_b32", k2_4<0>, d, w);
        run("v_xor_b32", k2_5<0>, d, w);
        run("v_add_u32", k2_6<0>, d, w);
        run("v_sub_u32", k2_7<0>, d, w);
        run("v_lshlrev_b32", k2_8<0>, d, w);
        run("v_ashrrev_i32", k2_9<0>, d, w);
        run("v_max_f32", k2_10<0>, d, w);
        run("v_add_f32", k2_11<0>, d, w);
        run("v_mul_f32", k2_12<0>, d, w);
        run("v_max_i16", k2_13<0>, d, w);
        run("v_max_u16", k2_14<0>, d, w);
        run("v_add_u16", k2_15<0>, d, w);
        run("v_pk_max_i16", k2_16<0>, d, w);
        run("v_pk_add_i16", k2_17<0>, d, w);
        run("v_pk_max_u16", k2_18<0>, d, w);
        run("v_mul_lo_u32", k2_19<0>, d, w);
        run("v_mul_u32_u24", k2_20<0>, d, w);
        run("v_max3_i32", k3_0<0>, d, w);
        run("v_max3_u32", k3_1<0>, d, w);
        run("v_max3_f32", k3_2<0>, d, w);
        run("v_med3_i32", k3_3<0>, d, w);
        run("v_bfi_b32", k3_4<0>, d, w);
        run("v_and_or_b32", k3_5<0>, d, w);
        run("v_or3_b32", k3_6<0>, d, w);
        run("v_lshl_or_b32", k3_7<0>, d, w);
        run("v_add3_u32", k3_8<0>, d, w);
        run("v_lshl_add_u32", k3_9<0>, d, w);
        run("v_xad_u32", k3_10<0>, d, w);
        run("v_perm_b32", k3_11<0>, d, w);
        run("v_mad_u32_u24", k3_12<0>, d, w);
        run("v_mad_i32_i24", k3_13<0>, d, w);
        run("v_pk_mad_u16", k3_14<0>, d, w);
        run("v_pk_mad_i16", k3_15<0>, d, w);
        run("v_fma_f32", k3_16<0>, d, w);
        run("v_max3_i16", k3_17<0>, d, w);
        run("v_add_lshl_u32", k3_18<0>, d, w);
        run("v_alignbit_b32", k3_19<0>, d, w);
        run("v_bfe_u32", k3_20<0>, d, w);
        run("v_bfe_i32", k3_21<0>, d, w);
        run("v_sad_u32", k3_22<0>, d, w);
        run("v_mov_b32_dpp wave_shr", k_dpp<0>, d, w);
        run("v_cndmask_b32 vcc", k_cnd<0>, d, w);
    }
    return 0;
}
