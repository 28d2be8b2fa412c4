// K1 instantiations for tabulated match functions (closures / BLOSUM / PAM, scores/mod.rs):
// SCORE_LDS keeps the compacted A x A table in LDS, SCORE_GLOBAL (A > 64) reads it from HBM/L2.
#include <type_traits>
#include "sw_fill.inc"
namespace bgsw {
sw_fill_fn get_fill_matrix_local(int lp, int r);  // sw_fill_matrix_local.hip
sw_fill_fn get_fill_matrix(int lp, int r, int sm, bool narrow, bool local) {
    if (local && narrow && sm == SCORE_LDS)
        if (sw_fill_fn f = get_fill_matrix_local(lp, r)) return f;
#define CASE(LP, R)                                                                   \
    if (lp == LP && r == R)                                                           \
        return sm == SCORE_LDS ? (narrow ? sw_fill_kernel<R, LP, SCORE_LDS, false, true> : sw_fill_kernel<R, LP, SCORE_LDS, false, false>) \
                               : (narrow ? sw_fill_kernel<R, LP, SCORE_GLOBAL, false, true> : sw_fill_kernel<R, LP, SCORE_GLOBAL, false, false>);
    CASE(16, 6) CASE(16, 12) CASE(32, 12) CASE(64, 8)
#undef CASE
#define CASE(LP, R) \
    if (lp == LP && r == R && sm == SCORE_LDS && narrow) return sw_fill_kernel<R, LP, SCORE_LDS, false, true>;
    CASE(16, 8) CASE(16, 10)
#undef CASE
    return nullptr;
}
}  // namespace bgsw
