// K1's LF flavour (sw_fill.inc): Aligner::local under MatchParams, scaled keys, reads of one strip.
#include "sw_fill.inc"
namespace bgsw {
sw_fill_fn get_fill_params_lf(int lp, int r) {
#define CASE(LP, R) if (lp == LP && r == R) return sw_fill_kernel<R, LP, SCORE_PARAMS, true, true, true>;
    CASE(16, 2) CASE(16, 4) CASE(16, 6) CASE(16, 8) CASE(16, 10) CASE(16, 12)
    CASE(32, 8) CASE(32, 10) CASE(32, 12)
    CASE(64, 8)
#undef CASE
    return nullptr;
}
}  // namespace bgsw
