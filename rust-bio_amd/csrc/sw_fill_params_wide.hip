// K1 instantiations for MatchParams scoring (Scoring::from_scores, mod.rs:259-278).
#include "sw_fill.inc"
namespace bgsw {
template <bool LOCAL, bool NARROW>
static sw_fill_fn pick(int lp, int r) {
#define CASE(LP, R) if (lp == LP && r == R) return sw_fill_kernel<R, LP, SCORE_PARAMS, LOCAL, NARROW>;
    CASE(16, 2) CASE(16, 4) CASE(16, 6) CASE(16, 8) CASE(16, 10) CASE(16, 12)
    CASE(32, 8) CASE(32, 10) CASE(32, 12)
    CASE(64, 8)
#undef CASE
    return nullptr;
}
sw_fill_fn get_fill_params_wide(int lp, int r, bool local) { return local ? pick<true, false>(lp, r) : pick<false, false>(lp, r); }
}  // namespace bgsw
