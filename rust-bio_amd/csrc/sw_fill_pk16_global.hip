// K1p instantiations for Aligner::global (mod.rs:934-938): all four clip penalties MIN_SCORE.
#include "sw_fill_pk16.inc"
namespace bgsw {
sw_fill_fn get_fill_pk16_global(int lp, int r, int which) {
    constexpr int XP_ = pk16::CI, XS_ = pk16::CI, YP_ = pk16::CI, YS_ = pk16::CI;
    constexpr bool LF_ = false;
    BG_PK16_CASE(16, 2) BG_PK16_CASE(16, 4) BG_PK16_CASE(16, 5) BG_PK16_CASE(16, 6) BG_PK16_CASE(16, 8) BG_PK16_CASE(16, 10) BG_PK16_CASE(16, 12)
    BG_PK16_CASE(32, 8) BG_PK16_CASE(32, 10) BG_PK16_CASE(32, 12)
    return nullptr;
}
}  // namespace bgsw
