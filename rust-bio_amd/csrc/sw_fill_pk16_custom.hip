// K1p instantiations for Aligner::custom (mod.rs:591): any clip penalties.
#include "sw_fill_pk16.inc"
namespace bgsw {
sw_fill_fn get_fill_pk16_custom(int lp, int r, int which) {
    constexpr int XP_ = pk16::CF, XS_ = pk16::CF, YP_ = pk16::CF, YS_ = pk16::CF;
    constexpr bool LF_ = false;
    BG_PK16_CASE(16, 2) BG_PK16_CASE(16, 4) BG_PK16_CASE(16, 5) BG_PK16_CASE(16, 6) BG_PK16_CASE(16, 8) BG_PK16_CASE(16, 10) BG_PK16_CASE(16, 12)
    BG_PK16_CASE(32, 8) BG_PK16_CASE(32, 10) BG_PK16_CASE(32, 12)
    return nullptr;
}
}  // namespace bgsw
