// K1p instantiations for Aligner::semiglobal (mod.rs:963-967): x clips MIN_SCORE, y clips 0.
#include "sw_fill_pk16.inc"
namespace bgsw {
sw_fill_fn get_fill_pk16_semiglobal(int lp, int r, int which) {
    constexpr int XP_ = pk16::CI, XS_ = pk16::CI, YP_ = pk16::CZ, YS_ = pk16::CZ;
    constexpr bool LF_ = false;
    BG_PK16_CASE(16, 2) BG_PK16_CASE(16, 4) BG_PK16_CASE(16, 5) BG_PK16_CASE(16, 6) BG_PK16_CASE(16, 8) BG_PK16_CASE(16, 10) BG_PK16_CASE(16, 12)
    BG_PK16_CASE(32, 8) BG_PK16_CASE(32, 10) BG_PK16_CASE(32, 12)
    return nullptr;
}
}  // namespace bgsw
