// K1p instantiations for Aligner::local (mod.rs:995-999) with gap_open < 0 and mismatch < 0: the LF flavour
// (sw_fill_pk16.inc) — the fold of the x-suffix clip only where column n is computed, the floor 0 by saturation.
#include "sw_fill_pk16.inc"
namespace bgsw {
sw_fill_fn get_fill_pk16_localfast(int lp, int r, int which) {
    constexpr int XP_ = pk16::CZ, XS_ = pk16::CZ, YP_ = pk16::CZ, YS_ = pk16::CZ;
    constexpr bool LF_ = true;
    BG_PK16_CASE(16, 2) BG_PK16_CASE(16, 3) BG_PK16_CASE(16, 4) BG_PK16_CASE(16, 5) BG_PK16_CASE(16, 6) BG_PK16_CASE(16, 7)
    BG_PK16_CASE(16, 8) BG_PK16_CASE(16, 9) BG_PK16_CASE(16, 10) BG_PK16_CASE(16, 11) BG_PK16_CASE(16, 12)
    BG_PK16_CASE(32, 7) BG_PK16_CASE(32, 8) BG_PK16_CASE(32, 9) BG_PK16_CASE(32, 10) BG_PK16_CASE(32, 11) BG_PK16_CASE(32, 12)
    return nullptr;
}
}  // namespace bgsw
