// K1 instantiations for tabulated match functions under Aligner::local (all four clip penalties 0, mod.rs:995-999):
// the LOCAL specialisation of the cell (no clip adds, no y-prefix-clip candidate) with the score table in LDS —
// what a BLOSUM62 / PAM local alignment of protein reads runs.
#include <type_traits>
#include "sw_fill.inc"
namespace bgsw {
sw_fill_fn get_fill_matrix_local(int lp, int r) {
#define CASE(LP, R) if (lp == LP && r == R) return sw_fill_kernel<R, LP, SCORE_LDS, true, true>;
    CASE(16, 6) CASE(16, 8) CASE(16, 10) CASE(16, 12) CASE(32, 12) CASE(64, 8)
#undef CASE
    return nullptr;
}
}  // namespace bgsw
