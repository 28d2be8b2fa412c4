// placeholder until the banded aligner (K3/K4) lands
#include "bg_common.h"
extern "C" int bg_align_banded_batch(bg_ctx*, const bg_scoring_t*, int, uint32_t, uint32_t, uint64_t,
                                     const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*,
                                     bg_alignment_t*, uint8_t*, uint64_t, uint64_t*, uint64_t*) {
    return BG_ERR_UNSUPPORTED;
}
