// One instantiation of K3v2's fill kernel (banded_fill2.inc) per unit: compile time.
#include "banded_fill2.inc"

namespace bgband_dev {
void launch_fill2_narrow(const BandArgs& a, dim3 grid, hipStream_t st) {
    banded_fill2_kernel<BF2_R, BF2_LP, true, false><<<grid, dim3(256), 0, st>>>(a);
}
}  // namespace bgband_dev
