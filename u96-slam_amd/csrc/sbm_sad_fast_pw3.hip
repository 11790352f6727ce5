// sbm_sad_fast_pw3.hip -- translation unit of the interior SAD kernel (sbm_sad_fast.hip): the windows 29, 31 (1-column vertical sums) and 27 (3-column sums).
// The kernel's ~270 instantiations compile in four parts side by side (make -j) instead of several minutes in one piece.
// gfx950 only.
#include "sbm_sad_fast_kernel.h"

namespace sbm {

hipError_t launch_sad_fast_pw3(const FastArgs& a, int wsz, bool border, bool split, hipStream_t s) {
  switch (wsz) {
    case 27: return launch_nd<9, 3>(a, border, split, s);
    case 29: return launch_nd<29, 1>(a, border, split, s);
    case 31: return launch_nd<31, 1>(a, border, split, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace sbm
