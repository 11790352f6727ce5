// sbm_sad_fast_pw1.hip -- translation unit of the interior SAD kernel (sbm_sad_fast.hip): the windows 5, 7, 11, 13 (1-column vertical sums) and 9 (3-column sums).
// The kernel's ~270 instantiations compile in four parts side by side (make -j) instead of several minutes in one piece.
// gfx950 only.
#include "sbm_sad_fast_kernel.h"

namespace sbm {

hipError_t launch_sad_fast_pw1(const FastArgs& a, int wsz, bool border, bool split, hipStream_t s) {
  switch (wsz) {
    case 9: return launch_nd<3, 3>(a, border, split, s);
    case 5: return launch_nd<5, 1>(a, border, split, s);
    case 7: return launch_nd<7, 1>(a, border, split, s);
    case 11: return launch_nd<11, 1>(a, border, split, s);
    case 13: return launch_nd<13, 1>(a, border, split, s);
    default: return launch_sad_fast_pw2(a, wsz, border, split, s);
  }
}

}  // namespace sbm
