// sbm_sad_fast_pw2.hip -- translation unit of the interior SAD kernel (sbm_sad_fast.hip): the windows 17, 19, 23, 25 (1-column vertical sums).
// The kernel's ~270 instantiations compile in four parts side by side (make -j) instead of several minutes in one piece.
// gfx950 only.
#include "sbm_sad_fast_kernel.h"

namespace sbm {

hipError_t launch_sad_fast_pw2(const FastArgs& a, int wsz, bool border, bool split, hipStream_t s) {
  switch (wsz) {
    case 17: return launch_nd<17, 1>(a, border, split, s);
    case 19: return launch_nd<19, 1>(a, border, split, s);
    case 23: return launch_nd<23, 1>(a, border, split, s);
    case 25: return launch_nd<25, 1>(a, border, split, s);
    default: return launch_sad_fast_pw3(a, wsz, border, split, s);
  }
}

}  // namespace sbm
