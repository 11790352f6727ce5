// sbm_sad_fast_pp.hip -- the FALLBACK build of the interior SAD kernel: two accumulator arrays in ping-pong (no
// v_mqsad_pk_u16_u8 ever writes a register it reads: what LLVM's early-clobber constraint prescribes), the register-staged
// strip of sbm_sad_fast_pp_strip.h. Taken when the device self-test of the in-place accumulate (mqsad_inplace_ok(),
// sbm_sad_fast.hip) does not pass, or with SBM_FAST_INPLACE=0 (GPU tests). Windows 5..27, up to 256 disparities in the
// 64-disparity cooperating layouts, masked-count kernels only; sad_fast_supported() hands windows 29 / 31 and 257..512
// disparities to the sliding-sum kernel on such a device (8-25x slower: include/sbm.h says so). gfx950 only.
#define SBM_FAST_PINGPONG 1
#include "sbm_sad_fast_kernel.h"

namespace sbm {

hipError_t launch_sad_fast_pp(const FastArgs& a, int wsz, bool border, bool split, hipStream_t s) {
  switch (wsz) {
    case 9: return launch_nd<3, 3>(a, border, split, s);
    case 15: return launch_nd<5, 3>(a, border, split, s);
    case 21: return launch_nd<7, 3>(a, border, split, s);
    case 27: return launch_nd<9, 3>(a, border, split, s);
    case 5: return launch_nd<5, 1>(a, border, split, s);
    case 7: return launch_nd<7, 1>(a, border, split, s);
    case 11: return launch_nd<11, 1>(a, border, split, s);
    case 13: return launch_nd<13, 1>(a, border, split, s);
    case 17: return launch_nd<17, 1>(a, border, split, s);
    case 19: return launch_nd<19, 1>(a, border, split, s);
    case 23: return launch_nd<23, 1>(a, border, split, s);
    case 25: return launch_nd<25, 1>(a, border, split, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace sbm
