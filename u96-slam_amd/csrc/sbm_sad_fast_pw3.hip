// sbm_sad_fast_pw3.hip -- fourth translation unit of the interior SAD kernel: the windows 29 and 31 (1-column vertical sums,
// 29 / 31 terms: 36 / 34 of a wavefront's 64 lanes produce), reached from launch_sad_fast_pw2() through launch_sad_fast_pw3().
// See sbm_sad_fast_pw1.hip. gfx950 only.
#define SBM_FAST_TU 3
#include "sbm_sad_fast.hip"
