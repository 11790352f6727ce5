// sbm_sad_fast_pw2.hip -- third translation unit of the interior SAD kernel: the windows 17, 19, 23, 25 (1-column vertical
// sums), reached from launch_sad_fast_pw1() through launch_sad_fast_pw2(). See sbm_sad_fast_pw1.hip. gfx950 only.
#define SBM_FAST_TU 2
#include "sbm_sad_fast.hip"
