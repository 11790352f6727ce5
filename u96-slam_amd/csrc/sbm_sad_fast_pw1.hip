// sbm_sad_fast_pw1.hip -- second translation unit of the interior SAD kernel: the windows 5, 7, 11, 13 (1-column vertical
// sums), reached from launch_sad_fast() through launch_sad_fast_pw1(). A build of sbm_sad_fast.hip like sbm_sad_fast_pp.hip:
// the kernel's ~190 instantiations compile in three parts side by side (make -j) instead of 2 1/2 minutes in one piece.
// gfx950 only.
#define SBM_FAST_TU 1
#include "sbm_sad_fast.hip"
