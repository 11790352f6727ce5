// sbm_sad_fast_pp.hip -- second build of the interior SAD kernel with TWO accumulator arrays in ping-pong (no
// v_mqsad_pk_u16_u8 ever writes a register it reads: what LLVM's early-clobber constraint prescribes). Used when the
// device self-test of the in-place accumulate (sbm_sad_fast.hip, mqsad_inplace_ok) does not pass, or with
// SBM_FAST_INPLACE=0 for A/B measurements. gfx950 only.
#define SBM_FAST_PINGPONG 1
#include "sbm_sad_fast.hip"
