// sbm_sad_fast_dev.h -- every development knob of the interior SAD kernel's host side, in one place. Product builds see the
// constants below; a development build (-DSBM_DEV: tools/exp/r05_devlib.sh) reads them once from the environment. Compile-time:
// -DSBM_DEV_FEW instantiates the bench workloads' windows (15, 21) only, which is what makes one-macro A/B builds quick.
#pragma once
#include "sbm_common.h"

namespace sbm {

struct FastTune {
  int nseg;         // SBM_FAST_NSEG         force the row-segment count of the interior strips (0 = the model in launch_sad_fast)
  int taper;        // SBM_FAST_TAPER        0 = equal row segments instead of the tapered tail
  int uniq_plain;   // SBM_FAST_UNIQ_PLAIN   0 = saturating deficit sums everywhere
  int split;        // SBM_FAST_SPLIT        0 = one-pair launches keep the regular layouts
  int seg_c1000;    // SBM_DEV_SEG_C         1000 x the constant c of the segment model
  int small_rows;   // SBM_DEV_SMALL_ROWS    shortest row segment of a launch that does not fill the chip
  int fill;         // SBM_DEV_FILL          workgroups such a launch is cut into
  int bseg;         // SBM_DEV_BSEG          rows per border row segment (0 = the model in launch_t)
  int print;        // SBM_DEV_PRINT         print the launch geometry
  int border_only;  // SBM_DEV_BORDER_ONLY   launch the border wavefronts alone (timing; results are wrong by construction)
};

inline const FastTune& fast_tune() {
  static const FastTune t = {SBM_TUNE("SBM_FAST_NSEG", 0), SBM_TUNE("SBM_FAST_TAPER", 1), SBM_TUNE("SBM_FAST_UNIQ_PLAIN", 1), SBM_TUNE("SBM_FAST_SPLIT", 1),
                             SBM_TUNE("SBM_DEV_SEG_C", 196), SBM_TUNE("SBM_DEV_SMALL_ROWS", 8), SBM_TUNE("SBM_DEV_FILL", 5000), SBM_TUNE("SBM_DEV_BSEG", 0),
                             SBM_TUNE("SBM_DEV_PRINT", 0), SBM_TUNE("SBM_DEV_BORDER_ONLY", 0)};
  return t;
}

}  // namespace sbm
