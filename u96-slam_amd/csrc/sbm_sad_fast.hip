// sbm_sad_fast.hip -- fast SAD/WTA kernel (placeholder until the mqsad kernel lands).
#include "sbm_common.h"
namespace sbm {
bool sad_fast_supported(const Geom&) { return false; }
hipError_t launch_sad_fast(const uint8_t*, const uint8_t*, int16_t*, int32_t*, const Geom&, int* xa, int* xb, hipStream_t) {
  *xa = *xb = 0;
  return hipSuccess;
}
}  // namespace sbm
