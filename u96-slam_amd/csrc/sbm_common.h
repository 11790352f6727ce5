// sbm_common.h -- shared declarations of the HIP stereo block-matching engine (internal, gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sbm.h"

#include <stdlib.h>

namespace sbm {

// Environment switches of the library -- the complete list, documented for integrators in include/sbm.h ("Environment").
//   env_switch(): read in every build. They select a tested fallback or a code path that the GPU tests compare with the
//                 default one: SBM_FAST_INPLACE, SBM_FAST_PFSHIFT, SBM_FAST_CS3, SBM_SPECKLE_LISTS, SBM_SPECKLE_BAND,
//                 SBM_SPECKLE_SEG, SBM_HOST_ZEROCOPY, SBM_WIDE, SBM_CV_READING.
//   SBM_TUNE():   tuning knobs behind the sweeps of tools/exp (SBM_FAST_NSEG, SBM_FAST_TAPER,
//                 SBM_FAST_UNIQ_PLAIN, SBM_FAST_SPLIT, SBM_PF_ROWS, SBM_HOST_CHUNK, SBM_HOST_PIPELINE, SBM_DEV_*): compiled in
//                 only with -DSBM_DEV (the development library of tools/exp/r05_devlib.sh; sbm_sad_fast_dev.h lists the interior
//                 kernel's); the product ignores them.
inline int env_switch(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
#ifdef SBM_DEV
#define SBM_TUNE(name, dflt) ::sbm::env_switch(name, dflt)
#else
#define SBM_TUNE(name, dflt) (dflt)
#endif

// Geometry of one launch, derived on the host from sbm_params and the image size. Naming follows
// cv::StereoBM (calib3d stereobm.cpp): lofs/rofs/width1, buffer index d <-> true disparity nd-1-d+mindisp.
struct Geom {
  int W, H;            // image size
  int n;               // pairs in the batch
  int pitch;           // byte pitch of the padded prefiltered planes
  int padl;            // bytes of left padding in front of column 0 of a prefiltered row
  int plane;           // bytes per prefiltered plane (pitch * H)
  int nd, mindisp, wsz, w2, cap;
  int lofs, rofs, width1, xend;  // xend = min(width1, W - lofs): computed columns are X = lofs + x, x in [0,xend)
  int tex, uniq;
  int filtered;        // (mindisp - 1) * 16
  int row0, row1;      // valid-ROI rows [row0,row1)
  int col0, col1;      // valid-ROI columns [col0,col1)
  int want_cost;       // disp12_max_diff >= 0
  int cost16;          // the cost plane holds uint16 (fast + border kernels: sums <= 65534) instead of int32
  int pfshift;         // the prefiltered planes hold (value << pfshift) + 1 (0 unless the fast path asks for it)
  int reading;         // SBM_CV_READING: alternative readings of cv::StereoBM behaviours nobody could pin (kRead* bits; 0 = default)
};

// The bits of SBM_CV_READING (include/sbm.h): every behaviour of cv::StereoBM that this engine restates from
// memory and that the reference's data cannot pin has its alternative reading behind one bit here AND in the CPU oracle
// (sbmo_set_reading), so that the day somebody runs tools/verify_with_opencv.py the fix is a default flip.
constexpr int kReadRoiMinusMinD = 1;      // getValidDisparityROI: xmax = min(roi1 right edge, roi2 right edge - minDisparity) - w/2 (2.4 lineage)
constexpr int kReadCostShort = 2;         // validateDisparity sees the block-matching cost plane as `short` (wraps beyond 32767)
constexpr int kReadSpeckleX16 = 4;        // filterSpeckles receives speckleRange * 16 (StereoSGBM's convention)
constexpr int kReadOddRowComputed = 8;    // prefilterXSobel: the last row of an odd-height image is computed, not filled with cap
constexpr int kReadLrTieLater = 16;       // validateDisparity: on equal cost the LATER x takes the slot

// Prefiltered planes store value+1 (range 1..2*cap+1 <= 127) so that 0 can act as the "masked byte" of
// v_mqsad_pk_u16_u8; padding bytes are 0. sbm_debug_fetch() removes the bias again.
// With Geom::pfshift = 2 the planes store 4*value+1 (<= 253): every absolute difference, hence every SAD, is a multiple
// of 4, which leaves the two low bits of the packed 16-bit sums free for a register tag in the interior kernel's
// winner search (sbm_sad_fast.hip; chosen by sad_fast_pfshift() when 4*maxS still fits 16 bits; pfshift = 1: 2*value+1,
// one tag bit, where only 2*maxS fits). The border wavefronts of the same launch work on the scaled sums too; only the
// uniqueness threshold and the stored cost go back to the unscaled sum.
constexpr int kPfBias = 1;

hipError_t launch_prefilter(const uint8_t* d_left, const uint8_t* d_right, uint8_t* pf_l, uint8_t* pf_r,
                            const Geom& g, hipStream_t s);

// PREFILTER_NORMALIZED_RESPONSE (cv prefilterNorm); vsum = 2*n*W*H uint16 scratch (column sums).
hipError_t launch_prefilter_norm(const uint8_t* d_left, const uint8_t* d_right, uint8_t* pf_l, uint8_t* pf_r, uint16_t* vsum,
                                 const Geom& g, int winsize, hipStream_t s);

// Generic SAD/WTA for output columns x in [xa,xb) (x relative to lofs) and rows [row0,row1): any block size,
// any disparity count, clamped border windows. Used for the border columns and as the fallback path.
hipError_t launch_sad_generic(const uint8_t* pf_l, const uint8_t* pf_r, int16_t* disp, int32_t* cost, const Geom& g,
                              int xa, int xb, hipStream_t s);

// The same definition and outputs with sliding sums in both directions (sbm_sad_wide.hip): the fallback outside the fast
// envelope for up to 2048 disparities; SBM_WIDE=0 selects the per-column kernel above instead.
bool sad_wide_supported(const Geom& g);
hipError_t launch_sad_wide(const uint8_t* pf_l, const uint8_t* pf_r, int16_t* disp, int32_t* cost, const Geom& g, int xa, int xb,
                           hipStream_t s);

// Fast path (interior columns, odd block sizes 5..31, up to 512 disparities, 16-bit sums: sad_fast_supported()). Outside its
// envelope the launch does nothing and returns hipSuccess; *xa,*xb receive the column range it covered.
// name of the SAD kernel instantiation of the calling thread's last launch (template tuple; sbm_last_kernel_name())
extern thread_local char g_sad_kernel_name[96];
bool sad_fast_supported(const Geom& g);
bool mqsad_inplace_ok(hipStream_t s);   // device self-test behind the in-place v_mqsad accumulate (cached per device)
constexpr int kFastNdMax = 512;   // disparities the interior kernel takes (four cooperating 128-disparity wavefronts)
bool sad_fast_borders_in_launch(const Geom& g);   // the clamped border columns ride in the interior launch (up to 256 disparities)
int sad_fast_pfshift(const Geom& g);   // 2 or 1 when the interior kernel wants pre-scaled planes (see kPfBias), else 0
// border: the w/2 clamped columns on each side of [xa,xb) are computed by extra wavefronts of the same launch
// (sbm_sad_border_wave.h); without it the launch leaves them untouched.
hipError_t launch_sad_fast(const uint8_t* pf_l, const uint8_t* pf_r, int16_t* disp, int32_t* cost, const Geom& g,
                           int* xa, int* xb, bool border, hipStream_t s);

// Left-right consistency (cv validateDisparity) + invalid rows/columns fill. Reads disp_pre/cost, writes disp_out.
hipError_t launch_lrcheck(const int16_t* disp_pre, const int32_t* cost, int16_t* disp_out, const Geom& g,
                          int disp12_max_diff, hipStream_t s);

// cv filterSpeckles as parallel connected components (union-find). Scratch, for n pairs of W x H (the band walk cuts a row into
// up to kSpkMaxSeg column segments with books of their own, hence the padding): runs: 16 * n * H * (W + kSpkRecordPad) bytes (run
// records of the band walk, or the per-pixel labels + sizes of the row-walking kernels); nheads: n * H * kSpkMaxSeg int32 (runs
// per row and segment; null -> row-walking kernels); seam: n * ceil(H/2) * (W + kSpkSeamPad) uint32, nseam: n * ceil(H/2) *
// kSpkMaxSeg int32 (contacts across band seams).
constexpr int kSpkMaxSeg = 4, kSpkRecordPad = 288, kSpkSeamPad = 512;
hipError_t launch_speckle(int16_t* disp, void* runs, int32_t* nheads, uint32_t* seam, int32_t* nseam, const Geom& g, int max_size,
                          int max_diff, hipStream_t s);

// Stand-alone prefilter of dense images (either flavour) and the rectifier in front of it (sbm_rectify.hip).
hipError_t launch_prefilter_dense(const uint8_t* d_src, uint8_t* d_dst, int n, int W, int H, int rtl, int cap,
                                  hipStream_t s);
hipError_t launch_rect_map(const sbm_rect_cam& cam, int W, int H, int16_t* d_map, hipStream_t s);
hipError_t launch_rect_remap(const uint8_t* d_src, const int16_t* d_map, uint8_t* d_dst, int n, int W, int H,
                             hipStream_t s);

// FPGA-flavour matcher (sbm_fpga.hip). rec: n*sad_hgt*sad_wdt*8 bytes (touched beyond 128 disparities only); flag: n ints,
// zero when allocated; gen: grows with every call on these buffers (> 0).
hipError_t launch_fpga_bm(const uint8_t* xl, const uint8_t* xr, void* rec, int* flag, int gen, int16_t* disp, int n,
                          const sbm_fpga_params& p, hipStream_t s);

// GFTT minimum-eigenvalue map of the PL (sbm_gftt.hip): eig = n*H*W uint16, maxv = n uint32 (`Max` register per image).
hipError_t launch_gftt_eig(const uint8_t* img, uint16_t* eig, unsigned* maxv, int n, int W, int H, hipStream_t s);

// Consumers of the map (sbm_consume.hip): decimation, reprojection, keypoint depth.
hipError_t launch_disp_to_float(const int16_t* disp, float* out, size_t count, hipStream_t s);
hipError_t launch_decimate(const int16_t* disp, int16_t* out, int n, int W, int H, int scale, hipStream_t s);
hipError_t launch_reproject(const int16_t* disp, float* xyz, int n, int W, int H, int scale, const sbm_stereo_model& m,
                            int apply_local, hipStream_t s);
hipError_t launch_keypoints3d(const int16_t* disp, const float* kp, float* xyz, int W, int H, int nk,
                              const sbm_stereo_model& m, float min_depth, float max_depth, hipStream_t s);

}  // namespace sbm
