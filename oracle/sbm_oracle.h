/*
 * sbm_oracle.h -- CPU restatement of the stereo block-matching path. TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this library.
 * It is never linked into libsbm_hip.so and the product path never falls back to it.
 *
 * What it restates
 *   The reference obtains its dense disparity from cv::StereoBM (src/slam/src/core/main.cpp:197-217).
 *   That algorithm lives in OpenCV calib3d (modules/calib3d/src/stereobm.cpp, plus validateDisparity /
 *   filterSpeckles / getValidDisparityROI in stereosgbm.cpp), a third-party dependency that is NOT
 *   vendored under /root/reference and whose version the reference does not pin (API usage implies
 *   OpenCV >= 3.0; restated here from the 4.x sources). Each function below names the OpenCV routine
 *   it follows.
 *
 * Pinning status
 *   * Prefilter: PINNED by the reference's own golden vectors. data/ref_xsbl_{l,r} is the RTL x-Sobel
 *     (src/dvp/rtl/xsbl2.v:183-198,661-874) of data/ref_rect_{l,r}; sbmo_prefilter_xsobel_fpga()
 *     reproduces it exactly and the OpenCV-flavour prefilter satisfies
 *     cv[y][x] == max(fpga[y][x]-1, 0) on the interior (tests/test_oracle_golden.py).
 *   * Block-matching output (SAD/WTA/uniqueness/LR/speckle): PARITY UNPINNED -- the reference holds no
 *     golden disparity map and OpenCV cannot be built or imported in this environment. The restatement
 *     is cross-checked only against an independent brute-force evaluation of the same definitions.
 *     Recall risks (SURVEY.md A.7 and one more), each with a case in the pin kit (tests/golden/pin_kit.npz,
 *     tools/verify_with_opencv.py settles them on any box with OpenCV):
 *       1. speckleRange is compared unscaled (1/16 px units) by filterSpeckles for BM output;
 *       2. the BM cost plane is `short` in OpenCV: sums above 32767 (w^2 * 2 * cap) wrap there, not here;
 *       3. odd image height: the last prefilter row is `cap`;
 *       4. validateDisparity: strict '>' claim, minX1 = max(minD + nd, 0) start column;
 *       5. IPP builds may route filterSpeckles through IPP (claimed identical);
 *       6. non-empty ROI1 / ROI2 change the valid ROI;
 *       7. getValidDisparityROI: the 2.4 lineage subtracts minDisparity in xmax (roi2.x + roi2.width - minD),
 *          this restatement follows 4.x and does not -- invisible at minDisparity 0 (the reference's call site
 *          and every BASELINE config), decides the right border otherwise.
 */
#ifndef SBM_ORACLE_H_
#define SBM_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#include "../include/sbm.h" /* sbm_params + status codes (shared vocabulary, no code) */

#ifdef __cplusplus
extern "C" {
#endif

/* OpenCV prefilterXSobel (stereobm.cpp). src/dst strides in bytes. */
void sbmo_prefilter_xsobel(const uint8_t* src, size_t sstride, uint8_t* dst, size_t dstride, int width, int height,
                           int cap);

/* OpenCV prefilterNorm (PREFILTER_NORMALIZED_RESPONSE). Not used by the reference; parity unpinned. */
void sbmo_prefilter_norm(const uint8_t* src, size_t sstride, uint8_t* dst, size_t dstride, int width, int height,
                         int winsize, int cap);

/* FPGA flavour: src/dvp/rtl/xsbl2.v (limit() 185-198, horizontal diff 684-698, line buffers 787-857,
 * edge substitution 869-872, row index hcnt-1 1020-1025). Rows 0 and H-1 are left at `fill` (the
 * testbench zero-fills them, src/dvp/sim/sim_dvp.v:524-542). */
void sbmo_prefilter_xsobel_fpga(const uint8_t* src, size_t sstride, uint8_t* dst, size_t dstride, int width,
                                int height, uint8_t fill);

/* OpenCV getValidDisparityROI (stereosgbm.cpp). roi = {x,y,w,h}; out all-zero when empty. */
/* Alternative readings of cv::StereoBM behaviours recalled from memory that nothing in the reference pins (SURVEY.md A.7 +
 * getValidDisparityROI's 2.4 lineage). 0 = the default reading. Process-global; the engine has the same bits (SBM_CV_READING). */
#define SBMO_READ_ROI_MINUS_MIND 1     /* getValidDisparityROI: roi2's right edge loses minDisparity */
#define SBMO_READ_COST_SHORT 2         /* validateDisparity reads the cost plane as `short` (wraps beyond 32767) */
#define SBMO_READ_SPECKLE_X16 4        /* filterSpeckles gets speckleRange * 16 */
#define SBMO_READ_ODD_ROW_COMPUTED 8   /* prefilterXSobel computes the last row of an odd-height image */
#define SBMO_READ_LR_TIE_LATER 16      /* validateDisparity: equal cost -> the later x takes the slot */
void sbmo_set_reading(int mask);
int sbmo_get_reading(void);

/* u16-vectorised timing variant of the correspondence stage (sbm_oracle_simd.c: what cv::StereoBM's "useShorts" SIMD path
 * delivers on a CPU). sbmo_set_simd(1) makes sbmo_compute / sbmo_compute_batch use it where sbmo_simd_ok(p); results are
 * bit-identical to the scalar restatement (tests/test_oracle_properties.py), which stays the checker. */
void sbmo_set_simd(int on);
int sbmo_get_simd(void);
int sbmo_simd_ok(const sbm_params* p);
void sbmo_find_correspondence_u16(const uint8_t* left_full, const uint8_t* right_full, size_t stride, int width, int height_full,
                                  int row0, int row1, const sbm_params* p, int16_t* disp_full, size_t dstride, int32_t* cost_full,
                                  size_t cstride);

void sbmo_valid_roi(const int32_t roi1[4], const int32_t roi2[4], int min_disparity, int num_disparities,
                    int block_size, int32_t out[4]);

/* OpenCV findStereoCorrespondenceBM for the row range [row0,row1) of prefiltered images (full-height
 * pointers; the function itself offsets to row0 exactly like FindStereoCorrespInvoker's rowRange()).
 * disp: int16 plane (stride in elements), cost: int32 plane or NULL (stride in elements).
 * Writes columns [0,width) of rows [row0,row1). */
void sbmo_find_correspondence(const uint8_t* left, const uint8_t* right, size_t stride, int width, int height,
                              int row0, int row1, const sbm_params* p, int16_t* disp, size_t dstride, int32_t* cost,
                              size_t cstride);

/* Independent brute-force evaluation of the same per-pixel definition, O(w^2)
 * per pixel-disparity. Used only to cross-check sbmo_find_correspondence on small images. */
void sbmo_find_correspondence_bruteforce(const uint8_t* left, const uint8_t* right, size_t stride, int width,
                                         int height, int row0, int row1, const sbm_params* p, int16_t* disp,
                                         size_t dstride, int32_t* cost, size_t cstride);

/* OpenCV validateDisparity (stereosgbm.cpp) on `rows` rows. */
void sbmo_validate_disparity(int16_t* disp, size_t dstride, const int32_t* cost, size_t cstride, int width, int rows,
                             int min_disparity, int num_disparities, int disp12_max_diff);

/* OpenCV filterSpeckles (stereosgbm.cpp, filterSpecklesImpl<short>). */
void sbmo_filter_speckles(int16_t* img, size_t stride, int width, int height, int new_val, int max_speckle_size,
                          int max_diff);

/* cv::StereoBM::compute() end to end. Returns an SBM_* status. If pre_lr / cost_out / pf_l / pf_r are non-NULL
 * (dense width*height planes) the intermediate stages are copied out for stage-wise parity tests. */
int sbmo_compute(const sbm_params* p, const uint8_t* left, size_t lstride, const uint8_t* right, size_t rstride,
                 int width, int height, int16_t* disp, size_t dstride_bytes, uint8_t* pf_l, uint8_t* pf_r,
                 int16_t* pre_lr, int32_t* cost_out);

/* n dense pairs, OpenMP across pairs with `threads` threads (cpu_baseline leg of bench.py). Returns status. */
int sbmo_compute_batch(const sbm_params* p, int n, const uint8_t* left, const uint8_t* right, int width, int height,
                       int16_t* disp, int threads);

/* Consumers of the map, restated from the reference's own C++:
 * SensorData.cpp:50-58 (decimation), Stereo.cpp:157-199 (projectDisparityTo3D / isFinite / transformPoint),
 * main.cpp:522-553 (reprojection of the decimated map), Stereo.cpp:53-117 (keypoints, dense-map branch). */
void sbmo_decimate(const int16_t* disp, int width, int height, int scale, int16_t* out);
void sbmo_reproject(const int16_t* disp, int width, int height, int scale, const sbm_stereo_model* m, int apply_local,
                    float* xyz);
void sbmo_keypoints3d(const int16_t* disp, int width, int height, const float* kpts, int nk, const sbm_stereo_model* m,
                      float min_depth, float max_depth, float* xyz);

/* Producers in front of the path, restated from the reference's firmware C and RTL:
 * inverse rectification map = rect_remap(), src/StereoBM/src/fpga.c:303-366; bilinear resampling with 5-bit
 * fractions = src/dvp/rtl/rect_intp.v:285-404. PARITY UNPINNED for the resampled image: the reference ships no raw
 * camera frame (data/ref_rect_* are already rectified), so there is no golden vector for this stage; the map function
 * is integer C in the reference and is restated expression by expression. Taps outside the source image read as 0. */
void sbmo_rect_map(const sbm_rect_cam* cam, int width, int height, int16_t* map);
void sbmo_rect_remap(const uint8_t* src, const int16_t* map, int width, int height, uint8_t* dst);

/* ---- FPGA flavour of the matcher, sbm_oracle_fpga.c. PARITY UNPINNED: restated from
 * the RTL (src/dvp/rtl/bm_calc_sad.v, bm_calc_det.v, bm_calc_frac.v, bm_calc_upd.v, bm_calc_uni.v, bm_calc.v, bm_obuf2.v,
 * diven.v, scheduling bm_ibuf.v:143-286); the reference holds the RTL's stimulus (data/ref_xsbl_*) but no output. */
/* diven.v with parameters (DW, VW, QW, MSB_INV): the pipelined non-restoring divider, bit-serially. */
uint32_t sbmo_rtl_diven(int DW, int VW, int QW, int MSB_INV, uint32_t dividend, uint32_t divisor);
/* bm_calc_det.v: sad[0] and sad[33] are the guard lanes, the minimum is searched over sad[1..32] (idx 0..31). */
void sbmo_rtl_det(const uint16_t sad[34], uint16_t* min1, uint16_t* min2, int* idx1, int* idx2, uint16_t* det_l,
                  uint16_t* det_r);
/* bm_calc_frac.v: signed 8-bit sub-pixel fraction (s-1.8) from centre / left (D-1) / right (D+1) SADs. */
uint8_t sbmo_rtl_frac(uint16_t c, uint16_t l, uint16_t r);
/* bm_obuf2.v:119-154: (integer disparity, fraction) -> s11.4, 0xFFFF when negative or zero. */
int16_t sbmo_rtl_pack_disparity(uint8_t disp, uint8_t frac);
/* struct FPGA_REG_BM (src/StereoBM/src/fpga.h:154-169) as decoded by bm.v:172-193:
 * out = {wdt, hgt, wsz, ndisp, uni_enb, uni_mode, uni_thr}. */
int sbmo_fpga_regs_decode(uint32_t image_size, uint32_t bm_setting, uint32_t uni_filt_ctrl, int32_t out[7]);
/* Limits of the RTL (field widths, whole 32-disparity phases, odd window) as SBM_* status codes. */
int sbmo_fpga_check(int width, int height, int wsz, int ndisp);
/* The matcher on dense x-Sobel planes (what data/ref_xsbl_{l,r} are); disp dense int16, fully written (-1 = none). */
int sbmo_fpga_bm(const uint8_t* xl, const uint8_t* xr, int width, int height, int wsz, int ndisp, int uni_enb,
                 int uni_mode, int uni_thr, int16_t* disp);
/* xsbl2.v prefilter of both rectified images (rows 0 / H-1 = 0) followed by the matcher. */
int sbmo_fpga_compute(const uint8_t* left, const uint8_t* right, int width, int height, int wsz, int ndisp, int uni_enb,
                      int uni_mode, int uni_thr, int16_t* disp);

/* GFTT minimum-eigenvalue map of the PL (gftt_sbl.v, gftt_box.v, gftt_eig.v, gftt_obuf.v). eig: dense
 * height*width uint16 (rows 0,1,H-2,H-1 = 0), max_out: the value of the GFTT `Max` register. PARITY UNPINNED; the square
 * root is taken as the exact floor (the CORDIC core's last bit is unspecified: +-1 LSB tolerance downstream). */
int sbmo_gftt_eig(const uint8_t* img, int width, int height, uint16_t* eig, uint32_t* max_out);

int sbmo_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
