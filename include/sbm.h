/*
 * sbm.h -- C-ABI of the MI355X stereo block-matching disparity engine (libsbm_hip.so).
 *
 * This is the drop-in boundary for the dense-disparity provider of sdoira/U96-SLAM:
 *
 *   reference call site               src/slam/src/core/main.cpp:197-217
 *       cv::Ptr<cv::StereoBM> bm = cv::StereoBM::create(16, 9);     (main.cpp:201)
 *       bm->setROI1 ... bm->setDisp12MaxDiff(1);                    (main.cpp:202-212)
 *       bm->compute(left, right, disp);                             (main.cpp:215)
 *   alternative provider, same output  src/slam/src/core/FPGA.cpp:270-279 (receiveDepthMap)
 *   consumers of the output contract   src/slam/src/core/Stereo.cpp:79-83, SensorData.cpp:50-58,
 *                                      main.cpp:529-530
 *
 * Output contract (identical to cv::StereoBM with a CV_16SC1 destination): int16, value =
 * 16 * disparity (4 fractional bits); every rejected / uncomputable pixel holds
 * (minDisparity - 1) * 16.
 *
 * All entry points are plain C: pointers, sizes, int status codes. No exceptions, no aborts,
 * no spinning (contrast the reference's Logger.cpp:52-55). The C++ adaptor that restores the
 * cv::StereoBM spelling (create / 11 setters / compute) is include/sbm_stereobm.hpp.
 *
 * There is NO CPU backend behind this ABI: if no HIP device is usable sbm_create() fails with
 * SBM_ERR_NO_DEVICE. The CPU restatement used by the tests lives in oracle/ and is never linked here.
 *
 * Exactness. Every parameter set cv::StereoBM accepts is computed (block sizes 5..255, any minDisparity / ROI, any multiple of 16
 * disparities up to the limits below); sets inside the fast envelope -- odd block size 5..31, numDisparities <= 512, blockSize^2
 * * 2 * preFilterCap <= 65534 -- run the hand-tuned kernels (4 T pixel-disparities/s), everything else a sliding-sum kernel with
 * 32-bit sums (0.15-0.5 T, same results; up to 2048 disparities, beyond that a per-column kernel ~10x slower again).
 * Bit-exactness against cv::StereoBM is CLAIMED for blockSize^2 * 2 * preFilterCap <= 32767 only (the reference's 21 x 21
 * at cap 31 is 27 342): OpenCV keeps its block-matching cost plane as `short`, so beyond that bound its left-right check
 * would see a wrapped cost where this engine (and its oracle) keep the true one.
 *
 * Limits (SBM_ERR_UNSUPPORTED beyond them; cv::StereoBM itself has none of these): numDisparities <= 4096, at most 32 767 pairs
 * per call, image height <= 65 535. The speckle filter's band walk serves images up to 65 535 columns and (W + 288) * H < 2^27 and
 * speckleWindowSize up to 2048; larger images or windows take its row-walking kernels (same results, ~3x the stage time).
 *
 * The hand-tuned kernels accumulate in place with v_mqsad_pk_u16_u8 (vdst == src2), which the hardware does right and the
 * compiler's register model forbids; a device self-test (once per device and process, on the first handle's stream: pseudo-random
 * operands, single instructions and dependent chains, at 1 / 4 / 8 wavefronts per SIMD) guards it. What the self-test does NOT
 * cover is a device that miscomputes only in instruction mixes it does not generate. If it fails, or with SBM_FAST_INPLACE=0, the
 * two-accumulator build runs instead: same results, 64-disparity layouts (~1.3x slower at 128 disparities), block sizes up to 27
 * and up to 256 disparities -- block sizes 29 / 31 and 257..512 disparities then fall to the sliding-sum kernel (8-25x slower; the
 * kernel name, sbm_last_kernel_name(), then reads "sad_wide_kernel [in-place accumulate unavailable]").
 *
 * Environment. The library reads these nine variables (nothing else); an integrator never needs to set any of them:
 *   variable            default  read      who sets it, and what for
 *   SBM_FAST_INPLACE    1        once      0 = run the two-accumulator build of the SAD kernel (the fallback that is taken
 *                                          automatically when the device self-test of the in-place v_mqsad accumulate
 *                                          fails); set by the GPU tests to check that fallback
 *   SBM_FAST_PFSHIFT    2        once      0 = unscaled prefiltered planes (plain winner search), 1 = at most one tag bit;
 *                                          GPU tests
 *   SBM_FAST_CS3        1        per call  0 = plain column strips only (no column-stride-3 strips); GPU tests
 *   SBM_SPECKLE_LISTS   1        per call  0 = the speckle filter's row-walking kernels (one wavefront per row, per-pixel labels)
 *                                          instead of the band walk + run records; GPU tests
 *   SBM_SPECKLE_BAND    auto     per call  2 / 4 = band height of the speckle filter's band walk, 0 = row-walking kernels;
 *                                          GPU tests
 *   SBM_SPECKLE_SEG     auto     per call  1 / 2 / 4 = column segments per band of the band walk (wavefronts of one workgroup that
 *                                          walk a band together; automatic: 4 for one-pair calls, 1 for frame batches); GPU tests
 *   SBM_HOST_ZEROCOPY   1        per call  0 = small host-buffer calls (sbm_compute / sbm_compute_batch up to 8 MB of maps) into
 *                                          pageable memory return their maps through a D2H copy + stream synchronisation instead of
 *                                          the copy kernel that writes pinned host memory and raises a flag the host polls; GPU
 *                                          tests / A-B measurements
 *   SBM_WIDE            1        per call  0 = configurations outside the fast envelope run the per-column kernel
 *                                          (sbm_sad_generic.hip) instead of the sliding-sum one (sbm_sad_wide.hip); GPU tests
 *   SBM_CV_READING      0        per call  bit mask of ALTERNATIVE readings of cv::StereoBM behaviours that this engine restates from
 *                                          memory and nothing in the reference can pin: 1 = getValidDisparityROI
 *                                          subtracts minDisparity from roi2's right edge (2.4 lineage), 2 = the LR check reads the cost
 *                                          plane as `short`, 4 = speckleRange * 16, 8 = the last row of an odd-height image is
 *                                          prefiltered instead of filled with preFilterCap, 16 = LR check: equal cost -> the later x
 *                                          wins. The oracle has the same bits; tests/golden/pin_kit.npz holds this engine's outputs
 *                                          under both readings of each, and tools/verify_with_opencv.py (numpy + cv2 only) names the
 *                                          reading a given OpenCV implements -- adopting it is a default flip here, not a rewrite
 * Tuning knobs of the measurement scripts (SBM_FAST_NSEG, SBM_FAST_TAPER, SBM_FAST_UNIQ_PLAIN,
 * SBM_FAST_SPLIT, SBM_PF_ROWS, SBM_HOST_CHUNK, SBM_HOST_PIPELINE, SBM_DEV_*; the interior kernel's are listed in
 * u96-slam_amd/csrc/sbm_sad_fast_dev.h) exist only in development builds (-DSBM_DEV, tools/exp/r05_devlib.sh); this library
 * ignores them. The Python mirror adds SBM_LIB_AB (file name of another build of this
 * library inside u96-slam_amd/lib/, A-B measurements only); bench.py reads SBM_BENCH_BACKEND / SBM_BENCH_FEED /
 * SBM_BENCH_SG_FAULT (tests of its multi-process control flow).
 */
#ifndef SBM_H_
#define SBM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SBM_VERSION_MAJOR 0
#define SBM_VERSION_MINOR 1

/* status codes (0 = ok). The negative "parameter" codes map 1:1 onto the CV_Error() checks at the
 * top of cv::StereoBM::compute (OpenCV calib3d, stereobm.cpp). */
enum {
  SBM_OK = 0,
  SBM_ERR_NULL = -1,             /* null handle / pointer argument                                   */
  SBM_ERR_SIZE = -2,             /* width/height <= 0, stride < width, left/right size mismatch       */
  SBM_ERR_PREFILTER_TYPE = -3,   /* preFilterType must be NORMALIZED_RESPONSE or XSOBEL               */
  SBM_ERR_PREFILTER_SIZE = -4,   /* preFilterSize must be odd and within 5..255                       */
  SBM_ERR_PREFILTER_CAP = -5,    /* preFilterCap must be within 1..63                                 */
  SBM_ERR_BLOCK_SIZE = -6,       /* blockSize must be odd, within 5..255 and < min(width,height)      */
  SBM_ERR_NUM_DISPARITIES = -7,  /* numDisparities must be > 0 and divisible by 16                    */
  SBM_ERR_TEXTURE = -8,          /* textureThreshold must be >= 0                                     */
  SBM_ERR_UNIQUENESS = -9,       /* uniquenessRatio must be >= 0                                      */
  SBM_ERR_NO_DEVICE = -20,       /* no usable HIP device / device index out of range                  */
  SBM_ERR_HIP = -21,             /* a HIP runtime call failed (see sbm_last_hip_error)                */
  SBM_ERR_NOMEM = -22,           /* device or host allocation failed                                  */
  SBM_ERR_UNSUPPORTED = -23,     /* valid OpenCV parameters this build cannot run (documented limits) */
  SBM_ERR_BATCH = -24            /* batch count <= 0                                                  */
};

#define SBM_PREFILTER_NORMALIZED_RESPONSE 0
#define SBM_PREFILTER_XSOBEL 1

/* Parameter block: one field per cv::StereoBM setter used at main.cpp:202-212 plus the ones left at
 * their OpenCV defaults there (preFilterType, preFilterSize). roi = {x, y, width, height}; a roi with
 * width <= 0 or height <= 0 is "empty" (cv::Rect()) and means "whole image", as at main.cpp:200-203. */
typedef struct sbm_params {
  int32_t prefilter_type;      /* setPreFilterType      default SBM_PREFILTER_XSOBEL                  */
  int32_t prefilter_size;      /* setPreFilterSize      default 9 (unused by XSOBEL)                  */
  int32_t prefilter_cap;       /* setPreFilterCap       default 31                                    */
  int32_t block_size;          /* setBlockSize          default 21  (SADWindowSize)                   */
  int32_t min_disparity;       /* setMinDisparity       default 0                                     */
  int32_t num_disparities;     /* setNumDisparities     default 64                                    */
  int32_t texture_threshold;   /* setTextureThreshold   default 10                                    */
  int32_t uniqueness_ratio;    /* setUniquenessRatio    default 15                                    */
  int32_t speckle_window_size; /* setSpeckleWindowSize  default 0 (off)                               */
  int32_t speckle_range;       /* setSpeckleRange       default 0                                     */
  int32_t disp12_max_diff;     /* setDisp12MaxDiff      default -1 (off)                              */
  int32_t roi1[4];             /* setROI1                                                             */
  int32_t roi2[4];             /* setROI2                                                             */
} sbm_params;

typedef struct sbm_handle sbm_handle; /* opaque; owns a HIP stream and lazily sized device scratch */

/* Fill *p with cv::StereoBM::create(numDisparities, blockSize) defaults (0 -> 64 resp. 21). */
void sbm_params_default(sbm_params* p, int num_disparities, int block_size);

/* Check p against an image size exactly as cv::StereoBM::compute does before doing any work. */
int sbm_params_validate(const sbm_params* p, int width, int height);

/* Create an engine on HIP device `device` (>= 0). Parameters are copied. */
int sbm_create(sbm_handle** out, const sbm_params* p, int device);
void sbm_destroy(sbm_handle* h);

/* Destroyed handles are parked (streams, events and up to 512 MB of device scratch each, at most 4) so that the
 * reference's pattern -- a new matcher per frame, src/slam/src/core/main.cpp:201 -- does not pay ~2 ms of set-up per
 * frame. sbm_trim() frees everything that is parked (e.g. before the process hands the GPU to someone else). */
void sbm_trim(void);

/* Replace the parameter block (the cv::StereoBM setters). Cheap; scratch is re-sized lazily. */
int sbm_set_params(sbm_handle* h, const sbm_params* p);
int sbm_get_params(const sbm_handle* h, sbm_params* p);

/* cv::StereoBM::compute() for one pair in HOST memory (the main.cpp:215 shape): strided, row-major
 * 8-bit inputs (cv::Mat::data / cv::Mat::step), strided int16 output. Strides are in BYTES.
 * Synchronous: on return `disp` is filled. */
int sbm_compute(sbm_handle* h, const uint8_t* left, size_t left_stride, const uint8_t* right, size_t right_stride,
                int width, int height, int16_t* disp, size_t disp_stride);

/* Same for n independent pairs in host memory (array-of-pointers, common size and strides). */
int sbm_compute_batch(sbm_handle* h, int n, const uint8_t* const* left, size_t left_stride,
                      const uint8_t* const* right, size_t right_stride, int width, int height,
                      int16_t* const* disp, size_t disp_stride);

/* n pairs already resident in DEVICE memory, densely packed: left/right = n*height*width bytes,
 * disp = n*height*width int16. Asynchronous on the handle's stream unless `sync` is non-zero.
 * This is the entry point bench.py times (inputs resident in HBM when the timed region starts).
 * Ordering: the handle's stream is a hipStreamNonBlocking stream -- it does NOT order behind the null stream or any
 * other stream of the caller. The inputs must be complete before the call (synchronise the producing stream, or make
 * sbm_stream() wait on an event of the producer with hipStreamWaitEvent), and with sync == 0 the buffers must stay
 * alive and untouched until sbm_synchronize() or an event recorded on sbm_stream() after the call has completed.
 * Every kernel of the call runs on the handle's stream (one SAD launch: the clamped border columns are extra wavefronts
 * of it). */
int sbm_compute_device(sbm_handle* h, int n, const void* d_left, const void* d_right, int width, int height,
                       void* d_disp, int sync);

/* Asynchronous dense feed (what a per-GPU feeder thread of a multi-GPU job uses; bench.py --feed host): sbm_submit_dense
 * queues one dense batch -- left/right: n*height*width bytes each, disp: n*height*width int16, all in HOST memory that
 * should be pinned -- and returns at once; at most three submissions are in flight (a fourth call first waits for the oldest).
 * Inputs cross PCIe on an H2D stream, the kernels run on the handle's stream, the maps return on a D2H stream, so batch k+1
 * arrives while batch k computes and batch k-1 leaves (two device staging sets, re-used through stream dependencies). sbm_wait_oldest blocks until the oldest outstanding submission's maps
 * are in `disp` (returns SBM_OK at once when nothing is outstanding). The buffers of a submission must stay untouched until
 * it has been waited for. The reference's call is synchronous (main.cpp:215); this pair exists because a feeder that keeps
 * the GPU busy across calls needs it. */
int sbm_submit_dense(sbm_handle* h, int n, const uint8_t* left, const uint8_t* right, int width, int height, int16_t* disp);
int sbm_wait_oldest(sbm_handle* h);

/* Block until everything queued on the handle's stream has finished and every outstanding submission of the asynchronous
 * feed has delivered its maps. */
int sbm_synchronize(sbm_handle* h);

/* sbm_destroy() and the asynchronous feed: copies that are already queued finish first (so the `disp` buffer of every
 * submission that was followed by another submission or by a wait must still exist when the handle is destroyed); the maps of
 * the NEWEST submission, whose trip home is only queued by the next submission or by a wait, are dropped -- destroy never
 * starts a write into caller memory. Callers that want those maps call sbm_synchronize() first. */

/* One dense host batch spread over several engines -- normally one handle per GPU of the node, created with
 * sbm_create(&h[k], &params, k) ("pair batches shard embarrassingly across the GPUs"; the reference's
 * caller is a single C++ thread, src/slam/src/core/main.cpp:149-216). left / right: n*height*width bytes, disp:
 * n*height*width int16, HOST memory that should be pinned (pageable memory works, but the runtime then stages every copy
 * synchronously and the devices run one after the other). Handle k computes the contiguous block of pairs
 * [n*k/K, n*(k+1)/K) through its asynchronous feed (two submissions per block); all blocks are queued before any is waited
 * for. Blocking: on return every map is in `disp`. The handles must be distinct (SBM_ERR_BATCH otherwise), may sit on the same
 * or on different devices and keep their own parameter blocks; results equal sbm_compute_batch() of each block on its handle.
 * Every handle is drained even when one of them fails; the first failure is returned. */
int sbm_compute_batch_multi(sbm_handle* const* handles, int n_handles, int n, const uint8_t* left, const uint8_t* right,
                            int width, int height, int16_t* disp);

/* Intermediate planes of the LAST sbm_compute_device call, for stage-by-stage parity tests.
 * which: 0 = prefiltered left (u8), 1 = prefiltered right (u8), 2 = WTA cost plane (int32, valid only
 * where the pre-LR disparity is valid and disp12_max_diff >= 0), 3 = disparity before LR/speckle (int16).
 * Copies n*height*width elements into host memory `dst`. */
int sbm_debug_fetch(sbm_handle* h, int which, void* dst, size_t dst_bytes);

/* Per-stage device time in ms, measured with HIP events recorded on the handle's stream around each stage.
 * enabled = 1: every sbm_compute_device call synchronises and sbm_get_profile returns the LAST call's times;
 * enabled = 2: events are recorded without synchronising (use inside a timed region) and sbm_get_profile
 *              (which synchronises) returns the average over the calls made since enabling (last 64 at most).
 * enabled = 3: as 2, but only every 4th call is instrumented (the five event records cost ~20 us per call at the bench size;
 *              sampling keeps a timed region close to the un-instrumented rate).
 * names: "prefilter", "sad" (the SAD/WTA kernels of the call: the interior kernel -- plus, beyond 256 disparities, the two
 * launches for its clamped border columns -- or the sliding-sum / per-column kernel when the fast path is off;
 * sbm_last_kernel_name() says which), "lrcheck", "speckle", "total"; "border" is still accepted and reads 0 (up to 256
 * disparities the clamped border columns have been wavefronts of the SAD launch since round 4: no second kernel to time). */
int sbm_set_profiling(sbm_handle* h, int enabled);
int sbm_get_profile(sbm_handle* h, const char* name, float* ms);

/* Which SAD kernel the LAST sbm_compute_device call launched, as text: the template instantiation of the interior
 * kernel ("sad_fast_kernel<128,1,5,3,true> pfshift=2"; "sad_fast_pp_kernel<...>" = its two-accumulator fallback build)
 * or "sad_wide_kernel" / "sad_generic_kernel" when the configuration is outside the fast envelope. bench.py compares it with the kernel the
 * committed counter profile was taken on, so that stale counters are never attached to a different kernel. */
int sbm_last_kernel_name(sbm_handle* h, char* dst, size_t dst_bytes);

/* ---- consumers of the disparity map --------------------------------------------
 * Device-side versions of what the reference does with the map right after compute(), so that only the small
 * results have to cross PCIe:
 *   decimation      SensorData::setFeatures, src/slam/src/core/SensorData.cpp:50-58  (keeps every scale-th pixel)
 *   reprojection    projectDisparityTo3D, src/slam/src/core/Stereo.cpp:157-182, as used on the decimated map by
 *                   buildOccupancyGridMap, src/slam/src/core/main.cpp:522-553 (pt2d = (col*scale, row*scale))
 *   keypoint depth  generateKeypoints3DStereo (dense-map branch), src/slam/src/core/Stereo.cpp:53-117
 * Arithmetic follows the C++ source operation by operation (float / double exactly where the reference uses them,
 * no fused multiply-add), so results are bit-identical to the reference's expressions. Invalid points are NaN. */
typedef struct sbm_stereo_model {
  double fx_l, fy_l, cx_l, cy_l, Tx_l;  /* StereoCameraModel P[0] entries (include/core/StereoCameraModel.h:25-29) */
  double fx_r, fy_r, cx_r, Tx_r;        /* P[1] entries (:30-34)                                                   */
  float local[12];                      /* localTransform r11 r12 r13 o14 / r21.. / r31.. (Transform.h:38-41)       */
  int32_t has_local;                    /* 0 = localTransform().isNull()                                            */
} sbm_stereo_model;

/* The CV_32F form of the result (cv::StereoBM::compute into a CV_32F destination: disp16.convertTo(dst, CV_32F, 1./16)):
 * out = disp / 16 as float, exact; FILTERED pixels become (minDisparity - 1). n*height*width floats. */
int sbm_disparity_to_float_device(sbm_handle* h, int n, const void* d_disp, int width, int height, void* d_out, int sync);

/* out[p][r][c] = disp[p][r*scale][c*scale]; out planes are (height/scale) x (width/scale), densely packed. */
int sbm_decimate_device(sbm_handle* h, int n, const void* d_disp, int width, int height, int scale, void* d_out, int sync);

/* xyz[p][r][c] = projectDisparityTo3D((c*scale, r*scale), disp[p][r][c]/16.0f, model), then localTransform if
 * apply_local != 0 and the model has one; NaN triple where the disparity is <= 0 or the point is not finite.
 * width/height are those of the (possibly decimated) map; d_xyz holds n*height*width*3 floats. */
int sbm_reproject_device(sbm_handle* h, int n, const void* d_disp, int width, int height, int scale,
                         const sbm_stereo_model* model, int apply_local, void* d_xyz, int sync);

/* 3-D points of nk keypoints of ONE full-resolution disparity plane: for keypoint (x,y) the disparity is
 * disp[(int)y][(int)x]/16.0f (negative -> 0 -> invalid), range-checked with min_depth / max_depth exactly as
 * generateKeypoints3DStereo does, then localTransform. d_kpts: nk*2 floats (x,y); d_xyz: nk*3 floats. */
int sbm_keypoints3d_device(sbm_handle* h, const void* d_disp, int width, int height, const void* d_kpts, int nk,
                           const sbm_stereo_model* model, float min_depth, float max_depth, void* d_xyz, int sync);

/* ---- producers in front of the path (SURVEY.md 8f rank 2 and the prefilter half of rank 3) -------------------------
 * Device-resident versions of the two stages the reference's FPGA flavour runs before block matching, so raw camera
 * frames can enter the engine without a host round trip:
 *   rectification   inverse map: rect_remap(), src/StereoBM/src/fpga.c:303-366 (s1.24 fixed point; the RTL twin is
 *                   src/dvp/rtl/rect_rmp.v:339-572); bilinear resampling with 5-bit fractions:
 *                   src/dvp/rtl/rect_intp.v:285-404
 *   x-Sobel         stand-alone prefilter of dense images, either flavour: cv prefilterXSobel (what sbm_compute runs
 *                   internally) or the RTL's src/dvp/rtl/xsbl2.v:185-198,661-874 (clip to [-32,31], +32; rows 0 and
 *                   H-1 unwritten = 0) -- the flavour data/ref_xsbl_{l,r} was produced with.
 * Integer arithmetic throughout; results are bit-identical to the reference's C / RTL expressions. */
typedef struct sbm_rect_cam {   /* struct RECT_PARAM_CH, src/StereoBM/src/fpga.h:250-256 (one camera)                  */
  int32_t f[2];                 /* focal length x, y of the source camera, u10.16                                      */
  int32_t c[2];                 /* principal point x, y of the source camera, integer pixels                           */
  int32_t f2inv[2];             /* 1 / f2 of the rectified camera, u-8.32                                              */
  int32_t c2_f2[2];             /* c2 / f2 of the rectified camera, u0.24                                              */
  int32_t rot[3][3];            /* inverted rotation, s0.24, indexed as the firmware does (rot[row][col])              */
} sbm_rect_cam;

/* d_map: height*width*2 int16, (x, y) interleaved, source coordinates in 1/32 px (what rect_remap() stores in
 * MAT2S.data[0] / data[1]). Depends on the camera only: build once, reuse for every frame. */
int sbm_rect_map_device(sbm_handle* h, const sbm_rect_cam* cam, int width, int height, void* d_map, int sync);

/* dst[i] = bilinear(src[i], map) for n dense u8 images sharing one map:
 *   ((UL*(32-xf)*(32-yf) + UR*xf*(32-yf) + DL*(32-xf)*yf + DR*xf*yf) >> 9) + 1) >> 1, taps at (x>>5, y>>5) and +1.
 * Taps outside the source image read as 0 (the RTL reads stale line-buffer contents there; calibrated rigs keep the
 * valid region inside). */
int sbm_rect_remap_device(sbm_handle* h, int n, const void* d_src, const void* d_map, int width, int height, void* d_dst,
                          int sync);

#define SBM_PREFILTER_FLAVOUR_CV 0  /* clip(s,-cap,cap)+cap, reflect-101 rows, odd H: last row = cap */
#define SBM_PREFILTER_FLAVOUR_RTL 1 /* clip(s,-32,31)+32, rows 0 and H-1 = 0 (cap ignored)           */
/* n dense u8 images -> n dense u8 planes. */
int sbm_prefilter_device(sbm_handle* h, int n, const void* d_src, int width, int height, int flavour, int cap,
                         void* d_dst, int sync);

/* ---- the reference's own matcher: FPGA flavour (SURVEY.md 8f rank 3 + 8a row a8) -------------------------------------
 * Bit-level restatement of the block matcher the reference runs in programmable logic (src/dvp/rtl/bm*.v, fed by
 * xsbl2.v), for consumers of DEPTH_METHOD_FPGA_BM (src/slam/src/core/FPGA.cpp:270-279): 6-bit x-Sobel inputs, 10-bit
 * saturating column sums, 32-disparity phases, tournament minimum, divider-based sub-pixel fraction, optional min1/min2
 * ratio filter, int16 s11.4 output with -1 (0xFFFF) for "no disparity" and for the never-computed borders
 * (hwsz rows top and bottom, ndisp + hwsz + 1 columns left, hwsz columns right). It differs from cv::StereoBM in every
 * stage; no texture threshold, LR check or speckle filter exists in this flavour.
 * Parameters are the fields of the BM register block, struct FPGA_REG_BM (src/StereoBM/src/fpga.h:154-169) as decoded by
 * src/dvp/rtl/bm.v:172-193; the firmware programs ImageSize = 480 << 16 | 640, BmSetting = 0x00150040 (window 21,
 * 64 disparities) and leaves UniFiltCtrl at 0 (src/StereoBM/src/fpga.c:150-160). */
typedef struct sbm_fpga_params {
  int32_t width;            /* ImageSize [9:0]    */
  int32_t height;           /* ImageSize [24:16]  */
  int32_t block_size;       /* BmSetting [20:16]  wsz: odd, 3..31 */
  int32_t num_disparities;  /* BmSetting [8:0]    ndisp: multiple of 32, 32..256 (whole disparity phases) */
  int32_t uni_enable;       /* UniFiltCtrl [31]   */
  int32_t uni_mode;         /* UniFiltCtrl [16]   0: rejected pixels read 0xFFFF, 1: they read disparity 255 + 255/256 */
  int32_t uni_threshold;    /* UniFiltCtrl [9:0]  reject when min1/min2 (u0.10) > threshold */
} sbm_fpga_params;

/* Register words -> parameters, exactly the bit fields of bm.v:172-193 (no validation). */
int sbm_fpga_params_from_regs(uint32_t image_size, uint32_t bm_setting, uint32_t uni_filt_ctrl, sbm_fpga_params* out);
/* Read-back value of SAD_Size [1724h] = sad_hgt << 16 | sad_wdt (bm.v:208,249-255). */
uint32_t sbm_fpga_sad_size_reg(const sbm_fpga_params* p);
/* Limits of the RTL as status codes: field widths (width <= 1023, height <= 511), odd window 3..31, ndisp a positive
 * multiple of 32 up to 256, at least one output pixel; SBM_ERR_UNSUPPORTED for (ndisp + hwsz + 1) % 32 == 0, where the
 * RTL's output sequencer (bm_obuf2.v:239) never leaves its fill state. */
int sbm_fpga_params_validate(const sbm_fpga_params* p);
/* The matcher on n dense pairs of x-Sobel planes (what data/ref_xsbl_{l,r} are: xsbl2.v output, 0..63) resident in
 * device memory; d_disp = n*height*width int16. Asynchronous on the handle's stream unless sync != 0. */
int sbm_fpga_bm_device(sbm_handle* h, int n, const void* d_xsbl_l, const void* d_xsbl_r, const sbm_fpga_params* p,
                       void* d_disp, int sync);
/* xsbl2.v prefilter (SBM_PREFILTER_FLAVOUR_RTL) of n dense rectified pairs followed by the matcher: the whole PL
 * pipeline behind Fpga::receiveDepthMap. */
int sbm_fpga_compute_device(sbm_handle* h, int n, const void* d_left, const void* d_right, const sbm_fpga_params* p,
                            void* d_disp, int sync);

/* Host-memory form of sbm_fpga_compute_device for ONE pair, shaped like the frame Fpga::receiveDepthMap hands out
 * (src/slam/src/core/FPGA.cpp:270-279: a dense height x width CV_16SC1 image): strided 8-bit rectified inputs, strided int16
 * output, strides in bytes. Synchronous. */
int sbm_fpga_compute(sbm_handle* h, const uint8_t* left, size_t left_stride, const uint8_t* right, size_t right_stride,
                     const sbm_fpga_params* p, int16_t* disp, size_t disp_stride);

/* ---- GFTT minimum-eigenvalue map of the PL ----------------------------------------------------------
 * The dense half of the reference's FPGA feature detector (src/dvp/rtl/gftt_sbl.v, gftt_box.v, gftt_eig.v, gftt_obuf.v):
 * per image a height*width uint16 map of (a + c) - sqrt((a - c)^2 + 4 b^2) over 3x3 boxes of the Sobel products, rows
 * 0, 1, H-2, H-1 and columns 0, W-1 = 0, plus the maximum of the map (the GFTT `Max` register) -- exactly what
 * generateKeypoints2() consumes (src/slam/src/core/GFTT.cpp:41-170, fed by FPGA.cpp:283-291). Fixed-point steps and
 * limiters follow the RTL; its CORDIC square root is specified to +-1 LSB, this engine returns the exact floor.
 * d_img: n dense u8 images (the rectified left frames already on the device); d_eig: n*height*width uint16;
 * d_max: n uint32. width 3..1023, height 5..511 (the RTL's field widths). */
int sbm_gftt_eig_device(sbm_handle* h, int n, const void* d_img, int width, int height, void* d_eig, void* d_max, int sync);
/* Host-memory form for one image (what FPGA.cpp:283-291 builds: a CV_16UC1 map and the Max register). Synchronous. */
int sbm_gftt_eig(sbm_handle* h, const uint8_t* img, size_t img_stride, int width, int height, uint16_t* eig, size_t eig_stride,
                 uint32_t* max_out);

/* The raw HIP stream (hipStream_t) as void*, so callers can order their own work behind ours (record an event on it
 * after sbm_compute_device(..., sync = 0)) or ours behind theirs (hipStreamWaitEvent on it before the call). Every entry
 * point selects the handle's device for the duration of the call and restores the caller's current device on return. */
void* sbm_stream(sbm_handle* h);

const char* sbm_strerror(int code);
int sbm_last_hip_error(const sbm_handle* h); /* hipError_t of the last failing runtime call, else 0 */
int sbm_version(void);                        /* major * 1000 + minor */

#ifdef __cplusplus
}
#endif
#endif /* SBM_H_ */
