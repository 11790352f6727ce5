// sbm_stereobm.hpp -- header-only C++ adaptor that restores the cv::StereoBM spelling on top of the C-ABI (sbm.h).
//
// It exists so that the one call site of the reference,
//
//     cv::Ptr<cv::StereoBM> bm = cv::StereoBM::create(16, 9);      // src/slam/src/core/main.cpp:201
//     bm->setROI1(roi1); ... bm->setDisp12MaxDiff(1);              // main.cpp:202-212
//     bm->compute(data.imageLeft(), data.imageRight(), disp);      // main.cpp:215
//
// compiles against the MI355X engine with a one-line type swap (sbm::StereoBM instead of cv::StereoBM); see
// INTEGRATION.md.  Setter names, argument meaning, defaults and failure behaviour follow cv::StereoBM: parameter
// violations throw (cv::Exception when OpenCV headers are present, sbm::Error otherwise) from compute().
#ifndef SBM_STEREOBM_HPP_
#define SBM_STEREOBM_HPP_

#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>

#include <vector>

#include "sbm.h"

#if defined(SBM_WITH_OPENCV) || (defined(__has_include) && __has_include(<opencv2/core.hpp>))
#include <opencv2/core.hpp>
#define SBM_HAVE_OPENCV 1
#endif

namespace sbm {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

class StereoBM {
 public:
  enum { PREFILTER_NORMALIZED_RESPONSE = SBM_PREFILTER_NORMALIZED_RESPONSE, PREFILTER_XSOBEL = SBM_PREFILTER_XSOBEL };

  // cv::StereoBM::create(numDisparities = 0, blockSize = 21); `device` selects the HIP device (default 0).
  static std::shared_ptr<StereoBM> create(int numDisparities = 0, int blockSize = 21, int device = 0) {
    return std::shared_ptr<StereoBM>(new StereoBM(numDisparities, blockSize, device));
  }

  // (outstanding computeAsync() calls are drained first: sbm_destroy alone lets the queued copies finish and DROPS the newest
  // submission's maps -- include/sbm.h, "sbm_destroy() and the asynchronous feed")
  ~StereoBM() {
    if (pending_ > 0) sbm_synchronize(h_);
    sbm_destroy(h_);
  }
  StereoBM(const StereoBM&) = delete;
  StereoBM& operator=(const StereoBM&) = delete;

  int getPreFilterType() const { return p_.prefilter_type; }
  void setPreFilterType(int v) { p_.prefilter_type = v; push(); }
  int getPreFilterSize() const { return p_.prefilter_size; }
  void setPreFilterSize(int v) { p_.prefilter_size = v; push(); }
  int getPreFilterCap() const { return p_.prefilter_cap; }
  void setPreFilterCap(int v) { p_.prefilter_cap = v; push(); }
  int getBlockSize() const { return p_.block_size; }
  void setBlockSize(int v) { p_.block_size = v; push(); }
  int getMinDisparity() const { return p_.min_disparity; }
  void setMinDisparity(int v) { p_.min_disparity = v; push(); }
  int getNumDisparities() const { return p_.num_disparities; }
  void setNumDisparities(int v) { p_.num_disparities = v; push(); }
  int getTextureThreshold() const { return p_.texture_threshold; }
  void setTextureThreshold(int v) { p_.texture_threshold = v; push(); }
  int getUniquenessRatio() const { return p_.uniqueness_ratio; }
  void setUniquenessRatio(int v) { p_.uniqueness_ratio = v; push(); }
  int getSpeckleWindowSize() const { return p_.speckle_window_size; }
  void setSpeckleWindowSize(int v) { p_.speckle_window_size = v; push(); }
  int getSpeckleRange() const { return p_.speckle_range; }
  void setSpeckleRange(int v) { p_.speckle_range = v; push(); }
  int getDisp12MaxDiff() const { return p_.disp12_max_diff; }
  void setDisp12MaxDiff(int v) { p_.disp12_max_diff = v; push(); }
  void setROI1(int x, int y, int w, int h) { p_.roi1[0] = x; p_.roi1[1] = y; p_.roi1[2] = w; p_.roi1[3] = h; push(); }
  void setROI2(int x, int y, int w, int h) { p_.roi2[0] = x; p_.roi2[1] = y; p_.roi2[2] = w; p_.roi2[3] = h; push(); }

  // Raw-pointer compute: strides in bytes (cv::Mat::step). Output int16, 1/16 px, invalid = (minDisparity-1)*16.
  void compute(const uint8_t* left, size_t lstep, const uint8_t* right, size_t rstep, int width, int height, int16_t* disp,
               size_t dstep) {
    check(sbm_compute(h_, left, lstep, right, rstep, width, height, disp, dstep));
  }

  // Opt-in asynchronous form for callers that keep frames in flight (double buffering), e.g. a capture loop that fetches frame
  // k+1 while frame k is matched: computeAsync() queues one DENSE pair (stride == width; left / right / disp should be pinned,
  // hipHostMalloc / hipHostRegister, and must stay valid until the matching wait() returns) and returns at once -- the inputs of
  // this call cross PCIe while the previous call computes and the one before sends its map home. wait() blocks until the OLDEST
  // outstanding call has delivered its map. At most three calls are in flight (a fourth computeAsync() first waits for the oldest).
  // The reference's own loop (main.cpp:201-216: compute, then use the map) keeps calling compute().
  void computeAsync(const uint8_t* left, const uint8_t* right, int width, int height, int16_t* disp) {
    check(sbm_submit_dense(h_, 1, left, right, width, height, disp));
    if (pending_ < 3) pending_++;
  }
  void wait() {
    if (pending_ > 0) {
      check(sbm_wait_oldest(h_));
      pending_--;
    }
  }
  int pending() const { return pending_; }

#ifdef SBM_HAVE_OPENCV
  void setROI1(cv::Rect r) { setROI1(r.x, r.y, r.width, r.height); }
  void setROI2(cv::Rect r) { setROI2(r.x, r.y, r.width, r.height); }
  cv::Rect getROI1() const { return cv::Rect(p_.roi1[0], p_.roi1[1], p_.roi1[2], p_.roi1[3]); }
  cv::Rect getROI2() const { return cv::Rect(p_.roi2[0], p_.roi2[1], p_.roi2[2], p_.roi2[3]); }

  // cv::StereoMatcher::compute(InputArray left, InputArray right, OutputArray disparity): CV_16SC1, or CV_32FC1 if fixed.
  void compute(cv::InputArray leftarr, cv::InputArray rightarr, cv::OutputArray disparr) {
    if (leftarr.size() != rightarr.size()) CV_Error(cv::Error::StsUnmatchedSizes, "All the images must have the same size");
    if (leftarr.type() != CV_8UC1 || rightarr.type() != CV_8UC1)
      CV_Error(cv::Error::StsUnsupportedFormat, "Both input images must have CV_8UC1");
    cv::Mat left = leftarr.getMat(), right = rightarr.getMat();
    // like cv::StereoBM: a destination with a fixed CV_32F type receives disparity / 16 as float, anything else CV_16SC1
    const bool want_f32 = disparr.fixedType() && disparr.type() == CV_32FC1;
    cv::Mat disp16;
    if (want_f32) disp16.create(left.size(), CV_16SC1);
    else disparr.create(left.size(), CV_16SC1);
    cv::Mat disp = want_f32 ? disp16 : disparr.getMat();
    int st = sbm_compute(h_, left.ptr<uint8_t>(), left.step, right.ptr<uint8_t>(), right.step, left.cols, left.rows,
                         disp.ptr<int16_t>(), disp.step);
    if (st != SBM_OK) CV_Error(st <= SBM_ERR_NO_DEVICE ? cv::Error::StsError : cv::Error::StsOutOfRange, message(st));
    if (want_f32) disp16.convertTo(disparr, CV_32F, 1. / 16);
  }
#endif

  sbm_handle* handle() const { return h_; }

  // One dense host batch (n*height*width bytes per side, n*height*width int16 out, pinned memory recommended) over several
  // matchers -- normally one per GPU, created with create(nd, bs, device): contiguous pair blocks, every device busy at once
  // from this one thread (sbm_compute_batch_multi; INTEGRATION.md "Several GPUs from one C++ process").
  static void computeBatch(const std::vector<std::shared_ptr<StereoBM>>& matchers, int n, const uint8_t* left, const uint8_t* right,
                           int width, int height, int16_t* disp) {
    std::vector<sbm_handle*> hs;
    for (const auto& m : matchers) hs.push_back(m ? m->h_ : nullptr);
    const int st = sbm_compute_batch_multi(hs.data(), (int)hs.size(), n, left, right, width, height, disp);
    if (st != SBM_OK) throw Error(st, sbm_strerror(st));
  }

 private:
  StereoBM(int nd, int bs, int device) : h_(nullptr), pending_(0) {
    sbm_params_default(&p_, nd, bs);
    check(sbm_create(&h_, &p_, device));
  }
  void push() { check(sbm_set_params(h_, &p_)); }
  std::string message(int st) const {
    std::string m = sbm_strerror(st);
    if (st == SBM_ERR_HIP) m += " (hipError " + std::to_string(sbm_last_hip_error(h_)) + ")";
    return m;
  }
  void check(int st) const {
    if (st != SBM_OK) throw Error(st, message(st));
  }
  sbm_params p_;
  sbm_handle* h_;
  int pending_;   // computeAsync() calls whose wait() is still due
};

}  // namespace sbm
#endif
