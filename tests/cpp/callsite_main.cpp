// The reference's call site (src/slam/src/core/main.cpp:197-217) with the one-line type swap of INTEGRATION.md:
// a matcher is created INSIDE the frame loop (main.cpp:201), configured with the 11 setters (main.cpp:202-212) and
// asked for one disparity map (main.cpp:215). Frames come from raw 8-bit files instead of cv::imread so that the
// program builds without OpenCV; with OpenCV headers present -- the real ones, or the mock of the few names involved
// under tests/cpp/mock_opencv, which pins nothing about OpenCV's arithmetic -- the cv::InputArray / cv::OutputArray
// overload is what computes (compile with -DSBM_TEST_WITH_OPENCV): plain cv::Mat destination, a fixed CV_32F destination
// and the error -> cv::Exception mapping.
//
//   callsite_main <width> <height> <nframes> <left.raw> <right.raw> <disp_out.raw> [async]
//
// "async": ONE matcher outside the loop and the opt-in computeAsync() / wait() pair of the adaptor, two frames in flight.
//
// left/right hold nframes dense frames; disp_out receives nframes dense int16 maps. Exit code 0 on success.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "sbm_stereobm.hpp"

static bool read_all(const char* path, std::vector<uint8_t>& buf) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return false;
  const size_t got = std::fread(buf.data(), 1, buf.size(), f);
  std::fclose(f);
  return got == buf.size();
}

int main(int argc, char** argv) {
  if (argc != 7 && argc != 8) return 2;
  const bool async = argc == 8;
  const int W = std::atoi(argv[1]), H = std::atoi(argv[2]), N = std::atoi(argv[3]);
  const size_t npix = (size_t)W * H;
  std::vector<uint8_t> left(npix * N), right(npix * N);
  std::vector<int16_t> disp(npix * N);
  if (!read_all(argv[4], left) || !read_all(argv[5], right)) return 3;
  try {
    if (async) {
      auto bm = sbm::StereoBM::create(64, 21);
      bm->setPreFilterCap(31); bm->setMinDisparity(0); bm->setTextureThreshold(10); bm->setUniquenessRatio(10);
      bm->setSpeckleWindowSize(50); bm->setSpeckleRange(32); bm->setDisp12MaxDiff(1);
      for (int i = 0; i < N; i++) {
        if (bm->pending() == 2) bm->wait();                          // frame i-2 is complete: its map may be consumed here
        bm->computeAsync(left.data() + i * npix, right.data() + i * npix, W, H, disp.data() + i * npix);
      }
      // the matcher goes out of scope with up to two frames outstanding: its destructor drains them
    }
    for (int i = 0; i < N && !async; i++) {                           // while(1) of main.cpp:149
      // --- main.cpp:198-212, cv::StereoBM -> sbm::StereoBM ---------------------------------------------------------
      auto bm = sbm::StereoBM::create(16, 9);
      bm->setROI1(0, 0, 0, 0);                                       // cv::Rect roi1, roi2 are empty (main.cpp:199-203)
      bm->setROI2(0, 0, 0, 0);
      bm->setPreFilterCap(31);
      bm->setBlockSize(21);
      bm->setMinDisparity(0);
      bm->setNumDisparities(64);
      bm->setTextureThreshold(10);
      bm->setUniquenessRatio(10);
      bm->setSpeckleWindowSize(50);
      bm->setSpeckleRange(32);
      bm->setDisp12MaxDiff(1);
      // --- main.cpp:215 ---------------------------------------------------------------------------------------------
#ifdef SBM_TEST_WITH_OPENCV
      cv::Mat l(H, W, CV_8UC1, left.data() + i * npix), r(H, W, CV_8UC1, right.data() + i * npix), d;
      bm->compute(l, r, d);
      if (d.type() != CV_16SC1 || d.rows != H || d.cols != W) return 5;
      for (int y = 0; y < H; y++) std::copy(d.ptr<int16_t>(y), d.ptr<int16_t>(y) + W, disp.data() + i * npix + (size_t)y * W);
      if (i == 0) {
        // a destination with a fixed CV_32F type receives disparity / 16 as float (what SensorData::setImageDepth and
        // Stereo.cpp:79-83 of the reference would read)
        cv::Mat_<float> df;
        bm->compute(l, r, df);
        if (df.type() != CV_32FC1 || df.rows != H || df.cols != W) return 7;
        for (int y = 0; y < H; y++)
          for (int x = 0; x < W; x++)
            if (df.ptr<float>(y)[x] != (float)d.ptr<int16_t>(y)[x] / 16.f) return 8;
      }
#else
      bm->compute(left.data() + i * npix, (size_t)W, right.data() + i * npix, (size_t)W, W, H, disp.data() + i * npix,
                  (size_t)W * sizeof(int16_t));
#endif
    }
    // parameter violations surface from compute(), as cv::StereoBM's CV_Error does (numDisparities % 16 != 0)
    auto bad = sbm::StereoBM::create(20, 9);
    bool threw = false;
    try {
      bad->compute(left.data(), (size_t)W, right.data(), (size_t)W, W, H, disp.data(), (size_t)W * 2);
    } catch (const sbm::Error& e) {
      threw = e.code == SBM_ERR_NUM_DISPARITIES;
    }
#ifdef SBM_HAVE_OPENCV
    catch (const cv::Exception&) { threw = true; }
#endif
    if (!threw) return 6;
#ifdef SBM_TEST_WITH_OPENCV
    // the same through the cv::Mat overload: a cv::Exception, as cv::StereoBM::compute's CV_Error raises
    {
      cv::Mat l(H, W, CV_8UC1, left.data()), r(H, W, CV_8UC1, right.data()), d;
      bool cvthrew = false;
      try {
        bad->compute(l, r, d);
      } catch (const cv::Exception&) { cvthrew = true; }
      if (!cvthrew) return 9;
      // images of different sizes / of the wrong type are refused before anything reaches the engine
      cv::Mat small(H - 1, W, CV_8UC1, right.data());
      cvthrew = false;
      try {
        sbm::StereoBM::create(16, 9)->compute(l, small, d);
      } catch (const cv::Exception&) { cvthrew = true; }
      if (!cvthrew) return 10;
    }
#endif
  } catch (const std::exception& e) {
    std::fprintf(stderr, "callsite_main: %s\n", e.what());
    return 4;
  }
  FILE* f = std::fopen(argv[6], "wb");
  if (!f) return 3;
  std::fwrite(disp.data(), sizeof(int16_t), disp.size(), f);
  std::fclose(f);
  std::printf("ok %d frames %dx%d\n", N, W, H);
  return 0;
}
