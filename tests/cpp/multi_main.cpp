// The C++ caller's multi-GPU recipe (INTEGRATION.md "Several GPUs from one process"; SURVEY.md section 8e: "one process, one
// stream set per device") and the thread-safety promise of include/sbm.h, as one program built against the C-ABI alone:
//
//   multi_main <width> <height> <npairs> <left.raw> <right.raw> <out_dir> [handles_per_device]
//
// left/right hold npairs dense 8-bit frames. The program writes
//   <out_dir>/single.raw   sbm_compute_batch() of the whole batch on ONE handle (device 0)
//   <out_dir>/multi.raw    sbm_compute_batch_multi() over one handle on EVERY visible device (x handles_per_device, default 1;
//                          2 on a one-GPU box still exercises the block partition with two engines on device 0), pinned host
//                          memory, the parameters of main.cpp:204-212
//   <out_dir>/threads.raw  two host threads, each with its OWN handle on device 0 and its own half of the batch, one
//                          sbm_compute() per pair (the reference's one-pair-per-call pattern), both loops running at once
// and exits 0 only if all three hold the same maps and every call returned SBM_OK. The Python test compares single.raw with
// the oracle, so "same" means bit-exact there too.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "sbm.h"
#include "sbm_stereobm.hpp"

static bool read_all(const char* path, void* buf, size_t bytes) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return false;
  const size_t got = std::fread(buf, 1, bytes, f);
  std::fclose(f);
  return got == bytes;
}
static bool write_all(const std::string& path, const void* buf, size_t bytes) {
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) return false;
  const size_t put = std::fwrite(buf, 1, bytes, f);
  std::fclose(f);
  return put == bytes;
}
#define CHECK(call)                                                                   \
  do {                                                                                \
    const int st_ = (call);                                                           \
    if (st_ != SBM_OK) { std::fprintf(stderr, "%s -> %d (%s)\n", #call, st_, sbm_strerror(st_)); return 10; } \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 7) return 2;
  const int W = std::atoi(argv[1]), H = std::atoi(argv[2]), N = std::atoi(argv[3]);
  const std::string out = argv[6];
  const int per_dev = argc > 7 ? std::atoi(argv[7]) : 1;
  const size_t npix = (size_t)W * H;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return 3;

  // pinned host memory: what lets the devices' copies run asynchronously (and side by side)
  uint8_t *left = nullptr, *right = nullptr;
  int16_t *d_single = nullptr, *d_multi = nullptr, *d_thr = nullptr;
  if (hipHostMalloc((void**)&left, npix * N, 0) != hipSuccess || hipHostMalloc((void**)&right, npix * N, 0) != hipSuccess ||
      hipHostMalloc((void**)&d_single, npix * N * 2, 0) != hipSuccess || hipHostMalloc((void**)&d_multi, npix * N * 2, 0) != hipSuccess ||
      hipHostMalloc((void**)&d_thr, npix * N * 2, 0) != hipSuccess)
    return 4;
  if (!read_all(argv[4], left, npix * N) || !read_all(argv[5], right, npix * N)) return 5;
  std::memset(d_single, 0x55, npix * N * 2); std::memset(d_multi, 0x66, npix * N * 2); std::memset(d_thr, 0x77, npix * N * 2);

  sbm_params p;
  sbm_params_default(&p, 64, 21);                 // main.cpp:204-212
  p.prefilter_cap = 31; p.texture_threshold = 10; p.uniqueness_ratio = 10;
  p.speckle_window_size = 50; p.speckle_range = 32; p.disp12_max_diff = 1;

  // --- one handle, whole batch ---------------------------------------------------------------------------------------------
  {
    sbm_handle* h = nullptr;
    CHECK(sbm_create(&h, &p, 0));
    std::vector<const uint8_t*> lp(N), rp(N);
    std::vector<int16_t*> dp(N);
    for (int i = 0; i < N; i++) { lp[i] = left + i * npix; rp[i] = right + i * npix; dp[i] = d_single + i * npix; }
    CHECK(sbm_compute_batch(h, N, lp.data(), W, rp.data(), W, W, H, dp.data(), (size_t)W * 2));
    sbm_destroy(h);
  }
  // --- one handle per device (x per_dev), contiguous pair blocks -------------------------------------------------------------
  {
    std::vector<sbm_handle*> hs;
    for (int d = 0; d < ndev; d++)
      for (int k = 0; k < per_dev; k++) {
        sbm_handle* h = nullptr;
        CHECK(sbm_create(&h, &p, d));
        hs.push_back(h);
      }
    CHECK(sbm_compute_batch_multi(hs.data(), (int)hs.size(), N, left, right, W, H, d_multi));
    // a second call on the same handles (staging sets are re-used) must give the same maps
    std::vector<int16_t> again(npix * N);
    int16_t* pin2 = nullptr;
    if (hipHostMalloc((void**)&pin2, npix * N * 2, 0) != hipSuccess) return 4;
    CHECK(sbm_compute_batch_multi(hs.data(), (int)hs.size(), N, left, right, W, H, pin2));
    if (std::memcmp(pin2, d_multi, npix * N * 2) != 0) { std::fprintf(stderr, "second multi call differs\n"); return 11; }
    (void)hipHostFree(pin2);
    // the same handle twice is refused
    if (hs.size() >= 1) {
      sbm_handle* twice[2] = {hs[0], hs[0]};
      if (sbm_compute_batch_multi(twice, 2, N, left, right, W, H, d_multi) != SBM_ERR_BATCH) return 12;
    }
    for (sbm_handle* h : hs) sbm_destroy(h);
    std::printf("multi: %d device(s) x %d handle(s)\n", ndev, per_dev);
  }
  // --- the same through the C++ adaptor: one sbm::StereoBM per device, StereoBM::computeBatch -----------------------------------
  {
    std::vector<std::shared_ptr<sbm::StereoBM>> ms;
    for (int d = 0; d < ndev; d++) {
      auto bm = sbm::StereoBM::create(16, 9, d);                     // main.cpp:201-212
      bm->setPreFilterCap(31); bm->setBlockSize(21); bm->setMinDisparity(0); bm->setNumDisparities(64); bm->setTextureThreshold(10);
      bm->setUniquenessRatio(10); bm->setSpeckleWindowSize(50); bm->setSpeckleRange(32); bm->setDisp12MaxDiff(1);
      ms.push_back(bm);
    }
    int16_t* pin3 = nullptr;
    if (hipHostMalloc((void**)&pin3, npix * N * 2, 0) != hipSuccess) return 4;
    try {
      sbm::StereoBM::computeBatch(ms, N, left, right, W, H, pin3);
    } catch (const sbm::Error& e) {
      std::fprintf(stderr, "computeBatch: %s\n", e.what());
      return 14;
    }
    if (std::memcmp(pin3, d_multi, npix * N * 2) != 0) { std::fprintf(stderr, "adaptor batch differs\n"); return 15; }
    (void)hipHostFree(pin3);
  }
  // --- two host threads, two handles on device 0, one pair per call -----------------------------------------------------------
  {
    std::atomic<int> bad{0};
    auto worker = [&](int t) {
      sbm_handle* h = nullptr;
      if (sbm_create(&h, &p, 0) != SBM_OK) { bad++; return; }
      const int i0 = t == 0 ? 0 : N / 2, i1 = t == 0 ? N / 2 : N;
      for (int rep = 0; rep < 3; rep++)
        for (int i = i0; i < i1; i++) {
          // (the handle is re-created every few frames, like the matcher of main.cpp:201: the pool of parked handles is shared state)
          if (rep > 0 && (i & 3) == 0) { sbm_destroy(h); h = nullptr; if (sbm_create(&h, &p, 0) != SBM_OK) { bad++; return; } }
          if (sbm_compute(h, left + i * npix, W, right + i * npix, W, W, H, d_thr + i * npix, (size_t)W * 2) != SBM_OK) bad++;
          char name[96];
          if (sbm_last_kernel_name(h, name, sizeof(name)) != SBM_OK || std::strstr(name, "sad_") == nullptr) bad++;
        }
      sbm_destroy(h);
    };
    std::thread a(worker, 0), b(worker, 1);
    a.join(); b.join();
    if (bad.load() != 0) { std::fprintf(stderr, "thread workers: %d failures\n", bad.load()); return 13; }
  }
  sbm_trim();
  if (!write_all(out + "/single.raw", d_single, npix * N * 2) || !write_all(out + "/multi.raw", d_multi, npix * N * 2) ||
      !write_all(out + "/threads.raw", d_thr, npix * N * 2))
    return 6;
  const bool same = std::memcmp(d_single, d_multi, npix * N * 2) == 0 && std::memcmp(d_single, d_thr, npix * N * 2) == 0;
  std::printf("%s\n", same ? "OK" : "MISMATCH");
  (void)hipHostFree(left); (void)hipHostFree(right); (void)hipHostFree(d_single); (void)hipHostFree(d_multi); (void)hipHostFree(d_thr);
  return same ? 0 : 20;
}
