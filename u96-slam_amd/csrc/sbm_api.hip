// sbm_api.hip -- C-ABI of libsbm_hip.so (declared in include/sbm.h): parameter checks, device scratch,
// kernel orchestration.  Replaces cv::StereoBM::compute at src/slam/src/core/main.cpp:201-216.
//
// Stage order (same as cv::StereoBM::compute): prefilter both images -> SAD/WTA on the valid-ROI rows
// (fast kernel: interior columns + the clamped border columns as extra wavefronts of the same launch; generic kernel otherwise) -> LR check + invalid
// row/column fill -> speckle filter.  Everything is enqueued on the handle's stream; no host sync inside.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <mutex>
#include <new>
#include <thread>

#include "sbm_common.h"

using namespace sbm;
namespace sbm { thread_local char g_sad_kernel_name[96] = ""; }

struct sbm_handle {
  sbm_params p;
  int device;
  hipStream_t stream;
  int last_hip;
  // scratch, sized for (cap_n, cap_W, cap_H, cap pitch)
  int cap_n, cap_W, cap_H, cap_pitch;
  uint8_t *pf_l, *pf_r;
  int16_t* disp_pre;
  int32_t* cost;
  void* spk_runs;        // speckle filter: 16 bytes per pixel (run records of the band walk / labels + sizes of the row-walking kernels)
  int32_t* spk_nheads;   // cap_n * H (runs per row)
  uint32_t* spk_seam;    // contacts across band seams (band walk of the speckle filter): cap_n * ceil(H/2) * W entries
  int32_t* spk_nseam;    // cap_n * ceil(H/2)
  uint16_t* vsum;      // column sums of PREFILTER_NORMALIZED_RESPONSE (2 * cap_n * W * H), allocated on first use
  // FPGA-flavour matcher scratch (allocated on first use, sized for fp_n pairs of fp_W x fp_H)
  int fp_n, fp_W, fp_H;
  uint8_t *fp_xs_l, *fp_xs_r;
  int fp_gen;              // call counter of the FPGA-flavour matcher: generation stamp of its saturation flags
  void* fp_rec;
  int* fp_flag;
  // staging for the host-buffer entry points
  int st_n, st_W, st_H;
  uint8_t *st_l, *st_r;
  int16_t* st_d;
  uint8_t* pin;        // pinned host staging for strided caller images (rows packed / unpacked on the CPU)
  size_t pin_bytes;
  // small host-buffer calls (the reference's one pair per call): the maps leave through a copy kernel that writes pinned,
  // device-mapped host memory and raises a flag there; the host polls the flag instead of synchronising the stream
  int16_t* zc_out;     // pinned + mapped host staging of the maps
  size_t zc_bytes;
  unsigned* zc_flag;   // pinned + mapped: sequence number of the last call whose maps are complete in zc_out
  unsigned* zc_cnt;    // device: workgroups of the copy kernel that have finished (the last one raises the flag and clears it)
  unsigned zc_seq;
  // copy streams + per-chunk events of the pipelined host batch path (created on first use)
  hipStream_t stream_in, stream_out;
  // asynchronous dense feed (sbm_submit_dense / sbm_wait_oldest): two device staging sets, up to three submissions in
  // flight (one arriving, one computing, one leaving); events are indexed by submission number & 3
  uint8_t *fq_l[2], *fq_r[2];
  int16_t* fq_d[2];
  int fq_n, fq_W, fq_H;
  hipEvent_t ev_fq_in[4], ev_fq_done[4], ev_fq_out[4];
  bool fq_ok;
  unsigned fq_submitted, fq_waited;
  int16_t* fq_pending_dst;     // maps of the newest submission not yet queued for their trip home (see sbm_submit_dense)
  size_t fq_pending_bytes;
  static constexpr int kChunks = 64;
  hipEvent_t ev_in[kChunks], ev_done[kChunks];
  bool pipe_ok;
  char last_kernel[128];   // SAD kernel of the last sbm_compute_device call (sbm_last_kernel_name)
  // last launch (for sbm_debug_fetch)
  Geom last;
  bool have_last;
  // profiling: mode 1 = sync after every call and keep that call's stage times; mode 2 = record stage events of
  // every call into a ring WITHOUT syncing (bench.py's timed region); sbm_get_profile then averages the ring.
  // mode 3 = mode 2 on every 4th call only (six event records cost ~25 us per call: sampling keeps the timed region honest)
  int profiling;
  unsigned ncall;      // calls since profiling was (re)enabled
  bool instr;          // this call records events
  static constexpr int kRing = 64, kMarks = 6;
  hipEvent_t ev[kRing][kMarks];
  bool ev_ok;
  unsigned calls;  // calls recorded since profiling was (re)enabled
  float ms_prefilter, ms_sad, ms_border, ms_lr, ms_speckle, ms_total;
};

// Entry points select the handle's device and put the caller's current device back on return.
struct DeviceScope {
  int prev, dev;
  bool have;
  explicit DeviceScope(int d) : prev(-1), dev(d), have(false) { have = hipGetDevice(&prev) == hipSuccess; }
  hipError_t enter() { return (have && prev == dev) ? hipSuccess : hipSetDevice(dev); }
  ~DeviceScope() {
    if (have && prev != dev) hipSetDevice(prev);
  }
};

#ifdef SBM_DEV   // development builds: wall-clock stamps of the host-buffer entry point's phases (tools/exp/r05_host_attrib.py)
static double g_hp_acc[8];
static unsigned long long g_hp_calls;
struct HostProf {
  std::chrono::steady_clock::time_point t;
  HostProf() : t(std::chrono::steady_clock::now()) {}
  void stamp(int i) {
    const auto n = std::chrono::steady_clock::now();
    g_hp_acc[i] += std::chrono::duration<double, std::micro>(n - t).count();
    t = n;
  }
};
#define HP_BEGIN() HostProf hp_; g_hp_calls++
#define HP(i) hp_.stamp(i)
extern "C" int sbm_dev_host_prof(double* out8, unsigned long long* calls) {
  for (int i = 0; i < 8; i++) { out8[i] = g_hp_acc[i]; g_hp_acc[i] = 0; }
  *calls = g_hp_calls; g_hp_calls = 0;
  return 0;
}
#else
#define HP_BEGIN() do { } while (0)
#define HP(i) do { } while (0)
#endif

static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  asm volatile("yield");
#else
  std::this_thread::yield();
#endif
}

#define HIPCHK(h, call)                         \
  do {                                          \
    hipError_t e_ = (call);                     \
    if (e_ != hipSuccess) {                     \
      (h)->last_hip = (int)e_;                  \
      return e_ == hipErrorOutOfMemory ? SBM_ERR_NOMEM : SBM_ERR_HIP; \
    }                                           \
  } while (0)

extern "C" {

void sbm_params_default(sbm_params* p, int num_disparities, int block_size) {
  if (!p) return;
  memset(p, 0, sizeof(*p));
  p->prefilter_type = SBM_PREFILTER_XSOBEL;
  p->prefilter_size = 9;
  p->prefilter_cap = 31;
  p->block_size = block_size > 0 ? block_size : 21;
  p->min_disparity = 0;
  p->num_disparities = num_disparities > 0 ? num_disparities : 64;
  p->texture_threshold = 10;
  p->uniqueness_ratio = 15;
  p->speckle_window_size = 0;
  p->speckle_range = 0;
  p->disp12_max_diff = -1;
}

int sbm_params_validate(const sbm_params* p, int width, int height) {
  if (!p) return SBM_ERR_NULL;
  if (width <= 0 || height <= 0) return SBM_ERR_SIZE;
  if (p->prefilter_type != SBM_PREFILTER_NORMALIZED_RESPONSE && p->prefilter_type != SBM_PREFILTER_XSOBEL)
    return SBM_ERR_PREFILTER_TYPE;
  if (p->prefilter_size < 5 || p->prefilter_size > 255 || p->prefilter_size % 2 == 0) return SBM_ERR_PREFILTER_SIZE;
  if (p->prefilter_cap < 1 || p->prefilter_cap > 63) return SBM_ERR_PREFILTER_CAP;
  if (p->block_size < 5 || p->block_size > 255 || p->block_size % 2 == 0 || p->block_size >= std::min(width, height))
    return SBM_ERR_BLOCK_SIZE;
  if (p->num_disparities <= 0 || p->num_disparities % 16 != 0) return SBM_ERR_NUM_DISPARITIES;
  if (p->texture_threshold < 0) return SBM_ERR_TEXTURE;
  if (p->uniqueness_ratio < 0) return SBM_ERR_UNIQUENESS;
  return SBM_OK;
}

const char* sbm_strerror(int code) {
  switch (code) {
    case SBM_OK: return "ok";
    case SBM_ERR_NULL: return "null argument";
    case SBM_ERR_SIZE: return "bad image size or stride (all the images must have the same size)";
    case SBM_ERR_PREFILTER_TYPE: return "preFilterType must be PREFILTER_NORMALIZED_RESPONSE or PREFILTER_XSOBEL";
    case SBM_ERR_PREFILTER_SIZE: return "preFilterSize must be odd and be within 5..255";
    case SBM_ERR_PREFILTER_CAP: return "preFilterCap must be within 1..63";
    case SBM_ERR_BLOCK_SIZE: return "SADWindowSize must be odd, be within 5..255 and be not larger than image width or height";
    case SBM_ERR_NUM_DISPARITIES: return "numDisparities must be positive and divisible by 16";
    case SBM_ERR_TEXTURE: return "texture threshold must be non-negative";
    case SBM_ERR_UNIQUENESS: return "uniqueness ratio must be non-negative";
    case SBM_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU backend)";
    case SBM_ERR_HIP: return "HIP runtime error (see sbm_last_hip_error)";
    case SBM_ERR_NOMEM: return "out of memory";
    case SBM_ERR_UNSUPPORTED: return "configuration outside this build's limits";
    case SBM_ERR_BATCH: return "batch count must be positive";
    default: return "unknown status";
  }
}

int sbm_version(void) { return SBM_VERSION_MAJOR * 1000 + SBM_VERSION_MINOR; }

static void free_scratch(sbm_handle* h) {
  hipFree(h->pf_l); hipFree(h->pf_r); hipFree(h->disp_pre); hipFree(h->cost);
  hipFree(h->spk_runs); hipFree(h->spk_nheads); hipFree(h->spk_seam); hipFree(h->spk_nseam);
  h->spk_runs = nullptr; h->spk_nheads = nullptr; h->spk_seam = nullptr; h->spk_nseam = nullptr;
  hipFree(h->vsum);
  h->vsum = nullptr;
  h->pf_l = h->pf_r = nullptr; h->disp_pre = nullptr; h->cost = nullptr;
  h->cap_n = h->cap_W = h->cap_H = h->cap_pitch = 0;
}

static void free_fpga(sbm_handle* h) {
  hipFree(h->fp_xs_l); hipFree(h->fp_xs_r); hipFree(h->fp_rec); hipFree(h->fp_flag);
  h->fp_xs_l = h->fp_xs_r = nullptr; h->fp_rec = nullptr; h->fp_flag = nullptr;
  h->fp_n = h->fp_W = h->fp_H = 0;
}

// The device sets of the asynchronous dense feed have their own life: submissions may be outstanding (the newest one's
// trip home not even queued yet) while a synchronous host entry point resizes ITS staging, so only the feed's own realloc
// path (drained first) and the handle's end of life free them.
static void free_feed(sbm_handle* h) {
  for (int k = 0; k < 2; k++) { hipFree(h->fq_l[k]); hipFree(h->fq_r[k]); hipFree(h->fq_d[k]); h->fq_l[k] = h->fq_r[k] = nullptr; h->fq_d[k] = nullptr; }
  h->fq_n = h->fq_W = h->fq_H = 0;
}

static void free_staging(sbm_handle* h) {
  hipFree(h->st_l); hipFree(h->st_r); hipFree(h->st_d);
  if (h->pin) hipHostFree(h->pin);
  h->pin = nullptr; h->pin_bytes = 0;
  if (h->zc_out) hipHostFree(h->zc_out);
  if (h->zc_flag) hipHostFree(h->zc_flag);
  hipFree(h->zc_cnt);
  h->zc_out = nullptr; h->zc_flag = nullptr; h->zc_cnt = nullptr; h->zc_bytes = 0;
  h->st_l = h->st_r = nullptr; h->st_d = nullptr; h->st_n = h->st_W = h->st_H = 0;
}

// The reference re-creates its matcher for every frame (cv::StereoBM::create inside the loop, main.cpp:201). Streams,
// ~400 events and the device scratch make a cold handle cost ~2 ms -- ten times the frame itself -- so destroyed handles
// are parked (a few, with at most kPoolScratch bytes of scratch each) and sbm_create re-arms one for the same device.
static std::mutex g_pool_mu;
static constexpr int kPool = 4;
static constexpr size_t kPoolScratch = (size_t)512 << 20;
static sbm_handle* g_pool[kPool];
static int g_pool_n = 0;

static size_t scratch_bytes(const sbm_handle* h) {
  const size_t npix = (size_t)h->cap_n * h->cap_W * h->cap_H, plane = (size_t)h->cap_n * h->cap_pitch * h->cap_H;
  size_t b = 2 * plane + npix * 2;
  if (h->cost) b += npix * 4;
  if (h->spk_runs) b += (size_t)h->cap_n * h->cap_H * (16 * ((size_t)h->cap_W + kSpkRecordPad) + 2 * ((size_t)h->cap_W + kSpkSeamPad) + 24);   // run records, seam lists, run / contact counts
  if (h->vsum) b += 2 * npix * sizeof(uint16_t);
  b += (size_t)h->st_n * h->st_W * h->st_H * 4 + h->pin_bytes;
  b += (size_t)h->fq_n * h->fq_W * h->fq_H * 8;
  b += (size_t)h->fp_n * h->fp_W * h->fp_H * 10;
  return b;
}

static void destroy_now(sbm_handle* h);

void sbm_trim(void) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  for (int i = 0; i < g_pool_n; i++) destroy_now(g_pool[i]);
  g_pool_n = 0;
}

int sbm_create(sbm_handle** out, const sbm_params* p, int device) {
  if (!out || !p) return SBM_ERR_NULL;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return SBM_ERR_NO_DEVICE;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (int i = 0; i < g_pool_n; i++)
      if (g_pool[i]->device == device) {
        sbm_handle* h = g_pool[i];
        g_pool[i] = g_pool[--g_pool_n];
        h->p = *p;
        h->last_hip = 0;
        h->have_last = false;
        h->profiling = 0;
        h->calls = 0;
        h->ms_prefilter = h->ms_sad = h->ms_border = h->ms_lr = h->ms_speckle = h->ms_total = 0.f;
        *out = h;
        return SBM_OK;
      }
  }
  sbm_handle* h = new (std::nothrow) sbm_handle();
  if (!h) return SBM_ERR_NOMEM;
  memset(h, 0, sizeof(*h));
  h->p = *p;
  h->device = device;
  DeviceScope dscope(device);
  if (dscope.enter() != hipSuccess) {
    delete h;
    return SBM_ERR_NO_DEVICE;
  }
  // every failure below goes through destroy_now(), which tolerates the members that were never created (null)
  bool ok = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) == hipSuccess;
  if (!ok) {
    destroy_now(h);
    return SBM_ERR_HIP;
  }
  h->ev_ok = true;
  for (int r = 0; r < sbm_handle::kRing; r++)
    for (int i = 0; i < sbm_handle::kMarks; i++) h->ev_ok &= hipEventCreate(&h->ev[r][i]) == hipSuccess;
  *out = h;
  return SBM_OK;
}

static void sync_all_streams(sbm_handle* h) {
  if (h->stream) hipStreamSynchronize(h->stream);
  if (h->stream_in) hipStreamSynchronize(h->stream_in);
  if (h->stream_out) hipStreamSynchronize(h->stream_out);
}

void sbm_destroy(sbm_handle* h) {
  if (!h) return;
  DeviceScope dscope(h->device);
  dscope.enter();
  // submissions of the asynchronous feed nobody waited for: the copies that are already queued finish (their destination
  // must still exist, as for any submission that has not been waited for); the newest submission's maps, whose trip home is
  // only queued by a wait or by the next submission, are DROPPED -- destroy never starts a write into caller memory
  h->fq_pending_dst = nullptr;
  h->fq_waited = h->fq_submitted;
  sync_all_streams(h);
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (g_pool_n < kPool) {
      if (scratch_bytes(h) > kPoolScratch) {
        free_scratch(h);
        free_staging(h);
        free_feed(h);
        free_fpga(h);
      }
      g_pool[g_pool_n++] = h;
      return;
    }
  }
  destroy_now(h);
}

// Frees whatever the handle owns; members that were never created are null (the handle is zero-initialised), so this is
// also the failure path of sbm_create. Restores the caller's current device.
static void destroy_now(sbm_handle* h) {
  DeviceScope dscope(h->device);
  dscope.enter();
  sync_all_streams(h);
  free_scratch(h);
  free_staging(h);
  free_feed(h);
  free_fpga(h);
  for (int r = 0; r < sbm_handle::kRing; r++)
    for (int i = 0; i < sbm_handle::kMarks; i++)
      if (h->ev[r][i]) hipEventDestroy(h->ev[r][i]);
  for (int i = 0; i < sbm_handle::kChunks; i++) {
    if (h->ev_in[i]) hipEventDestroy(h->ev_in[i]);
    if (h->ev_done[i]) hipEventDestroy(h->ev_done[i]);
  }
  for (int k = 0; k < 4; k++) {
    if (h->ev_fq_in[k]) hipEventDestroy(h->ev_fq_in[k]);
    if (h->ev_fq_done[k]) hipEventDestroy(h->ev_fq_done[k]);
    if (h->ev_fq_out[k]) hipEventDestroy(h->ev_fq_out[k]);
  }
  if (h->stream_in) hipStreamDestroy(h->stream_in);
  if (h->stream_out) hipStreamDestroy(h->stream_out);
  if (h->stream) hipStreamDestroy(h->stream);
  delete h;
}

int sbm_set_params(sbm_handle* h, const sbm_params* p) {
  if (!h || !p) return SBM_ERR_NULL;
  h->p = *p;
  return SBM_OK;
}

int sbm_get_params(const sbm_handle* h, sbm_params* p) {
  if (!h || !p) return SBM_ERR_NULL;
  *p = h->p;
  return SBM_OK;
}

void* sbm_stream(sbm_handle* h) { return h ? (void*)h->stream : nullptr; }
int sbm_last_hip_error(const sbm_handle* h) { return h ? h->last_hip : 0; }

int sbm_synchronize(sbm_handle* h) {
  if (!h) return SBM_ERR_NULL;
  DeviceScope dscope(h->device);
  HIPCHK(h, dscope.enter());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  while (h->fq_waited != h->fq_submitted) {   // submissions of the asynchronous feed count as pending work too
    const int st = sbm_wait_oldest(h);
    if (st != SBM_OK) return st;
  }
  return SBM_OK;
}

int sbm_set_profiling(sbm_handle* h, int enabled) {
  if (!h) return SBM_ERR_NULL;
  h->profiling = enabled;
  h->calls = 0;
  h->ncall = 0;
  h->instr = false;
  h->ms_prefilter = h->ms_sad = h->ms_border = h->ms_lr = h->ms_speckle = h->ms_total = 0.f;
  return SBM_OK;
}

// cv getValidDisparityROI (calib3d stereosgbm.cpp) with cv::StereoBM's "empty rect = whole image" substitution.
static void valid_roi(const sbm_params& p, int W, int H, int reading, int roi[4]) {
  int full[4] = {0, 0, W, H};
  const int* r1 = (p.roi1[2] > 0 && p.roi1[3] > 0) ? p.roi1 : full;
  const int* r2 = (p.roi2[2] > 0 && p.roi2[3] > 0) ? p.roi2 : full;
  const int sw2 = p.block_size / 2, maxd = p.min_disparity + p.num_disparities - 1;
  const int xmin = std::max(r1[0], r2[0] + maxd) + sw2;
  const int xmax = std::min(r1[0] + r1[2], r2[0] + r2[2] - ((reading & kReadRoiMinusMinD) ? p.min_disparity : 0)) - sw2;
  const int ymin = std::max(r1[1], r2[1]) + sw2;
  const int ymax = std::min(r1[1] + r1[3], r2[1] + r2[3]) - sw2;
  if (xmax - xmin > 0 && ymax - ymin > 0) {
    roi[0] = xmin; roi[1] = ymin; roi[2] = xmax - xmin; roi[3] = ymax - ymin;
  } else {
    roi[0] = roi[1] = roi[2] = roi[3] = 0;
  }
}

static int ensure_scratch(sbm_handle* h, int n, int W, int H, int pitch, bool need_cost, bool need_speckle) {
  const bool fits = n <= h->cap_n && W == h->cap_W && H == h->cap_H && pitch == h->cap_pitch && h->pf_l;
  if (!fits) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    free_scratch(h);
    const size_t npix = (size_t)n * W * H, pfbytes = (size_t)n * pitch * H + 4096;
    HIPCHK(h, hipMalloc((void**)&h->pf_l, pfbytes));
    HIPCHK(h, hipMalloc((void**)&h->pf_r, pfbytes));
    HIPCHK(h, hipMalloc((void**)&h->disp_pre, npix * sizeof(int16_t)));
    // padding bytes must read as 0 (the masked value of the fast kernel); the prefilter never writes them
    HIPCHK(h, hipMemsetAsync(h->pf_l, 0, pfbytes, h->stream));
    HIPCHK(h, hipMemsetAsync(h->pf_r, 0, pfbytes, h->stream));
    h->cap_n = n; h->cap_W = W; h->cap_H = H; h->cap_pitch = pitch;
  }
  const size_t npix = (size_t)h->cap_n * W * H;
  if (need_cost && !h->cost) HIPCHK(h, hipMalloc((void**)&h->cost, npix * sizeof(int32_t)));
  if (need_speckle && !h->spk_nseam) {   // keyed on the LAST buffer of the set: an attempt that failed half way is redone
    hipFree(h->spk_runs); hipFree(h->spk_nheads); hipFree(h->spk_seam);
    h->spk_runs = nullptr; h->spk_nheads = nullptr; h->spk_seam = nullptr;
    // sizes: launch_speckle (sbm_common.h)
    HIPCHK(h, hipMalloc(&h->spk_runs, (size_t)h->cap_n * H * ((size_t)W + kSpkRecordPad) * 16));
    HIPCHK(h, hipMalloc((void**)&h->spk_nheads, (size_t)h->cap_n * H * kSpkMaxSeg * sizeof(int32_t)));
    const size_t seams = (size_t)h->cap_n * ((H + 1) / 2);
    HIPCHK(h, hipMalloc((void**)&h->spk_seam, seams * ((size_t)W + kSpkSeamPad) * sizeof(uint32_t)));
    HIPCHK(h, hipMalloc((void**)&h->spk_nseam, seams * kSpkMaxSeg * sizeof(int32_t)));
  }
  return SBM_OK;
}

static inline void mark(sbm_handle* h, int i) {
  if (h->instr) hipEventRecord(h->ev[h->calls % sbm_handle::kRing][i], h->stream);
}

// mode 2: average stage times over the recorded calls (at most the last kRing); mode 1: the last call only.
// The stream must be idle.
static void collect_profile(sbm_handle* h) {
  unsigned nrec = std::min<unsigned>(h->calls, sbm_handle::kRing), first = 0;
  if (h->profiling == 1 && h->calls > 0) { first = (h->calls - 1) % sbm_handle::kRing; nrec = 1; }
  // events 0..5 with 3 unused (it used to close the border kernel's side stream; one launch carries those columns now)
  static const int kFrom[4] = {0, 1, 2, 4}, kTo[4] = {1, 2, 4, 5};
  float acc[4] = {0, 0, 0, 0}, tot = 0.f;
  for (unsigned r = first; r < first + nrec; r++) {
    for (int i = 0; i < 4; i++) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, h->ev[r][kFrom[i]], h->ev[r][kTo[i]]) == hipSuccess) acc[i] += ms;
    }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->ev[r][0], h->ev[r][5]) == hipSuccess) tot += ms;
  }
  const float inv = nrec ? 1.f / nrec : 0.f;
  h->ms_prefilter = acc[0] * inv; h->ms_sad = acc[1] * inv; h->ms_border = 0.f; h->ms_lr = acc[2] * inv;
  h->ms_speckle = acc[3] * inv; h->ms_total = tot * inv;
}

int sbm_compute_device(sbm_handle* h, int n, const void* d_left, const void* d_right, int width, int height,
                       void* d_disp, int sync) {
  if (!h || !d_left || !d_right || !d_disp) return SBM_ERR_NULL;
  if (n <= 0) return SBM_ERR_BATCH;
  int st = sbm_params_validate(&h->p, width, height);
  if (st != SBM_OK) return st;
  if (n > 32767 || height > 65535) return SBM_ERR_UNSUPPORTED;
  const sbm_params& p = h->p;
  if (p.num_disparities > 4096) return SBM_ERR_UNSUPPORTED;
  DeviceScope dscope(h->device);
  HIPCHK(h, dscope.enter());

  Geom g;
  memset(&g, 0, sizeof(g));
  g.W = width; g.H = height; g.n = n;
  g.nd = p.num_disparities; g.mindisp = p.min_disparity; g.wsz = p.block_size; g.w2 = p.block_size / 2;
  g.cap = p.prefilter_cap; g.tex = p.texture_threshold; g.uniq = p.uniqueness_ratio;
  g.filtered = (p.min_disparity - 1) * 16;
  g.lofs = std::max(g.nd - 1 + g.mindisp, 0);
  g.rofs = -std::min(g.nd - 1 + g.mindisp, 0);
  g.width1 = width - g.rofs - g.nd + 1;
  g.xend = std::min(g.width1, width - g.lofs);
  g.want_cost = p.disp12_max_diff >= 0;
  // padded prefiltered planes: the fast kernel stages 16-byte pieces that may start up to nd+64 bytes left of
  // column 0 and end up to 96 bytes right of column W-1
  g.padl = ((g.nd + 64 + 63) / 64) * 64;
  g.pitch = ((g.padl + width + 128 + 63) / 64) * 64;
  g.plane = g.pitch * height;
  int16_t* out = (int16_t*)d_disp;

  int roi[4];
  g.reading = env_switch("SBM_CV_READING", 0);
  valid_roi(p, width, height, g.reading, roi);
  g.row0 = std::max(roi[1], 0); g.row1 = std::min(roi[1] + roi[3], height);
  g.col0 = std::max(std::min(roi[0], width), 0); g.col1 = std::max(std::min(roi[0] + roi[2], width), 0);
  const bool range_fits = !(g.lofs >= width || g.rofs >= width || g.width1 < 1);
  const bool any_rows = range_fits && roi[2] > 0 && roi[3] > 0 && g.row1 > g.row0;
  if (!any_rows) { g.row0 = g.row1 = 0; }
  const bool speckle = p.speckle_range >= 0 && p.speckle_window_size > 0;

  st = ensure_scratch(h, n, width, height, g.pitch, g.want_cost, speckle);
  if (st != SBM_OK) return st;
  h->last = g; h->have_last = true;

  // 16-bit cost plane when every producer is a 16-bit-sum kernel (fast interior + border kernels, w/2 clamped columns on
  // each side); the generic kernel needs int32
  // (the once-per-device self-test behind the in-place accumulate runs on THIS handle's stream, before the envelope predicate
  // below can trigger it on the legacy stream)
  const bool inplace = mqsad_inplace_ok(h->stream);
  const bool fast = any_rows && sad_fast_supported(g);
  int fa = 0, fb = 0;
  if (fast) {
    const int xhi = std::min(g.W - g.lofs - 1, g.W - g.rofs - g.nd);
    fa = g.w2; fb = xhi - g.w2 + 1;   // the interior range launch_sad_fast covers; xend - fb == w/2 by construction
    g.cost16 = 1;
    g.pfshift = sad_fast_pfshift(g);   // pre-scaled planes for the interior kernel's tagged winner search
  }
  h->last = g;
  // columns left and right of the fast range: clamped windows. They only matter if they can influence the output:
  // through the LR check or when inside the valid ROI.
  const bool borders_visible = g.want_cost || g.col0 < g.lofs + fa || g.col1 > g.lofs + fb;
  // ... and then they ride in the interior launch as extra wavefronts (sbm_sad_border_wave.h): one SAD launch, one stream
  const bool border = fast && borders_visible;
  // Stages run one after the other on the main stream. Overlapping LR + speckle of one sub-batch with the SAD kernel of
  // the next was built and measured in round 2 (profiles/r02_subbatch_pipeline.md; the code is in the history at commit
  // "Engine: device guard ..."): 1.47 ms -> 1.54 / 1.73 ms with 2 / 4 sub-batches, because four 126-VGPR wavefronts per
  // SIMD leave no registers for a guest wavefront -- the post-filters only overlap once the SAD kernel runs at 3
  // wavefronts per SIMD, which costs it 27 %.
  const uint8_t* dl = (const uint8_t*)d_left;
  const uint8_t* dr = (const uint8_t*)d_right;

  h->instr = h->profiling && h->ev_ok && (h->profiling != 3 || (h->ncall & 3u) == 0);
  mark(h, 0);
  if (any_rows) {
    if (p.prefilter_type == SBM_PREFILTER_XSOBEL) {
      HIPCHK(h, launch_prefilter(dl, dr, h->pf_l, h->pf_r, g, h->stream));
    } else {
      if (!h->vsum) HIPCHK(h, hipMalloc((void**)&h->vsum, (size_t)2 * h->cap_n * width * height * sizeof(uint16_t)));
      HIPCHK(h, launch_prefilter_norm(dl, dr, h->pf_l, h->pf_r, h->vsum, g, p.prefilter_size, h->stream));
    }
  }
  mark(h, 1);
  if (any_rows) {
    if (fast) {
      int xa = 0, xb = 0;
      HIPCHK(h, launch_sad_fast(h->pf_l, h->pf_r, h->disp_pre, h->cost, g, &xa, &xb, border, h->stream));
      snprintf(h->last_kernel, sizeof(h->last_kernel), "%s", g_sad_kernel_name);
      if (border && !sad_fast_borders_in_launch(g)) {   // beyond 256 disparities: the clamped columns from the sliding-sum kernel
        HIPCHK(h, launch_sad_wide(h->pf_l, h->pf_r, h->disp_pre, h->cost, g, 0, xa, h->stream));
        HIPCHK(h, launch_sad_wide(h->pf_l, h->pf_r, h->disp_pre, h->cost, g, xb, g.xend, h->stream));
      }
    } else if (sad_wide_supported(g) && env_switch("SBM_WIDE", 1)) {
      HIPCHK(h, launch_sad_wide(h->pf_l, h->pf_r, h->disp_pre, h->cost, g, 0, g.xend, h->stream));
      // (say so when this configuration only left the interior kernel's envelope because the in-place accumulate is off or
      // its device self-test failed: windows 29 / 31 and 257..512 disparities are 8-25x slower here, see include/sbm.h)
      const bool narrowed = !inplace && g.wsz <= 31 && g.nd <= kFastNdMax && (g.wsz > 27 || g.nd > 256);
      snprintf(h->last_kernel, sizeof(h->last_kernel), narrowed ? "sad_wide_kernel [in-place accumulate unavailable]" : "sad_wide_kernel");
    } else {
      HIPCHK(h, launch_sad_generic(h->pf_l, h->pf_r, h->disp_pre, h->cost, g, 0, g.xend, h->stream));
      snprintf(h->last_kernel, sizeof(h->last_kernel), "sad_generic_kernel");
    }
  }
  mark(h, 2);
  HIPCHK(h, launch_lrcheck(h->disp_pre, h->cost, out, g, p.disp12_max_diff, h->stream));
  mark(h, 4);
  if (speckle)
    HIPCHK(h, launch_speckle(out, h->spk_runs, h->spk_nheads, h->spk_seam, h->spk_nseam, g, p.speckle_window_size, p.speckle_range, h->stream));
  mark(h, 5);
  if (h->instr) h->calls++;
  h->ncall++;
  if (sync || h->profiling == 1) HIPCHK(h, hipStreamSynchronize(h->stream));
  return SBM_OK;
}

int sbm_rect_map_device(sbm_handle* h, const sbm_rect_cam* cam, int width, int height, void* d_map, int sync) {
  if (!h || !cam || !d_map) return SBM_ERR_NULL;
  if (width <= 0 || height <= 0 || width > 32767 || height > 32767) return SBM_ERR_SIZE;
  DeviceScope dscope(h->device);
  HIPCHK(h, dscope.enter());
  HIPCHK(h, launch_rect_map(*cam, width, height, (int16_t*)d_map, h->stream));
  if (sync) HIPCHK(h, hipStreamSynchronize(h->stream));
  return SBM_OK;
}

int sbm_rect_remap_device(sbm_handle* h, int n, const void* d_src, const void* d_map, int width, int height, void* d_dst,
                          int sync) {
  if (!h || !d_src || !d_map || !d_dst) return SBM_ERR_NULL;
  if (n <= 0) return SBM_ERR_BATCH;
  if (width <= 0 || height <= 0 || width > 32767 || height > 32767) return SBM_ERR_SIZE;
  if (((size_t)width * height + 1023) / 1024 > 65535) return SBM_ERR_UNSUPPORTED;
  DeviceScope dscope(h->device);
  HIPCHK(h, dscope.enter());
  HIPCHK(h, launch_rect_remap((const uint8_t*)d_src, (const int16_t*)d_map, (uint8_t*)d_dst, n, width, height, h->stream));
  if (sync) HIPCHK(h, hipStreamSynchronize(h->stream));
  return SBM_OK;
}

int sbm_prefilter_device(sbm_handle* h, int n, const void* d_src, int width, int height, int flavour, int cap,
                         void* d_dst, int sync) {
  if (!h || !d_src || !d_dst) return SBM_ERR_NULL;
  if (n <= 0) return SBM_ERR_BATCH;
  if (width <= 0 || height <= 0) return SBM_ERR_SIZE;
  if (flavour != SBM_PREFILTER_FLAVOUR_CV && flavour != SBM_PREFILTER_FLAVOUR_RTL) return SBM_ERR_PREFILTER_TYPE;
  if (flavour == SBM_PREFILTER_FLAVOUR_CV && (cap < 1 || cap > 63)) return SBM_ERR_PREFILTER_CAP;
  if (n > 65534 || height > 65535 * 4) return SBM_ERR_UNSUPPORTED;
  DeviceScope dscope(h->device);
  HIPCHK(h, dscope.enter());
  HIPCHK(h, launch_prefilter_dense((const uint8_t*)d_src, (uint8_t*)d_dst, n, width, height,
                                   flavour == SBM_PREFILTER_FLAVOUR_RTL, cap, h->stream));
  if (sync) HIPCHK(h, hipStreamSynchronize(h->stream));
  return SBM_OK;
}

// ---- FPGA flavour: register decode (bm.v:172-193), limits, entry points -----------------------------------------------------
int sbm_fpga_params_from_regs(uint32_t image_size, uint32_t bm_setting, uint32_t uni_filt_ctrl, sbm_fpga_params* out) {
  if (!out) return SBM_ERR_NULL;
  out->width = (int32_t)(image_size & 0x3ffu);
  out->height = (int32_t)((image_size >> 16) & 0x1ffu);
  out->block_size = (int32_t)((bm_setting >> 16) & 0x1fu);
  out->num_disparities = (int32_t)(bm_setting & 0x1ffu);
  out->uni_enable = (int32_t)((uni_filt_ctrl >> 31) & 1u);
  out->uni_mode = (int32_t)((uni_filt_ctrl >> 16) & 1u);
  out->uni_threshold = (int32_t)(uni_filt_ctrl & 0x3ffu);
  return SBM_OK;
}

uint32_t sbm_fpga_sad_size_reg(const sbm_fpga_params* p) {
  if (!p) return 0;
  const uint32_t hwsz = ((uint32_t)p->block_size >> 1) & 0xfu;
  const uint32_t hsad_wdt = ((uint32_t)p->width - (uint32_t)p->num_disparities - 1u) & 0x3ffu;   // bm.v:249
  const uint32_t sad_wdt = (hsad_wdt - 2u * hwsz) & 0x3ffu;                                        // bm.v:252
  const uint32_t sad_hgt = ((uint32_t)p->height - 2u * hwsz) & 0x1ffu;                             // bm.v:255
  return (sad_hgt << 16) | sad_wdt;
}

int sbm_fpga_params_validate(const sbm_fpga_params* p) {
  if (!p) return SBM_ERR_NULL;
  if (p->width <= 0 || p->height <= 0 || p->width > 1023 || p->height > 511) return SBM_ERR_SIZE;
  if (p->block_size < 3 || p->block_size > 31 || (p->block_size & 1) == 0) return SBM_ERR_BLOCK_SIZE;
  if (p->num_disparities < 32 || p->num_disparities > 256 || (p->num_disparities & 31)) return SBM_ERR_NUM_DISPARITIES;
  const int hwsz = p->block_size >> 1;
  if (p->width - p->num_disparities - 1 - 2 * hwsz < 1 || p->height - 2 * hwsz < 1) return SBM_ERR_SIZE;
  if (((p->num_disparities + hwsz + 1) & 31) == 0) return SBM_ERR_UNSUPPORTED;
  return SBM_OK;
}

static int ensure_fpga(sbm_handle* h, int n, int W, int H, bool need_xs) {
  const bool fits = n <= h->fp_n && W == h->fp_W && H == h->fp_H && h->fp_flag;
  if (!fits) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    free_fpga(h);
    const size_t npix = (size_t)n * W * H;
    HIPCHK(h, hipMalloc(&h->fp_rec, npix * 8));
    HIPCHK(h, hipMalloc((void**)&h->fp_flag, (size_t)n * sizeof(int)));
    HIPCHK(h, hipMemsetAsync(h->fp_flag, 0, (size_t)n * sizeof(int), h->stream));   // generation stamps: 0 = never saturated
    h->fp_gen = 0;
    h->fp_n = n; h->fp_W = W; h->fp_H = H;
  }
  if (need_xs && !h->fp_xs_l) {
    const size_t npix = (size_t)h->fp_n * W * H;
    HIPCHK(h, hipMalloc((void**)&h->fp_xs_l, npix + 64));
    HIPCHK(h, hipMalloc((void**)&h->fp_xs_r, npix + 64));
  }
  return SBM_OK;
}

int sbm_fpga_bm_device(sbm_handle* h, int n, const void* d_xsbl_l, const void* d_xsbl_r, const sbm_fpga_params* p,
                       void* d_disp, int sync) {
  if (!h || !d_xsbl_l || !d_xsbl_r || !p || !d_disp) return SBM_ERR_NULL;
  if (n <= 0) return SBM_ERR_BATCH;
  int st = sbm_fpga_params_validate(p);
  if (st != SBM_OK) return st;
  if (n > 65535) return SBM_ERR_UNSUPPORTED;
  DeviceScope dscope(h->device);
  HIPCHK(h, dscope.enter());
  st = ensure_fpga(h, n, p->width, p->height, false);
  if (st != SBM_OK) return st;
  HIPCHK(h, launch_fpga_bm((const uint8_t*)d_xsbl_l, (const uint8_t*)d_xsbl_r, h->fp_rec, h->fp_flag, ++h->fp_gen, (int16_t*)d_disp, n, *p,
                           h->stream));
  if (sync) HIPCHK(h, hipStreamSynchronize(h->stream));
  return SBM_OK;
}

int sbm_fpga_compute_device(sbm_handle* h, int n, const void* d_left, const void* d_right, const sbm_fpga_params* p,
                            void* d_disp, int sync) {
  if (!h || !d_left || !d_right || !p || !d_disp) return SBM_ERR_NULL;
  if (n <= 0) return SBM_ERR_BATCH;
  int st = sbm_fpga_params_validate(p);
  if (st != SBM_OK) return st;
  if (n > 65534) return SBM_ERR_UNSUPPORTED;
  DeviceScope dscope(h->device);
  HIPCHK(h, dscope.enter());
  st = ensure_fpga(h, n, p->width, p->height, true);
  if (st != SBM_OK) return st;
  HIPCHK(h, launch_prefilter_dense((const uint8_t*)d_left, h->fp_xs_l, n, p->width, p->height, 1, 31, h->stream));
  HIPCHK(h, launch_prefilter_dense((const uint8_t*)d_right, h->fp_xs_r, n, p->width, p->height, 1, 31, h->stream));
  HIPCHK(h, launch_fpga_bm(h->fp_xs_l, h->fp_xs_r, h->fp_rec, h->fp_flag, ++h->fp_gen, (int16_t*)d_disp, n, *p, h->stream));
  if (sync) HIPCHK(h, hipStreamSynchronize(h->stream));
  return SBM_OK;
}

static int ensure_staging(sbm_handle* h, int n, int W, int H);

// host-memory forms of the PL blocks for one frame: staged through the handle's device staging buffers (2-D copies
// take care of the caller's strides)
int sbm_fpga_compute(sbm_handle* h, const uint8_t* left, size_t left_stride, const uint8_t* right, size_t right_stride,
                     const sbm_fpga_params* p, int16_t* disp, size_t disp_stride) {
  if (!h || !left || !right || !p || !disp) return SBM_ERR_NULL;
  int st = sbm_fpga_params_validate(p);
  if (st != SBM_OK) return st;
  const int W = p->width, H = p->height;
  if (left_stride < (size_t)W || right_stride < (size_t)W || disp_stride < (size_t)W * 2) return SBM_ERR_SIZE;
  DeviceScope dscope(h->device);
  HIPCHK(h, dscope.enter());
  st = ensure_staging(h, 1, W, H);
  if (st != SBM_OK) return st;
  HIPCHK(h, hipMemcpy2DAsync(h->st_l, W, left, left_stride, W, H, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpy2DAsync(h->st_r, W, right, right_stride, W, H, hipMemcpyHostToDevice, h->stream));
  st = sbm_fpga_compute_device(h, 1, h->st_l, h->st_r, p, h->st_d, 0);
  if (st != SBM_OK) return st;
  HIPCHK(h, hipMemcpy2DAsync(disp, disp_stride, h->st_d, (size_t)W * 2, (size_t)W * 2, H, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return SBM_OK;
}

int sbm_gftt_eig(sbm_handle* h, const uint8_t* img, size_t img_stride, int width, int height, uint16_t* eig, size_t eig_stride,
                 uint32_t* max_out) {
  if (!h || !img || !eig) return SBM_ERR_NULL;
  if (width < 3 || height < 5 || width > 1023 || height > 511) return SBM_ERR_SIZE;
  if (img_stride < (size_t)width || eig_stride < (size_t)width * 2) return SBM_ERR_SIZE;
  DeviceScope dscope(h->device);
  HIPCHK(h, dscope.enter());
  int st = ensure_staging(h, 1, width, height);
  if (st != SBM_OK) return st;
  // st_l: image, st_d: map, st_r: the Max word
  HIPCHK(h, hipMemcpy2DAsync(h->st_l, width, img, img_stride, width, height, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, launch_gftt_eig(h->st_l, reinterpret_cast<uint16_t*>(h->st_d), reinterpret_cast<unsigned*>(h->st_r), 1, width, height, h->stream));
  HIPCHK(h, hipMemcpy2DAsync(eig, eig_stride, h->st_d, (size_t)width * 2, (size_t)width * 2, height, hipMemcpyDeviceToHost, h->stream));
  uint32_t mx = 0;
  HIPCHK(h, hipMemcpyAsync(&mx, h->st_r, sizeof(mx), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (max_out) *max_out = mx;
  return SBM_OK;
}

int sbm_gftt_eig_device(sbm_handle* h, int n, const void* d_img, int width, int height, void* d_eig, void* d_max, int sync) {
  if (!h || !d_img || !d_eig || !d_max) return SBM_ERR_NULL;
  if (n <= 0) return SBM_ERR_BATCH;
  if (width < 3 || height < 5 || width > 1023 || height > 511) return SBM_ERR_SIZE;
  if (n > 65535) return SBM_ERR_UNSUPPORTED;
  DeviceScope dscope(h->device);
  HIPCHK(h, dscope.enter());
  HIPCHK(h, launch_gftt_eig((const uint8_t*)d_img, (uint16_t*)d_eig, (unsigned*)d_max, n, width, height, h->stream));
  if (sync) HIPCHK(h, hipStreamSynchronize(h->stream));
  return SBM_OK;
}

int sbm_disparity_to_float_device(sbm_handle* h, int n, const void* d_disp, int width, int height, void* d_out, int sync) {
  if (!h || !d_disp || !d_out) return SBM_ERR_NULL;
  if (n <= 0) return SBM_ERR_BATCH;
  if (width <= 0 || height <= 0) return SBM_ERR_SIZE;
  DeviceScope dscope(h->device);
  HIPCHK(h, dscope.enter());
  HIPCHK(h, launch_disp_to_float((const int16_t*)d_disp, (float*)d_out, (size_t)n * width * height, h->stream));
  if (sync) HIPCHK(h, hipStreamSynchronize(h->stream));
  return SBM_OK;
}

int sbm_decimate_device(sbm_handle* h, int n, const void* d_disp, int width, int height, int scale, void* d_out, int sync) {
  if (!h || !d_disp || !d_out) return SBM_ERR_NULL;
  if (n <= 0) return SBM_ERR_BATCH;
  if (width <= 0 || height <= 0 || scale <= 0) return SBM_ERR_SIZE;
  DeviceScope dscope(h->device);
  HIPCHK(h, dscope.enter());
  HIPCHK(h, launch_decimate((const int16_t*)d_disp, (int16_t*)d_out, n, width, height, scale, h->stream));
  if (sync) HIPCHK(h, hipStreamSynchronize(h->stream));
  return SBM_OK;
}

int sbm_reproject_device(sbm_handle* h, int n, const void* d_disp, int width, int height, int scale,
                         const sbm_stereo_model* model, int apply_local, void* d_xyz, int sync) {
  if (!h || !d_disp || !d_xyz || !model) return SBM_ERR_NULL;
  if (n <= 0) return SBM_ERR_BATCH;
  if (width <= 0 || height <= 0 || scale <= 0) return SBM_ERR_SIZE;
  DeviceScope dscope(h->device);
  HIPCHK(h, dscope.enter());
  HIPCHK(h, launch_reproject((const int16_t*)d_disp, (float*)d_xyz, n, width, height, scale, *model, apply_local, h->stream));
  if (sync) HIPCHK(h, hipStreamSynchronize(h->stream));
  return SBM_OK;
}

int sbm_keypoints3d_device(sbm_handle* h, const void* d_disp, int width, int height, const void* d_kpts, int nk,
                           const sbm_stereo_model* model, float min_depth, float max_depth, void* d_xyz, int sync) {
  if (!h || !d_disp || !model || (nk > 0 && (!d_kpts || !d_xyz))) return SBM_ERR_NULL;
  if (width <= 0 || height <= 0 || nk < 0) return SBM_ERR_SIZE;
  DeviceScope dscope(h->device);
  HIPCHK(h, dscope.enter());
  HIPCHK(h, launch_keypoints3d((const int16_t*)d_disp, (const float*)d_kpts, (float*)d_xyz, width, height, nk, *model,
                               min_depth, max_depth, h->stream));
  if (sync) HIPCHK(h, hipStreamSynchronize(h->stream));
  return SBM_OK;
}

int sbm_get_profile(sbm_handle* h, const char* name, float* ms) {
  if (!h || !name || !ms) return SBM_ERR_NULL;
  DeviceScope dscope(h->device);
  HIPCHK(h, dscope.enter());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->profiling && h->ev_ok) collect_profile(h);
  if (!strcmp(name, "prefilter")) *ms = h->ms_prefilter;
  else if (!strcmp(name, "sad")) *ms = h->ms_sad;
  else if (!strcmp(name, "border")) *ms = h->ms_border;
  else if (!strcmp(name, "lrcheck")) *ms = h->ms_lr;
  else if (!strcmp(name, "speckle")) *ms = h->ms_speckle;
  else if (!strcmp(name, "total")) *ms = h->ms_total;
  else return SBM_ERR_UNSUPPORTED;
  return SBM_OK;
}

int sbm_last_kernel_name(sbm_handle* h, char* dst, size_t dst_bytes) {
  if (!h || !dst || dst_bytes == 0) return SBM_ERR_NULL;
  snprintf(dst, dst_bytes, "%s", h->last_kernel);
  return SBM_OK;
}

int sbm_debug_fetch(sbm_handle* h, int which, void* dst, size_t dst_bytes) {
  if (!h || !dst) return SBM_ERR_NULL;
  if (!h->have_last) return SBM_ERR_UNSUPPORTED;
  const Geom& g = h->last;
  DeviceScope dscope(h->device);
  HIPCHK(h, dscope.enter());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const size_t npix = (size_t)g.n * g.W * g.H;
  if (which == 0 || which == 1) {
    if (dst_bytes < npix) return SBM_ERR_SIZE;
    const uint8_t* src = which == 0 ? h->pf_l : h->pf_r;
    HIPCHK(h, hipMemcpy2D(dst, g.W, src + g.padl, g.pitch, g.W, (size_t)g.n * g.H, hipMemcpyDeviceToHost));
    uint8_t* d = (uint8_t*)dst;
    for (size_t i = 0; i < npix; i++) d[i] = (uint8_t)((d[i] - kPfBias) >> g.pfshift);
    return SBM_OK;
  }
  if (which == 2) {
    if (!h->cost) return SBM_ERR_UNSUPPORTED;
    if (dst_bytes < npix * sizeof(int32_t)) return SBM_ERR_SIZE;
    if (g.cost16) {
      uint16_t* tmp = (uint16_t*)malloc(npix * sizeof(uint16_t));
      if (!tmp) return SBM_ERR_NOMEM;
      hipError_t e = hipMemcpy(tmp, h->cost, npix * sizeof(uint16_t), hipMemcpyDeviceToHost);
      if (e == hipSuccess)
        for (size_t i = 0; i < npix; i++) ((int32_t*)dst)[i] = tmp[i];
      free(tmp);
      HIPCHK(h, e);
      return SBM_OK;
    }
    HIPCHK(h, hipMemcpy(dst, h->cost, npix * sizeof(int32_t), hipMemcpyDeviceToHost));
    return SBM_OK;
  }
  if (which == 3) {
    if (dst_bytes < npix * sizeof(int16_t)) return SBM_ERR_SIZE;
    HIPCHK(h, hipMemcpy(dst, h->disp_pre, npix * sizeof(int16_t), hipMemcpyDeviceToHost));
    // rows outside the valid ROI and the never-matchable column bands are not produced on the device
    int16_t* d = (int16_t*)dst;
    for (int i = 0; i < g.n; i++)
      for (int y = 0; y < g.H; y++) {
        int16_t* row = d + ((size_t)i * g.H + y) * g.W;
        const bool live = y >= g.row0 && y < g.row1;
        for (int x = 0; x < g.W; x++)
          if (!live || x < g.lofs || x >= g.lofs + g.xend) row[x] = (int16_t)g.filtered;
      }
    return SBM_OK;
  }
  return SBM_ERR_UNSUPPORTED;
}

static int ensure_staging(sbm_handle* h, int n, int W, int H) {
  if (n <= h->st_n && W == h->st_W && H == h->st_H && h->st_l) return SBM_OK;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  free_staging(h);
  const size_t npix = (size_t)n * W * H;
  HIPCHK(h, hipMalloc((void**)&h->st_l, npix + 64));
  HIPCHK(h, hipMalloc((void**)&h->st_r, npix + 64));
  HIPCHK(h, hipMalloc((void**)&h->st_d, npix * sizeof(int16_t)));
  h->st_n = n; h->st_W = W; h->st_H = H;
  return SBM_OK;
}

static int ensure_pipe(sbm_handle* h) {
  if (h->pipe_ok) return SBM_OK;
  HIPCHK(h, hipStreamCreateWithFlags(&h->stream_in, hipStreamNonBlocking));
  HIPCHK(h, hipStreamCreateWithFlags(&h->stream_out, hipStreamNonBlocking));
  for (int i = 0; i < sbm_handle::kChunks; i++) {
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_in[i], hipEventDisableTiming));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_done[i], hipEventDisableTiming));
  }
  h->pipe_ok = true;
  return SBM_OK;
}

// Large dense host batches: chunks of pairs flow through three streams -- H2D copies, compute, D2H copies -- so the GPU
// works on chunk k while chunk k+1 arrives and chunk k-1 leaves. With pageable caller memory the copies themselves still
// run one after the other on the calling thread (the runtime stages them), but the compute disappears behind them; with
// pinned (hipHostMalloc / hipHostRegister) caller memory the two copy directions overlap as well.
// maps [i0, i1) device -> caller, one transfer per run of maps that are contiguous in the caller's memory
static hipError_t copy_out_runs(sbm_handle* h, int16_t* const* disp, int i0, int i1, size_t npix1) {
  for (int i = i0; i < i1;) {
    int j = i + 1;
    while (j < i1 && disp[j] == disp[j - 1] + npix1) j++;
    const hipError_t e = hipMemcpyAsync(disp[i], h->st_d + i * npix1, (size_t)(j - i) * npix1 * 2, hipMemcpyDeviceToHost, h->stream_out);
    if (e != hipSuccess) return e;
    i = j;
  }
  return hipSuccess;
}

static int pipelined_enqueue(sbm_handle* h, int n, const uint8_t* const* left, const uint8_t* const* right, int width,
                             int height, int16_t* const* disp) {
  const size_t npix1 = (size_t)width * height;
  // Chunk plan. Small chunks overlap more of the transfers but run the kernels on part-filled launches (8 KITTI pairs cost
  // 0.36 ms on the device, 64 pairs 1.2 ms), so: a small FIRST chunk (the computation starts after one short transfer), a
  // small LAST one (only its computation and its maps are left when the inputs have arrived) and large ones in between.
  // Measured on 64 KITTI pairs from pinned memory (profiles/r03_host_feed.json). SBM_HOST_CHUNK=<pairs>: uniform chunks.
  static const int chunk_env = SBM_TUNE("SBM_HOST_CHUNK", 0);
  int start[sbm_handle::kChunks + 1];
  int nch = 0;
  start[0] = 0;
  if (chunk_env > 0 || n < 32) {
    int chunk = chunk_env > 0 ? chunk_env : 8;
    while ((n + chunk - 1) / chunk > sbm_handle::kChunks) chunk *= 2;
    for (int i = 0; i < n; i += chunk) start[++nch] = std::min(n, i + chunk);
  } else {
    const int edge = 8, mid = n - 2 * edge;
    int nmid = std::max(1, (mid + 23) / 24);                     // middle chunks of at most 24 pairs
    nmid = std::min(nmid, sbm_handle::kChunks - 2);
    start[++nch] = edge;
    for (int k = 1; k <= nmid; k++) start[++nch] = edge + (int)((long)mid * k / nmid);
    start[++nch] = n;
  }
  for (int k = 0; k < nch; k++) {
    const int i0 = start[k], cnt = start[k + 1] - i0;
    // images that follow each other in the caller's memory (one (n,H,W) array) travel as one transfer per run
    for (int i = i0; i < i0 + cnt;) {
      int j = i + 1;
      while (j < i0 + cnt && left[j] == left[j - 1] + npix1) j++;
      HIPCHK(h, hipMemcpyAsync(h->st_l + i * npix1, left[i], (size_t)(j - i) * npix1, hipMemcpyHostToDevice, h->stream_in));
      i = j;
    }
    for (int i = i0; i < i0 + cnt;) {
      int j = i + 1;
      while (j < i0 + cnt && right[j] == right[j - 1] + npix1) j++;
      HIPCHK(h, hipMemcpyAsync(h->st_r + i * npix1, right[i], (size_t)(j - i) * npix1, hipMemcpyHostToDevice, h->stream_in));
      i = j;
    }
    HIPCHK(h, hipEventRecord(h->ev_in[k], h->stream_in));
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_in[k], 0));
    const int st = sbm_compute_device(h, cnt, h->st_l + i0 * npix1, h->st_r + i0 * npix1, width, height, h->st_d + i0 * npix1, 0);
    if (st != SBM_OK) return st;
    HIPCHK(h, hipEventRecord(h->ev_done[k], h->stream));
    if (k > 0) {   // the previous chunk leaves while this one computes
      HIPCHK(h, hipStreamWaitEvent(h->stream_out, h->ev_done[k - 1], 0));
      HIPCHK(h, copy_out_runs(h, disp, start[k - 1], start[k], npix1));
    }
  }
  HIPCHK(h, hipStreamWaitEvent(h->stream_out, h->ev_done[nch - 1], 0));
  HIPCHK(h, copy_out_runs(h, disp, start[nch - 1], n, npix1));
  return SBM_OK;
}

static int compute_batch_pipelined(sbm_handle* h, int n, const uint8_t* const* left, const uint8_t* const* right, int width,
                                   int height, int16_t* const* disp) {
  int st = ensure_pipe(h);
  if (st != SBM_OK) return st;
  HIPCHK(h, hipStreamSynchronize(h->stream));   // staging buffers of an earlier call are free
  st = pipelined_enqueue(h, n, left, right, width, height, disp);
  // success or not: nothing may still be reading or writing the caller's buffers when this returns
  const hipError_t e1 = hipStreamSynchronize(h->stream_in), e2 = hipStreamSynchronize(h->stream),
                   e3 = hipStreamSynchronize(h->stream_out);
  if (st != SBM_OK) return st;
  HIPCHK(h, e1);
  HIPCHK(h, e2);
  HIPCHK(h, e3);
  return SBM_OK;
}

// ---- asynchronous dense feed -------------------------------------------------------------------------------------
// What a per-GPU feeder thread uses: batch k+1 is submitted (its inputs start crossing PCIe on the H2D stream) while batch k
// computes and batch k-1's maps travel back on the D2H stream. Whole batches, no chunking: the kernels run on full launches
// and in steady state a step costs its slowest leg (profiles/r03_host_feed.json). Caller buffers should be pinned
// (hipHostMalloc / hipHostRegister) -- pageable memory works but the runtime then copies synchronously.
// The runtime executes the copies of all streams in the order they were queued (measured: an H2D transfer queued behind a
// D2H one does not start before it, whatever their streams -- profiles/r03_host_feed.json), and a D2H copy can only run when
// its batch has been computed. So the maps of submission k are queued for their trip home only AFTER the inputs of
// submission k+1 (or when somebody waits for k): the inputs of k+1 then cross PCIe while k computes.
static int fq_flush_pending(sbm_handle* h) {
  if (!h->fq_pending_dst) return SBM_OK;
  const unsigned k = h->fq_submitted - 1u, slot = k & 1u, e = k & 3u;
  HIPCHK(h, hipStreamWaitEvent(h->stream_out, h->ev_fq_done[e], 0));
  HIPCHK(h, hipMemcpyAsync(h->fq_pending_dst, h->fq_d[slot], h->fq_pending_bytes, hipMemcpyDeviceToHost, h->stream_out));
  HIPCHK(h, hipEventRecord(h->ev_fq_out[e], h->stream_out));
  h->fq_pending_dst = nullptr;
  return SBM_OK;
}

int sbm_wait_oldest(sbm_handle* h) {
  if (!h) return SBM_ERR_NULL;
  if (h->fq_waited == h->fq_submitted) return SBM_OK;
  DeviceScope dscope(h->device);
  HIPCHK(h, dscope.enter());
  if (h->fq_waited + 1u == h->fq_submitted) {   // the newest submission: its maps may not have been queued yet
    const int st = fq_flush_pending(h);
    if (st != SBM_OK) return st;
  }
  HIPCHK(h, hipEventSynchronize(h->ev_fq_out[h->fq_waited & 3u]));
  h->fq_waited++;
  return SBM_OK;
}

int sbm_submit_dense(sbm_handle* h, int n, const uint8_t* left, const uint8_t* right, int width, int height, int16_t* disp) {
  if (!h || !left || !right || !disp) return SBM_ERR_NULL;
  if (n <= 0) return SBM_ERR_BATCH;
  int st = sbm_params_validate(&h->p, width, height);
  if (st != SBM_OK) return st;
  DeviceScope dscope(h->device);
  HIPCHK(h, dscope.enter());
  st = ensure_pipe(h);
  if (st != SBM_OK) return st;
  if (!h->fq_ok) {
    for (int k = 0; k < 4; k++) {
      HIPCHK(h, hipEventCreateWithFlags(&h->ev_fq_in[k], hipEventDisableTiming));
      HIPCHK(h, hipEventCreateWithFlags(&h->ev_fq_done[k], hipEventDisableTiming));
      HIPCHK(h, hipEventCreateWithFlags(&h->ev_fq_out[k], hipEventDisableTiming));
    }
    h->fq_ok = true;
  }
  while (h->fq_submitted - h->fq_waited >= 3u) {   // queue depth: one batch arriving, one computing, one leaving
    st = sbm_wait_oldest(h);
    if (st != SBM_OK) return st;
  }
  const size_t npix = (size_t)n * width * height;
  if (!(n <= h->fq_n && width == h->fq_W && height == h->fq_H && h->fq_l[0])) {
    while (h->fq_waited != h->fq_submitted) {
      st = sbm_wait_oldest(h);
      if (st != SBM_OK) return st;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    free_feed(h);
    for (int k = 0; k < 2; k++) {
      HIPCHK(h, hipMalloc((void**)&h->fq_l[k], npix + 64));
      HIPCHK(h, hipMalloc((void**)&h->fq_r[k], npix + 64));
      HIPCHK(h, hipMalloc((void**)&h->fq_d[k], npix * sizeof(int16_t)));
    }
    h->fq_n = n; h->fq_W = width; h->fq_H = height;
  }
  // submission k uses device staging set k & 1. The set's previous user is submission k-2: its inputs are free once k-2 has
  // computed, its map buffer once k-2's maps have left -- both are stream dependencies, the host never blocks on them.
  const unsigned k = h->fq_submitted, slot = k & 1u, e = k & 3u;
  if (k >= 2 && k - 2 >= h->fq_waited) {
    HIPCHK(h, hipStreamWaitEvent(h->stream_in, h->ev_fq_done[(k - 2) & 3u], 0));
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_fq_out[(k - 2) & 3u], 0));
  }
  HIPCHK(h, hipMemcpyAsync(h->fq_l[slot], left, npix, hipMemcpyHostToDevice, h->stream_in));
  HIPCHK(h, hipMemcpyAsync(h->fq_r[slot], right, npix, hipMemcpyHostToDevice, h->stream_in));
  HIPCHK(h, hipEventRecord(h->ev_fq_in[e], h->stream_in));
  st = fq_flush_pending(h);                      // the previous submission's maps: queued behind this one's inputs
  if (st != SBM_OK) return st;
  HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_fq_in[e], 0));
  st = sbm_compute_device(h, n, h->fq_l[slot], h->fq_r[slot], width, height, h->fq_d[slot], 0);
  if (st != SBM_OK) return st;
  HIPCHK(h, hipEventRecord(h->ev_fq_done[e], h->stream));
  h->fq_pending_dst = disp;
  h->fq_pending_bytes = npix * sizeof(int16_t);
  h->fq_submitted++;
  return SBM_OK;
}

// Maps of a small host-buffer call on their way out (the reference's pattern: one 640x480 pair per call, main.cpp:201-216).
// A D2H copy into pageable memory costs the call ~70 us after the last kernel (the runtime stages it: DMA + CPU copy) and the
// stream synchronisation behind it another ~15 (profiles/r05_host_attrib.txt). Instead the last kernel of the call copies the
// maps into pinned, device-mapped host memory and raises a sequence flag there (last workgroup done, system-scope release);
// the host spins on the flag and copies the rows to the caller itself.
// The maps leave in up to kZcChunks contiguous chunks, each with its own arrival counter and flag: the host copies chunk k to
// the caller while the chunks behind it are still crossing PCIe (round 6: the 25 us CPU copy of a 640x480 map used to START
// when the last byte had landed).
constexpr int kZcChunks = 8;
__global__ void __launch_bounds__(256) maps_out_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16, size_t per_chunk, int bpc,
                                                       const int16_t* __restrict__ src_tail, int16_t* __restrict__ dst_tail, int ntail, unsigned* cnt,
                                                       unsigned* flag, unsigned seq) {
  const int c = blockIdx.x / bpc, bi = blockIdx.x - c * bpc;   // chunk, block within the chunk
  const size_t lo = (size_t)c * per_chunk, hi = lo + per_chunk < n16 ? lo + per_chunk : n16;
  for (size_t i = lo + (size_t)bi * 256 + threadIdx.x; i < hi; i += (size_t)bpc * 256) {
    const uint4 v = src[i];
    __builtin_nontemporal_store(v.x, &dst[i].x); __builtin_nontemporal_store(v.y, &dst[i].y);
    __builtin_nontemporal_store(v.z, &dst[i].z); __builtin_nontemporal_store(v.w, &dst[i].w);
  }
  if (blockIdx.x == gridDim.x - 1 && (int)threadIdx.x < ntail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];   // (the last chunk's last block)
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned done = __hip_atomic_fetch_add(cnt + c, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (done == (unsigned)bpc) {
      __hip_atomic_store(cnt + c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence_system();
      __hip_atomic_store(flag + c, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

static int ensure_zc(sbm_handle* h, size_t bytes) {
  if (h->zc_out && h->zc_flag && h->zc_cnt && h->zc_bytes >= bytes) return SBM_OK;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  // flag and counter first, the staging last: zc_bytes only ever describes a complete set (a failure half way leaves a state
  // the next call simply completes)
  if (!h->zc_flag) {
    HIPCHK(h, hipHostMalloc((void**)&h->zc_flag, 64, hipHostMallocMapped | hipHostMallocCoherent));
    for (int k = 0; k < kZcChunks; k++) h->zc_flag[k] = 0u;
    h->zc_seq = 0u;
  }
  if (!h->zc_cnt) {
    unsigned* cnt = nullptr;
    HIPCHK(h, hipMalloc((void**)&cnt, 64));
    const hipError_t e = hipMemsetAsync(cnt, 0, 64, h->stream);
    if (e != hipSuccess) {
      hipFree(cnt);
      HIPCHK(h, e);
    }
    h->zc_cnt = cnt;
  }
  if (!(h->zc_out && h->zc_bytes >= bytes)) {
    if (h->zc_out) hipHostFree(h->zc_out);
    h->zc_out = nullptr; h->zc_bytes = 0;
    HIPCHK(h, hipHostMalloc((void**)&h->zc_out, bytes + 64, hipHostMallocMapped | hipHostMallocCoherent));
    h->zc_bytes = bytes;
  }
  return SBM_OK;
}

// queue the copy kernel behind the call's kernels; the maps arrive in h->zc_out chunk by chunk and go to the caller (n dense maps
// of npix1 pixels at disp[i]) as they arrive
static int maps_out_to_caller(sbm_handle* h, const int16_t* d_src, int n, size_t npix1, int16_t* const* disp) {
  const size_t count = (size_t)n * npix1, bytes = count * sizeof(int16_t);
  int st = ensure_zc(h, bytes);
  if (st != SBM_OK) return st;
  const size_t n16 = bytes / 16;
  const int ntail = (int)((bytes - n16 * 16) / 2);
  const unsigned seq = ++h->zc_seq == 0u ? ++h->zc_seq : h->zc_seq;   // (0 is "nothing yet")
  const int nch = (int)std::min<size_t>(kZcChunks, std::max<size_t>(1, bytes >> 16));        // chunks of at least 64 KB
  const size_t per_chunk = (n16 + nch - 1) / nch;
  const int bpc = (int)std::min<size_t>(256 / nch, std::max<size_t>(1, (per_chunk + 511) / 512));   // blocks per chunk
  hipLaunchKernelGGL(maps_out_kernel, dim3(nch * bpc), dim3(256), 0, h->stream, reinterpret_cast<const uint4*>(d_src), reinterpret_cast<uint4*>(h->zc_out), n16,
                     per_chunk, bpc, d_src + n16 * 8, h->zc_out + n16 * 8, ntail, h->zc_cnt, h->zc_flag, seq);
  HIPCHK(h, hipGetLastError());
  // Poll the chunk flags in order: a short pure spin (a one-pair call ends within tens of microseconds of the launch), then spin
  // with yields so that a loaded host or many engines driven from many threads do not burn a core each, and after 2 ms the
  // runtime's own wait -- also the way out when the stream has failed and the flags will never be raised.
  const auto t0 = std::chrono::steady_clock::now();
  bool synced = false;
  size_t done = 0;   // int16 elements already with the caller
  for (int c = 0; c < nch; c++) {
    unsigned spins = 0;
    while (!synced && __atomic_load_n(h->zc_flag + c, __ATOMIC_ACQUIRE) != seq) {
      cpu_relax();
      if ((++spins & 0xffu) == 0u) {
        const auto dt = std::chrono::steady_clock::now() - t0;
        if (dt > std::chrono::milliseconds(2)) {
          HIPCHK(h, hipStreamSynchronize(h->stream));
          synced = true;
        } else if (dt > std::chrono::microseconds(150)) {
          std::this_thread::yield();
        }
      }
    }
    // elements [done, end) have landed: hand them to the maps they belong to
    const size_t end = c == nch - 1 ? count : std::min(count, (size_t)(c + 1) * per_chunk * 8);
    while (done < end) {
      const size_t i = done / npix1, off = done - i * npix1, len = std::min(end - done, npix1 - off);
      memcpy(disp[i] + off, h->zc_out + done, len * sizeof(int16_t));
      done += len;
    }
  }
  return SBM_OK;
}

// One dense host batch over several engines -- the C++ caller's form of "pair batches shard across the GPUs of a node"
// (one process, one stream set per device): handle k takes the contiguous block of pairs
// [n k / K, n (k + 1) / K), cut into at most two submissions of its asynchronous feed so that the second half's inputs cross
// PCIe while the first half computes; every device's submissions are queued before anything is waited for, so the devices run
// side by side from ONE host thread. Pairs are independent: no data-path collective, the blocks' maps land in `disp` in place.
int sbm_compute_batch_multi(sbm_handle* const* handles, int n_handles, int n, const uint8_t* left, const uint8_t* right,
                            int width, int height, int16_t* disp) {
  if (!handles || !left || !right || !disp) return SBM_ERR_NULL;
  if (n_handles <= 0 || n <= 0) return SBM_ERR_BATCH;
  for (int k = 0; k < n_handles; k++) {
    if (!handles[k]) return SBM_ERR_NULL;
    for (int j = 0; j < k; j++)
      if (handles[j] == handles[k]) return SBM_ERR_BATCH;   // a handle owns one feed: the same one twice would interleave its staging sets
  }
  const size_t npix1 = (size_t)width * height;
  int first_err = SBM_OK;
  for (int part = 0; part < 2 && first_err == SBM_OK; part++)
    for (int k = 0; k < n_handles && first_err == SBM_OK; k++) {
      const long b0 = (long)n * k / n_handles, b1 = (long)n * (k + 1) / n_handles;   // this engine's block
      const long half = (b1 - b0 + 1) / 2;
      const long c0 = part == 0 ? b0 : b0 + half, c1 = part == 0 ? b0 + half : b1;
      if (c1 <= c0) continue;
      first_err = sbm_submit_dense(handles[k], (int)(c1 - c0), left + c0 * npix1, right + c0 * npix1, width, height, disp + c0 * npix1);
    }
  // drain every engine even after a failure: what was queued writes into `disp`, which the caller may free on return
  for (int k = 0; k < n_handles; k++) {
    const int st = sbm_synchronize(handles[k]);
    if (first_err == SBM_OK) first_err = st;
  }
  return first_err;
}

int sbm_compute_batch(sbm_handle* h, int n, const uint8_t* const* left, size_t left_stride, const uint8_t* const* right,
                      size_t right_stride, int width, int height, int16_t* const* disp, size_t disp_stride) {
  if (!h || !left || !right || !disp) return SBM_ERR_NULL;
  if (n <= 0) return SBM_ERR_BATCH;
  int st = sbm_params_validate(&h->p, width, height);
  if (st != SBM_OK) return st;
  if (left_stride < (size_t)width || right_stride < (size_t)width || disp_stride < (size_t)width * 2) return SBM_ERR_SIZE;
  for (int i = 0; i < n; i++)
    if (!left[i] || !right[i] || !disp[i]) return SBM_ERR_NULL;
  HP_BEGIN();
  DeviceScope dscope(h->device);
  HIPCHK(h, dscope.enter());
  st = ensure_staging(h, n, width, height);
  if (st != SBM_OK) return st;
  HP(0);
  const size_t npix1 = (size_t)width * height;
  // Dense caller images (stride == width, what cv::Mat::isContinuous() gives) go through plain 1-D copies. Strided ones
  // are packed row by row into pinned staging on the CPU: a 2-D copy from pageable memory degenerates into one small
  // transfer per row (measured 5.6 ms per 1242x375 pair against 0.2 ms packed).
  const bool in_dense = left_stride == (size_t)width && right_stride == (size_t)width;
  const bool out_dense = disp_stride == (size_t)width * 2;
  static const int pipe_env = SBM_TUNE("SBM_HOST_PIPELINE", 1);
  if (in_dense && out_dense && n >= 16 && pipe_env && !h->profiling)
    return compute_batch_pipelined(h, n, left, right, width, height, disp);
  if (!in_dense || !out_dense) {
    const size_t need = (size_t)n * npix1 * 4;
    if (need > h->pin_bytes) {
      HIPCHK(h, hipStreamSynchronize(h->stream));
      if (h->pin) hipHostFree(h->pin);
      h->pin = nullptr; h->pin_bytes = 0;
      HIPCHK(h, hipHostMalloc((void**)&h->pin, need, hipHostMallocDefault));
      h->pin_bytes = need;
    }
  }
  uint8_t* pin_l = h->pin;
  uint8_t* pin_r = h->pin ? h->pin + (size_t)n * npix1 : nullptr;
  uint8_t* pin_d = h->pin ? h->pin + (size_t)n * npix1 * 2 : nullptr;
  if (in_dense) {
    for (int i = 0; i < n; i++) {
      HIPCHK(h, hipMemcpyAsync(h->st_l + i * npix1, left[i], npix1, hipMemcpyHostToDevice, h->stream));
      HIPCHK(h, hipMemcpyAsync(h->st_r + i * npix1, right[i], npix1, hipMemcpyHostToDevice, h->stream));
    }
  } else {
    for (int i = 0; i < n; i++)
      for (int y = 0; y < height; y++) {
        memcpy(pin_l + i * npix1 + (size_t)y * width, left[i] + (size_t)y * left_stride, width);
        memcpy(pin_r + i * npix1 + (size_t)y * width, right[i] + (size_t)y * right_stride, width);
      }
    HIPCHK(h, hipMemcpyAsync(h->st_l, pin_l, (size_t)n * npix1, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->st_r, pin_r, (size_t)n * npix1, hipMemcpyHostToDevice, h->stream));
  }
  HP(1);
  st = sbm_compute_device(h, n, h->st_l, h->st_r, width, height, h->st_d, 0);
  if (st != SBM_OK) return st;
  HP(2);
  // Small calls into PAGEABLE caller memory (what a cv::Mat is): copy kernel into pinned host memory + flag, then the rows go to
  // the caller from there (see maps_out_kernel; 640x480: 0.199 -> 0.187 ms per call). Pinned caller memory takes the D2H copy
  // below: the DMA engine writes it directly and nothing is left for the CPU to copy (0.158 against 0.180 ms through the kernel).
  bool zero_copy = false;
  if (out_dense && (size_t)n * npix1 * 2 <= ((size_t)8 << 20) && !h->profiling && env_switch("SBM_HOST_ZEROCOPY", 1)) {
    hipPointerAttribute_t attr;
    const hipError_t pe = hipPointerGetAttributes(&attr, disp[0]);
    if (pe != hipSuccess) (void)hipGetLastError();   // (pageable memory is unknown to the runtime: that is the answer, not an error)
    zero_copy = !(pe == hipSuccess && (attr.type == hipMemoryTypeHost || attr.type == hipMemoryTypeManaged || attr.type == hipMemoryTypeDevice));
  }
  if (zero_copy) {
    st = maps_out_to_caller(h, h->st_d, n, npix1, disp);
    if (st != SBM_OK) return st;
    HP(4);
  } else if (out_dense) {
    for (int i = 0; i < n; i++)
      HIPCHK(h, hipMemcpyAsync(disp[i], h->st_d + i * npix1, npix1 * 2, hipMemcpyDeviceToHost, h->stream));
    HP(3);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HP(4);
  } else {
    HIPCHK(h, hipMemcpyAsync(pin_d, h->st_d, (size_t)n * npix1 * 2, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < n; i++)
      for (int y = 0; y < height; y++)
        memcpy((uint8_t*)disp[i] + (size_t)y * disp_stride, pin_d + (i * npix1 + (size_t)y * width) * 2, (size_t)width * 2);
  }
  return SBM_OK;
}

int sbm_compute(sbm_handle* h, const uint8_t* left, size_t left_stride, const uint8_t* right, size_t right_stride, int width,
                int height, int16_t* disp, size_t disp_stride) {
  const uint8_t* l[1] = {left};
  const uint8_t* r[1] = {right};
  int16_t* d[1] = {disp};
  if (!left || !right || !disp) return SBM_ERR_NULL;
  return sbm_compute_batch(h, 1, l, left_stride, r, right_stride, width, height, d, disp_stride);
}

}  // extern "C"
