// sbm_sad_generic.hip -- parameter-complete SAD / WTA / texture / uniqueness / sub-pixel kernel, gfx950.
//
// Device counterpart of findStereoCorrespondenceBM (OpenCV calib3d stereobm.cpp) as reached from
// src/slam/src/core/main.cpp:215; in-tree hardware twins: src/dvp/rtl/bm_calc_sad.v (SAD), bm_calc_det.v (WTA),
// bm_calc_frac.v (sub-pixel).  It evaluates the per-pixel definition for ANY block size, disparity count,
// minDisparity and for the clamped windows of the border columns:
//
//   AD(x',y,d) = |Lp[y][lofs + clamp(x',-lofs,W-lofs-1)] - Rp[y][rofs + clamp(x',-rofs,W-rofs-nd) + d]|
//   SAD(x,y,d) = sum over the w x w window, rows clamped to the image (only reachable with a custom ROI)
//   first d attaining the minimum wins; texture / uniqueness reject; parabola-like sub-pixel; 32-bit sums.
//
// Role: (1) the border columns whose windows are clamped (needed bit-exactly because the LR check reads them
// before they are overwritten), (2) fallback for configurations outside the fast kernel's envelope.
// Layout: one 64-lane wavefront per output column, lanes = disparities (d = lane + 64k), marching down a row
// segment with vertical sliding sums held in LDS; WTA is a wavefront min-reduction on 64-bit (sad,d) keys.
#include "sbm_common.h"

namespace sbm {

constexpr int kGenCols = 4;   // columns (= wavefronts) per workgroup
constexpr int kGenSeg = 32;   // output rows per workgroup

struct GenArgs {
  const uint8_t* pf_l;
  const uint8_t* pf_r;
  int16_t* disp;
  int32_t* cost;
  int W, H, pitch, padl, plane;
  int nd, mindisp, wsz, cap, lofs, rofs, tex, uniq, filtered;
  int row0, row1, xa, xb;
};

__device__ __forceinline__ int iclampd(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// horizontal window sum of one row for one (column, disparity index)
__device__ __forceinline__ int hrow(const uint8_t* lrow, const uint8_t* rrow, int x, int d, int w2, int lofs, int rofs,
                                    int W, int nd) {
  int s = 0;
  for (int dx = -w2; dx <= w2; dx++) {
    const int xp = x + dx;
    const int lv = lrow[lofs + iclampd(xp, -lofs, W - lofs - 1)];
    const int rv = rrow[rofs + iclampd(xp, -rofs, W - rofs - nd) + d];
    const int t = lv - rv;
    s += t < 0 ? -t : t;
  }
  return s;
}

__device__ __forceinline__ int trow(const uint8_t* lrow, int x, int w2, int lofs, int W, int capb) {
  int s = 0;
  for (int dx = -w2; dx <= w2; dx++) {
    const int t = (int)lrow[lofs + iclampd(x + dx, -lofs, W - lofs - 1)] - capb;
    s += t < 0 ? -t : t;
  }
  return s;
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    unsigned long long o = __shfl_xor(v, off, 64);
    v = o < v ? o : v;
  }
  return v;
}

extern __shared__ __attribute__((aligned(16))) int gen_lds[];  // [kGenCols][nd + 2] running SAD per column

__global__ void __launch_bounds__(64 * kGenCols) sad_generic_kernel(GenArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int x = a.xa + blockIdx.x * kGenCols + wave;  // column relative to lofs
  const bool active = x < a.xb;
  const int xc = active ? x : a.xa;                   // idle waves shadow a valid column (barriers stay uniform)
  const int ys = a.row0 + blockIdx.y * kGenSeg;
  const int ye = min(ys + kGenSeg, a.row1);
  const int pair = blockIdx.z;
  const int w2 = a.wsz / 2;
  const int capb = a.cap + kPfBias;
  const uint8_t* pl = a.pf_l + (size_t)pair * a.plane + a.padl;
  const uint8_t* pr = a.pf_r + (size_t)pair * a.plane + a.padl;
  int* sad = gen_lds + wave * (a.nd + 2) + 1;  // sad[-1], sad[nd] are the mirrored ends
  const size_t obase = (size_t)pair * a.W * a.H;

  // prime with the window of row ys-1: rows ys-w2-1 .. ys+w2-1 (clamped to the image)
  int tsum = 0;
  for (int d = lane; d < a.nd; d += 64) sad[d] = 0;
  for (int yy = ys - w2 - 1; yy <= ys + w2 - 1; yy++) {
    const int yc = iclampd(yy, 0, a.H - 1);
    const uint8_t* lrow = pl + (size_t)yc * a.pitch;
    const uint8_t* rrow = pr + (size_t)yc * a.pitch;
    for (int d = lane; d < a.nd; d += 64) sad[d] += hrow(lrow, rrow, xc, d, w2, a.lofs, a.rofs, a.W, a.nd);
    tsum += trow(lrow, xc, w2, a.lofs, a.W, capb);
  }

  for (int y = ys; y < ye; y++) {
    const int ya = iclampd(y + w2, 0, a.H - 1), yb = iclampd(y - w2 - 1, 0, a.H - 1);
    const uint8_t* la = pl + (size_t)ya * a.pitch;
    const uint8_t* ra = pr + (size_t)ya * a.pitch;
    const uint8_t* lb = pl + (size_t)yb * a.pitch;
    const uint8_t* rb = pr + (size_t)yb * a.pitch;
    unsigned long long best = ~0ull;
    for (int d = lane; d < a.nd; d += 64) {
      const int s = sad[d] + hrow(la, ra, xc, d, w2, a.lofs, a.rofs, a.W, a.nd) -
                    hrow(lb, rb, xc, d, w2, a.lofs, a.rofs, a.W, a.nd);
      sad[d] = s;
      const unsigned long long key = ((unsigned long long)(unsigned)s << 32) | (unsigned)d;
      best = key < best ? key : best;
    }
    tsum += trow(la, xc, w2, a.lofs, a.W, capb) - trow(lb, xc, w2, a.lofs, a.W, capb);
    best = wave_min_u64(best);
    __syncthreads();  // running sums visible to the whole wavefront (and keeps LDS traffic ordered)
    const int minsad = (int)(best >> 32), mind = (int)(best & 0xffffffffu);
    int out = a.filtered;
    bool ok = tsum >= a.tex;
    if (ok && a.uniq > 0) {
      const int thresh = minsad + (minsad * a.uniq / 100);
      bool hit = false;
      for (int d = lane; d < a.nd; d += 64) hit |= (d < mind - 1 || d > mind + 1) && sad[d] <= thresh;
      ok = __ballot(hit) == 0ull;
    }
    if (ok) {
      const int p = mind + 1 < a.nd ? sad[mind + 1] : sad[a.nd - 2];
      const int n = mind - 1 >= 0 ? sad[mind - 1] : sad[1];
      const int ad = p > n ? p - n : n - p;
      const int den = p + n - 2 * minsad + ad;
      out = ((a.nd - mind - 1 + a.mindisp) * 256 + (den != 0 ? (p - n) * 256 / den : 0) + 15) >> 4;
    }
    if (active && lane == 0) {
      const size_t o = obase + (size_t)y * a.W + a.lofs + x;
      a.disp[o] = (int16_t)out;
      if (a.cost && ok) a.cost[o] = minsad;
    }
    __syncthreads();  // reads of sad[] done before the next row overwrites it
  }
}

hipError_t launch_sad_generic(const uint8_t* pf_l, const uint8_t* pf_r, int16_t* disp, int32_t* cost, const Geom& g,
                              int xa, int xb, hipStream_t s) {
  if (xb <= xa || g.row1 <= g.row0) return hipSuccess;
  GenArgs a;
  a.pf_l = pf_l; a.pf_r = pf_r; a.disp = disp; a.cost = g.want_cost ? cost : nullptr;
  a.W = g.W; a.H = g.H; a.pitch = g.pitch; a.padl = g.padl; a.plane = g.plane;
  a.nd = g.nd; a.mindisp = g.mindisp; a.wsz = g.wsz; a.cap = g.cap; a.lofs = g.lofs; a.rofs = g.rofs;
  a.tex = g.tex; a.uniq = g.uniq; a.filtered = g.filtered;
  a.row0 = g.row0; a.row1 = g.row1; a.xa = xa; a.xb = xb;
  dim3 grid((xb - xa + kGenCols - 1) / kGenCols, (g.row1 - g.row0 + kGenSeg - 1) / kGenSeg, g.n);
  size_t lds = (size_t)kGenCols * (g.nd + 2) * sizeof(int);
  hipLaunchKernelGGL(sad_generic_kernel, grid, dim3(64 * kGenCols), lds, s, a);
  return hipGetLastError();
}

}  // namespace sbm
