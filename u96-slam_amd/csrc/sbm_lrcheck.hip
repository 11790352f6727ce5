// sbm_lrcheck.hip -- left-right consistency check + valid-ROI fill of the disparity map. gfx950.
//
// Device counterpart of validateDisparity (OpenCV calib3d stereosgbm.cpp), switched on by setDisp12MaxDiff(1) at
// src/slam/src/core/main.cpp:212; also writes the rows / columns that can never hold a match as FILTERED (the fill cv::StereoBM
// does around its stripes). No FPGA twin in the reference.
#include <stdlib.h>

#include <algorithm>

#include "sbm_common.h"

namespace sbm {

// ---------------------------------------------------------------------------------------------------------
// LR check.  cv semantics per row: pass 1 walks x upward and lets each valid left pixel claim the right-view
// column x2 = x - round(d) if its cost is strictly smaller than the current claimant's  => the winner of a slot is
// the claimant with minimum (cost, x).  That is order-free: one LDS atomicMin on the 64-bit key cost<<32|x per
// pixel.  Pass 2 is per-pixel.  One workgroup owns one image row; the same kernel writes the never-valid rows and
// columns (outside the valid ROI) as FILTERED, which cv does after validateDisparity.
// ---------------------------------------------------------------------------------------------------------
struct LrArgs {
  const int16_t* disp_pre;
  const void* cost;   // uint16 plane when cost16, else int32
  int16_t* disp_out;
  int cost16;
  int W, H, mindisp, nd, tol, filtered, row0, row1, col0, col1, do_lr;
  int cx0, cx1;  // columns [cx0,cx1) of disp_pre were computed; the rest reads as FILTERED
  int cost_short, tie_later;   // alternative readings (sbm_common.h kRead*): generic kernel only
};

extern __shared__ __attribute__((aligned(16))) unsigned long long lr_keys[];

__global__ void __launch_bounds__(256) lrcheck_kernel(LrArgs a) {
  const int y = blockIdx.x;
  const size_t base = ((size_t)blockIdx.y * a.H + y) * a.W;
  int16_t* out = a.disp_out + base;
  if (y < a.row0 || y >= a.row1) {
    for (int x = threadIdx.x; x < a.W; x += 256) out[x] = (int16_t)a.filtered;
    return;
  }
  const int16_t* dp = a.disp_pre + base;
  if (!a.do_lr) {
    const int lo = max(a.col0, a.cx0), hi = min(a.col1, a.cx1);
    for (int x = threadIdx.x; x < a.W; x += 256) out[x] = (x >= lo && x < hi) ? dp[x] : (int16_t)a.filtered;
    return;
  }
  const uint16_t* cp16 = static_cast<const uint16_t*>(a.cost) + base;
  const int32_t* cp32 = static_cast<const int32_t*>(a.cost) + base;
  const int INV = a.filtered;
  const int minX1 = max(max(a.mindisp + a.nd, 0), a.cx0), maxX1 = min(a.W + min(a.mindisp, 0), a.cx1);
  for (int x = threadIdx.x; x < a.W; x += 256) lr_keys[x] = ~0ull;
  __syncthreads();
  for (int x = minX1 + threadIdx.x; x < maxX1; x += 256) {
    const int d = dp[x];
    if (d == INV) continue;
    const int x2 = x - ((d + 8) >> 4);
    if (x2 >= 0 && x2 < a.W) {
      unsigned c = a.cost16 ? (unsigned)cp16[x] : (unsigned)cp32[x];
      if (a.cost_short) c = (unsigned)((int)(short)c + 32768);            // order of the wrapped `short`
      // cheapest claimant, then the earliest x (cv's strict '>'); tie_later: then the latest x
      atomicMin(&lr_keys[x2], ((unsigned long long)c << 32) | (unsigned)(a.tie_later ? a.W - 1 - x : x));
    }
  }
  __syncthreads();
  for (int x = threadIdx.x; x < a.W; x += 256) {
    int d = (x >= a.cx0 && x < a.cx1) ? dp[x] : INV;
    if (x < a.col0 || x >= a.col1) {
      d = INV;
    } else if (d != INV && x >= minX1 && x < maxX1) {
      const int xa = x - (d >> 4), xb = x - ((d + 15) >> 4);
      bool bad_a = false, bad_b = false;
      if (xa >= 0 && xa < a.W) {
        const unsigned long long k = lr_keys[xa];
        if (k != ~0ull) {
          const int xw = (int)(k & 0xffffffffu);
          const int d2 = dp[a.tie_later ? a.W - 1 - xw : xw];
          bad_a = abs(d2 - d) > a.tol;
        }
      }
      if (xb >= 0 && xb < a.W) {
        const unsigned long long k = lr_keys[xb];
        if (k != ~0ull) {
          const int xw = (int)(k & 0xffffffffu);
          const int d2 = dp[a.tie_later ? a.W - 1 - xw : xw];
          bad_b = abs(d2 - d) > a.tol;
        }
      }
      if (bad_a && bad_b) d = INV;
    }
    out[x] = (int16_t)d;
  }
}

// Fast variant for the 16-bit cost plane (fast interior strips + border wavefronts) and rows of at most 4096 columns: the row's
// disparities and costs are loaded once, up front, and the claim on a right-view column is ONE 32-bit LDS word,
// (cost << 16) | (disparity ^ 0x8000), taken with ds_min_u32. The winner's disparity rides in the key: among the claimants of
// one column x2 = x - round(d/16) a smaller x means a strictly smaller d, so "lowest cost, then lowest d" picks the pixel that
// cv's "lowest cost, then lowest x" picks -- no second look-up of the winner's disparity, no copy of the row in LDS.
// Layout: key[0] front pad (column -1), key[1 + x] column x, key[W+1], key[W+2] "no claimant" (targets outside the row read
// here), then 64 per-lane dummy slots that absorb the claims of pixels that make none (branch-free ds_min).
//
// The kernel is bound by VALU issue (round 3: 50 vector instructions per pixel slot, 0.9 VALU busy), so the work went into the
// instruction count of the common case -- a wavefront whose 128 columns lie inside every column range that matters (computed
// columns, checked columns, valid ROI, the row itself) -- which now runs without a single range test:
//  * the SAD kernels store cost 0xffff with every filtered pixel, so such a pixel's own key is >= 0xffff0000 = "no claimant":
//    it may take part in the ds_min like any other (its target column x + 1 - minDisparity is inside the key array);
//  * a filtered pixel's verdict does not matter (it stays filtered), and a valid pixel's two targets x - floor(d/16),
//    x - ceil(d/16) are inside the row by the definition of the checked column range;
//  * |dw - d| > tol as ONE 16-bit subtraction per target: t = (dw - d + tol) mod 2^16 > 2 tol (disparities of one map are less
//    than 16384 apart here and tol is clamped to 16384: no aliasing), straight on the key's low half.
// Two adjacent pixels per thread and iteration (4-byte loads / stores; rows are only 2-byte aligned: unaligned ones): half the
// LDS bank conflicts of the four-pixel layout, a third of its unpacking.
extern __shared__ __attribute__((aligned(16))) unsigned lr_lds32[];

template <int NIT, int PX>   // PX = 2 or 4 adjacent pixels per thread and iteration
__global__ void __launch_bounds__(320) lrcheck16_kernel(LrArgs a) {
  const int BS = blockDim.x;   // 64..320 threads (a multiple of 64): narrow rows get a narrower block
  const int y = blockIdx.x;
  const size_t base = ((size_t)blockIdx.y * a.H + y) * a.W;
  int16_t* out = a.disp_out + base;
  if (y < a.row0 || y >= a.row1) {
    for (int x = threadIdx.x; x < a.W; x += BS) out[x] = (int16_t)a.filtered;
    return;
  }
  unsigned* const key = lr_lds32;
  const int W = a.W;
  const int16_t* dp = a.disp_pre + base;
  const uint16_t* cp = static_cast<const uint16_t*>(a.cost) + base;
  const int INV = a.filtered;
  const int minX1 = max(max(a.mindisp + a.nd, 0), a.cx0), maxX1 = min(W + min(a.mindisp, 0), a.cx1);
  constexpr unsigned NONE = 0xffffffffu;   // real keys stay below 0xffff0000 (costs <= 65534 in this envelope)
  const unsigned tol2 = 2u * (unsigned)a.tol;            // (a.tol <= 16384, see launch_lrcheck)
  const int dbias = 0x8000 - a.tol;
  const int wbase = PX * __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
  const int lo = max(max(minX1, a.cx0), a.col0), hi = min(min(maxX1, a.cx1), min(a.col1, W));
  bool inner[NIT];
#pragma unroll
  for (int k = 0; k < NIT; k++) inner[k] = wbase + PX * BS * k >= lo && wbase + PX * BS * k + 64 * PX <= hi;
  int dv[NIT][PX];
  unsigned kv[NIT][PX];
  auto load = [&](auto inner_t, const int k) {
    constexpr bool IN = decltype(inner_t)::value;
    const int x0 = PX * (threadIdx.x + BS * k);
    unsigned dd[PX / 2], cc[PX / 2];
#pragma unroll
    for (int j = 0; j < PX / 2; j++) dd[j] = cc[j] = 0;
    if (IN || x0 + PX <= W) {
      __builtin_memcpy(dd, dp + x0, 2 * PX);
      __builtin_memcpy(cc, cp + x0, 2 * PX);
    } else {
      for (int i = 0; x0 + i < W; i++) {
        dd[i >> 1] |= (unsigned)(unsigned short)dp[x0 + i] << (16 * (i & 1));
        cc[i >> 1] |= (unsigned)cp[x0 + i] << (16 * (i & 1));
      }
    }
#pragma unroll
    for (int j = 0; j < PX / 2; j++) {
      const unsigned db = dd[j] ^ 0x80008000u;
      dv[k][2 * j] = (int)(short)(dd[j] & 0xffffu);
      dv[k][2 * j + 1] = (int)dd[j] >> 16;
      kv[k][2 * j] = __builtin_amdgcn_perm(cc[j], db, 0x05040100u);       // cost.lo : biased disparity.lo
      kv[k][2 * j + 1] = __builtin_amdgcn_perm(cc[j], db, 0x07060302u);   // cost.hi : biased disparity.hi
    }
    if (!IN) {
#pragma unroll
      for (int i = 0; i < PX; i++)
        if (!(x0 + i >= a.cx0 && x0 + i < a.cx1)) dv[k][i] = INV;
    }
  };
  const int dummy = W + 3 + (threadIdx.x & 63);
  auto claim = [&](auto inner_t, const int k) {
    constexpr bool IN = decltype(inner_t)::value;
#pragma unroll
    for (int i = 0; i < PX; i++) {
      const int x = PX * (threadIdx.x + BS * k) + i;
      const int d = dv[k][i];
      const int x2 = x - ((d + 8) >> 4);
      if (IN) {
        __hip_atomic_fetch_min(&key[x2 + 1], kv[k][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
        const bool claims = d != INV && x >= minX1 && x < maxX1 && (unsigned)x2 < (unsigned)W;
        __hip_atomic_fetch_min(&key[claims ? x2 + 1 : dummy], kv[k][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  };
  auto check = [&](auto inner_t, const int k) {
    constexpr bool IN = decltype(inner_t)::value;
    const int x0 = PX * (threadIdx.x + BS * k);
    if (!IN && x0 >= W) return;
    // targets x - floor(d/16) and x - ceil(d/16): the same column or two adjacent ones -> one two-word read. A pixel that
    // is not checked, or whose floor target lies outside the row, reads "no claimant" twice (it cannot fail then: both
    // targets must disagree); a ceil target of -1 reads the front pad. The reads are issued before the first use, and the
    // verdict is plain arithmetic (no short-circuit: a branch per pixel would serialise the LDS latencies).
    unsigned k0[PX], k1[PX];
#pragma unroll
    for (int i = 0; i < PX; i++) {
      const int x = x0 + i;
      const int d = dv[k][i];
      const int xa = x - (d >> 4);
      int p = xa;
      if (!IN) {
        const bool checked = ((int)(d != INV) & (int)(x >= minX1 && x < maxX1) & (int)((unsigned)xa < (unsigned)W)) != 0;
        p = checked ? xa : W + 1;
      }
      k0[i] = key[p];            // claim of column xa-1
      k1[i] = key[p + 1];        // claim of column xa
    }
    unsigned res[PX];
#pragma unroll
    for (int i = 0; i < PX; i++) {
      const int x = x0 + i;
      const int d = dv[k][i];
      const unsigned ka = k1[i], kb = (d & 15) ? k0[i] : k1[i];
      const unsigned dbt = (unsigned)(d + dbias);
      const unsigned ta = (ka - dbt) & 0xffffu, tb = (kb - dbt) & 0xffffu;    // (winner's disparity - d + tol) mod 2^16
      const bool bad = ((int)(max(ka, kb) < 0xffff0000u) & (int)(min(ta, tb) > tol2)) != 0;
      res[i] = (unsigned)(((!IN && (x < a.col0 || x >= a.col1)) || bad) ? INV : d);
    }
    if (IN || x0 + PX <= W) {
      unsigned r2[PX / 2];
#pragma unroll
      for (int j = 0; j < PX / 2; j++) r2[j] = __builtin_amdgcn_perm(res[2 * j + 1], res[2 * j], 0x05040100u);
      __builtin_memcpy(out + x0, r2, 2 * PX);
    } else {
      for (int i = 0; x0 + i < W; i++) out[x0 + i] = (int16_t)res[i];
    }
  };
#pragma unroll
  for (int k = 0; k < NIT; k++) {
    if (inner[k]) load(std::true_type{}, k);
    else load(std::false_type{}, k);
  }
  for (int i = 4 * threadIdx.x; i < W + 3; i += 4 * BS) *reinterpret_cast<uint4*>(key + i) = make_uint4(NONE, NONE, NONE, NONE);
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NIT; k++) {
    if (inner[k]) claim(std::true_type{}, k);
    else claim(std::false_type{}, k);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NIT; k++) {
    if (inner[k]) check(std::true_type{}, k);
    else check(std::false_type{}, k);
  }
}

hipError_t launch_lrcheck(const int16_t* disp_pre, const int32_t* cost, int16_t* disp_out, const Geom& g,
                          int disp12_max_diff, hipStream_t s) {
  LrArgs a;
  a.disp_pre = disp_pre; a.cost = cost; a.disp_out = disp_out; a.cost16 = g.cost16;
  a.W = g.W; a.H = g.H; a.mindisp = g.mindisp; a.nd = g.nd; a.tol = disp12_max_diff * 16; a.filtered = g.filtered;
  a.row0 = g.row0; a.row1 = g.row1; a.col0 = g.col0; a.col1 = g.col1; a.do_lr = disp12_max_diff >= 0;
  a.cx0 = g.lofs; a.cx1 = g.lofs + g.xend;
  a.cost_short = (g.reading & kReadCostShort) != 0; a.tie_later = (g.reading & kReadLrTieLater) != 0;
  // (the 16-bit verdict of lrcheck16_kernel: disparities of one map less than 16384 apart, tolerance clamped to that -- a larger
  // one passes everything either way)
  if (a.do_lr && g.cost16 && g.W <= 4096 && (g.nd + 1) * 16 <= 16384 && !a.cost_short && !a.tie_later) {
    a.tol = std::min(a.tol, 16384);
    const size_t lds16 = (size_t)(g.W + 3 + 64 + 8) * sizeof(unsigned);   // claims, pads, per-lane dummy slots
    // PX pixels per thread and iteration, blocks of at most BSMAX threads, the fewest iterations that cover the row with them and
    // then the narrowest block (a multiple of 64) that does. Measured (profiles/r04_lr_sweep.txt): what counts is how few idle
    // pixel slots the cover leaves and, for rows up to ~1300 columns, few wavefronts per row -- 2 pixels x 5 iterations x one
    // (640 columns) or two (1242) wavefronts; wider rows do best with 4 pixels x 256 threads x 2..4 iterations.
    int px = g.W <= 1280 ? 2 : 4;
    int bsmax = px == 4 ? 256 : (g.W <= 640 ? 64 : 128);
    if (SBM_TUNE("SBM_DEV_LR_PX", 0)) px = SBM_TUNE("SBM_DEV_LR_PX", 0) == 2 ? 2 : 4;
    if (SBM_TUNE("SBM_DEV_LR_BS", 0)) bsmax = std::max(64, std::min(320, SBM_TUNE("SBM_DEV_LR_BS", 0) / 64 * 64));
    const int groups = (g.W + px - 1) / px;
    const int nit = (groups + bsmax - 1) / bsmax;
    const int bs = (((groups + nit - 1) / nit + 63) / 64) * 64;
    const dim3 grid(g.H, g.n), block(bs);
    bool launched = true;
#define SBM_LR_CASE(N, P) case N * 8 + P: hipLaunchKernelGGL((lrcheck16_kernel<N, P>), grid, block, lds16, s, a); break;
    switch (nit * 8 + px) {
      SBM_LR_CASE(1, 2) SBM_LR_CASE(2, 2) SBM_LR_CASE(3, 2) SBM_LR_CASE(4, 2) SBM_LR_CASE(5, 2)   // up to 1280 columns
      SBM_LR_CASE(2, 4) SBM_LR_CASE(3, 4) SBM_LR_CASE(4, 4)                                       // 1281 .. 4096 columns
#ifdef SBM_DEV
      SBM_LR_CASE(1, 4) SBM_LR_CASE(5, 4) SBM_LR_CASE(6, 4) SBM_LR_CASE(7, 4) SBM_LR_CASE(8, 4) SBM_LR_CASE(6, 2) SBM_LR_CASE(7, 2) SBM_LR_CASE(8, 2)
#endif
      default: launched = false;
    }
#undef SBM_LR_CASE
    if (launched) return hipGetLastError();
    a.tol = disp12_max_diff * 16;   // (no such instantiation: the generic kernel below)
  }
  size_t lds = a.do_lr ? (size_t)g.W * sizeof(unsigned long long) : 0;
  hipLaunchKernelGGL(lrcheck_kernel, dim3(g.H, g.n), dim3(256), lds, s, a);
  return hipGetLastError();
}

}  // namespace sbm
