// sbm_post.hip -- post-filters of the disparity map: left-right consistency + ROI fill, speckle filter. gfx950.
//
// Device counterparts of validateDisparity / filterSpeckles (OpenCV calib3d stereosgbm.cpp), switched on by
// setDisp12MaxDiff(1), setSpeckleWindowSize(50), setSpeckleRange(32) at src/slam/src/core/main.cpp:210-212.
// Neither has an FPGA twin in the reference.
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
  const int32_t* cost;
  int16_t* disp_out;
  int W, H, mindisp, nd, tol, filtered, row0, row1, col0, col1, do_lr;
  int cx0, cx1;  // columns [cx0,cx1) of disp_pre were computed; the rest reads as FILTERED
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
  const int32_t* cp = a.cost + base;
  const int INV = a.filtered;
  const int minX1 = max(max(a.mindisp + a.nd, 0), a.cx0), maxX1 = min(a.W + min(a.mindisp, 0), a.cx1);
  for (int x = threadIdx.x; x < a.W; x += 256) lr_keys[x] = ~0ull;
  __syncthreads();
  for (int x = minX1 + threadIdx.x; x < maxX1; x += 256) {
    const int d = dp[x];
    if (d == INV) continue;
    const int x2 = x - ((d + 8) >> 4);
    if (x2 >= 0 && x2 < a.W)
      atomicMin(&lr_keys[x2], ((unsigned long long)(unsigned)cp[x] << 32) | (unsigned)x);
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
          const int d2 = dp[(int)(k & 0xffffffffu)];
          bad_a = abs(d2 - d) > a.tol;
        }
      }
      if (xb >= 0 && xb < a.W) {
        const unsigned long long k = lr_keys[xb];
        if (k != ~0ull) {
          const int d2 = dp[(int)(k & 0xffffffffu)];
          bad_b = abs(d2 - d) > a.tol;
        }
      }
      if (bad_a && bad_b) d = INV;
    }
    out[x] = (int16_t)d;
  }
}

hipError_t launch_lrcheck(const int16_t* disp_pre, const int32_t* cost, int16_t* disp_out, const Geom& g,
                          int disp12_max_diff, hipStream_t s) {
  LrArgs a;
  a.disp_pre = disp_pre; a.cost = cost; a.disp_out = disp_out;
  a.W = g.W; a.H = g.H; a.mindisp = g.mindisp; a.nd = g.nd; a.tol = disp12_max_diff * 16; a.filtered = g.filtered;
  a.row0 = g.row0; a.row1 = g.row1; a.col0 = g.col0; a.col1 = g.col1; a.do_lr = disp12_max_diff >= 0;
  a.cx0 = g.lofs; a.cx1 = g.lofs + g.xend;
  size_t lds = a.do_lr ? (size_t)g.W * sizeof(unsigned long long) : 0;
  hipLaunchKernelGGL(lrcheck_kernel, dim3(g.H, g.n), dim3(256), lds, s, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Speckle filter.  cv's raster-order flood fill yields plain 4-connected components of the relation
// "both != newVal and |a-b| <= maxDiff" (order-independent), so it is computed here as
// run-based union-find:
//   1. runs    one wavefront per image row: every pixel gets the index of the first pixel of its horizontal run
//              (ballot + count-leading-zeros, carry across 64-pixel chunks); run heads are their own parents.
//   2. merge   one thread per pixel: union the runs of vertically connected pixels, skipping contacts that the
//              pixel to the left already made (same two runs) -- so the number of atomics ~ number of run contacts.
//   3. count   root of every pixel; a wavefront adds each equal-root lane segment with ONE atomic, and stops
//              adding to a component once it is known to exceed maxSpeckleSize (kills contention on large regions).
//   4. apply   components with count <= maxSpeckleSize become newVal.
// labels doubles as the parent array (indices within the pair's plane, -1 = invalid pixel).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int uf_find(const int* L, int i) {
  int r = i;
  for (;;) {
    const int p = __hip_atomic_load(L + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p == r) return r;
    r = p;
  }
}

__device__ __forceinline__ void uf_union(int* L, int a, int b) {
  for (;;) {
    a = uf_find(L, a);
    b = uf_find(L, b);
    if (a == b) return;
    if (a > b) { const int t = a; a = b; b = t; }
    const int old = atomicMin(L + b, a);
    if (old == b) return;
    b = old;
  }
}

// grid: (ceil(H/4), n), block 256 = 4 wavefronts = 4 rows
__global__ void __launch_bounds__(256) speckle_runs_kernel(const int16_t* __restrict__ disp, int* __restrict__ labels,
                                                            int* __restrict__ counts, int W, int H, int newval,
                                                            int maxdiff) {
  const int lane = threadIdx.x & 63;
  const int y = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (y >= H) return;
  const size_t po = (size_t)blockIdx.y * W * H + (size_t)y * W;
  const int16_t* d = disp + po;
  int* L = labels + po;
  int* C = counts + po;
  int carry = -1;         // run start of the pixel just left of this chunk (only used when connected to it)
  int prev_last = newval; // value of the pixel just left of this chunk
  for (int cb = 0; cb < W; cb += 64) {
    const int x = cb + lane;
    const bool in = x < W;
    const int v = in ? (int)d[x] : newval;
    int pv = __shfl_up(v, 1, 64);
    if (lane == 0) pv = prev_last;
    const bool valid = v != newval;
    const bool joined = valid && pv != newval && abs(v - pv) <= maxdiff;  // connected to the left neighbour
    const bool head = valid && !joined;
    const unsigned long long m = __ballot(head) & ((2ull << lane) - 1ull);
    const int start = m ? cb + (63 - __clzll((long long)m)) : carry;
    if (in) {
      L[x] = valid ? y * W + start : -1;
      C[x] = 0;
    }
    carry = __shfl(valid ? start : -1, 63, 64);
    prev_last = __shfl(v, 63, 64);
  }
}

__global__ void __launch_bounds__(256) speckle_merge_kernel(const int16_t* __restrict__ disp, int* __restrict__ labels,
                                                             int W, int H, int newval, int maxdiff) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int npix = W * H;
  if (i >= npix - W) return;  // last row has no row below
  const size_t po = (size_t)blockIdx.y * npix;
  const int16_t* d = disp + po;
  int* L = labels + po;
  const int v = d[i], u = d[i + W];
  if (v == newval || u == newval || abs(v - u) > maxdiff) return;
  const int la = L[i], lb = L[i + W];
  const int x = i % W;
  if (x > 0) {
    const int v1 = d[i - 1], u1 = d[i - 1 + W];
    if (v1 != newval && u1 != newval && abs(v1 - u1) <= maxdiff && L[i - 1] == la && L[i - 1 + W] == lb) return;
  }
  uf_union(L, la, lb);
}

__global__ void __launch_bounds__(256) speckle_count_kernel(int* __restrict__ labels, int* __restrict__ counts, int npix,
                                                             int maxsize) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const size_t po = (size_t)blockIdx.y * npix;
  int* L = labels + po;
  int r = -1;
  if (i < npix) {
    const int l = L[i];
    if (l >= 0) {
      // the merge kernel has finished: parents are final, so plain (L1-cacheable) loads are safe here; concurrent
      // shortcut stores below only ever write a node's final root
      r = l;
      for (int p2 = L[r]; p2 != r; p2 = L[r]) r = p2;
      L[i] = r;
    }
  }
  int pr = __shfl_up(r, 1, 64);
  const bool seg = lane == 0 || pr != r;
  const unsigned long long m = __ballot(seg);
  if (seg && r >= 0) {
    const unsigned long long above = lane == 63 ? 0ull : (m >> (lane + 1));
    const int len = above ? __ffsll((long long)above) : 64 - lane;
    int* c = counts + po + r;
    if (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= maxsize) atomicAdd(c, len);
  }
}

__global__ void __launch_bounds__(256) speckle_apply_kernel(int16_t* __restrict__ disp, const int* __restrict__ labels,
                                                             const int* __restrict__ counts, int npix, int newval,
                                                             int maxsize) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= npix) return;
  const size_t po = (size_t)blockIdx.y * npix;
  const int r = labels[po + i];
  if (r >= 0 && counts[po + r] <= maxsize) disp[po + i] = (int16_t)newval;
}

hipError_t launch_speckle(int16_t* disp, int32_t* labels, int32_t* counts, const Geom& g, int max_size, int max_diff,
                          hipStream_t s) {
  const int npix = g.W * g.H;
  dim3 grid((npix + 255) / 256, g.n);
  hipLaunchKernelGGL(speckle_runs_kernel, dim3((g.H + 3) / 4, g.n), dim3(256), 0, s, disp, labels, counts, g.W, g.H,
                     g.filtered, max_diff);
  hipLaunchKernelGGL(speckle_merge_kernel, grid, dim3(256), 0, s, disp, labels, g.W, g.H, g.filtered, max_diff);
  hipLaunchKernelGGL(speckle_count_kernel, grid, dim3(256), 0, s, labels, counts, npix, max_size);
  hipLaunchKernelGGL(speckle_apply_kernel, grid, dim3(256), 0, s, disp, labels, counts, npix, g.filtered, max_size);
  return hipGetLastError();
}

}  // namespace sbm
