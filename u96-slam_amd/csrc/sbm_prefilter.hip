// sbm_prefilter.hip -- x-Sobel prefilter (cv prefilterXSobel semantics), gfx950.
//
// Replaces, on the device, the first stage of cv::StereoBM::compute as called from
// src/slam/src/core/main.cpp:215 (in-tree hardware twin: src/dvp/rtl/xsbl2.v:661-874, which clips to
// [-32,31] instead of [-cap,cap]).  out = clip(d(y-1) + 2 d(y) + d(y+1), -cap, cap) + cap with
// d(r) = r[x+1] - r[x-1]; rows mirrored (reflect-101) at the top and, for even H, at the bottom; columns 0 and
// W-1 = cap; for odd H the last row is all cap.
//
// HBM-bound: 1 byte read + 1 byte written per pixel (the three source rows of a strip hit in L2). Each thread
// produces 16 adjacent pixels from three unaligned 16+4-byte loads (SWAR vertical sums) and stores 16 bytes. The result is written with
// a +1 bias into a zero-padded plane (see kPfBias in sbm_common.h).
#include <stdlib.h>

#include "sbm_common.h"

namespace sbm {

__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t* p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}

// value mapping of one flavour: out = clip(s, lo, hi) + off (+ bias of the destination plane)
//   cv  (prefilterXSobel):  lo = -cap, hi = cap, off = cap; reflect-101 rows; odd H: last row = cap
//   rtl (xsbl2.v:185-198):  lo = -32,  hi = 31,  off = 32;  rows 0 and H-1 are never written by the RTL (= 0)
struct PfMap {
  int lo, hi, off, bias, rtl;
  int shift;   // out = ((clip + off) << shift) + bias  (engine planes of the pre-scaled fast path, else 0)
  int odd_row_cap;   // cv flavour: the last row of an odd-height image is all cap (0: computed like any other row, SBM_CV_READING bit 3)
};
__device__ __forceinline__ int clipmap(int v, const PfMap& m) { return (v < m.lo ? m.lo : (v > m.hi ? m.hi : v)) + m.off; }

__device__ __forceinline__ uint4 load_u128_ua(const uint8_t* p) {
  uint4 v;
  __builtin_memcpy(&v, p, 16);
  return v;
}

typedef short s16x2 __attribute__((ext_vector_type(2)));

// bytes (b0,b1,b2,b3) of a word -> packed u16 pairs: even = (b0, b2), odd = (b1, b3)
__device__ __forceinline__ uint32_t even_bytes(uint32_t w) { return w & 0x00ff00ffu; }
__device__ __forceinline__ uint32_t odd_bytes(uint32_t w) { return __builtin_amdgcn_perm(0u, w, 0x0c030c01u); }

// packed (2 x i16): (clip(a - b, lo, hi) + off) * mul + bias   (one v_pk_mad_i16 for scale and bias)
__device__ __forceinline__ uint32_t diff_clip(uint32_t a, uint32_t b, s16x2 lo, s16x2 hi, s16x2 off, s16x2 mul, s16x2 bias) {
  s16x2 d = __builtin_bit_cast(s16x2, a) - __builtin_bit_cast(s16x2, b);
  d = (__builtin_elementwise_min(__builtin_elementwise_max(d, lo), hi) + off) * mul + bias;
  return __builtin_bit_cast(uint32_t, d);
}

// Each thread produces a 16-pixel x PF_ROWS-row tile: PF_ROWS + 2 source row pieces are loaded once and every
// vertical 1-2-1 sum reuses them (1.5 loads per output row instead of 3).
// grid: x = ceil(ceil(H/PF_ROWS)*ceil(W/16)/256), y = 2*n (image index: even = left, odd = right)
template <int PF_ROWS>
__global__ void __launch_bounds__(256) prefilter_kernel(const uint8_t* __restrict__ left, const uint8_t* __restrict__ right,
                                                        uint8_t* __restrict__ pf_l, uint8_t* __restrict__ pf_r, int W, int H,
                                                        int pitch, int padl, size_t sstride, size_t plane, size_t extent_l,
                                                        size_t extent_r,
                                                        PfMap m) {
  // threads are flattened over (row group, 16-pixel piece) of one image so that every lane has work whatever the width
  const int npiece = (W + 15) / 16;
  const int ngroup = (H + PF_ROWS - 1) / PF_ROWS;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= npiece * ngroup) return;
  const int yg = t / npiece;
  const int y0 = yg * PF_ROWS;
  const int x0 = (t - yg * npiece) * 16;
  const int img = blockIdx.y >> 1;
  const uint8_t* src = ((blockIdx.y & 1) ? right : left) + (size_t)img * sstride;
  uint8_t* dst0 = ((blockIdx.y & 1) ? pf_r : pf_l) + img * plane + padl + x0;
  // Row pieces at the image edges read a few bytes of the neighbouring row (or image) instead of taking a clamped,
  // divergent path: those bytes only feed the forced edge columns. Only a piece whose 20-byte window would leave the
  // caller's buffer (`extent` bytes from the side's base pointer: first row of the first image, last row of the last
  // one) assembles its window byte by byte.
  const size_t ioff = (size_t)img * sstride;
  const long lo = (long)ioff + (long)(y0 > 0 ? y0 - 1 : 0) * W + x0 - 1;
  const long hi = (long)ioff + (long)(y0 + PF_ROWS < H ? y0 + PF_ROWS : H - 1) * W + x0 + 19;
  const bool interior = lo >= 0 && hi <= (long)((blockIdx.y & 1) ? extent_r : extent_l);

  // packed pieces of source rows y0-1 .. y0+PF_ROWS (reflect-101 at the image border): bytes x0-1 .. x0+18
  uint32_t rw[PF_ROWS + 2][5];
#pragma unroll
  for (int k = 0; k < PF_ROWS + 2; k++) {
    int yy = y0 - 1 + k;
    yy = yy < 0 ? (H > 1 ? 1 : 0) : (yy > H - 1 ? (yy == H ? (H > 1 ? H - 2 : 0) : H - 1) : yy);
    const uint8_t* r = src + (size_t)yy * W;
    if (interior) {
      const uint4 a = load_u128_ua(r + x0 - 1);
      rw[k][0] = a.x; rw[k][1] = a.y; rw[k][2] = a.z; rw[k][3] = a.w;
      rw[k][4] = load_u32_unaligned(r + x0 + 15);
    } else {
#pragma unroll
      for (int w = 0; w < 5; w++) {
        uint32_t v = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          int x = x0 - 1 + 4 * w + i;
          x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);  // value unused where clamped (edge columns are forced to cap)
          v |= (uint32_t)r[x] << (8 * i);
        }
        rw[k][w] = v;
      }
    }
  }

  // unpack every source row piece once: E[k][w] = columns (4w, 4w+2), O[k][w] = columns (4w+1, 4w+3) of the piece,
  // as packed u16 pairs (column index relative to x0-1)
  uint32_t E[PF_ROWS + 2][5], O[PF_ROWS + 2][5];
#pragma unroll
  for (int k = 0; k < PF_ROWS + 2; k++)
#pragma unroll
    for (int w = 0; w < 5; w++) {
      E[k][w] = even_bytes(rw[k][w]);
      O[k][w] = odd_bytes(rw[k][w]);
    }
  const s16x2 vlo = {(short)m.lo, (short)m.lo}, vhi = {(short)m.hi, (short)m.hi};
  const s16x2 voff = {(short)m.off, (short)m.off};
  const s16x2 vmul = {(short)(1 << m.shift), (short)(1 << m.shift)}, vbias = {(short)m.bias, (short)m.bias};
  const uint32_t edge = (uint32_t)((m.off << m.shift) + m.bias);

#pragma unroll
  for (int j = 0; j < PF_ROWS; j++) {
    const int y = y0 + j;
    if (y >= H) break;
    uint32_t out[4];
    if (m.rtl ? (y == 0 || y == H - 1) : (m.odd_row_cap && (H & 1) && y == H - 1)) {
      out[0] = out[1] = out[2] = out[3] = (uint32_t)(((m.rtl ? 0 : m.off) << m.shift) + m.bias) * 0x01010101u;
    } else {
      // vertical 1-2-1 sums (max 4*255: no carry between the halves), then the horizontal difference
      // s[i+2] - s[i] on packed pairs: pixels (4q, 4q+2) from the even sums, (4q+1, 4q+3) from the odd sums
      uint32_t SE[5], SO[5];
#pragma unroll
      for (int w = 0; w < 5; w++) {
        SE[w] = E[j][w] + 2u * E[j + 1][w] + E[j + 2][w];
        SO[w] = O[j][w] + 2u * O[j + 1][w] + O[j + 2][w];
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t pe = diff_clip(__builtin_amdgcn_alignbit(SE[q + 1], SE[q], 16), SE[q], vlo, vhi, voff, vmul, vbias);
        const uint32_t po = diff_clip(__builtin_amdgcn_alignbit(SO[q + 1], SO[q], 16), SO[q], vlo, vhi, voff, vmul, vbias);
        out[q] = pe | (po << 8);
      }
      // image columns 0 and W-1 carry the offset value
      if (x0 == 0) out[0] = (out[0] & 0xffffff00u) | edge;
      const int xl = W - 1 - x0;
      if (xl >= 0 && xl < 16) {
#pragma unroll
        for (int q = 0; q < 4; q++)
          if ((xl >> 2) == q) out[q] = (out[q] & ~(0xffu << (8 * (xl & 3)))) | (edge << (8 * (xl & 3)));
      }
    }
    uint8_t* dst = dst0 + (size_t)y * pitch;
    if (x0 + 16 <= W) {
      if (((pitch | padl) & 15) == 0) {
        *reinterpret_cast<uint4*>(dst) = make_uint4(out[0], out[1], out[2], out[3]);  // engine planes: 16-byte aligned
      } else {
        const uint4 o4 = make_uint4(out[0], out[1], out[2], out[3]);
        __builtin_memcpy(dst, &o4, 16);                                                // dense caller-owned plane
      }
    } else if (pitch >= padl + x0 + 16) {
      // engine planes: the bytes right of column W-1 are padding that must read as 0 (the masked value of the SAD kernel)
      const int nb = W - x0;  // 1..15 valid bytes
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int r = nb - 4 * q;
        out[q] = r >= 4 ? out[q] : (r <= 0 ? 0u : (out[q] & (0xffffffffu >> (8 * (4 - r)))));
      }
      *reinterpret_cast<uint4*>(dst) = make_uint4(out[0], out[1], out[2], out[3]);
    } else {
      // dense plane, 1..15 valid bytes: at most four partial stores (8 + 4 + 2 + 1)
      const int nb = W - x0;
      int q = 0;
      uint8_t* d = dst;
      if (nb & 8) { const uint2 o2 = make_uint2(out[0], out[1]); __builtin_memcpy(d, &o2, 8); d += 8; q = 2; }
      if (nb & 4) { const uint32_t o1 = q ? out[2] : out[0]; __builtin_memcpy(d, &o1, 4); d += 4; q++; }
      const uint32_t rest = q == 0 ? out[0] : (q == 1 ? out[1] : (q == 2 ? out[2] : out[3]));
      if (nb & 2) { const uint16_t o16 = (uint16_t)rest; __builtin_memcpy(d, &o16, 2); d += 2; }
      if (nb & 1) *d = (uint8_t)((nb & 2) ? (rest >> 16) : rest);
    }
  }
}

// Rows per thread tile (PF_ROWS + 2 row pieces are loaded for PF_ROWS output rows). Measured on MI355X
// (tools/bench_prefilter.py, profiles/r02_prefilter.json): 4 rows is best while source + destination fit the 256 MB
// Infinity Cache (4.3 vs 4.0 TB/s at 119 MB), 8 rows beyond it (4.4-4.9 vs 4.1-4.5 TB/s at 1.9-2.1 GB, i.e. 98-105 % of
// the runtime's device-to-device copy of the same bytes). SBM_PF_ROWS=2|4|8 overrides.
static int pf_rows(size_t bytes_in_out) {
  static const int env = [] { const int v = SBM_TUNE("SBM_PF_ROWS", 0); return (v == 2 || v == 4 || v == 8) ? v : 0; }();
  if (env) return env;
  return bytes_in_out > ((size_t)256 << 20) ? 8 : 4;
}

hipError_t launch_prefilter(const uint8_t* d_left, const uint8_t* d_right, uint8_t* pf_l, uint8_t* pf_r,
                            const Geom& g, hipStream_t s) {
  const int npiece = (g.W + 15) / 16;
  const int rows = pf_rows((size_t)4 * g.n * g.W * g.H);
  const int ngroup = (g.H + rows - 1) / rows;
  dim3 grid((npiece * ngroup + 255) / 256, 2 * g.n);
  const PfMap m{-g.cap, g.cap, g.cap, kPfBias, 0, g.pfshift, (g.reading & kReadOddRowComputed) ? 0 : 1};
#define SBM_PF(R) hipLaunchKernelGGL(prefilter_kernel<R>, grid, dim3(256), 0, s, d_left, d_right, pf_l, pf_r, g.W, g.H, g.pitch, g.padl, \
                     (size_t)g.W * g.H, (size_t)g.plane, (size_t)g.n * g.W * g.H, (size_t)g.n * g.W * g.H, m)
  if (rows == 2) SBM_PF(2); else if (rows == 8) SBM_PF(8); else SBM_PF(4);
#undef SBM_PF
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// PREFILTER_NORMALIZED_RESPONSE (cv prefilterNorm; SURVEY.md A.2 -- not used by the reference, built so that the whole
// cv::StereoBM parameter surface runs on the device).  Two passes over each image:
//   1. column sums of the winsize rows around y (rows replicated at the border): one thread per column slides down a
//      block of rows (coalesced across columns), uint16 plane in scratch;
//   2. per pixel: horizontal sum of winsize column sums (columns replicated), centre-weighted term, clip to [0, 2 cap],
//      written with the +1 bias into the padded plane like the x-Sobel output.
// Both are small next to the SAD kernel at the default winsize 9 (O(winsize) loads per pixel in pass 2).
// ---------------------------------------------------------------------------------------------------------
constexpr int PFN_ROWS = 32;

// grid: (ceil(W/256), ceil(H/PFN_ROWS), 2n)
__global__ void __launch_bounds__(256) pf_norm_vsum_kernel(const uint8_t* __restrict__ left, const uint8_t* __restrict__ right,
                                                           uint16_t* __restrict__ vs, int W, int H, int wsz2) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= W) return;
  const int img = blockIdx.z >> 1;
  const uint8_t* src = ((blockIdx.z & 1) ? right : left) + (size_t)img * W * H + x;
  uint16_t* out = vs + (size_t)blockIdx.z * W * H + x;
  const int y0 = blockIdx.y * PFN_ROWS, y1 = min(y0 + PFN_ROWS, H);
  int sum = 0;
  for (int dy = -wsz2; dy <= wsz2; dy++) sum += src[(size_t)min(max(y0 + dy, 0), H - 1) * W];
  for (int y = y0; y < y1; y++) {
    out[(size_t)y * W] = (uint16_t)sum;
    sum += (int)src[(size_t)min(y + 1 + wsz2, H - 1) * W] - (int)src[(size_t)max(y - wsz2, 0) * W];
  }
}

// grid: (ceil(W/256), H, 2n)
__global__ void __launch_bounds__(256) pf_norm_resp_kernel(const uint8_t* __restrict__ left, const uint8_t* __restrict__ right,
                                                           const uint16_t* __restrict__ vs, uint8_t* __restrict__ pf_l,
                                                           uint8_t* __restrict__ pf_r, int W, int H, int pitch, int padl,
                                                           int plane, int wsz2, int sg, int ss, int cap, int pfshift) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const int img = blockIdx.z >> 1;
  const uint8_t* src = ((blockIdx.z & 1) ? right : left) + (size_t)img * W * H;
  const uint16_t* v = vs + ((size_t)blockIdx.z * H + y) * W;
  int box = 0;
  for (int dx = -wsz2; dx <= wsz2; dx++) box += v[min(max(x + dx, 0), W - 1)];
  const uint8_t* prev = src + (size_t)max(y - 1, 0) * W;
  const uint8_t* curr = src + (size_t)y * W;
  const uint8_t* next = src + (size_t)min(y + 1, H - 1) * W;
  const int c = curr[x], lft = x > 0 ? curr[x - 1] : c, rgt = x < W - 1 ? curr[x + 1] : c;   // missing neighbour = centre
  const int centre = 4 * c + lft + rgt + prev[x] + next[x];
  const int val = (centre * sg - box * ss) >> 10;
  const int o = val < -cap ? 0 : (val > cap ? 2 * cap : val + cap);
  uint8_t* dst = ((blockIdx.z & 1) ? pf_r : pf_l) + (size_t)img * plane + (size_t)y * pitch + padl;
  dst[x] = (uint8_t)((o << pfshift) + kPfBias);
}

hipError_t launch_prefilter_norm(const uint8_t* d_left, const uint8_t* d_right, uint8_t* pf_l, uint8_t* pf_r, uint16_t* vsum,
                                 const Geom& g, int winsize, hipStream_t s) {
  const int wsz2 = winsize / 2;
  const int g0 = winsize * winsize / 8;
  const int ss = g0 > 0 ? (1024 + g0) / (g0 * 2) : 0;
  const int sg = g0 * ss;
  const unsigned gx = (unsigned)((g.W + 255) / 256);
  hipLaunchKernelGGL(pf_norm_vsum_kernel, dim3(gx, (g.H + PFN_ROWS - 1) / PFN_ROWS, 2 * g.n), dim3(256), 0, s, d_left, d_right,
                     vsum, g.W, g.H, wsz2);
  hipLaunchKernelGGL(pf_norm_resp_kernel, dim3(gx, g.H, 2 * g.n), dim3(256), 0, s, d_left, d_right, vsum, pf_l, pf_r, g.W, g.H,
                     g.pitch, g.padl, g.plane, wsz2, sg, ss, g.cap, g.pfshift);
  return hipGetLastError();
}

// Stand-alone prefilter of n dense images into n dense planes (no padding, no bias), either flavour.
hipError_t launch_prefilter_dense(const uint8_t* d_src, uint8_t* d_dst, int n, int W, int H, int rtl, int cap,
                                  hipStream_t s) {
  const int npiece = (W + 15) / 16;
  const int rows = pf_rows((size_t)2 * n * W * H);
  const int ngroup = (H + rows - 1) / rows;
  // the kernel addresses images as (pair, side): image j = pair j/2, side j&1, so a dense array of images is a
  // sequence of pairs with a stride of two images; an odd last image goes in a second launch of one side
  const PfMap m = rtl ? PfMap{-32, 31, 32, 0, 1, 0, 1} : PfMap{-cap, cap, cap, 0, 0, 0, 1};
  const size_t img = (size_t)W * H;
  const unsigned gx = (unsigned)((npiece * ngroup + 255) / 256);
#define SBM_PFD(R)                                                                                                          \
  do {                                                                                                                      \
    if (n >= 2)                                                                                                             \
      hipLaunchKernelGGL(prefilter_kernel<R>, dim3(gx, 2 * (n / 2)), dim3(256), 0, s, d_src, d_src + img, d_dst, d_dst + img, \
                         W, H, W, 0, 2 * img, 2 * img, (size_t)(n & ~1) * img, (size_t)(n & ~1) * img - img, m);            \
    if (n & 1) {                                                                                                            \
      const size_t o = (size_t)(n - 1) * img;                                                                               \
      hipLaunchKernelGGL(prefilter_kernel<R>, dim3(gx, 1), dim3(256), 0, s, d_src + o, d_src + o, d_dst + o, d_dst + o, W, H, \
                         W, 0, img, img, img, img, m);                                                                      \
    }                                                                                                                       \
  } while (0)
  if (rows == 2) SBM_PFD(2); else if (rows == 8) SBM_PFD(8); else SBM_PFD(4);
#undef SBM_PFD
  return hipGetLastError();
}

}  // namespace sbm
