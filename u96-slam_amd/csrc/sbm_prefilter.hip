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
};
__device__ __forceinline__ int clipmap(int v, const PfMap& m) { return (v < m.lo ? m.lo : (v > m.hi ? m.hi : v)) + m.off; }

__device__ __forceinline__ uint4 load_u128_ua(const uint8_t* p) {
  uint4 v;
  __builtin_memcpy(&v, p, 16);
  return v;
}

// vertical 1-2-1 sums of 4 packed bytes, as two registers of 2 x u16 (even bytes / odd bytes): max 4*255 fits u16
__device__ __forceinline__ void vsum4(uint32_t a, uint32_t b, uint32_t c, uint32_t& even, uint32_t& odd) {
  const uint32_t m = 0x00ff00ffu;
  even = (a & m) + 2u * (b & m) + (c & m);
  odd = ((a >> 8) & m) + 2u * ((b >> 8) & m) + ((c >> 8) & m);
}

// Each thread produces a 16-pixel x PF_ROWS-row tile: PF_ROWS + 2 source row pieces are loaded once and every
// vertical 1-2-1 sum reuses them (1.5 loads per output row instead of 3).
constexpr int PF_ROWS = 4;

// grid: x = ceil(ceil(H/PF_ROWS)*ceil(W/16)/256), y = 2*n (image index: even = left, odd = right)
__global__ void __launch_bounds__(256) prefilter_kernel(const uint8_t* __restrict__ left, const uint8_t* __restrict__ right,
                                                        uint8_t* __restrict__ pf_l, uint8_t* __restrict__ pf_r, int W, int H,
                                                        int pitch, int padl, size_t sstride, size_t plane, PfMap m) {
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
  const bool interior = x0 >= 16 && x0 + 32 <= W;

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

#pragma unroll
  for (int j = 0; j < PF_ROWS; j++) {
    const int y = y0 + j;
    if (y >= H) break;
    uint32_t out[4];
    if (m.rtl ? (y == 0 || y == H - 1) : ((H & 1) && y == H - 1)) {
      out[0] = out[1] = out[2] = out[3] = (uint32_t)((m.rtl ? 0 : m.off) + m.bias) * 0x01010101u;
    } else {
      int s[18];  // vertical 1-2-1 sums of columns x0-1 .. x0+16
#pragma unroll
      for (int k = 0; k < 5; k++) {
        uint32_t ev, od;
        vsum4(rw[j][k], rw[j + 1][k], rw[j + 2][k], ev, od);
        if (4 * k + 0 < 18) s[4 * k + 0] = (int)(ev & 0xffffu);
        if (4 * k + 1 < 18) s[4 * k + 1] = (int)(od & 0xffffu);
        if (4 * k + 2 < 18) s[4 * k + 2] = (int)(ev >> 16);
        if (4 * k + 3 < 18) s[4 * k + 3] = (int)(od >> 16);
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int x = x0 + 4 * k + i;
          const int v = (x == 0 || x >= W - 1) ? m.off : clipmap(s[4 * k + i + 2] - s[4 * k + i], m);
          o |= (uint32_t)(v + m.bias) << (8 * i);
        }
        out[k] = o;
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
    } else {
      for (int i = 0; x0 + i < W; i++) dst[i] = (uint8_t)(out[i >> 2] >> (8 * (i & 3)));
    }
  }
}

hipError_t launch_prefilter(const uint8_t* d_left, const uint8_t* d_right, uint8_t* pf_l, uint8_t* pf_r,
                            const Geom& g, hipStream_t s) {
  const int npiece = (g.W + 15) / 16;
  const int ngroup = (g.H + PF_ROWS - 1) / PF_ROWS;
  dim3 grid((npiece * ngroup + 255) / 256, 2 * g.n);
  const PfMap m{-g.cap, g.cap, g.cap, kPfBias, 0};
  hipLaunchKernelGGL(prefilter_kernel, grid, dim3(256), 0, s, d_left, d_right, pf_l, pf_r, g.W, g.H, g.pitch, g.padl,
                     (size_t)g.W * g.H, (size_t)g.plane, m);
  return hipGetLastError();
}

// Stand-alone prefilter of n dense images into n dense planes (no padding, no bias), either flavour.
hipError_t launch_prefilter_dense(const uint8_t* d_src, uint8_t* d_dst, int n, int W, int H, int rtl, int cap,
                                  hipStream_t s) {
  const int npiece = (W + 15) / 16;
  const int ngroup = (H + PF_ROWS - 1) / PF_ROWS;
  // the kernel addresses images as (pair, side): image j = pair j/2, side j&1, so a dense array of images is a
  // sequence of pairs with a stride of two images; an odd last image goes in a second launch of one side
  const PfMap m = rtl ? PfMap{-32, 31, 32, 0, 1} : PfMap{-cap, cap, cap, 0, 0};
  const size_t img = (size_t)W * H;
  const unsigned gx = (unsigned)((npiece * ngroup + 255) / 256);
  if (n >= 2)
    hipLaunchKernelGGL(prefilter_kernel, dim3(gx, 2 * (n / 2)), dim3(256), 0, s, d_src, d_src + img, d_dst, d_dst + img,
                       W, H, W, 0, 2 * img, 2 * img, m);
  if (n & 1) {
    const size_t o = (size_t)(n - 1) * img;
    hipLaunchKernelGGL(prefilter_kernel, dim3(gx, 1), dim3(256), 0, s, d_src + o, d_src + o, d_dst + o, d_dst + o, W, H, W,
                       0, img, img, m);
  }
  return hipGetLastError();
}

}  // namespace sbm
