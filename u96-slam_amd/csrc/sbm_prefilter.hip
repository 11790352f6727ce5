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

__device__ __forceinline__ int clipcap(int v, int cap) { return (v < -cap ? -cap : (v > cap ? cap : v)) + cap; }

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

// grid: x = ceil(H*ceil(W/16)/256), y = 2*n (image index: even = left, odd = right); 16 pixels per thread
__global__ void __launch_bounds__(256) prefilter_kernel(const uint8_t* __restrict__ left, const uint8_t* __restrict__ right,
                                                        uint8_t* __restrict__ pf_l, uint8_t* __restrict__ pf_r, int W, int H,
                                                        int pitch, int padl, int plane, int cap) {
  // threads are flattened over (row, 16-pixel piece) of one image so that every lane has work whatever the width
  const int npiece = (W + 15) / 16;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= npiece * H) return;
  const int y = t / npiece;
  const int x0 = (t - y * npiece) * 16;
  const int img = blockIdx.y >> 1;
  const uint8_t* src = ((blockIdx.y & 1) ? right : left) + (size_t)img * W * H;
  uint8_t* dst = ((blockIdx.y & 1) ? pf_r : pf_l) + (size_t)img * plane + (size_t)y * pitch + padl + x0;

  uint32_t out[4];
  if ((H & 1) && y == H - 1) {
    out[0] = out[1] = out[2] = out[3] = (uint32_t)(cap + kPfBias) * 0x01010101u;
  } else {
    const int ym = y > 0 ? y - 1 : (H > 1 ? 1 : 0);
    const int yp = y < H - 1 ? y + 1 : (H > 1 ? H - 2 : 0);
    const uint8_t* r0 = src + (size_t)ym * W;
    const uint8_t* r1 = src + (size_t)y * W;
    const uint8_t* r2 = src + (size_t)yp * W;
    int s[18];  // vertical 1-2-1 sums of columns x0-1 .. x0+16
    if (x0 >= 16 && x0 + 32 <= W) {
      // interior: bytes x0-1 .. x0+18 of each row from one 16-byte and one 4-byte unaligned load
      const uint4 a = load_u128_ua(r0 + x0 - 1), b = load_u128_ua(r1 + x0 - 1), c = load_u128_ua(r2 + x0 - 1);
      const uint32_t a4 = load_u32_unaligned(r0 + x0 + 15), b4 = load_u32_unaligned(r1 + x0 + 15),
                     c4 = load_u32_unaligned(r2 + x0 + 15);
      const uint32_t aw[5] = {a.x, a.y, a.z, a.w, a4}, bw[5] = {b.x, b.y, b.z, b.w, b4}, cw[5] = {c.x, c.y, c.z, c.w, c4};
#pragma unroll
      for (int k = 0; k < 5; k++) {
        uint32_t ev, od;
        vsum4(aw[k], bw[k], cw[k], ev, od);
        if (4 * k + 0 < 18) s[4 * k + 0] = (int)(ev & 0xffffu);
        if (4 * k + 1 < 18) s[4 * k + 1] = (int)(od & 0xffffu);
        if (4 * k + 2 < 18) s[4 * k + 2] = (int)(ev >> 16);
        if (4 * k + 3 < 18) s[4 * k + 3] = (int)(od >> 16);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 18; i++) {
        int x = x0 - 1 + i;
        x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);  // value unused where clamped (edge columns are forced to cap)
        s[i] = (int)r0[x] + 2 * (int)r1[x] + (int)r2[x];
      }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      uint32_t o = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int x = x0 + 4 * k + i;
        const int v = (x == 0 || x >= W - 1) ? cap : clipcap(s[4 * k + i + 2] - s[4 * k + i], cap);
        o |= (uint32_t)(v + kPfBias) << (8 * i);
      }
      out[k] = o;
    }
  }
  if (x0 + 16 <= W) {
    *reinterpret_cast<uint4*>(dst) = make_uint4(out[0], out[1], out[2], out[3]);  // pitch, padl multiples of 16
  } else {
    for (int i = 0; x0 + i < W; i++) dst[i] = (uint8_t)(out[i >> 2] >> (8 * (i & 3)));
  }
}

hipError_t launch_prefilter(const uint8_t* d_left, const uint8_t* d_right, uint8_t* pf_l, uint8_t* pf_r,
                            const Geom& g, hipStream_t s) {
  const int npiece = (g.W + 15) / 16;
  dim3 grid((npiece * g.H + 255) / 256, 2 * g.n);
  hipLaunchKernelGGL(prefilter_kernel, grid, dim3(256), 0, s, d_left, d_right, pf_l, pf_r, g.W, g.H, g.pitch, g.padl,
                     g.plane, g.cap);
  return hipGetLastError();
}

}  // namespace sbm
