// sbm_prefilter.hip -- x-Sobel prefilter (cv prefilterXSobel semantics), gfx950.
//
// Replaces, on the device, the first stage of cv::StereoBM::compute as called from
// src/slam/src/core/main.cpp:215 (in-tree hardware twin: src/dvp/rtl/xsbl2.v:661-874, which clips to
// [-32,31] instead of [-cap,cap]).  out = clip(d(y-1) + 2 d(y) + d(y+1), -cap, cap) + cap with
// d(r) = r[x+1] - r[x-1]; rows mirrored (reflect-101) at the top and, for even H, at the bottom; columns 0 and
// W-1 = cap; for odd H the last row is all cap.
//
// HBM-bound: 1 byte read + 1 byte written per pixel (the three source rows of a strip hit in L2). Each thread
// produces 4 adjacent pixels from three unaligned dword triples and stores one dword. The result is written with
// a +1 bias into a zero-padded plane (see kPfBias in sbm_common.h).
#include "sbm_common.h"

namespace sbm {

__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t* p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}

__device__ __forceinline__ int clipcap(int v, int cap) { return (v < -cap ? -cap : (v > cap ? cap : v)) + cap; }

// grid: x = ceil(W/4/256), y = H, z = 2*n (image index: even = left, odd = right)
__global__ void __launch_bounds__(256) prefilter_kernel(const uint8_t* __restrict__ left, const uint8_t* __restrict__ right,
                                                         uint8_t* __restrict__ pf_l, uint8_t* __restrict__ pf_r, int W, int H,
                                                         int pitch, int padl, int plane, int cap) {
  const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (x0 >= W) return;
  const int y = blockIdx.y;
  const int img = blockIdx.z >> 1;
  const uint8_t* src = ((blockIdx.z & 1) ? right : left) + (size_t)img * W * H;
  uint8_t* dst = ((blockIdx.z & 1) ? pf_r : pf_l) + (size_t)img * plane + (size_t)y * pitch + padl + x0;

  uint32_t out;
  if ((H & 1) && y == H - 1) {
    out = (uint32_t)(cap + kPfBias) * 0x01010101u;
  } else {
    const int ym = y > 0 ? y - 1 : (H > 1 ? 1 : 0);
    const int yp = y < H - 1 ? y + 1 : (H > 1 ? H - 2 : 0);
    const uint8_t* r0 = src + (size_t)ym * W;
    const uint8_t* r1 = src + (size_t)y * W;
    const uint8_t* r2 = src + (size_t)yp * W;
    int s[6];  // vertical 1-2-1 sums of columns x0-1 .. x0+4
    if (x0 >= 4 && x0 + 8 <= W) {
      // interior: two unaligned dword loads per row cover bytes x0-1 .. x0+6
      uint32_t a0 = load_u32_unaligned(r0 + x0 - 1), a1 = load_u32_unaligned(r0 + x0 + 3);
      uint32_t b0 = load_u32_unaligned(r1 + x0 - 1), b1 = load_u32_unaligned(r1 + x0 + 3);
      uint32_t c0 = load_u32_unaligned(r2 + x0 - 1), c1 = load_u32_unaligned(r2 + x0 + 3);
#pragma unroll
      for (int i = 0; i < 4; i++)
        s[i] = (int)((a0 >> (8 * i)) & 255) + 2 * (int)((b0 >> (8 * i)) & 255) + (int)((c0 >> (8 * i)) & 255);
#pragma unroll
      for (int i = 0; i < 2; i++)
        s[4 + i] = (int)((a1 >> (8 * i)) & 255) + 2 * (int)((b1 >> (8 * i)) & 255) + (int)((c1 >> (8 * i)) & 255);
    } else {
#pragma unroll
      for (int i = 0; i < 6; i++) {
        int x = x0 - 1 + i;
        x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);  // value unused where clamped (edge columns are forced to cap)
        s[i] = (int)r0[x] + 2 * (int)r1[x] + (int)r2[x];
      }
    }
    out = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int x = x0 + i;
      int v = (x == 0 || x >= W - 1) ? cap : clipcap(s[i + 2] - s[i], cap);
      out |= (uint32_t)(v + kPfBias) << (8 * i);
    }
  }
  if (x0 + 4 <= W) {
    __builtin_memcpy(dst, &out, 4);  // pitch and padl are multiples of 4 -> aligned dword store
  } else {
    for (int i = 0; x0 + i < W; i++) dst[i] = (uint8_t)(out >> (8 * i));
  }
}

hipError_t launch_prefilter(const uint8_t* d_left, const uint8_t* d_right, uint8_t* pf_l, uint8_t* pf_r,
                            const Geom& g, hipStream_t s) {
  dim3 grid((g.W + 1023) / 1024, g.H, 2 * g.n);
  hipLaunchKernelGGL(prefilter_kernel, grid, dim3(256), 0, s, d_left, d_right, pf_l, pf_r, g.W, g.H, g.pitch, g.padl,
                     g.plane, g.cap);
  return hipGetLastError();
}

}  // namespace sbm
