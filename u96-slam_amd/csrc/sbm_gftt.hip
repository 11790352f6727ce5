// sbm_gftt.hip -- the PL's GFTT minimum-eigenvalue map on gfx950.
//
// Device counterpart of src/dvp/rtl/gftt_sbl.v:113-204 (3x3 Sobel, first / last column forced to 0), gftt_eig.v:122-143
// (|dx|^2 >> 6, |dy|^2 >> 6, |dx||dy| >> 6), gftt_box.v (3x3 box sums, edge columns forced to 0, 16-bit limiter),
// gftt_eig.v:226-362 ((a + c) - sqrt(((a-c)^2 >> 10) + (b^2 >> 8)) with its limiters) and gftt_obuf.v:101-130,295-305
// (rows 2..H-3 of a dense uint16 map, `Max` register); consumer: src/slam/src/core/GFTT.cpp:41-170 via FPGA.cpp:283-291.
// The RTL takes the square root in a Xilinx CORDIC core whose last bit is unspecified; this kernel takes the exact floor.
//
// HBM-bound by construction (1 B read + 2 B written per pixel): a workgroup stages a (TH+4) x (TW+4) pixel tile in LDS,
// forms the three product planes of the (TH+2) x (TW+2) Sobel samples there, and every thread sums 3x3 neighbourhoods.
#include "sbm_common.h"

namespace sbm {

constexpr int GF_TW = 64, GF_TH = 16;

__global__ void __launch_bounds__(256) gftt_eig_kernel(const uint8_t* __restrict__ img, uint16_t* __restrict__ eig,
                                                       unsigned* __restrict__ maxv, int W, int H) {
  __shared__ uint8_t px[GF_TH + 4][GF_TW + 4];
  __shared__ unsigned short vxx[GF_TH + 2][GF_TW + 2], vyy[GF_TH + 2][GF_TW + 2], vxy[GF_TH + 2][GF_TW + 2];
  const int n = blockIdx.z;
  const int x0 = blockIdx.x * GF_TW, y0 = blockIdx.y * GF_TH;
  const uint8_t* src = img + (size_t)n * W * H;
  for (int i = threadIdx.x; i < (GF_TH + 4) * (GF_TW + 4); i += 256) {
    const int ty = i / (GF_TW + 4), tx = i - ty * (GF_TW + 4);
    const int y = y0 - 2 + ty, x = x0 - 2 + tx;
    px[ty][tx] = (y >= 0 && y < H && x >= 0 && x < W) ? src[(size_t)y * W + x] : 0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (GF_TH + 2) * (GF_TW + 2); i += 256) {
    const int ty = i / (GF_TW + 2), tx = i - ty * (GF_TW + 2);
    const int y = y0 - 1 + ty, x = x0 - 1 + tx;     // Sobel sample (y, x); its pixels sit at px[ty..ty+2][tx..tx+2]
    unsigned ax = 0, ay = 0;
    if (y >= 1 && y <= H - 2 && x >= 1 && x <= W - 2) {
      const int dx = (px[ty][tx + 2] - px[ty][tx]) + 2 * (px[ty + 1][tx + 2] - px[ty + 1][tx]) + (px[ty + 2][tx + 2] - px[ty + 2][tx]);
      const int dy = (px[ty + 2][tx] - px[ty][tx]) + 2 * (px[ty + 2][tx + 1] - px[ty][tx + 1]) + (px[ty + 2][tx + 2] - px[ty][tx + 2]);
      ax = (unsigned)(dx < 0 ? -dx : dx);
      ay = (unsigned)(dy < 0 ? -dy : dy);
    }
    vxx[ty][tx] = (unsigned short)((ax * ax) >> 6);
    vyy[ty][tx] = (unsigned short)((ay * ay) >> 6);
    vxy[ty][tx] = (unsigned short)((ax * ay) >> 6);
  }
  __syncthreads();
  unsigned mx = 0;
  for (int i = threadIdx.x; i < GF_TH * GF_TW; i += 256) {
    const int ty = i / GF_TW, tx = i - ty * GF_TW;
    const int y = y0 + ty, x = x0 + tx;
    if (y >= H || x >= W) continue;
    unsigned out = 0;
    if (y >= 2 && y <= H - 3 && x >= 1 && x <= W - 2) {   // the horizontal sums of the first / last column are forced to 0
      unsigned a = 0, c = 0, b = 0;
#pragma unroll
      for (int j = 0; j < 3; j++)
#pragma unroll
        for (int k = 0; k < 3; k++) {
          a += vxx[ty + j][tx + k];
          c += vyy[ty + j][tx + k];
          b += vxy[ty + j][tx + k];
        }
      a = min(a, 0xffffu); c = min(c, 0xffffu); b = min(b, 0xffffu);
      const unsigned apc = a + c, amc = a > c ? a - c : c - a;
      const unsigned amc2 = (unsigned)(((unsigned long long)amc * amc) >> 10) & 0x3fffffu;
      const unsigned b2 = (unsigned)(((unsigned long long)b * b) >> 8) & 0xffffffu;
      const unsigned s = min(amc2 + b2, 0x3fffffu);
      const unsigned long long rad = (unsigned long long)s << 10;
      unsigned r = (unsigned)__builtin_sqrtf((float)rad);
      while ((unsigned long long)r * r > rad) r--;
      while ((unsigned long long)(r + 1) * (r + 1) <= rad) r++;
      const int e = (int)apc - (int)(r & 0xffffu);
      out = e < 0 ? 0u : (e > 0xffff ? 0xffffu : (unsigned)e);
    }
    eig[((size_t)n * H + y) * W + x] = (unsigned short)out;
    mx = max(mx, out);
  }
  for (int off = 32; off > 0; off >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, off, 64));
  if ((threadIdx.x & 63) == 0 && mx) atomicMax(maxv + n, mx);
}

hipError_t launch_gftt_eig(const uint8_t* img, uint16_t* eig, unsigned* maxv, int n, int W, int H, hipStream_t s) {
  hipError_t e = hipMemsetAsync(maxv, 0, (size_t)n * sizeof(unsigned), s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(gftt_eig_kernel, dim3((W + GF_TW - 1) / GF_TW, (H + GF_TH - 1) / GF_TH, n), dim3(256), 0, s, img, eig, maxv, W, H);
  return hipGetLastError();
}

}  // namespace sbm
