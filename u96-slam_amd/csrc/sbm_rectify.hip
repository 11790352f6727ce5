// sbm_rectify.hip -- stereo rectification in front of the prefilter, gfx950.
//
// Device counterparts of the reference's FPGA-flavour rectifier:
//   rect_map_kernel    rect_remap(), src/StereoBM/src/fpga.c:303-366 (RTL twin src/dvp/rtl/rect_rmp.v:339-572):
//                      destination pixel -> normalised rectified ray -> inverse rotation -> perspective divide ->
//                      source pixel, s1.24 fixed point, stored as s10.5 (x, y) pairs.
//   rect_remap_kernel  src/dvp/rtl/rect_intp.v:285-404: bilinear resampling with 5-bit fractions,
//                      ((UL*(32-xf)*(32-yf) + UR*xf*(32-yf) + DL*(32-xf)*yf + DR*xf*yf) >> 9) + 1) >> 1.
//
// The map depends on the camera only and is built once (one thread per pixel; the 64-bit division runs once per
// pixel per calibration, not per frame). The resampler is HBM-bound by construction: per destination pixel 4 B of
// map + 1 B written + the source taps, which are gathers with 2-D locality served by L2 (algorithmic 1 B). A thread
// produces 4 adjacent pixels from one 16-byte map load and issues one 4-byte store; all images of a batch share
// the map, so the grid is ordered map-tile-major and the map tile stays in L2 across the batch.
#include "sbm_common.h"

namespace sbm {

__device__ __forceinline__ long long rect_coord(long long num, long long lw_inv, int f, int c) {
  const long long n2 = (num * lw_inv) >> 24;  // s1.24 * s1.24 -> s1.24
  const long long nf = (n2 * (long long)f) >> 34;  // s1.24 * u10.16 -> s10.6
  const long long v = nf + ((long long)c << 6);
  return (v + 1) >> 1;  // s10.5
}

__global__ void __launch_bounds__(256) rect_map_kernel(sbm_rect_cam cam, int W, int H, short2* __restrict__ map) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= W * H) return;
  const int yd = t / W, xd = t - yd * W;
  const long long xn = (((long long)xd * (long long)cam.f2inv[0]) >> 8) - (long long)cam.c2_f2[0];
  const long long yn = (((long long)yd * (long long)cam.f2inv[1]) >> 8) - (long long)cam.c2_f2[1];
  long long l[3];
#pragma unroll
  for (int k = 0; k < 3; k++)
    l[k] = (((long long)cam.rot[0][k] * xn) >> 24) + (((long long)cam.rot[1][k] * yn) >> 24) + (long long)cam.rot[2][k];
  // the firmware divides (1ull << 48) by lw converted to unsigned
  const unsigned long long den = (unsigned long long)l[2];
  const long long lw_inv = den ? (long long)((1ull << 48) / den) : 0;
  map[t] = make_short2((short)rect_coord(l[0], lw_inv, cam.f[0], cam.c[0]), (short)rect_coord(l[1], lw_inv, cam.f[1], cam.c[1]));
}

__device__ __forceinline__ int tap(const uint8_t* __restrict__ src, int W, int H, int x, int y) {
  return ((unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H) ? (int)src[(size_t)y * W + x] : 0;
}

__device__ __forceinline__ uint32_t interp(const uint8_t* __restrict__ src, int W, int H, uint32_t m) {
  const int mx = (short)(m & 0xffffu), my = (short)(m >> 16);
  const int xi = mx >> 5, yi = my >> 5, xf = mx & 31, yf = my & 31;
  int ul, ur, dl, dr;
  if (xi >= 0 && xi + 1 < W && yi >= 0 && yi + 1 < H) {
    // the two taps of a row are neighbours: one unaligned 2-byte load per row instead of two byte loads
    const uint8_t* p = src + (size_t)yi * W + xi;
    unsigned short a, b;
    __builtin_memcpy(&a, p, 2);
    __builtin_memcpy(&b, p + W, 2);
    ul = a & 0xff; ur = a >> 8; dl = b & 0xff; dr = b >> 8;
  } else {
    ul = tap(src, W, H, xi, yi); ur = tap(src, W, H, xi + 1, yi);
    dl = tap(src, W, H, xi, yi + 1); dr = tap(src, W, H, xi + 1, yi + 1);
  }
  // UL (32-xf)(32-yf) + UR xf (32-yf) + DL (32-xf) yf + DR xf yf, factored (same integer, no rounding in between):
  // top = 32 UL + (UR - UL) xf, bot = 32 DL + (DR - DL) xf, acc = 32 top + (bot - top) yf
  const int top = (ul << 5) + (ur - ul) * xf, bot = (dl << 5) + (dr - dl) * xf;
  const int acc = (top << 5) + (bot - top) * yf;
  return (uint32_t)(((acc >> 9) + 1) >> 1);
}

// grid: x = images (fastest: the workgroups that share a map tile are dispatched together), y = map tiles of 1024 px
__global__ void __launch_bounds__(256) rect_remap_kernel(const uint8_t* __restrict__ src, const uint32_t* __restrict__ map,
                                                         uint8_t* __restrict__ dst, int W, int H) {
  const size_t npix = (size_t)W * H;
  const size_t t4 = ((size_t)blockIdx.y * 256 + threadIdx.x) * 4;
  if (t4 >= npix) return;
  const uint8_t* s = src + (size_t)blockIdx.x * npix;
  uint8_t* d = dst + (size_t)blockIdx.x * npix + t4;
  if (t4 + 4 <= npix) {
    const uint4 m = *reinterpret_cast<const uint4*>(map + t4);  // 4 (x,y) pairs; map is 16-byte aligned, t4 % 4 == 0
    const uint32_t o = interp(s, W, H, m.x) | (interp(s, W, H, m.y) << 8) | (interp(s, W, H, m.z) << 16) |
                       (interp(s, W, H, m.w) << 24);
    if ((npix & 3) == 0) {
      *reinterpret_cast<uint32_t*>(d) = o;
    } else {
      __builtin_memcpy(d, &o, 4);  // image planes are only 4-byte aligned when W*H is a multiple of 4
    }
  } else {
    for (size_t i = t4; i < npix; i++) dst[(size_t)blockIdx.x * npix + i] = (uint8_t)interp(s, W, H, map[i]);
  }
}

hipError_t launch_rect_map(const sbm_rect_cam& cam, int W, int H, int16_t* d_map, hipStream_t s) {
  const unsigned blocks = (unsigned)(((size_t)W * H + 255) / 256);
  hipLaunchKernelGGL(rect_map_kernel, dim3(blocks), dim3(256), 0, s, cam, W, H, reinterpret_cast<short2*>(d_map));
  return hipGetLastError();
}

hipError_t launch_rect_remap(const uint8_t* d_src, const int16_t* d_map, uint8_t* d_dst, int n, int W, int H,
                             hipStream_t s) {
  const size_t quads = ((size_t)W * H + 3) / 4;
  dim3 grid((unsigned)n, (unsigned)((quads + 255) / 256));
  hipLaunchKernelGGL(rect_remap_kernel, grid, dim3(256), 0, s, d_src, reinterpret_cast<const uint32_t*>(d_map), d_dst, W, H);
  return hipGetLastError();
}

}  // namespace sbm
