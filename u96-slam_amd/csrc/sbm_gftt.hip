// sbm_gftt.hip -- the PL's GFTT minimum-eigenvalue map on gfx950.
//
// Device counterpart of src/dvp/rtl/gftt_sbl.v:113-204 (3x3 Sobel, first / last column forced to 0), gftt_eig.v:122-143
// (|dx|^2 >> 6, |dy|^2 >> 6, |dx||dy| >> 6), gftt_box.v (3x3 box sums, edge columns forced to 0, 16-bit limiter),
// gftt_eig.v:226-362 ((a + c) - sqrt(((a-c)^2 >> 10) + (b^2 >> 8)) with its limiters) and gftt_obuf.v:101-130,295-305
// (rows 2..H-3 of a dense uint16 map, `Max` register); consumer: src/slam/src/core/GFTT.cpp:41-170 via FPGA.cpp:283-291.
// The RTL takes the square root in a Xilinx CORDIC core whose last bit is unspecified; this kernel takes the exact floor.
//
// HBM-bound by construction (1 B read + 2 B written per pixel). Lane = image column, a wavefront marches down a row
// segment like the RTL's line buffers do: three pixel rows (own column and both neighbours, one unaligned 4-byte load per
// row) give the Sobel pair of the middle row, the three products meet their horizontal neighbours through DPP wave
// shifts, and the last three rows of horizontal sums stay in registers for the vertical sum. Lanes 0 and 63 only serve as
// neighbours (62 outputs per wavefront).
#include "sbm_common.h"

namespace sbm {

constexpr int GF_NV = 62;    // output columns per wavefront
constexpr int GF_SEG = 64;   // output rows per wavefront

__device__ __forceinline__ unsigned gf_shr1(unsigned v) {   // value of lane-1 (lane 0: 0)
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ unsigned gf_shl1(unsigned v) {   // value of lane+1 (lane 63: 0)
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}

__global__ void __launch_bounds__(64) gftt_eig_kernel(const uint8_t* __restrict__ img, uint16_t* __restrict__ eig,
                                                      unsigned* __restrict__ maxv, int W, int H) {
  const int lane = threadIdx.x;
  const int n = blockIdx.z;
  const int x = blockIdx.x * GF_NV + lane - 1;            // lane 0 is the left neighbour of the first output column
  const int y0 = blockIdx.y * GF_SEG, y1 = min(y0 + GF_SEG, H);
  const uint8_t* src = img + (size_t)n * W * H;
  uint16_t* dst = eig + (size_t)n * W * H;
  const bool in_img = x >= 0 && x < W;
  const bool sob_ok = x >= 1 && x <= W - 2;               // gftt_sbl.v: first_r / last_r force the edge samples to 0
  const bool writes = lane >= 1 && lane <= GF_NV && in_img;
  const int xl = min(max(x - 1, 0), W - 1), xc = min(max(x, 0), W - 1), xr = min(max(x + 1, 0), W - 1);

  // pixel rows y-1, y, y+1 of the Sobel row in flight: (left, centre, right) each
  // (one unaligned 4-byte load where the Sobel sample exists and the fourth byte is still inside the image; the edge
  // lanes -- whose samples are forced to 0 anyway -- and the very last pixels of the image read byte by byte)
  auto load3 = [&](int y, int& l, int& c, int& r) {
    const int yy = min(max(y, 0), H - 1);
    const uint8_t* p = src + (size_t)yy * W;
    if (sob_ok && (x + 2 < W || yy < H - 1)) {
      unsigned v;
      __builtin_memcpy(&v, p + x - 1, 4);
      l = (int)(v & 0xffu); c = (int)((v >> 8) & 0xffu); r = (int)((v >> 16) & 0xffu);
    } else {
      l = p[xl]; c = p[xc]; r = p[xr];
    }
  };
  // horizontal 3-sums of the three products of Sobel row ys (0 outside rows 1..H-2 and in the edge columns)
  auto hsums = [&](int ys, int l0, int c0, int r0, int l1, int r1, int l2, int c2, int r2, unsigned& ha, unsigned& hc, unsigned& hb) {
    unsigned ax = 0, ay = 0;
    if (sob_ok && ys >= 1 && ys <= H - 2) {
      const int dx = (r0 - l0) + 2 * (r1 - l1) + (r2 - l2);
      const int dy = (l2 - l0) + 2 * (c2 - c0) + (r2 - r0);
      ax = (unsigned)(dx < 0 ? -dx : dx);
      ay = (unsigned)(dy < 0 ? -dy : dy);
    }
    const unsigned vxx = (ax * ax) >> 6, vyy = (ay * ay) >> 6, vxy = (ax * ay) >> 6;
    ha = sob_ok ? gf_shr1(vxx) + vxx + gf_shl1(vxx) : 0u;   // gftt_box.v: col_start_r | col_end_r -> 0
    hc = sob_ok ? gf_shr1(vyy) + vyy + gf_shl1(vyy) : 0u;
    hb = sob_ok ? gf_shr1(vxy) + vxy + gf_shl1(vxy) : 0u;
  };

  int pl[3], pc[3], pr[3];                                 // pixel rows ys-1, ys, ys+1 (rolling)
  unsigned ha[3], hc[3], hb[3];                            // horizontal sums of Sobel rows y-1, y, y+1 (rolling)
  // prime: Sobel rows y0-1 and y0
  load3(y0 - 2, pl[0], pc[0], pr[0]);
  load3(y0 - 1, pl[1], pc[1], pr[1]);
  load3(y0, pl[2], pc[2], pr[2]);
  hsums(y0 - 1, pl[0], pc[0], pr[0], pl[1], pr[1], pl[2], pc[2], pr[2], ha[0], hc[0], hb[0]);
  pl[0] = pl[1]; pc[0] = pc[1]; pr[0] = pr[1]; pl[1] = pl[2]; pc[1] = pc[2]; pr[1] = pr[2];
  load3(y0 + 1, pl[2], pc[2], pr[2]);
  hsums(y0, pl[0], pc[0], pr[0], pl[1], pr[1], pl[2], pc[2], pr[2], ha[1], hc[1], hb[1]);
  unsigned mx = 0;
  for (int y = y0; y < y1; y++) {
    // Sobel row y+1 from pixel rows y, y+1, y+2
    pl[0] = pl[1]; pc[0] = pc[1]; pr[0] = pr[1]; pl[1] = pl[2]; pc[1] = pc[2]; pr[1] = pr[2];
    load3(y + 2, pl[2], pc[2], pr[2]);
    hsums(y + 1, pl[0], pc[0], pr[0], pl[1], pr[1], pl[2], pc[2], pr[2], ha[2], hc[2], hb[2]);
    unsigned out = 0;
    if (y >= 2 && y <= H - 3) {                            // gftt_obuf.v:295-305: four border lines are never written
      const unsigned a = min(ha[0] + ha[1] + ha[2], 0xffffu), c = min(hc[0] + hc[1] + hc[2], 0xffffu),
                     b = min(hb[0] + hb[1] + hb[2], 0xffffu);
      const unsigned apc = a + c, amc = a > c ? a - c : c - a;
      const unsigned amc2 = (amc * amc) >> 10, b2 = (b * b) >> 8;   // both operands < 2^16: the squares fit 32 bits
      const unsigned s = min(amc2 + b2, 0x3fffffu);
      // floor(sqrt(s << 10)): s < 2^22 is exact in float, the estimate is within 1 and its square fits 32 bits
      const unsigned rad = s << 10;
      unsigned r = (unsigned)(__builtin_sqrtf((float)s) * 32.0f);
      r = min(r, 65535u);
      if (r * r > rad) r--;
      if (r < 65535u && (r + 1) * (r + 1) <= rad) r++;
      const int e = (int)apc - (int)r;
      out = e < 0 ? 0u : (e > 0xffff ? 0xffffu : (unsigned)e);
    }
    if (writes) dst[(size_t)y * W + x] = (unsigned short)out;
    mx = max(mx, writes ? out : 0u);
    ha[0] = ha[1]; hc[0] = hc[1]; hb[0] = hb[1];
    ha[1] = ha[2]; hc[1] = hc[2]; hb[1] = hb[2];
  }
  for (int off = 32; off > 0; off >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, off, 64));
  if (lane == 0 && mx) atomicMax(maxv + n, mx);
}

hipError_t launch_gftt_eig(const uint8_t* img, uint16_t* eig, unsigned* maxv, int n, int W, int H, hipStream_t s) {
  hipError_t e = hipMemsetAsync(maxv, 0, (size_t)n * sizeof(unsigned), s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(gftt_eig_kernel, dim3((W + GF_NV - 1) / GF_NV, (H + GF_SEG - 1) / GF_SEG, n), dim3(64), 0, s, img, eig, maxv, W, H);
  return hipGetLastError();
}

}  // namespace sbm
