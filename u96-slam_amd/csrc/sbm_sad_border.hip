// sbm_sad_border.hip -- SAD/WTA for the w/2 border columns on each side of the fast kernel's range.  gfx950.
//
// These columns have CLAMPED windows: cv::StereoBM computes them, lets them
// take part in validateDisparity, and only then overwrites them with FILTERED, so they must be bit-exact whenever the
// LR check is on (setDisp12MaxDiff(1), src/slam/src/core/main.cpp:212).
//
// A window column x' maps to the pair (left column lofs+clamp(x',-lofs,W-lofs-1), right base rofs+clamp(x',-rofs,
// W-rofs-nd)); a border column's SAD is the sum of w such "virtual columns".  Per side there are only 3*(w/2)
// distinct virtual columns, so the kernel keeps, per disparity (= per thread), the vertical sliding sum C[v] of
// |L - R| for each of them in registers (one abs-diff in, one out per row) and forms the w/2 outputs as sliding
// sums over v.  That is O(1) work per (virtual column, disparity, row) instead of the generic kernel's O(w).
// One workgroup = one side of one row segment of one pair; threads = disparities; WTA per output column is a
// wavefront reduction over LDS-resident sums.  Same envelope as the fast kernel (16-bit-safe sums, nd <= 128).
#include <stdlib.h>

#include <algorithm>

#include "sbm_common.h"

namespace sbm {

struct BorderArgs {
  const uint8_t* pf_l;
  const uint8_t* pf_r;
  int16_t* disp;
  void* cost;      // uint16 plane when cost16, else int32
  int cost16;
  int W, H, pitch, padl, plane;
  int nd, mindisp, lofs, rofs, tex, uniq, filtered, capb;
  int row0, row1, seg;
  int xo[2];  // first output column (relative to lofs) of the left / right side; each side has w/2 columns
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned o = __shfl_xor(v, off, 64);
    v = o < v ? o : v;
  }
  return v;
}

template <int W2>
__global__ void __launch_bounds__(256) sad_border_kernel(BorderArgs a) {
  constexpr int NVC = 3 * W2, WSZ = 2 * W2 + 1;
  constexpr int RSPAN = NVC + 256;          // right bytes staged per row: rb(0) .. rb(0)+NVC+nd
  __shared__ uint8_t Lbuf[2][2][NVC + 1];   // [parity][enter/leave][virtual column] (clamp already applied)
  __shared__ uint8_t Rbuf[2][2][RSPAN];
  __shared__ int Sbuf[W2][256];
  __shared__ int Tcol[NVC];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int side = blockIdx.x & 1;
  const int ys = a.row0 + (blockIdx.x >> 1) * a.seg;
  const int ye = min(ys + a.seg, a.row1);
  const int pair = blockIdx.y;
  const int xfirst = a.xo[side] - W2;       // window column of virtual column 0
  const uint8_t* pl = a.pf_l + (size_t)pair * a.plane + a.padl;
  const uint8_t* pr = a.pf_r + (size_t)pair * a.plane + a.padl;
  const int rb0 = a.rofs + clampi(xfirst, -a.rofs, a.W - a.rofs - a.nd);
  const int d = tid;
  const bool dact = d < a.nd;

  int C[NVC];
#pragma unroll
  for (int v = 0; v < NVC; v++) C[v] = 0;
  int Ct = 0;  // thread v < NVC: vertical sum of |L - cap| of virtual column v

  // Staging is split into fetch (global -> registers, issued one row ahead so its latency hides behind the previous
  // row's arithmetic) and commit (registers -> LDS).  Per thread: up to RPT right bytes and one left byte per row.
  constexpr int RPT = (NVC + 256 + 63) / 64;   // enough for the smallest block (64 threads)
  struct Staged { uint8_t r[RPT]; uint8_t l; };
  const int nstage = NVC + a.nd;
  const int lcol = a.lofs + clampi(xfirst + (tid < NVC ? tid : 0), -a.lofs, a.W - a.lofs - 1);
  auto fetch = [&](int y) {
    Staged g;
    const uint8_t* lrow = pl + (size_t)y * a.pitch;
    const uint8_t* rrow = pr + (size_t)y * a.pitch + rb0;
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const int i = tid + k * (int)blockDim.x;
      g.r[k] = i < nstage ? rrow[i] : (uint8_t)0;
    }
    g.l = lrow[lcol];
    return g;
  };
  auto commit = [&](const Staged& g, int par, int which) {
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const int i = tid + k * (int)blockDim.x;
      if (i < nstage) Rbuf[par][which][i] = g.r[k];
    }
    if (tid < NVC) Lbuf[par][which][tid] = g.l;
  };
  // add (sign=+1) or remove (-1) one staged row
  auto accumulate = [&](int par, int which, int sign) {
    if (dact) {
#pragma unroll
      for (int v = 0; v < NVC; v++) {
        const int ov = a.rofs + clampi(xfirst + v, -a.rofs, a.W - a.rofs - a.nd) - rb0;
        // v_sad_u8 on zero-extended bytes: |l - r| + accumulator in one instruction
        const unsigned l = Lbuf[par][which][v], r = Rbuf[par][which][ov + d];
        if (sign > 0) C[v] = (int)__builtin_amdgcn_sad_u8(l, r, (unsigned)C[v]);
        else C[v] -= (int)__builtin_amdgcn_sad_u8(l, r, 0u);
      }
    }
    if (tid < NVC) {
      const int t = (int)Lbuf[par][which][tid] - a.capb;
      Ct += sign * (t < 0 ? -t : t);
    }
  };

  int par = 0;
  Staged ge = fetch(ys - W2);
  for (int yy = ys - W2; yy < ys + W2; yy++) {
    commit(ge, par, 0);
    __syncthreads();
    ge = fetch(yy + 1);          // the last one fetched here is row ys+W2, the first output row's entering row
    accumulate(par, 0, +1);
    par ^= 1;
  }

  Staged gl = ge;  // placeholder until there is a leaving row
  for (int y = ys; y < ye; y++) {
    commit(ge, par, 0);
    if (y > ys) commit(gl, par, 1);
    __syncthreads();
    if (y + 1 < ye) {            // next iteration's rows: entering y+1+W2, leaving y-W2
      ge = fetch(y + 1 + W2);
      gl = fetch(y - W2);
    }
    accumulate(par, 0, +1);
    if (y > ys) accumulate(par, 1, -1);
    par ^= 1;
    // sliding sums over the virtual columns -> the W2 border outputs of this disparity
    if (dact) {
      int s = 0;
#pragma unroll
      for (int v = 0; v < WSZ; v++) s += C[v];
#pragma unroll
      for (int j = 0; j < W2; j++) {
        Sbuf[j][d] = s;
        if (j + 1 < W2) s += C[j + WSZ] - C[j];
      }
    }
    if (tid < NVC) Tcol[tid] = Ct;
    __syncthreads();
    for (int j = wave; j < W2; j += nwaves) {
      unsigned bk = 0xffffffffu;
      for (int dd = lane; dd < a.nd; dd += 64) {
        const unsigned k = ((unsigned)Sbuf[j][dd] << 16) | (unsigned)dd;
        bk = k < bk ? k : bk;
      }
      const unsigned best = wave_min_u32(bk);
      const int minsad = (int)(best >> 16), mind = (int)(best & 0xffffu);
      int tsum = 0;
      for (int v = 0; v < WSZ; v++) tsum += Tcol[j + v];
      bool ok = tsum >= a.tex;
      if (a.uniq > 0) {
        const int thresh = minsad + (minsad * a.uniq / 100);
        bool hit = false;
        for (int dd = lane; dd < a.nd; dd += 64) hit |= (dd < mind - 1 || dd > mind + 1) && Sbuf[j][dd] <= thresh;
        ok = ok && __ballot(hit) == 0ull;
      }
      if (lane == 0) {
        int out = a.filtered;
        const size_t o = (size_t)pair * a.W * a.H + (size_t)y * a.W + a.lofs + a.xo[side] + j;
        if (ok) {
          const int p = mind + 1 < a.nd ? Sbuf[j][mind + 1] : Sbuf[j][a.nd - 2];
          const int n = mind - 1 >= 0 ? Sbuf[j][mind - 1] : Sbuf[j][1];
          const int ad = p > n ? p - n : n - p;
          const int den = p + n - 2 * minsad + ad;
          out = ((a.nd - mind - 1 + a.mindisp) * 256 + (den != 0 ? (p - n) * 256 / den : 0) + 15) >> 4;
          if (a.cost) {
            if (a.cost16) static_cast<uint16_t*>(a.cost)[o] = (uint16_t)minsad;
            else static_cast<int32_t*>(a.cost)[o] = minsad;
          }
        }
        a.disp[o] = (int16_t)out;
      }
    }
    // the next iteration's __syncthreads (after staging into the other parity) orders Sbuf/Tcol reuse
  }
}

bool sad_border_supported(const Geom& g) { return sad_fast_supported(g); }

hipError_t launch_sad_border(const uint8_t* pf_l, const uint8_t* pf_r, int16_t* disp, int32_t* cost, const Geom& g,
                             int xa, int xb, hipStream_t s) {
  // left side = columns [0,xa), right side = [xb,xend); both hold exactly w/2 columns when the fast range exists
  if (xa != g.w2 || g.xend - xb != g.w2) return hipErrorInvalidValue;
  BorderArgs a;
  a.pf_l = pf_l; a.pf_r = pf_r; a.disp = disp; a.cost = g.want_cost ? cost : nullptr; a.cost16 = g.cost16;
  a.W = g.W; a.H = g.H; a.pitch = g.pitch; a.padl = g.padl; a.plane = g.plane;
  a.nd = g.nd; a.mindisp = g.mindisp; a.lofs = g.lofs; a.rofs = g.rofs; a.tex = g.tex; a.uniq = g.uniq;
  a.filtered = g.filtered; a.capb = g.cap + kPfBias;
  a.row0 = g.row0; a.row1 = g.row1;
  a.xo[0] = 0; a.xo[1] = xb;
  const int rows = g.row1 - g.row0;
  // short segments: each row costs a latency-bound staging round trip, so favour many concurrent workgroups
  static const int seg_rows_env = [] { const char* e = getenv("SBM_BORDER_SEG"); return e ? atoi(e) : 0; }();
  // The kernel runs on a side stream concurrently with the VALU-bound interior kernel, so what it costs is its
  // instruction total (priming = w-1 extra rows per segment) -- as long as its critical path, (seg + w - 1) row steps
  // of ~6 us each (two barriers per row), stays inside the interior kernel's duration. Estimate that duration from
  // the interior kernel's measured rate (~3e12 pixel-disparities/s) and spend 60 % of it.
  int seg_rows = seg_rows_env;
  if (seg_rows <= 0) {
    const double t_interior_us = (double)g.n * g.W * (g.row1 - g.row0) * g.nd / 3.0e12 * 1e6;
    seg_rows = (int)(0.6 * t_interior_us / 6.0) - (g.wsz - 1);
    seg_rows = std::max(12, std::min(seg_rows, rows));
  }
  int nseg = std::max(1, rows / seg_rows);
  a.seg = (rows + nseg - 1) / nseg;
  nseg = (rows + a.seg - 1) / a.seg;
  dim3 grid(2 * nseg, g.n);
  dim3 block(64 * ((g.nd + 63) / 64));
  switch (g.w2) {
    case 4: hipLaunchKernelGGL(sad_border_kernel<4>, grid, block, 0, s, a); break;
    case 7: hipLaunchKernelGGL(sad_border_kernel<7>, grid, block, 0, s, a); break;
    case 10: hipLaunchKernelGGL(sad_border_kernel<10>, grid, block, 0, s, a); break;
    case 13: hipLaunchKernelGGL(sad_border_kernel<13>, grid, block, 0, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace sbm
