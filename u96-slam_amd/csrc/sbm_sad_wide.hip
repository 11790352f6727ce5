// sbm_sad_wide.hip -- SAD / WTA / texture / uniqueness / sub-pixel for configurations outside the fast kernel's envelope
// (block sizes above 27, more than 512 disparities, sums beyond 16 bits) and for the clamped border columns of the fast kernel
// beyond 256 disparities, gfx950.
//
// Same definition as sbm_sad_generic.hip (findStereoCorrespondenceBM of OpenCV calib3d stereobm.cpp as reached from
// src/slam/src/core/main.cpp:215; clamped windows, 32-bit sums, first minimum wins -- SURVEY.md Appendix A.3/A.4), evaluated
// with sliding sums in BOTH directions instead of a w-wide row sum per (pixel, disparity, row):
//
//   workgroup = (pair, row segment, tile of TX output columns); wavefront wv owns the 64-disparity chunks wv, wv + NW, ...;
//   lanes = disparities.  S[x][d] (window sums of the current row, 32 bit) lives in LDS.
//   per row:  A  every wavefront walks the tile left to right with D(d) = H_enter(x, d) - H_leave(x, d) in a register per
//                chunk (H = horizontal window sum of one row; one step = two bytes in, two bytes out per row), adds it to
//                S[x][d] and reduces its chunks to a partial winner (one DPP minimum over keys sum << 6 | lane); wavefront 0
//                slides the texture sums along (the left bytes are wavefront-uniform);
//             B  a thread per column merges the partial winners and derives the uniqueness threshold;
//             C  every wavefront tests its chunks against the threshold (ballot);
//             D  a thread per column: neighbours from S, sub-pixel fit, stores.
//   Cost: ~80-110 wavefront-instructions per (pixel, 64 disparities), independent of the block size (the per-column kernel:
//   ~4 w per pixel-disparity).
#include <algorithm>
#include <type_traits>

#include "sbm_common.h"

namespace sbm {

typedef unsigned int u32;
typedef unsigned long long u64;

constexpr int kWideWaves = 8;        // wavefronts per workgroup (at most; one per 64-disparity chunk below that)

struct WideArgs {
  const uint8_t* pf_l;
  const uint8_t* pf_r;
  int16_t* disp;
  int32_t* cost;
  int W, H, pitch, padl, plane;
  int nd, mindisp, wsz, cap, lofs, rofs, tex, uniq, filtered;
  int row0, row1, xa, xb;
  int tx, seg;
  int pfshift, cost16;   // planes hold (value << pfshift) + 1; the cost plane holds uint16 (0xffff = filtered) -- border columns of the fast path
};

__device__ __forceinline__ int wclamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// minimum over the wavefront (result wavefront-uniform): butterfly inside each row of 16 lanes, then lane 15 of a row into the
// next row, lane 31 into the upper half -- lane 63 ends up with the minimum
__device__ __forceinline__ u32 wave_min_u32(u32 v) {
  v = min(v, (u32)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, false));               // quad_perm [1,0,3,2]
  v = min(v, (u32)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, false));               // quad_perm [2,3,0,1]
  v = min(v, (u32)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xf, 0xf, false));              // row_half_mirror
  v = min(v, (u32)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xf, 0xf, false));              // row_mirror
  v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xa, 0xf, false));   // row_bcast:15 into rows 1 and 3
  v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xc, 0xf, false));   // row_bcast:31 into rows 2 and 3
  return __builtin_amdgcn_readlane(v, 63);
}

extern __shared__ __attribute__((aligned(16))) u32 wide_lds[];

template <int CPW>   // 64-disparity chunks per wavefront
__global__ void __launch_bounds__(64 * kWideWaves) sad_wide_kernel(WideArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nw = (int)(blockDim.x >> 6), nth = (int)blockDim.x, tid = (int)threadIdx.x;
  const int x0 = a.xa + (int)blockIdx.x * a.tx;             // first output column of the tile (relative to lofs)
  const int txn = min(a.tx, a.xb - x0);                     // <= 64: column xi of the tile also lives in lane xi
  const int ys = a.row0 + (int)blockIdx.y * a.seg, ye = min(ys + a.seg, a.row1);
  const int pair = (int)blockIdx.z;
  const int w2 = a.wsz / 2, nd = a.nd, nch = (nd + 63) >> 6, nds = nch * 64;
  const u32 capw = (u32)((a.cap << a.pfshift) + kPfBias);
  // row bases (column 0 of the padded planes); a left byte sits at lofs + clamped column, a right byte at rofs + clamped column + d
  const uint8_t* const pl = a.pf_l + (size_t)pair * a.plane + a.padl;
  const uint8_t* const pr = a.pf_r + (size_t)pair * a.plane + a.padl;
  const int llo = 0, lhi = a.W - 1, rlo = 0, rhi = a.W - nd;             // clamps of lofs + x, rofs + x
  const size_t obase = (size_t)pair * a.W * a.H;

  // LDS: S[tx][nds] (lanes beyond nd have slots of their own, never read), per-column results, partial winners [tx][nw],
  // texture sums, uniqueness flags
  u32* const S = wide_lds;
  int4* const fin = reinterpret_cast<int4*>(S + (size_t)a.tx * nds);   // {minsad, mind, thresh, texture ok}
  u64* const pm = reinterpret_cast<u64*>(fin + a.tx);
  int* const T = reinterpret_cast<int*>(pm + (size_t)a.tx * nw);
  u32* const uq = reinterpret_cast<u32*>(T + a.tx);

  // this lane's buffer index in each of the wavefront's chunks
  int dj[CPW];
  u32 dld[CPW];
  bool dv[CPW], cv[CPW];
#pragma unroll
  for (int j = 0; j < CPW; j++) {
    const int c = wave + nw * j;
    cv[j] = c < nch;                       // (wavefront-uniform)
    dj[j] = 64 * c + lane;
    dv[j] = cv[j] && dj[j] < nd;
    dld[j] = (u32)min(dj[j], nd - 1);      // loads of lanes without a disparity stay inside the row
  }
  int Tv = 0;                              // wavefront 0: lane xi holds the texture sum of column xi

  // One walk over the tile: S[x][d] += H(ya) - H(yb) (PRIME: + H(ya) only, no winner search); wavefront 0 slides the
  // texture sums |L - cap| along (the left bytes are wavefront-uniform: the same four loads)
  auto sweep = [&](auto prime_tag, const int ya, const int yb) {
    constexpr bool PRIME = decltype(prime_tag)::value;
    const uint8_t* const la = pl + (size_t)ya * a.pitch;
    const uint8_t* const ra = pr + (size_t)ya * a.pitch;
    const uint8_t* const lb = pl + (size_t)yb * a.pitch;
    const uint8_t* const rb = pr + (size_t)yb * a.pitch;
    u32 D[CPW], Dt = 0u;
#pragma unroll
    for (int j = 0; j < CPW; j++) D[j] = 0u;
    // the window of the tile's first column
    for (int dx = -w2; dx <= w2; dx++) {
      const int li = wclamp(a.lofs + x0 + dx, llo, lhi), ri = wclamp(a.rofs + x0 + dx, rlo, rhi);
      const u32 lav = la[li], lbv = PRIME ? 0u : (u32)lb[li];
      const uint8_t* const rpa = ra + ri;
      const uint8_t* const rpb = rb + ri;
#pragma unroll
      for (int j = 0; j < CPW; j++) {
        if (!cv[j]) continue;
        D[j] = __builtin_amdgcn_sad_u8(lav, (u32)rpa[dld[j]], D[j]);
        if (!PRIME) D[j] -= __builtin_amdgcn_sad_u8(lbv, (u32)rpb[dld[j]], 0u);
      }
      Dt = __builtin_amdgcn_sad_u8(lav, capw, Dt);
      if (!PRIME) Dt -= __builtin_amdgcn_sad_u8(lbv, capw, 0u);
    }
    u32 kS = 0xffffffffu, kD = 0u;   // lane xi: this wavefront's partial winner of column xi
    int Tadd = 0;                    // lane xi: the row's texture contribution to column xi
    for (int xi = 0; xi < txn; xi++) {
      const int x = x0 + xi;
      // the next step's bytes first: their latency runs under this column's update and reduction
      const int li_in = wclamp(a.lofs + x + 1 + w2, llo, lhi), ri_in = wclamp(a.rofs + x + 1 + w2, rlo, rhi);
      const int li_out = wclamp(a.lofs + x - w2, llo, lhi), ri_out = wclamp(a.rofs + x - w2, rlo, rhi);
      const u32 la_in = la[li_in], la_out = la[li_out];
      const u32 lb_in = PRIME ? 0u : (u32)lb[li_in], lb_out = PRIME ? 0u : (u32)lb[li_out];
      const uint8_t* const ra_ip = ra + ri_in;
      const uint8_t* const ra_op = ra + ri_out;
      const uint8_t* const rb_ip = rb + ri_in;
      const uint8_t* const rb_op = rb + ri_out;
      u32 ra_in[CPW], ra_out[CPW], rb_in[CPW], rb_out[CPW];
#pragma unroll
      for (int j = 0; j < CPW; j++) {
        ra_in[j] = ra_out[j] = rb_in[j] = rb_out[j] = 0;
        if (j > 0 && !cv[j]) continue;       // (a wavefront's first chunk always exists)
        ra_in[j] = ra_ip[dld[j]];
        ra_out[j] = ra_op[dld[j]];
        if (!PRIME) { rb_in[j] = rb_ip[dld[j]]; rb_out[j] = rb_op[dld[j]]; }
      }
      u32 bS = 0xffffffffu, bD = 0u;       // (wavefront-uniform) winner over this wavefront's chunks
#pragma unroll
      for (int j = 0; j < CPW; j++) {
        if (j > 0 && !cv[j]) continue;
        u32* const sp = S + (size_t)xi * nds + dj[j];
        const u32 sn = *sp + D[j];
        *sp = sn;
        if (!PRIME) {
          // one reduction for the minimum and its first lane: sums stay below 2^26 (255^2 x 126 < 2^23, pre-scaled by at most 4)
          const u32 key = dv[j] ? (sn << 6) | (u32)lane : 0xffffffffu;
          const u32 m = wave_min_u32(key);
          if ((m >> 6) < bS) { bS = m >> 6; bD = (u32)(64 * (wave + nw * j)) + (m & 63u); }   // chunks ascend: the first minimum stays
        }
      }
      if (!PRIME) {
        kS = lane == xi ? bS : kS;
        kD = lane == xi ? bD : kD;
      }
      if (wave == 0) Tadd = lane == xi ? (int)Dt : Tadd;
      // slide the window one column to the right
#pragma unroll
      for (int j = 0; j < CPW; j++) {
        if (j > 0 && !cv[j]) continue;
        if (PRIME) {
          D[j] = __builtin_amdgcn_sad_u8(la_in, ra_in[j], D[j]) - __builtin_amdgcn_sad_u8(la_out, ra_out[j], 0u);
        } else {
          const u32 p = __builtin_amdgcn_sad_u8(la_in, ra_in[j], __builtin_amdgcn_sad_u8(lb_out, rb_out[j], D[j]));
          const u32 q = __builtin_amdgcn_sad_u8(la_out, ra_out[j], __builtin_amdgcn_sad_u8(lb_in, rb_in[j], 0u));
          D[j] = p - q;
        }
      }
      if (PRIME) {
        Dt = __builtin_amdgcn_sad_u8(la_in, capw, Dt) - __builtin_amdgcn_sad_u8(la_out, capw, 0u);
      } else {
        Dt = __builtin_amdgcn_sad_u8(la_in, capw, __builtin_amdgcn_sad_u8(lb_out, capw, Dt)) -
             __builtin_amdgcn_sad_u8(la_out, capw, __builtin_amdgcn_sad_u8(lb_in, capw, 0u));
      }
    }
    Tv += Tadd;
    if (!PRIME) {
      if (lane < txn) pm[(size_t)lane * nw + wave] = ((u64)kS << 32) | kD;
      if (wave == 0 && lane < txn) T[lane] = Tv;
    }
  };

  // ---- prime with the window of row ys-1: rows ys-w2-1 .. ys+w2-1 (clamped to the image) ---------------------------------
  for (int i = tid; i < txn * nds; i += nth) S[i] = 0u;
  __syncthreads();
  for (int yy = ys - w2 - 1; yy <= ys + w2 - 1; yy++)
    sweep(std::true_type{}, wclamp(yy, 0, a.H - 1), 0);   // (a wavefront only touches its own chunks of S)

  for (int y = ys; y < ye; y++) {
    const int ya = wclamp(y + w2, 0, a.H - 1), yb = wclamp(y - w2 - 1, 0, a.H - 1);
    // ---- A: slide down one row, partial winners ----------------------------------------------------------------------
    sweep(std::false_type{}, ya, yb);
    __syncthreads();
    // ---- B: a thread per column: winner, texture, threshold -------------------------------------------------------------
    for (int xi = tid; xi < txn; xi += nth) {
      u64 best = pm[(size_t)xi * nw];
      for (int w = 1; w < nw; w++) {
        const u64 k = pm[(size_t)xi * nw + w];
        best = k < best ? k : best;
      }
      const int minsad = (int)(best >> 32), mind = (int)(best & 0xffffffffu);
      // (pre-scaled planes: every sum is a multiple of 1 << pfshift; the threshold is defined on the unscaled sum)
      const int ms = minsad >> a.pfshift;
      fin[xi] = make_int4(minsad, mind, (ms + (ms * a.uniq / 100)) << a.pfshift, (T[xi] >> a.pfshift) >= a.tex ? 1 : 0);
      uq[xi] = 0u;
    }
    __syncthreads();
    // ---- C: uniqueness: any d outside [mind-1, mind+1] with S[d] <= thresh rejects -------------------------------------
    if (a.uniq > 0) {
      for (int xi = 0; xi < txn; xi++) {
        const int4 f = fin[xi];
        if (!f.w) continue;
        bool hit = false;
#pragma unroll
        for (int j = 0; j < CPW; j++)
          if (dv[j]) hit |= (dj[j] < f.y - 1 || dj[j] > f.y + 1) && S[(size_t)xi * nds + dj[j]] <= (u32)f.z;
        if (__ballot(hit) != 0ull && lane == 0) uq[xi] = 1u;
      }
      __syncthreads();
    }
    // ---- D: a thread per column: sub-pixel fit, stores ------------------------------------------------------------------
    for (int xi = tid; xi < txn; xi += nth) {
      const int4 f = fin[xi];
      const bool ok = f.w && uq[xi] == 0u;
      int out = a.filtered;
      if (ok) {
        const u32* const Sx = S + (size_t)xi * nds;
        const int minsad = f.x, mind = f.y;
        const int p = mind + 1 < nd ? (int)Sx[mind + 1] : (int)Sx[nd - 2];
        const int n = mind - 1 >= 0 ? (int)Sx[mind - 1] : (int)Sx[1];
        const int ad = p > n ? p - n : n - p;
        const int den = p + n - 2 * minsad + ad;
        out = ((nd - mind - 1 + a.mindisp) * 256 + (den != 0 ? (p - n) * 256 / den : 0) + 15) >> 4;
      }
      const size_t o = obase + (size_t)y * a.W + a.lofs + x0 + xi;
      a.disp[o] = (int16_t)out;
      if (a.cost) {
        if (a.cost16) reinterpret_cast<uint16_t*>(a.cost)[o] = ok ? (uint16_t)(f.x >> a.pfshift) : (uint16_t)0xffffu;
        else if (ok) a.cost[o] = f.x;
      }
    }
    __syncthreads();   // S, pm, fin are rewritten by the next row
  }
}

// nd <= 64 * kWideWaves * 4 (= 2048); everything else about the parameters is free
bool sad_wide_supported(const Geom& g) { return g.nd >= 2 && g.nd <= 64 * kWideWaves * 4; }

hipError_t launch_sad_wide(const uint8_t* pf_l, const uint8_t* pf_r, int16_t* disp, int32_t* cost, const Geom& g, int xa, int xb,
                           hipStream_t s) {
  if (xb <= xa || g.row1 <= g.row0) return hipSuccess;
  WideArgs a;
  a.pf_l = pf_l; a.pf_r = pf_r; a.disp = disp; a.cost = g.want_cost ? cost : nullptr;
  a.W = g.W; a.H = g.H; a.pitch = g.pitch; a.padl = g.padl; a.plane = g.plane;
  a.nd = g.nd; a.mindisp = g.mindisp; a.wsz = g.wsz; a.cap = g.cap; a.lofs = g.lofs; a.rofs = g.rofs;
  a.tex = g.tex; a.uniq = g.uniq; a.filtered = g.filtered;
  a.row0 = g.row0; a.row1 = g.row1; a.xa = xa; a.xb = xb;
  a.pfshift = g.pfshift; a.cost16 = g.cost16;
  const int nch = (g.nd + 63) / 64, nw = std::min(kWideWaves, nch), cpw = (nch + nw - 1) / nw;
  const int rows = g.row1 - g.row0, cols = xb - xa;
  // tile: S of a workgroup sized for ~16 wavefronts per CU (160 KB of LDS: 10 KB per wavefront of the workgroup), at most 64
  // columns (column xi of a tile also lives in lane xi). The w - 1 columns a walk takes before its first output cost about a
  // seventh of an output column each. Segment: every segment primes w rows at about a fifth of a row's cost, so 2 w rows
  // and more -- shorter while the grid is small
  const int nds = nch * 64;
  int tx = std::max(4, std::min(std::min(64, 10240 * nw / (4 * nds + 8 * nw + 24)), cols));
  tx = (cols + (cols + tx - 1) / tx - 1) / ((cols + tx - 1) / tx);          // equal tiles
  const int tiles = (cols + tx - 1) / tx;
  int seg = std::min(rows, std::max(32, 2 * g.wsz));
  while (seg > 16 && (long)tiles * ((rows + seg - 1) / seg) * g.n < 2048) seg = (seg + 1) / 2;
  a.tx = tx; a.seg = seg;
  dim3 grid((unsigned)tiles, (unsigned)((rows + seg - 1) / seg), (unsigned)g.n);
  const size_t lds = (size_t)tx * nds * 4 + (size_t)tx * nw * 8 + (size_t)tx * (16 + 4 + 4);   // <= 10 KB x nw, + 2 KB at tx = 4
  auto go = [&](auto kern) {
    // (more than 64 KB of dynamic LDS has to be granted per kernel; idempotent, so unsynchronised repeats are harmless)
    if (lds > 64 * 1024) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, grid, dim3(64 * nw), lds, s, a);
    return hipGetLastError();
  };
  if (cpw <= 1) return go(sad_wide_kernel<1>);
  if (cpw == 2) return go(sad_wide_kernel<2>);
  return go(sad_wide_kernel<4>);
}

}  // namespace sbm
