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
  int nsub, npacked;   // quad kernel: row segments walked in lockstep by one workgroup; workgroups that do so (full segments only)
  int xo[2];  // first output column (relative to lofs) of the left / right side; each side has w/2 columns
  int pfshift; // the planes hold (value << pfshift) + 1 (pre-scaled for the interior kernel, sbm_common.h): every sum, capb and tex
               // are scaled alike; only the uniqueness threshold and the stored cost go back to the unscaled sum
};


__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
typedef unsigned short bu16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned bpk_sub_sat(unsigned x, unsigned y) {   // per 16-bit half: max(x - y, 0)
  return __builtin_bit_cast(unsigned, __builtin_elementwise_sub_sat(__builtin_bit_cast(bu16x2, x), __builtin_bit_cast(bu16x2, y)));
}
__device__ __forceinline__ unsigned bpk_min(unsigned x, unsigned y) {       // per 16-bit half: min(x, y)
  return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(bu16x2, x), __builtin_bit_cast(bu16x2, y)));
}

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
  // these wavefronts are latency-bound and share their SIMDs with the VALU-bound interior kernel: let them issue first
  __builtin_amdgcn_s_setprio(3);
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
        const int ms = minsad >> a.pfshift;
        const int thresh = (ms + (ms * a.uniq / 100)) << a.pfshift;
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
            if (a.cost16) static_cast<uint16_t*>(a.cost)[o] = (uint16_t)(minsad >> a.pfshift);
            else static_cast<int32_t*>(a.cost)[o] = minsad >> a.pfshift;
          }
        }
        a.disp[o] = (int16_t)out;
      }
    }
    // the next iteration's __syncthreads (after staging into the other parity) orders Sbuf/Tcol reuse
  }
}

// ---------------------------------------------------------------------------------------------------------
// Second-generation border kernel: the same virtual-column scheme, but a lane owns a disparity QUAD and both sides
// share one workgroup. One v_mqsad_pk_u16_u8 with the pattern (l,0,0,0) -- the three zero bytes are masked --
// yields |l - R[k]| for four consecutive right bytes, i.e. four disparities of one virtual column, accumulated into a
// packed 4 x u16 register pair; the right row piece is staged in LDS as 8-byte entries laid out [p & 3][p >> 2] so
// that the quads of a wavefront read consecutive entries (conflict-free ds_read_b64). About 3x fewer instructions per
// row than the per-disparity kernel above (kept as SBM_BORDER_V=1 for A/B and as documentation of the scheme).
// ---------------------------------------------------------------------------------------------------------
extern __shared__ __attribute__((aligned(16))) unsigned char border_lds[];

// INPLACE: the vertical sums are accumulated in place (v_mqsad_pk_u16_u8 with vdst == src2: right on gfx950, checked on the
// device by mqsad_inplace_ok() before it is used, see sbm_sad_fast.hip) -- one array of NVC register pairs instead of two
// in ping-pong. At w = 21 that is 76 instead of 136 VGPRs: the wavefront fits beside four 100-VGPR interior wavefronts of a
// SIMD from the start of the launch instead of waiting for two of them to retire.
template <int W2, bool INPLACE>
__global__ void __launch_bounds__(128) sad_border2_kernel(BorderArgs a) {
  // these wavefronts are latency-bound and share their SIMDs with the VALU-bound interior kernel: let them issue first
  __builtin_amdgcn_s_setprio(3);
  constexpr int NVC = 3 * W2, WSZ = 2 * W2 + 1;
  constexpr int RPT = 6;                    // staged 8-byte entries per thread per row: jobs*(NVC+nd) <= 6*T (host-checked)
  typedef unsigned long long u64;
  const int tid = threadIdx.x, T = blockDim.x;
  const int nq = a.nd >> 2;                 // disparity quads per side
  const int nsl = (NVC + a.nd) / 4 + 2;     // 8-byte entries per residue class of one staged piece
  const int npiece = NVC + a.nd;            // entries of one staged piece (entry p = bytes p .. p+7)
  // Few disparities leave most of a side's 32 lanes idle (nd = 64: 16 of 32), so a workgroup then walks nsub = 2 or 4 row
  // segments in lockstep: lane group `sub` of each half-wavefront owns segment seg0 + sub. Only full-length segments are
  // packed (the first a.npacked workgroups); the remaining ones run one per workgroup. A "job" is one (segment, side).
  const bool packed = (int)blockIdx.x < a.npacked;
  const int nsub = packed ? a.nsub : 1;
  const int njobs = 2 * nsub;
  const int seg0 = packed ? (int)blockIdx.x * a.nsub : a.npacked * a.nsub + ((int)blockIdx.x - a.npacked);
  const int glanes = 32 / nsub;             // lanes of one job
  // ---- LDS carve-up (sized on the host for a.nsub) ---------------------------------------------------------
  u64* Rb = reinterpret_cast<u64*>(border_lds);                               // [par][which][job][4*nsl]
  unsigned short* Sb = reinterpret_cast<unsigned short*>(Rb + 2 * 2 * njobs * 4 * nsl);  // [job][W2][nd]
  int* Tc = reinterpret_cast<int*>(Sb + njobs * W2 * a.nd);                   // [job][NVC]
  unsigned* Best = reinterpret_cast<unsigned*>(Tc + njobs * NVC);             // [job][W2]
  int* Hit = reinterpret_cast<int*>(Best + njobs * W2);                       // [job][W2]
  unsigned char* Lb = reinterpret_cast<unsigned char*>(Hit + njobs * W2);     // [par][which][job][NVC]

  // lanes 0..31 = left side, lanes 32..63 = right side; within a side, lane group sub = (lane & 31) / glanes, quad q
  const int side = (tid >> 5) & 1;
  const int sub = (tid & 31) / glanes;
  const int q = tid & (glanes - 1);
  const int job = sub * 2 + side;
  const bool act = tid < 64 && q < nq;
  const int ys = a.row0 + seg0 * a.seg;     // first row of segment seg0; segment seg0 + sub starts sub * a.seg rows below
  const int ye = min(ys + a.seg, a.row1);   // packed segments are full length: ye - ys == a.seg for every sub
  const int pair = blockIdx.y;
  const uint8_t* pl = a.pf_l + (size_t)pair * a.plane + a.padl;
  const uint8_t* pr = a.pf_r + (size_t)pair * a.plane + a.padl;
  const int xfirst0 = a.xo[0] - W2, xfirst1 = a.xo[1] - W2;
  const int rb00 = a.rofs + clampi(xfirst0, -a.rofs, a.W - a.rofs - a.nd);
  const int rb01 = a.rofs + clampi(xfirst1, -a.rofs, a.W - a.rofs - a.nd);
  const int xfirst = side ? xfirst1 : xfirst0, rb0 = side ? rb01 : rb00;

  // per virtual column: byte offset of this quad's window entry inside a staged piece
  int off[NVC];
#pragma unroll
  for (int v = 0; v < NVC; v++) {
    const int ov = a.rofs + clampi(xfirst + v, -a.rofs, a.W - a.rofs - a.nd) - rb0;   // 0 .. NVC-1
    off[v] = (__mul24(ov & 3, nsl) + (ov >> 2) + q) * 8;   // (24-bit multiplies: the 32-bit ones run at quarter rate)
  }
  // staging plan of this thread, fixed for the whole segment: RPT right-row entries and 2 left bytes per row. Entry e of
  // the njobs * npiece right entries belongs to job e / npiece; its source lies (sub * seg) rows below the row of segment
  // seg0 (offset folded into gofs), its LDS slot is [job][p & 3][p >> 2].
  unsigned gofs[RPT];                      // byte offset from the row of segment seg0, ~0u: no entry
  int lidx[RPT];
#pragma unroll
  for (int k = 0; k < RPT; k++) {
    const int e = tid + k * T;
    gofs[k] = ~0u; lidx[k] = 0;
    if (e < njobs * npiece) {
      int j = 0;                                   // e / npiece without a division (at most 7 jobs to step over)
      for (int m = 1; m < njobs; m++) j += e >= m * npiece;
      const int pp = e - __mul24(j, npiece);
      gofs[k] = (unsigned)(__mul24(j >> 1, a.seg * a.pitch) + ((j & 1) ? rb01 : rb00) + pp);
      lidx[k] = __mul24(j, 4 * nsl) + __mul24(pp & 3, nsl) + (pp >> 2);
    }
  }
  unsigned lofs2[2];                       // same for the left bytes
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int e = tid + k * T;
    lofs2[k] = ~0u;
    if (e < njobs * NVC) {
      int j = 0;
      for (int m = 1; m < njobs; m++) j += e >= m * NVC;
      const int v = e - j * NVC;
      lofs2[k] = (unsigned)(__mul24(j >> 1, a.seg * a.pitch) + a.lofs + clampi(((j & 1) ? xfirst1 : xfirst0) + v, -a.lofs, a.W - a.lofs - 1));
    }
  }
  // vertical sums, packed 4 x u16 per virtual column, in ping-pong (mqsad may not overwrite a source): an entering
  // row maps CA -> CB through the free accumulate, the leaving row maps CB -> CA with plain subtractions
  uint2 CA[INPLACE ? 1 : NVC];
  u64 CB[NVC];
#pragma unroll
  for (int v = 0; v < NVC; v++) {
    if constexpr (INPLACE) CB[v] = 0ull;
    else CA[v] = make_uint2(0u, 0u);
  }
  int Ct[2] = {0, 0};   // texture: this thread's entries e = tid, tid + T of the njobs*NVC (job, virtual column) pairs

  struct Staged { u64 r[RPT]; unsigned char l[2]; };
  auto fetch = [&](int y) {                 // y: row of segment seg0
    Staged g;
    const uint8_t* lrow = pl + (size_t)y * a.pitch;
    const uint8_t* rrow = pr + (size_t)y * a.pitch;
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      u64 v = 0;
      if (gofs[k] != ~0u) __builtin_memcpy(&v, rrow + (size_t)gofs[k], 8);
      g.r[k] = v;
    }
#pragma unroll
    for (int k = 0; k < 2; k++) g.l[k] = lofs2[k] != ~0u ? lrow[(size_t)lofs2[k]] : (unsigned char)0;
    return g;
  };
  auto commit = [&](const Staged& g, int par, int which) {
    u64* rb = Rb + (size_t)((par * 2 + which) * njobs) * 4 * nsl;
    unsigned char* lb = Lb + ((par * 2 + which) * njobs) * NVC;
#pragma unroll
    for (int k = 0; k < RPT; k++)
      if (gofs[k] != ~0u) rb[lidx[k]] = g.r[k];
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int e = tid + k * T;
      if (lofs2[k] != ~0u) lb[e] = g.l[k];   // e = job*NVC + v
    }
  };
  // mode 0: CB = CA + row (enter)   mode 1: CA = CB - row (leave)   mode 2: CA = CB + row (second of a prime pair)
  auto accumulate = [&](int par, int which, const int mode) {
    const unsigned char* rbb = reinterpret_cast<const unsigned char*>(Rb + (size_t)((par * 2 + which) * njobs + job) * 4 * nsl);
    const unsigned char* lb = Lb + ((par * 2 + which) * njobs + job) * NVC;
    if (act) {
#pragma unroll
      for (int v = 0; v < NVC; v++) {
        const unsigned l = lb[v];
        const u64 win = *reinterpret_cast<const u64*>(rbb + off[v]);
        if constexpr (INPLACE) {
          if (mode != 1) {
            asm("v_mqsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(CB[v]) : "v"(win), "v"(l));
          } else {
            const uint2 t = __builtin_bit_cast(uint2, __builtin_amdgcn_mqsad_pk_u16_u8(win, l, 0ull));
            uint2 cb = __builtin_bit_cast(uint2, CB[v]);
            cb.x -= t.x;   // no u16 borrows: sums are exact
            cb.y -= t.y;
            CB[v] = __builtin_bit_cast(u64, cb);
          }
        } else if (mode == 0) {
          CB[v] = __builtin_amdgcn_mqsad_pk_u16_u8(win, l, __builtin_bit_cast(u64, CA[v]));
        } else if (mode == 2) {
          CA[v] = __builtin_bit_cast(uint2, __builtin_amdgcn_mqsad_pk_u16_u8(win, l, CB[v]));
        } else {
          const uint2 t = __builtin_bit_cast(uint2, __builtin_amdgcn_mqsad_pk_u16_u8(win, l, 0ull));
          const uint2 cb = __builtin_bit_cast(uint2, CB[v]);
          CA[v].x = cb.x - t.x;   // no u16 borrows: sums are exact
          CA[v].y = cb.y - t.y;
        }
      }
    }
    const unsigned char* lall = Lb + ((par * 2 + which) * njobs) * NVC;
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int e = tid + k * T;
      if (lofs2[k] != ~0u) {
        const int t = (int)lall[e] - a.capb;
        const int at = t < 0 ? -t : t;
        Ct[k] += mode == 1 ? -at : at;
      }
    }
  };

  // prime: rows ys-W2 .. ys+W2-1 (an even count) in CA -> CB -> CA pairs, fetching one row ahead
  int par = 0;
  Staged ge = fetch(ys - W2);
  for (int yy = ys - W2; yy < ys + W2; yy += 2) {
    commit(ge, par, 0);
    Staged g1 = fetch(yy + 1);
    __syncthreads();
    accumulate(par, 0, 0);
    par ^= 1;
    commit(g1, par, 0);
    ge = fetch(yy + 2);             // the last one fetched is row ys+W2: the first output row's entering row
    __syncthreads();
    accumulate(par, 0, 2);
    par ^= 1;
  }

  Staged gl = fetch(ys - W2);       // leaving row of the first output row
  for (int y = ys; y < ye; y++) {
    commit(ge, par, 0);
    commit(gl, par, 1);
    __syncthreads();
    if (y + 1 < ye) {               // next iteration's rows: entering y+1+W2, leaving y+1-W2
      ge = fetch(y + 1 + W2);
      gl = fetch(y + 1 - W2);
    }
    accumulate(par, 0, 0);          // CB = window rows y-W2 .. y+W2

    // sliding sums over the virtual columns -> W2 outputs, 4 disparities each; publish the sums (sub-pixel lookup) and
    // reduce the best key over the job's lanes in registers (DPP butterfly + one cross-row exchange when a job has 32 lanes)
    u64 S[W2];
    unsigned bestk[W2];
    if (act) {
      unsigned lo = 0, hi = 0;
#pragma unroll
      for (int v = 0; v < WSZ; v++) { lo += (unsigned)CB[v]; hi += (unsigned)(CB[v] >> 32); }
#pragma unroll
      for (int j = 0; j < W2; j++) {
        S[j] = ((u64)hi << 32) | lo;
        *reinterpret_cast<u64*>(Sb + ((size_t)(job * W2 + j) * a.nd + 4 * q)) = S[j];
        const unsigned d0 = 4u * q;
        const unsigned k0 = (lo << 16) | d0, k1 = (lo & 0xffff0000u) | (d0 + 1);
        const unsigned k2 = (hi << 16) | (d0 + 2), k3 = (hi & 0xffff0000u) | (d0 + 3);
        bestk[j] = min(min(k0, k1), min(k2, k3));
        if (j + 1 < W2) {
          lo += (unsigned)CB[j + WSZ] - (unsigned)CB[j];
          hi += (unsigned)(CB[j + WSZ] >> 32) - (unsigned)(CB[j] >> 32);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < W2; j++) { S[j] = ~0ull; bestk[j] = 0xffffffffu; }
    }
#pragma unroll
    for (int j = 0; j < W2; j++) {
      unsigned b = bestk[j];
      b = min(b, (unsigned)__builtin_amdgcn_mov_dpp((int)b, 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
      b = min(b, (unsigned)__builtin_amdgcn_mov_dpp((int)b, 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
      b = min(b, (unsigned)__builtin_amdgcn_mov_dpp((int)b, 0x141, 0xf, 0xf, true));  // row_half_mirror: the job's 8 lanes
      if (glanes >= 16) b = min(b, (unsigned)__builtin_amdgcn_mov_dpp((int)b, 0x140, 0xf, 0xf, true));  // row_mirror: 16 lanes
      if (glanes == 32) {            // the other 16-lane row of the side: v_permlane16_swap (odd rows of one copy <-> even rows of
        const auto r = __builtin_amdgcn_permlane16_swap(b, b, false, false);   // the other) instead of an LDS-crossbar shuffle
        b = min(r[0], r[1]);
      }
      bestk[j] = b;
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int e = tid + k * T;
      if (lofs2[k] != ~0u) Tc[e] = Ct[k];
    }
    // uniqueness: "some disparity outside mind-1..mind+1 has a sum <= thresh" <=> the job holds more sums <= thresh than
    // mind's neighbourhood does. Every lane counts its four packed sums with two saturating packed subtractions; the counts
    // of four output columns share a register (one byte each: a job's total is at most 128) and are summed over the job's
    // lanes by the same butterfly as the keys; the epilogue lane, which reads S[mind-1] and S[mind+1] anyway, compares.
    // (The per-disparity index tests this replaces were 38 % of the kernel's vector instructions.)
    constexpr int NPK = (W2 + 3) / 4;
    unsigned pk[NPK];
#pragma unroll
    for (int i = 0; i < NPK; i++) pk[i] = 0;
    if (a.uniq > 0) {
#pragma unroll
      for (int j = 0; j < W2; j++) {
        // minsad * uniq / 100 with a 24-bit multiply and one mulhi (x / 100 == mulhi(x, 0x51EB851F) >> 5 for 32-bit x); the
        // plain expression costs two quarter-rate 32-bit multiplies per output column. The envelope keeps the product < 2^23.
        const unsigned minsad = (bestk[j] >> 16) >> a.pfshift;          // the threshold is defined on the unscaled sum
        const unsigned thresh = (minsad + (__umulhi(__umul24(minsad, (unsigned)a.uniq), 0x51EB851Fu) >> 5)) << a.pfshift;
        const unsigned Tq = min(thresh + 1u, 65535u);                  // sums are <= 65534: sv <= thresh <=> Tq - sv > 0
        const unsigned T2 = Tq | (Tq << 16);
        const unsigned c2 = bpk_min(bpk_sub_sat(T2, (unsigned)S[j]), 0x00010001u) +
                            bpk_min(bpk_sub_sat(T2, (unsigned)(S[j] >> 32)), 0x00010001u);   // 0..2 per 16-bit half
        pk[j >> 2] |= ((c2 & 0xffffu) + (c2 >> 16)) << (8 * (j & 3));  // 0..4 (idle lanes hold 0xffff sums: 0)
      }
#pragma unroll
      for (int i = 0; i < NPK; i++) {
        unsigned b = pk[i];
        b += (unsigned)__builtin_amdgcn_mov_dpp((int)b, 0xB1, 0xf, 0xf, true);
        b += (unsigned)__builtin_amdgcn_mov_dpp((int)b, 0x4E, 0xf, 0xf, true);
        b += (unsigned)__builtin_amdgcn_mov_dpp((int)b, 0x141, 0xf, 0xf, true);
        if (glanes >= 16) b += (unsigned)__builtin_amdgcn_mov_dpp((int)b, 0x140, 0xf, 0xf, true);
        if (glanes == 32) {
          const auto r = __builtin_amdgcn_permlane16_swap(b, b, false, false);
          b = r[0] + r[1];
        }
        pk[i] = b;
      }
    }
    if (q == 0 && tid < 64) {
#pragma unroll
      for (int j = 0; j < W2; j++) {
        Best[job * W2 + j] = bestk[j];
        Hit[job * W2 + j] = (int)((pk[j >> 2] >> (8 * (j & 3))) & 0xffu);
      }
    }
    accumulate(par, 1, 1);          // CA = CB - leaving row (independent of the WTA merge; overlaps its latency)
    par ^= 1;
    __syncthreads();
    if (tid < njobs * W2) {         // one lane per (job, output column)
      const int jb = tid / W2, j = tid - jb * W2;
      const int sd = jb & 1, yo = y + (jb >> 1) * a.seg;
      const unsigned best = Best[tid];
      const int minsad = (int)(best >> 16), mind = (int)(best & 0xffffu);
      int ts = 0;
#pragma unroll
      for (int v = 0; v < WSZ; v++) ts += Tc[jb * NVC + j + v];
      const unsigned short* sb = Sb + (size_t)(jb * W2 + j) * a.nd;
      const int p = mind + 1 < a.nd ? sb[mind + 1] : sb[a.nd - 2];
      const int n = mind - 1 >= 0 ? sb[mind - 1] : sb[1];
      bool ok = ts >= a.tex;
      if (a.uniq > 0) {
        const int ms = minsad >> a.pfshift;
        const int tl = min((ms + (ms * a.uniq / 100)) << a.pfshift, 65534);
        const int expected = 1 + (mind + 1 < a.nd && p <= tl) + (mind - 1 >= 0 && n <= tl);   // of mind-1, mind, mind+1
        ok = ok && Hit[tid] == expected;
      }
      int out = a.filtered;
      const size_t o = (size_t)pair * a.W * a.H + (size_t)yo * a.W + a.lofs + a.xo[sd] + j;
      if (ok) {
        const int ad = p > n ? p - n : n - p;
        const int den = p + n - 2 * minsad + ad;
        int frac = 0;
        if (den != 0) {               // |p-n|*256 < 2^24: the float estimate is within 1 of the quotient (C division truncates)
          const unsigned num = (unsigned)ad << 8;
          unsigned qv = (unsigned)((float)num / (float)den);
          while ((unsigned long long)qv * (unsigned)den > num) qv--;
          while ((unsigned long long)(qv + 1) * (unsigned)den <= num) qv++;
          frac = p >= n ? (int)qv : -(int)qv;
        }
        out = ((a.nd - mind - 1 + a.mindisp) * 256 + frac + 15) >> 4;
        if (a.cost) {
          if (a.cost16) static_cast<uint16_t*>(a.cost)[o] = (uint16_t)(minsad >> a.pfshift);
          else static_cast<int32_t*>(a.cost)[o] = minsad >> a.pfshift;
        }
      }
      a.disp[o] = (int16_t)out;
    }
    // the next iteration's first __syncthreads orders the reuse of Sb/Tc/Best/Hit
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
  a.nd = g.nd; a.mindisp = g.mindisp; a.lofs = g.lofs; a.rofs = g.rofs; a.tex = g.tex << g.pfshift; a.uniq = g.uniq;
  a.filtered = g.filtered; a.capb = (g.cap << g.pfshift) + kPfBias;
  a.row0 = g.row0; a.row1 = g.row1;
  a.xo[0] = 0; a.xo[1] = xb;
  a.pfshift = g.pfshift;
  const int rows = g.row1 - g.row0;
  // short segments: each row costs a latency-bound staging round trip, so favour many concurrent workgroups
  static const int seg_rows_env = [] { const char* e = getenv("SBM_BORDER_SEG"); return e ? atoi(e) : 0; }();
  // The kernel runs on a side stream concurrently with the interior kernel. It is register-heavy (one of its
  // wavefronts displaces two of the interior kernel's) and latency-bound (barriers + serial merge per row), so the
  // best split is many short segments: measured flat optimum 12-24 rows on KITTI w15, 640x480 w21 and 1080p w21.
  static const int version_env = [] { const char* e = getenv("SBM_BORDER_V"); return e ? atoi(e) : 0; }();
  // quad-per-lane kernel up to 128 disparities (one wavefront per workgroup); beyond that its two-wavefront workgroups
  // with 256 VGPRs become the tail of the step (measured at 1080p nd256: 3.75 vs 3.64 ms) -> per-disparity kernel
  const int version = g.nd > 128 ? 1 : (version_env ? version_env : 2);   // the quad kernel maps one side to 32 lanes
  int seg_rows = seg_rows_env;
  if (seg_rows <= 0) {
    if (version != 1) {
      seg_rows = 12;
    } else {
      const double t_interior_us = (double)g.n * g.W * (g.row1 - g.row0) * g.nd / 3.0e12 * 1e6;
      seg_rows = (int)(0.6 * t_interior_us / 6.0) - (g.wsz - 1);
    }
    seg_rows = std::max(12, std::min(seg_rows, rows));
    // small batches (one pair per call): the chip is mostly idle and this kernel's serial row loop is the latency of
    // the whole SAD stage -> shorter segments, down to 4 rows (measured 640x480, one pair: 0.21 -> 0.16 ms per call)
    const long wgs = (long)g.n * std::max(1, rows / seg_rows);
    if (wgs < 1024) seg_rows = (int)std::max(4L, std::min((long)seg_rows, (long)g.n * rows / 1024));
  }
  int nseg = std::max(1, rows / seg_rows);
  a.seg = (rows + nseg - 1) / nseg;
  nseg = (rows + a.seg - 1) / a.seg;
  if (version != 1) {
    const int NVC = 3 * g.w2, nq = g.nd / 4, nsl = (NVC + g.nd) / 4 + 2;
    // few disparities leave most of a side's 32 lanes idle: pack nsub = 4 / 2 full-length row segments into one workgroup
    // (limits: lanes, the 6 staged entries and 2 left bytes per thread, one epilogue lane per job and output column).
    // Throughput regime only -- small batches keep one segment per workgroup (more, shorter workgroups: latency).
    const int nsub_env = [] { const char* e = getenv("SBM_BORDER_NSUB"); return e ? atoi(e) : 0; }();   // read per call
    auto fits = [&](int c) { return nq * c <= 32 && 2 * c * (NVC + g.nd) <= 6 * 64 && 2 * c * NVC <= 2 * 64 && 2 * c * g.w2 <= 64; };
    int nsub = fits(4) ? 4 : fits(2) ? 2 : 1;
    if ((long)g.n * nseg < 2048) nsub = 1;
    // SBM_BORDER_NSUB=1|2|4 forces the packing (where it fits) whatever the batch size: tests and A/B runs
    if ((nsub_env == 1 || nsub_env == 2 || nsub_env == 4) && fits(nsub_env)) nsub = nsub_env;
    const int full = rows / a.seg;                       // segments of full length
    a.nsub = nsub;
    a.npacked = nsub > 1 ? full / nsub : 0;
    const int nblocks = a.npacked + (nseg - a.npacked * nsub);
    const int njobs = 2 * nsub;
    const size_t lds = (size_t)2 * 2 * njobs * 4 * nsl * 8 + (size_t)njobs * g.w2 * g.nd * 2 + (size_t)njobs * NVC * 4 +
                       (size_t)2 * njobs * g.w2 * 4 + (size_t)2 * 2 * njobs * NVC + 16;
    dim3 grid2(nblocks, g.n);
    dim3 block2(64 * ((2 * nq + 63) / 64));
    const bool inplace = mqsad_inplace_ok(s);
#define SBM_B2(W) case W: if (inplace) hipLaunchKernelGGL((sad_border2_kernel<W, true>), grid2, block2, lds, s, a); \
                          else hipLaunchKernelGGL((sad_border2_kernel<W, false>), grid2, block2, lds, s, a); break;
    switch (g.w2) {   // every odd window 5..27
      SBM_B2(2) SBM_B2(3) SBM_B2(4) SBM_B2(5) SBM_B2(6) SBM_B2(7) SBM_B2(8) SBM_B2(9) SBM_B2(10) SBM_B2(11) SBM_B2(12) SBM_B2(13)
      default: return hipErrorInvalidValue;
    }
#undef SBM_B2
    return hipGetLastError();
  }
  dim3 grid(2 * nseg, g.n);
  dim3 block(64 * ((g.nd + 63) / 64));
#define SBM_B1(W) case W: hipLaunchKernelGGL(sad_border_kernel<W>, grid, block, 0, s, a); break;
  switch (g.w2) {
    SBM_B1(2) SBM_B1(3) SBM_B1(4) SBM_B1(5) SBM_B1(6) SBM_B1(7) SBM_B1(8) SBM_B1(9) SBM_B1(10) SBM_B1(11) SBM_B1(12) SBM_B1(13)
    default: return hipErrorInvalidValue;
  }
#undef SBM_B1
  return hipGetLastError();
}

}  // namespace sbm
