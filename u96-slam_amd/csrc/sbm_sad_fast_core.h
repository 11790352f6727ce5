// sbm_sad_fast_core.h -- what every translation unit of the interior SAD kernel shares: launch arguments, small packed-math
// helpers, the LDS carve-up of a wavefront, the plan of the horizontal window sum, and the per-row tail of a strip (winner
// search, uniqueness, neighbour look-up, sub-pixel) that the LDS-direct strip (sbm_sad_fast_strip.h) and the two-accumulator
// fallback strip (sbm_sad_fast_pp_strip.h) have in common. gfx950 only. See sbm_sad_fast.hip for the kernel's description.
#pragma once
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <cmath>
#include <mutex>
#include <type_traits>

#include "sbm_common.h"

#ifndef SBM_FAST_PINGPONG   // 1 in sbm_sad_fast_pp.hip only: the fallback build with two accumulator arrays
#define SBM_FAST_PINGPONG 0
#endif

namespace sbm {

typedef unsigned int u32;
typedef unsigned long long u64;
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct FastArgs {
  const uint8_t* pf_l;
  const uint8_t* pf_r;
  int16_t* disp;
  uint16_t* cost;            // 16-bit cost plane (sums fit by the envelope)
  int W, H, pitch, padl, plane;
  int nd, mindisp, lofs, rofs, tex, uniq, filtered, capb;
  int row0, row1;            // rows [row0,row1)
  int segrow[66];            // row segment k = rows [segrow[k], segrow[k+1]); long segments first, short ones last
  int strips, nseg, npairs;  // grid decomposition (1-D grid of strips*nseg*npairs workgroups)
  int strips3;               // the first strips3 strips (a multiple of 3) have column stride 3, the others stride 1
  int uniq_plain;            // 8 * (maxS * uniq / 100 + 1) fits 16 bits: deficit partial sums need no saturating adds
  int xc0, xc1;              // interior centre columns [xc0,xc1) (relative to lofs); xc0 = w/2
  int pfshift;               // the planes hold (value << pfshift) + 1: every sum below is scaled by 1 << pfshift (0 or 2)
  // border jobs (sbm_sad_border_wave.h): the grid starts with nbseg x bord workgroups that carry the clamped border columns
  int bord;                  // border workgroups per border row segment (0: no border columns wanted)
  int bnw;                   // border wavefronts per border row segment: 2 sides x ceil(pairs / JW) pair groups
  int bseg, nbseg;           // the border jobs' own row segments: nbseg segments of bseg rows (the last one shorter)
};

__device__ __forceinline__ u32 pk_sub_sat(u32 a, u32 b) {
  u16x2 r = __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b));
  return __builtin_bit_cast(u32, r);
}
__device__ __forceinline__ u32 pk_add_sat(u32 a, u32 b) {
  u16x2 r = __builtin_elementwise_add_sat(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b));
  return __builtin_bit_cast(u32, r);
}
__device__ __forceinline__ u32 pk_min(u32 a, u32 b) {
  u16x2 r = __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b));
  return __builtin_bit_cast(u32, r);
}
// one v_min3_u32 (the compiler re-associates min(a, min(b, c)) chains and then only finds about half of them)
__device__ __forceinline__ u32 umin3(u32 a, u32 b, u32 c) {
  u32 r;
  asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

extern __shared__ __attribute__((aligned(16))) uint4 fast_lds[];  // the workgroup's LDS: per wavefront its strip areas, then the merge area

#include "sbm_sad_border_wave.h"   // (inside namespace sbm, after FastArgs and the helpers)

// s_setprio level of a wavefront while it is in its horizontal exchange (a chain of LDS round trips with a few adds behind
// each; the phases after it are hundreds of back-to-back vector instructions). With 3-4 wavefronts per SIMD and oldest-first
// issue, a wavefront in one of those long arithmetic phases keeps the port while its neighbour's adds wait, the neighbour's next
// LDS reads go out late and the CU's LDS pipe idles. KITTI x64 SAD stage 0.963 -> 0.902 ms, 640x480 nd 64 w 21 0.450 -> 0.430,
// 1080p nd 256 2.59 -> 2.52 (round 3). Also raising it during the mqsad phases gains nothing at nd <= 128 and costs 10 % with
// cooperating wavefronts.
constexpr int kFastPrioExchange = 2;

// LDS carve-up of one wavefront in 16-byte slots, shared by the strips and the launcher
template <int NDW, int NTERM, int PW, int CS>
struct FastLds {
  static constexpr int NQ = NDW / 4;
  static constexpr int NCH = (NQ + 15) / 16;   // chunks of 16 quads
  // staging: slot p holds bytes p..p+15 of the right row piece; the last window read of lane 63 is slot
  // CS*63 + 4 + 16 * (4 * (NCH - 1) + 3)
  static constexpr int NSLOT = ((CS * 63 + 4 + 16 * (4 * (NCH - 1) + 3) + 1) + 63) / 64 * 64;
  static constexpr int KS = CS == 3 ? 1 : PW;  // lane distance between two partners of the horizontal window
  // horizontal exchange of the register-staged strip (fallback build): XCH quads at a time, XS entries per quad pair (64 lanes
  // + the KS*(NTERM-1) halo). Chunk sizes from round 3: 4 quads for the cooperating 64-disparity wavefronts, 2 where one
  // wavefront holds 128 disparities.
  static constexpr int XCHMAX = NDW >= 128 ? 2 : 4;
  static constexpr int XCH = NQ < XCHMAX ? NQ : XCHMAX;
  static constexpr int XS = 64 + KS * (NTERM - 1);
  static constexpr int XSLOT = (XCH / 2) * XS + (XS * 4 + 15) / 16;
  static constexpr int WSLOT = NSLOT > XSLOT ? NSLOT : XSLOT;
  static_assert(CS == 1 || (CS == 3 && PW == 3), "column stride 3 goes with 3-column sums");
};

// Plan of the horizontal window sum (LDS-direct strips). The window is NTERM vertical sums V at lane distance KS. Summing
// them directly costs one publish (ds_write_b128), NTERM - 1 partner reads and (NTERM - 1) / 2 three-operand adds per entry of
// four registers; with an intermediate level -- T = S1 consecutive V, published to a second exchange area and read back
// shifted -- the window is NTT T's + NVV V's (greedy, left to right): w 21 = T(0) + T(3) + V(6) with T = 3 V: 2 publishes,
// 2 + 2 reads, 2 adds instead of 1, 6, 3; w 19 (1-column sums) = 6 T + V: 2 publishes, 2 + 6 reads, 4 adds instead of 1, 18, 9.
// A publish is the expensive part (~25-30 SIMD-cycles in this kernel's mix, more than a v_mqsad_pk_u16_u8: the store path
// moves address and data registers at 2 cycles per dword and holds the SIMD's register ports -- profiles/r05_sad_isa_budget.md),
// a partner read ~3.5, an add3 over the entry 19. S1 per window, each measured against its neighbours (KITTI x64 nd 128,
// profiles/r05_envelope.txt): 5 terms stay direct (a second publish costs more than the add it saves: w 15 0.781 -> 0.841 ms);
// 7, 9, 11, 13, 19 terms: T = 3 V; 17: T = 4 V; 23, 25: T = 5 V. A third level (U = 3 T) lost everywhere it was tried (w 19:
// 1.281 against 1.245 ms, w 23: 1.597 against 1.528), and so did, at 7 terms with two cooperating wavefronts, the direct sum,
// T = 2 V and exchange chunks of 4 / 8 quads (profiles/r06_w21_exchange.txt).
template <int NTERM, int PW>
struct HPlan {
  static constexpr int S1 = NTERM >= 23 ? 5 : (NTERM == 17 ? 4 : (NTERM >= 7 ? 3 : 1));
  static constexpr int NTT = S1 > 1 ? NTERM / S1 : 0;
  static constexpr int NVV = NTERM - NTT * S1;
  static constexpr bool PUB1 = NTT >= 2;            // T is read by other lanes
  static constexpr int NLEV = 1 + PUB1;             // exchange areas
};

// LDS of one wavefront of an LDS-direct strip, in bytes: two staged-row areas (the first doubles as exchange level 0; 64
// left-pattern dwords behind each) and the further exchange levels of the plan.
// LDS-direct staging is dword-granular: lane i of a load writes bytes i..i+3 of the row piece to dword slot i (the 4x-expanded
// layout, read back with ds_read2_b32). The 16-byte form into a 16x-expanded layout (round 4's first version) costs the texture
// path 64 CU-cycles per byte-misaligned wavefront-instruction against 16 (tools/ubench/lds_dma_rate.hip,
// profiles/r04_lds_dma_rate.txt) and WAS the bound at 64 disparities and below (KITTI x64 nd 32: 0.733 -> 0.466 ms per step).
template <int NDW, int NWAVES, int NTERM, int PW, int CS>
struct DmaLds {
  using L = FastLds<NDW, NTERM, PW, CS>;
  using P = HPlan<NTERM, PW>;
  static constexpr int XCH = L::NQ >= 2 ? 2 : L::XCH;                       // exchange chunk (quads)
  static constexpr int STAGE_B = L::NSLOT * 4;                              // one staged right row piece, 4x-expanded (dword slots)
  static constexpr int XLEV_B = ((XCH / 2) * L::XS + (L::XS * 4 + 15) / 16) * 16;   // one exchange level: quad entries + texture column
  static constexpr int PAT_OFS = STAGE_B > XLEV_B ? STAGE_B : XLEV_B;       // the left patterns of a staged row
  static constexpr int AREA_B = PAT_OFS + 256;
  static constexpr int WAVE_B = 2 * AREA_B + (P::NLEV - 1) * XLEV_B;
};

// ---------------------------------------------------------------------------------------------------------------------------
// The per-row tail of a strip, on the NR = NDW / 2 packed sums S of this wavefront's disparities (register j = buffer indices
// 2j | 2j+1 << 16, relative to the wavefront's first index).
// ---------------------------------------------------------------------------------------------------------------------------

// WTA: first index attaining the minimum, as (sum << 16 | index): keys carry a group-local index (inline constants for
// v_lshl_or_b32 / v_and_or_b32), four independent v_min3_u32 chains keep the dependency chains short. Ties: the smaller key is
// the smaller buffer index, as cv's strict '<' scan.
// On pre-scaled planes (sbm_common.h, pfshift) every sum is a multiple of 4 (2), so the low bits of each packed half can carry a
// register tag: registers j, j + NR/4, j + NR/2, j + 3NR/4 (tags 0..3 = the top two bits of the buffer index) are reduced with
// packed 16-bit minima first -- one OR (a full-rate instruction) and one v_pk_min_u16 per register instead of two key builds and
// a v_min3_u32 -- and only the NR/4 (NR/2 with one tag bit: windows 17..21, where only 2 maxS fits 16 bits) survivors get 32-bit
// keys; the smaller (sum, tag, low index bits) triple is the smaller buffer index. One tagged variant per instantiation, chosen
// by the window: a third alternative in the same body makes the register allocator spill hundreds of bytes everywhere.
template <int NR, int WSZ>
__device__ __forceinline__ u32 fast_first_min(const u32 (&S)[NR], const int pfshift) {
  u32 best = 0xffffffffu;
  constexpr int TSMAX = WSZ <= 15 ? 2 : 1;
  if (NR >= 16 && TSMAX == 2 && pfshift == 2) {
    constexpr int NG = NR / 4;
    u32 b[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
#pragma unroll
    for (int j = 0; j < NG; j++) {
      const u32 g01 = pk_min(S[j], S[j + NG] | 0x00010001u);
      const u32 g23 = pk_min(S[j + 2 * NG] | 0x00020002u, S[j + 3 * NG] | 0x00030003u);
      const u32 gm = pk_min(g01, g23);
      const u32 klo = (gm << 16) | (u32)(2 * j);
      const u32 khi = (gm & 0xffff0000u) | (u32)(2 * j + 1);
      b[j & 3] = umin3(b[j & 3], klo, khi);
    }
    const u32 bt = min(min(b[0], b[1]), min(b[2], b[3]));       // (4 S + tag) << 16 | low index bits
    best = (bt & 0xfffc0000u) | (((bt >> 16) & 3u) * (u32)(2 * NG) + (bt & 0xffffu));
  } else if (NR >= 16 && TSMAX == 1 && pfshift == 1) {
    constexpr int NG = NR / 2;
    u32 b[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    const u32 tag1 = 0x00010001u;
#pragma unroll
    for (int j = 0; j < NG; j++) {
      const u32 gm = pk_min(S[j], S[j + NG] | tag1);
      const u32 klo = (gm << 16) | (u32)(2 * j);
      const u32 khi = (gm & 0xffff0000u) | (u32)(2 * j + 1);
      b[j & 3] = umin3(b[j & 3], klo, khi);
    }
    const u32 bt = min(min(b[0], b[1]), min(b[2], b[3]));       // (2 S + tag) << 16 | low index bits
    best = (bt & 0xfffe0000u) | (((bt >> 16) & 1u) * (u32)(2 * NG) + (bt & 0xffffu));
  } else {
#pragma unroll
    for (int g0 = 0; g0 < NR; g0 += 32) {
      u32 b[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
#pragma unroll
      for (int j = g0; j < g0 + 32 && j < NR; j++) {
        const u32 klo = (S[j] << 16) | (u32)(2 * (j - g0));
        const u32 khi = (S[j] & 0xffff0000u) | (u32)(2 * (j - g0) + 1);
        b[j & 3] = umin3(b[j & 3], klo, khi);
      }
      const u32 bg = min(min(b[0], b[1]), min(b[2], b[3])) + (u32)(2 * g0);
      best = min(best, bg);
    }
  }
  return best;
}

// the uniqueness threshold on the (scaled) sums: T = thresh + 1 with cv's thresh = minsad + minsad * uniq / 100 on the unscaled sum
__device__ __forceinline__ u32 fast_uniq_threshold(const int minsad, const int uniq, const int pfshift) {
  const int ms = minsad >> pfshift;
  const int thresh = ms + (ms * uniq / 100);
  return (u32)min((thresh + 1) << pfshift, 65535);
}

// Uniqueness, part 1: saturating sum of the deficits max(T - S[d], 0), per 16-bit half. Independent accumulators of 8 registers
// each (a single chain is one dependent v_pk_sub -> v_pk_add pair per register with a wait state in between). Every deficit is
// at most T - minsad <= maxS*uniq/100 + 1; when 8 of them cannot reach 65536 (host check, uniq_plain) the partial sums are plain
// 32-bit adds of the packed halves -- no carry can cross -- and only the final combine saturates. Saturating adds give
// min(65535, sum) in any grouping.
template <int NR>
__device__ __forceinline__ u32 fast_deficits(const u32 (&S)[NR], const u32 T, const int uniq_plain) {
  const u32 T2 = T | (T << 16);
  constexpr int NACC = NR >= 32 ? NR / 8 : 4;
  u32 ac[NACC];
#pragma unroll
  for (int k = 0; k < NACC; k++) ac[k] = 0u;
  if (uniq_plain) {
#pragma unroll
    for (int j = 0; j < NR; j++) ac[j % NACC] += pk_sub_sat(T2, S[j]);
  } else {
    // (an opaque copy of the threshold: otherwise the compiler hoists the NR subtractions both paths share above
    // the branch and keeps all of them live at once -- 32 registers at the kernel's pressure peak)
    u32 T2s = T2;
    asm("" : "+v"(T2s));
#pragma unroll
    for (int j = 0; j < NR; j++) ac[j % NACC] = pk_add_sat(ac[j % NACC], pk_sub_sat(T2s, S[j]));
  }
#pragma unroll
  for (int n = NACC; n > 1; n >>= 1)
#pragma unroll
    for (int k = 0; k < n / 2; k++) ac[k] = pk_add_sat(ac[k], ac[k + n / 2]);
  return ac[0];
}

// first level of the neighbour selection: one v_perm_b32 per quad picks S[4q + (ln & 3)] (low half) and S[4q + (lp & 3)] (high half)
// -- byte selectors built arithmetically from the packed index pair lnp = ln | lp << 16 (compares + selects cost several times as much)
template <int NQ>
__device__ __forceinline__ void fast_neighbours_quads(const u32 (&S)[2 * NQ], const u32 lnp, u32 (&X)[NQ]) {
  // bytes (2a, 2a+1) with a = index & 3:  0x0100 + a * 0x0202 per half
  const u32 sel = __umul24(lnp & 0x00030003u, 0x0202u) + 0x01000100u;
#pragma unroll
  for (int q = 0; q < NQ; q++) X[q] = __builtin_amdgcn_perm(S[2 * q + 1], S[2 * q], sel);
}
// ... and the binary tree over the quads: bit `lvl` of each index picks the odd entry (src0 = bytes 4..7) or the even one (src1 =
// bytes 0..3): selector halves 0x0100 / 0x0504 (low, index ln) and 0x0302 / 0x0706 (high, index lp) = base + bit * 0x0404.
// Returns S[ln] | S[lp] << 16.
template <int NQ>
__device__ __forceinline__ u32 fast_neighbours_tree(u32 (&X)[NQ], const u32 lnp) {
  int lvl = 2;
#pragma unroll
  for (int n = NQ; n > 1; n >>= 1) {
    const u32 sel = __umul24((lnp >> lvl) & 0x00010001u, 0x0404u) + 0x03020100u;
#pragma unroll
    for (int m = 0; m < n / 2; m++) X[m] = __builtin_amdgcn_perm(X[2 * m + 1], X[2 * m], sel);
    lvl++;
  }
  return X[0];
}

// Uniqueness, part 2: any d outside [mind-1, mind+1] with S[d] <= thresh rejects -- the deficit sums of the two halves (even /
// odd buffer indices) must equal what the three neighbourhood entries alone account for.
__device__ __forceinline__ bool fast_unique(const u32 acc_lo, const u32 acc_hi, const u32 T, const int minsad, const int mind, const int nn,
                                            const int pp, const int nd) {
  const u32 dm = T - (u32)minsad;                                     // >= 1
  const u32 dn = (mind > 0 && (u32)nn < T) ? T - (u32)nn : 0u;         // S[mind-1] exists
  const u32 dp = (mind < nd - 1 && (u32)pp < T) ? T - (u32)pp : 0u;    // S[mind+1] exists
  const u32 e_same = dm, e_other = dn + dp;                            // mind's parity half / the other half
  const u32 exp_lo = (mind & 1) ? e_other : e_same, exp_hi = (mind & 1) ? e_same : e_other;
  return acc_lo == exp_lo && acc_hi == exp_hi;
}

// cv's sub-pixel step: ((nd - mind - 1 + minDisparity) * 256 + (p - n) * 256 / (p + n - 2 c + |p - n|) + 15) >> 4.
// den = (p + n - 2 minsad) + |p - n| >= |p - n|, so the quotient is at most 256: one reciprocal estimate (relative error 2^-22)
// is within 1 of it and one exact remainder settles which way (24-bit products).
__device__ __forceinline__ int fast_subpixel(const int nn, const int pp, const int minsad, const int mind, const int nd, const int mindisp) {
  const int ad = pp > nn ? pp - nn : nn - pp;
  const int den = pp + nn - 2 * minsad + ad;
  int frac = 0;
  if (den != 0) {
    const u32 num = (u32)ad << 8;
    u32 qv = (u32)((float)num * __builtin_amdgcn_rcpf((float)den));
    const int rem = (int)num - (int)__umul24(qv, (u32)den);
    qv = rem < 0 ? qv - 1 : (rem >= den ? qv + 1 : qv);
    frac = pp >= nn ? (int)qv : -(int)qv;          // C division truncates toward zero
  }
  return ((nd - mind - 1 + mindisp) * 256 + frac + 15) >> 4;
}

}  // namespace sbm
