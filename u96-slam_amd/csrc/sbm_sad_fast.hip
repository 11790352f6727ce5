// sbm_sad_fast.hip -- the hot kernel: windowed SAD over the disparity range + WTA + texture + uniqueness +
// sub-pixel, for the interior (unclamped) columns.  gfx950 / CDNA4 only.
//
// Device counterpart of findStereoCorrespondenceBM (OpenCV calib3d stereobm.cpp) reached from
// src/slam/src/core/main.cpp:215.  In-tree hardware twins: src/dvp/rtl/bm_calc_sad.v:391-605 (34-lane abs-diff,
// vertical running sums, horizontal running SAD), bm_calc_det.v:124-426 (WTA tree), bm_calc_frac.v (sub-pixel).
//
// Mapping (a workgroup = NWAVES cooperating wavefronts over the SAME 64 columns; wavefront k owns disparities
// [NDW*k, NDW*k+NDW); two barriers per row where NWAVES > 1, none otherwise):
//   lane      = one image column c (64 consecutive columns per workgroup), marching down a row segment
//   registers = V[d]: for every disparity the sum over the w window rows of the 3-column SAD
//               H3(c,y,d) = sum_{i<3} |Lp[y][c+i] - Rp[y][c+i-D]|, packed 4 x u16 per VGPR pair.
//               One v_mqsad_pk_u16_u8 produces H3 for 4 consecutive disparities AND accumulates (pattern = 3 left
//               bytes + a zero byte, which the instruction masks; sliding 8-byte window = right bytes).
//               Entering row: V = mqsad(R, L, V), accumulated IN PLACE (vdst == src2: right on gfx950, checked by a device
//               self-test; sbm_sad_fast_pp.hip is the two-array fallback).  Leaving row: V -= mqsad(R, L, 0).
//   LDS       = the right row piece of the wavefront, staged by LDS-direct loads (buffer_load_dword ... lds: no staging
//               registers) into a 4x-expanded layout -- dword slot p holds bytes p..p+3, so a quad's 8-byte window is the
//               dword pair (4q, 4q+4) behind the lane's slot, one conflict-free ds_read2_b32 (sad_fast_strip_dma: every
//               layout since round 5). The register-staged strip with its 16x-expanded layout (sad_fast_strip) remains for
//               the fallback build.
//   exchange  = horizontal window: S(c + w/2) = sum_k V(c + 3k), k < w/3: lanes publish V to LDS ([quad pair][lane],
//               16 B entries) and read the shifted copies back; from 7 terms on in two levels (HPlan: T = a few V,
//               published again, window = a few T + the remaining V). Lanes whose partners fall outside the wavefront
//               produce nothing and are recomputed by the next strip.
//   WTA       = in registers: min over (S << 16 | d) keys (first d wins ties, as cv's strict '<' scan; a tagged packed
//               search on pre-scaled planes where the sums leave bits free), uniqueness by a saturating deficit sum,
//               S[mind +- 1] by a v_perm_b32 selection tree; per-wavefront results merged through LDS, one wavefront
//               (alternating per row) finishes and stores.
// Column stride (template parameter CS, windows that are multiples of 3): with CS = 3 lane i takes column base + 3 i, so the
// partners of the horizontal window V(c), V(c+3), ... are the NEXT LANES and only NTERM - 1 lanes of a wavefront produce
// nothing (w 15: 60 of 64 lanes useful instead of 52; w 21: 58 instead of 46). Three such wavefronts (bases b, b+1, b+2)
// tile 3 * (65 - NTERM) contiguous columns; they are ordinary, independent strips. The right row piece a wavefront stages
// grows from 64 + nd to 190 + nd bytes (window reads at a lane stride of 48 bytes: still conflict-free), nothing else
// changes. Columns left over by the last full triple take CS = 1 strips -- both bodies live in one kernel, chosen by a
// workgroup-uniform branch on the strip index.
// Windows that are not multiples of 3 use 1-column sums (template parameter PW = 1: single-byte pattern, w-1 partners).
// Envelope (checked on the host, everything else takes the generic kernel): odd w in 5..27, nd <= 256,
// w*w*2*cap <= 65534 (16-bit sums), 2*(maxS*uniq/100+1) < 65535, valid-ROI rows inside [w/2, H-w/2).
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <cmath>
#include <mutex>
#include <type_traits>

#include "sbm_common.h"

#ifndef SBM_FAST_EXACT512   // exact-count kernels for 384 and 512 disparities (three / four full 128-disparity wavefronts)
#define SBM_FAST_EXACT512 1
#endif
#ifndef SBM_FAST_PRIO_XCH   // s_setprio level during the horizontal exchange (0 = off; development builds compare)
#define SBM_FAST_PRIO_XCH 2
#endif
#ifndef SBM_FAST_PINGPONG
#define SBM_FAST_PINGPONG 0
#endif

// Translation units of this file (the kernel has ~190 instantiations: windows x layouts x exact / masked disparity counts):
//   SBM_FAST_TU 0  sbm_sad_fast.hip itself: host side + the windows that are multiples of 3 (3-column sums)
//   SBM_FAST_TU 1  sbm_sad_fast_pw1.hip: the windows 5, 7, 11, 13 (1-column sums), reached through launch_sad_fast_pw1()
//   SBM_FAST_TU 2  sbm_sad_fast_pw2.hip: the windows 17, 19, 23, 25, reached through launch_sad_fast_pw2()
//   SBM_FAST_TU 3  sbm_sad_fast_pw3.hip: the windows 29, 31, reached through launch_sad_fast_pw3()
//   ping-pong      sbm_sad_fast_pp.hip: the two-accumulator fallback, every window, 64-disparity layouts only
#ifndef SBM_FAST_TU
#define SBM_FAST_TU 0
#endif
#if SBM_FAST_PINGPONG   // second build of this file (sbm_sad_fast_pp.hip): same kernels with two accumulator arrays
#define sad_fast_kernel sad_fast_pp_kernel
#define sad_fast_strip sad_fast_pp_strip
#define launch_sad_fast launch_sad_fast_pp
#define FastArgs FastArgsPP
#define fast_lds fast_pp_lds
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

__device__ __forceinline__ uint4 load_u128_unaligned(const uint8_t* p) {
  uint4 v;
  __builtin_memcpy(&v, p, 16);
  return v;
}
__device__ __forceinline__ u32 load_u32_ua(const uint8_t* p) {
  u32 v;
  __builtin_memcpy(&v, p, 4);
  return v;
}
__device__ __forceinline__ u32 bperm(int byte_addr, u32 v) { return (u32)__builtin_amdgcn_ds_bpermute(byte_addr, (int)v); }
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

extern __shared__ __attribute__((aligned(16))) uint4 fast_lds[];  // per wave NSLOT 16-byte slots, then exchange area

#include "sbm_sad_border_wave.h"

// NDW disparities per wavefront, NWAVES wavefronts per workgroup covering NDW*NWAVES >= nd disparities of the SAME
// 64 columns.  NWAVES > 1 keeps the register footprint of a wavefront at NDW/2 accumulators + NDW/2 sums (4 waves per
// SIMD at NDW = 64) at the price of two workgroup barriers per row for the WTA merge through LDS.
#if SBM_FAST_PINGPONG
#define SBM_FAST_WAVES_PER_EU
#else
// wavefronts per SIMD the register allocation aims at: 5 for the cooperating 64-disparity wavefronts (their two barriers
// per row want the extra wavefront to cover the waits), 4 for a lone wavefront (no barriers: measured slower at 5),
// 3 for the 128-disparity single-wavefront variant
// (two cooperating wavefronts of <= 64 disparities each only run small launches -- one pair per call -- and 48 disparities:
// 4, i.e. 128 VGPRs, keeps their border wavefronts free of scratch -- at 96 VGPRs their spill reloads were memory round trips
// inside the serial chain that IS the length of a one-pair SAD stage: 640x480 nd 64 w 21, one pair: 0.089 -> 0.071 ms)
#define SBM_FAST_WAVES_PER_EU __attribute__((amdgpu_waves_per_eu(NDW > 64 ? 3 : (NWAVES > 2 ? SBM_FAST_WPE : SBM_FAST_WPE1))))
#ifndef SBM_FAST_WPE
#define SBM_FAST_WPE 5
#endif
#ifndef SBM_FAST_WPE1
#define SBM_FAST_WPE1 4
#endif
#endif
// LDS carve-up of one wavefront in 16-byte slots, shared by the kernel and its launcher
template <int NDW, int NTERM, int PW, int CS>
struct FastLds {
  static constexpr int NQ = NDW / 4;
  static constexpr int NCH = (NQ + 15) / 16;   // chunks of 16 quads
  // staging: slot p holds bytes p..p+15 of the right row piece; the last window read of lane 63 is slot
  // CS*63 + 4 + 16 * (4 * (NCH - 1) + 3)
  static constexpr int NSLOT = ((CS * 63 + 4 + 16 * (4 * (NCH - 1) + 3) + 1) + 63) / 64 * 64;
  static constexpr int KS = CS == 3 ? 1 : PW;  // lane distance between two partners of the horizontal window
  // horizontal exchange: XCH quads at a time, XS entries per quad pair (64 lanes + the KS*(NTERM-1) halo)
  // (chunk size re-measured once the exchange had its issue priority: 8 quads per chunk -> 4: 1080p -2 %, 2160p -2 %, KITTI
  // -0.5 %; 2 quads: KITTI -1 % but +4 % / +8 % with cooperating wavefronts -- so 2 where one wavefront holds 128 disparities)
#ifndef SBM_FAST_XCH128   // exchange chunk (quads) of the 128-disparity wavefront / of the others (development builds compare)
#define SBM_FAST_XCH128 2
#endif
#ifndef SBM_FAST_XCH64
#define SBM_FAST_XCH64 4
#endif
  static constexpr int XCHMAX = NDW >= 128 ? SBM_FAST_XCH128 : SBM_FAST_XCH64;
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
// 1.281 against 1.245 ms, w 23: 1.597 against 1.528) and is not in the code.
#ifndef SBM_FAST_HPLAN   // 0: direct sums everywhere (the round-4 exchange; development builds compare)
#define SBM_FAST_HPLAN 1
#endif
template <int NTERM, int PW>
struct HPlan {
  static constexpr int S1 = !SBM_FAST_HPLAN ? 1 : (NTERM >= 23 ? 5 : (NTERM == 17 ? 4 : (NTERM >= 7 ? 3 : 1)));
  static constexpr int NTT = S1 > 1 ? NTERM / S1 : 0;
  static constexpr int NVV = NTERM - NTT * S1;
  static constexpr bool PUB1 = NTT >= 2;            // T is read by other lanes
  static constexpr int NLEV = 1 + PUB1;             // exchange areas
};
// Every layout stages its rows with LDS-direct loads (sad_fast_strip_dma): two staged rows per wavefront (entering / leaving),
// the first one doubling as exchange level 0. Round 4 kept the register-staged strip for the 64-disparity cooperating
// wavefronts (two 16x-expanded staged rows per wavefront cost them a workgroup per CU); with dword staging the areas are a
// quarter of that and the LDS-direct strip wins there too (1080p nd 192: 1.88 -> 1.65 ms, one 1080p pair: 0.248 -> 0.190).
// SBM_FAST_DMA_ALL=0 (development builds): single-wavefront workgroups and the two 128-disparity wavefronts only.
#ifndef SBM_FAST_DMA_ALL
#define SBM_FAST_DMA_ALL 1
#endif
// LDS-direct staging is dword-granular: lane i of a load writes bytes i..i+3 of the row piece to dword slot i (the 4x-expanded
// layout, read back with ds_read2_b32). The 16-byte form into a 16x-expanded layout (round 4's first version) costs the texture path
// 64 CU-cycles per byte-misaligned wavefront-instruction against 16 (tools/ubench/lds_dma_rate.hip, profiles/r04_lds_dma_rate.txt) and
// WAS the bound at 64 disparities and below (KITTI x64 nd 32: 0.733 -> 0.466 ms per step, nd 64: 0.721 -> 0.603, nd 128: 0.960 ->
// 0.944; 640x480 nd 64 w 21: 0.514 -> 0.489). Round 5: the two cooperating 128-disparity wavefronts take the dword form as well
// (1080p nd 256: 2.012 against 2.014 ms, 2160p: 2.165 against 2.19 -- profiles/r05_nbr_lds_negative.txt, run 2, dev_d2): their staged
// rows shrink from 5 KB to 1.3 KB, which is what lets the plan's extra exchange level in without costing a workgroup per CU.
constexpr bool fast_dma(int ndw, int nwaves) { return !SBM_FAST_PINGPONG && (SBM_FAST_DMA_ALL || nwaves == 1 || (ndw == 128 && nwaves == 2)); }

// LDS of one wavefront of an LDS-direct strip, in bytes: two staged-row areas (the first doubles as exchange level 0; 64
// left-pattern dwords behind each) and the further exchange levels of the plan.
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

// One strip of one row segment of one pair: lane i works on column cbase + CS * i (relative to lofs).
template <int NDW, int NWAVES, int NTERM, int PW, bool EXACT_ND, int CS>
__device__ __forceinline__ void sad_fast_strip(const FastArgs& a, const int cbase, const int segi, const int pair) {
  using L = FastLds<NDW, NTERM, PW, CS>;
  constexpr int NQ = NDW / 4;           // disparity quads of this wavefront (one u64 accumulator each)
  constexpr int NR = NDW / 2;           // packed pair registers
  constexpr int NSLOT = L::NSLOT;
  constexpr int NIT = NSLOT / 64;
  // PW = columns per vertical sum (the mqsad pattern width): 3 when the window is a multiple of 3, else 1
  constexpr int WSZ = PW * NTERM, W2 = WSZ / 2;
  constexpr int KS = L::KS;
  constexpr int NV = 64 - KS * (NTERM - 1);   // lanes that produce an output
  constexpr int XCH = L::XCH, XS = L::XS;

  const int lane = threadIdx.x & 63;
  const int wv = NWAVES > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
  const int d0 = wv * NDW;                              // first buffer index of this wavefront
  const int c = cbase + CS * lane;                      // this lane's column (relative to lofs): V covers c..c+2
  const int xc = c + W2;                                // centre column this lane produces
  const bool produces = lane < NV && xc >= a.xc0 && xc < a.xc1;
  const int ys = a.segrow[segi];
  const int ye = a.segrow[segi + 1];
  // wavefront-uniform bases (scalar registers; the per-row step is scalar arithmetic) + this lane's 32-bit offset
  const uint8_t* pl = a.pf_l + (size_t)pair * a.plane + a.padl + a.lofs + cbase;  // left bytes: + CS * lane
  // right piece of the wavefront: window of buffer index d starts at rofs + c + d
  const uint8_t* pr = a.pf_r + (size_t)pair * a.plane + a.padl + a.rofs + cbase + d0;
  const unsigned lane_u = (unsigned)lane;
  // raw buffer descriptors over the rest of this pair's planes (rows of one pair are < 2^31 bytes apart)
  const __amdgpu_buffer_rsrc_t rs_l = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(pl), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(pr), 0, 0x7fffffff, 0x00020000);
  // LDS carve-up (16-byte units): per wavefront one region of WSLOT slots that serves first as the staging area of a
  // row (apply) and then as the exchange area of the horizontal window -- never live together, and a wavefront's LDS
  // operations execute in order -- followed by the WTA merge area of the workgroup.
  constexpr int WSLOT = L::WSLOT;
  uint4* const stage_lds = fast_lds + wv * WSLOT;
  uint4* const xq = stage_lds;                                          // [XCH/2 quad pairs][XS lanes], 8 x u16 each
  u32* const xt = reinterpret_cast<u32*>(xq + (XCH / 2) * XS);          // [XS] texture column sums
  u32* const xkey = reinterpret_cast<u32*>(fast_lds + NWAVES * WSLOT);  // [2][NWAVES][64]
  uint2* const xacc = reinterpret_cast<uint2*>(xkey + 2 * NWAVES * 64); // [2][NWAVES][64]  (deficit acc, nn | pp<<16)
  const u32 capw = (u32)a.capb * 0x01010101u;

  // packed 4 x u16 per quad (low dword = indices 4q,4q+1, high dword = 4q+2,4q+3).
#if SBM_FAST_PINGPONG
  // Two arrays in ping-pong: the compiler never lets v_mqsad_pk_u16_u8 write a register it reads (vdst is early-clobber
  // against every source in LLVM), so an entering row maps VA -> VB through the instruction's free accumulate and the
  // leaving row maps VB -> VA with plain subtractions.
  uint2 VA[NQ];
  u64 VB[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) VA[q] = make_uint2(0u, 0u);
#else
  // ONE array, accumulated in place: on gfx950 the instruction gives the right result when vdst is its own accumulator
  // (src2) -- verified on the device by tools/ubench/mqsad_alias (9.4e9 results) and re-checked by
  // sbm_selftest_mqsad_inplace() when a handle is created (the ping-pong build of this kernel, sbm_sad_fast_pp.hip, is
  // the fallback). That frees 32 VGPRs: 5 wavefronts per SIMD instead of 4.
  u64 VB[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) VB[q] = 0ull;
#endif
  u32 Vt = 0;  // texture: window-row sum of the 3-column |L - cap|

  // One row of one image contributes in three phases, split so that global-load latency overlaps compute:
  //   fetch  : global -> registers (this wavefront's right row piece + this lane's left bytes)
  //   expand : registers -> LDS in the 16x expanded layout
  //   apply  : LDS -> mqsad -> V (add for an entering row, subtract for a leaving row)
  struct RowRegs { uint4 r[NIT]; u32 l; };
  auto fetch = [&](int y) {
    RowRegs g;
    // buffer loads: descriptor base + this lane's 32-bit offset + the row offset in a scalar register, so a row
    // costs no vector address arithmetic (flat 64-bit addressing cost two v_mad_u64_u32 per fetch)
    const int rowoff = __builtin_amdgcn_readfirstlane(y * a.pitch);
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_r, (int)(it * 64 + lane_u), rowoff, 0);
      g.r[it] = make_uint4(v.x, v.y, v.z, v.w);
    }
    g.l = __builtin_amdgcn_raw_buffer_load_b32(rs_l, (int)(CS * lane_u), rowoff, 0);
    return g;
  };
  // mode 0: VB = VA + row (enter)   mode 1: VA = VB - row (leave)   mode 2: VA = VB + row (second half of a prime pair)
  auto apply = [&](const RowRegs& g, const int mode) {
#pragma unroll
    for (int it = 0; it < NIT; it++) stage_lds[it * 64 + lane] = g.r[it];
    constexpr u32 PMASK = PW == 3 ? 0x00ffffffu : 0x000000ffu;
    const u32 pat = g.l & PMASK;  // remaining bytes = 0 -> masked by mqsad
    const u32 tv = __builtin_amdgcn_sad_u8(pat | (capw & ~PMASK), capw, 0u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // 16 quads (64 disparities) at a time: 4 + 4 ds_read_b128 cover their 17 window dwords in both alignments
    // (lane stride CS * 16 bytes: 16 consecutive lanes hit 64 distinct banks for CS = 1 and for CS = 3).
    const uint4* const win_lds = stage_lds + CS * lane;
#pragma unroll
    for (int q0 = 0; q0 < NQ; q0 += 16) {
      constexpr int NM = 4;
      uint4 ra[NM], rb[NM];
#pragma unroll
      for (int m = 0; m < NM; m++) {
        ra[m] = win_lds[16 * (q0 / 4 + m)];
        rb[m] = win_lds[4 + 16 * (q0 / 4 + m)];
      }
#pragma unroll
      for (int qq = 0; qq < 16 && q0 + qq < NQ; qq++) {
        const int q = q0 + qq;
        // window dwords (qq, qq+1) of this chunk: even qq from ra, odd qq from rb (same bytes shifted by one dword)
        u32 lo, hi;
        if ((qq & 1) == 0) {
          const uint4 v = ra[qq >> 2];
          lo = (qq & 2) ? v.z : v.x;
          hi = (qq & 2) ? v.w : v.y;
        } else {
          const uint4 v = rb[(qq - 1) >> 2];
          lo = ((qq - 1) & 2) ? v.z : v.x;
          hi = ((qq - 1) & 2) ? v.w : v.y;
        }
        const u64 win = ((u64)hi << 32) | lo;
#if SBM_FAST_PINGPONG
        if (mode == 0) {
          VB[q] = __builtin_amdgcn_mqsad_pk_u16_u8(win, pat, __builtin_bit_cast(u64, VA[q]));
        } else if (mode == 2) {
          VA[q] = __builtin_bit_cast(uint2, __builtin_amdgcn_mqsad_pk_u16_u8(win, pat, VB[q]));
        } else {
          const u64 t = __builtin_amdgcn_mqsad_pk_u16_u8(win, pat, 0ull);
          const uint2 vb = __builtin_bit_cast(uint2, VB[q]), tt = __builtin_bit_cast(uint2, t);
          VA[q].x = vb.x - tt.x;                                // no u16 lane borrows: every partial sum is exact
          VA[q].y = vb.y - tt.y;
        }
#else
        if (mode != 1) {
          asm("v_mqsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(VB[q]) : "v"(win), "v"(pat));
        } else {
          const uint2 tt = __builtin_bit_cast(uint2, __builtin_amdgcn_mqsad_pk_u16_u8(win, pat, 0ull));
          uint2 vb = __builtin_bit_cast(uint2, VB[q]);
          vb.x -= tt.x;                                         // no u16 lane borrows: every partial sum is exact
          vb.y -= tt.y;
          VB[q] = __builtin_bit_cast(u64, vb);
        }
#endif
      }
    }
    __builtin_amdgcn_wave_barrier();
    Vt = mode == 1 ? Vt - tv : Vt + tv;
  };

  // prime: rows ys-W2 .. ys+W2-1, then the entering row of the first output row is fetched ahead
  RowRegs g = fetch(ys - W2);
  for (int yy = ys - W2; yy < ys + W2; yy += 2) {   // 2*W2 rows: an even count, processed in VA->VB->VA pairs
    RowRegs n1 = fetch(yy + 1);
    apply(g, 0);
    RowRegs n2 = fetch(yy + 2);
    apply(n1, 2);
    g = n2;
  }
  // g now holds row ys+W2
  // outputs through buffer stores as well: per-pair descriptors, this lane's byte offset, the row in a scalar register
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(a.disp + (size_t)pair * a.W * a.H, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(a.cost + (size_t)pair * a.W * a.H, 0, 0x7fffffff, 0x00020000);
  const int ocol = 2 * (a.lofs + xc);
  for (int y = ys; y < ye; y++) {
    apply(g, 0);

    // Issue priority: the exchange below is a chain of LDS round trips with a few adds behind each; the phases after it are
    // hundreds of back-to-back vector instructions. With 3-4 wavefronts per SIMD and oldest-first issue, a wavefront in one of
    // those long arithmetic phases keeps the port while its neighbour's adds wait, the neighbour's next LDS reads go out late
    // and the CU's LDS pipe idles. Raising the priority of whoever is in the exchange keeps both pipes fed:
    // KITTI x64 SAD stage 0.963 -> 0.902 ms, 640x480 nd 64 w 21 0.450 -> 0.430, 1080p nd 256 2.59 -> 2.52 (bit-exact, of
    // course). Also raising it during the mqsad phases gains nothing at nd <= 128 and costs 10 % with cooperating
    // wavefronts (1080p 2.70 -> 2.96 ms).
    __builtin_amdgcn_s_setprio(SBM_FAST_PRIO_XCH);
    // ---- horizontal window across lanes ------------------------------------------------------------------
    // S(c + w/2) = sum_k V(c + 3k): every lane publishes its V quads to LDS ([quad][lane], 8-byte entries: both
    // ds_write_b64 and the shifted ds_read_b64 are conflict-free) and reads the NTERM-1 shifted copies back.
    // (ds_bpermute_b32 would do the same without the round trip, but it costs ~24 SIMD-cycles per 4 bytes/lane on
    // gfx950 -- measured with tools/ubench/isa_probe -- which made the kernel crossbar-bound.)
    // Lanes >= NV read beyond lane 63 (unwritten halo entries): their sums are garbage and never stored.
    // (partner entries through an opaque copy of the lane index, as in sad_fast_strip_dma: a load of [lane + k] must not be
    // merged with the previous chunk's load of the same address -- only OTHER lanes write it)
    u32 lx = lane_u;
    asm volatile("" : "+v"(lx));
    u32 S[NR];
    const int par = y & 1;
    const int orow = __builtin_amdgcn_readfirstlane(2 * y * a.W);
    xt[lane] = Vt;
#pragma unroll
    for (int q0 = 0; q0 < NQ; q0 += XCH) {
      // two quads (16 bytes) per LDS entry: ds_write_b128 / ds_read_b128 at lane stride 16 B
#pragma unroll
      for (int qq = 0; qq < XCH; qq += 2) {
        const uint2 v0 = __builtin_bit_cast(uint2, VB[q0 + qq]), v1 = __builtin_bit_cast(uint2, VB[q0 + qq + 1]);
        xq[(qq / 2) * XS + lane] = make_uint4(v0.x, v0.y, v1.x, v1.y);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int qq = 0; qq < XCH; qq += 2) {
        const uint2 v0 = __builtin_bit_cast(uint2, VB[q0 + qq]), v1 = __builtin_bit_cast(uint2, VB[q0 + qq + 1]);
        u32 s0 = v0.x, s1 = v0.y, s2 = v1.x, s3 = v1.y;
#pragma unroll
        for (int k = 1; k < NTERM; k++) {
          const uint4 r = xq[(qq / 2) * XS + lx + KS * k];
          s0 += r.x;               // packed u16 pairs: no carries, every sum stays below 65535
          s1 += r.y;
          s2 += r.z;
          s3 += r.w;
        }
        S[2 * (q0 + qq)] = s0;
        S[2 * (q0 + qq) + 1] = s1;
        S[2 * (q0 + qq) + 2] = s2;
        S[2 * (q0 + qq) + 3] = s3;
      }
      __builtin_amdgcn_wave_barrier();
    }
    if constexpr (!EXACT_ND) {
#pragma unroll
      for (int j = 0; j < NR; j++)
        if (d0 + 2 * j >= a.nd) S[j] = 0xffffffffu;
    }

    __builtin_amdgcn_s_setprio(0);
    // ---- WTA: first index attaining the minimum ------------------------------------------------------------
    // keys carry a group-local index 0..63 (inline constants for v_lshl_or_b32 / v_and_or_b32); the group base is
    // added once per group.  Four independent v_min3_u32 chains per group keep the dependency chains short.
    u32 best = 0xffffffffu;
    // (one tagged variant per instantiation, chosen by the window: a third alternative in the same loop body makes the
    // register allocator spill hundreds of bytes in every instantiation)
    constexpr int TSMAX = WSZ <= 15 ? 2 : 1;
    if (NR >= 16 && TSMAX == 2 && a.pfshift == 2) {
      // Pre-scaled planes (sbm_common.h): every sum is a multiple of 4, so the two low bits of each packed half can carry
      // a register tag. Registers j, j + NR/4, j + NR/2, j + 3NR/4 (tags 0..3 = the top two bits of the buffer index) are
      // reduced with packed 16-bit minima first -- one OR (a full-rate instruction) and one v_pk_min_u16 per register
      // instead of two key builds and a v_min3_u32 -- and only the NR/4 survivors get 32-bit keys. Ties: the smaller
      // (sum, tag, low index bits) triple is the smaller buffer index, as in the plain key scan.
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
    } else if (NR >= 16 && TSMAX == 1 && a.pfshift == 1) {
      // the same with one tag bit (planes hold 2 v + 1: windows whose 4 maxS does not fit 16 bits but 2 maxS does, e.g. 21 x 21
      // at cap 31): registers j and j + NR/2 meet in one v_pk_min_u16, NR/2 survivors get keys
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
    best += (u32)d0;
    const int mpar = par * NWAVES * 64;   // the merge arrays alternate by row parity
    if constexpr (NWAVES > 1) {
      xkey[mpar + wv * 64 + lane] = best;
      __syncthreads();
#pragma unroll
      for (int w = 0; w < NWAVES; w++) best = min(best, xkey[mpar + w * 64 + lane]);
    }
    const int minsad = (int)(best >> 16), mind = (int)(best & 0xffffu);

    // ---- uniqueness (part 1): saturating sum of the deficits max(T - S[d], 0), per 16-bit half --------------
    u32 acc = 0, T = 0;
    if (a.uniq > 0) {
      const int ms = minsad >> a.pfshift;                     // the threshold is defined on the unscaled sum
      const int thresh = ms + (ms * a.uniq / 100);
      T = (u32)min((thresh + 1) << a.pfshift, 65535);
      const u32 T2 = T | (T << 16);
      // independent accumulators of 8 registers each (a single chain is one dependent v_pk_sub -> v_pk_add pair per
      // register with a wait state in between). Every deficit is at most T - minsad <= maxS*uniq/100 + 1; when 8 of them
      // cannot reach 65536 (host check, uniq_plain) the partial sums are plain 32-bit adds of the packed halves -- no
      // carry can cross -- and only the final combine saturates. Saturating adds give min(65535, sum) in any grouping.
      constexpr int NACC = NR >= 32 ? NR / 8 : 4;
      u32 ac[NACC];
#pragma unroll
      for (int k = 0; k < NACC; k++) ac[k] = 0u;
      if (a.uniq_plain) {
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
      acc = ac[0];
    }

    // ---- neighbours S[mind-1], S[mind+1] (mirrored at the ends) via a byte-permute selection tree -------------
    const int in_ = mind > 0 ? mind - 1 : 1;
    const int ip_ = mind < a.nd - 1 ? mind + 1 : a.nd - 2;
    const int ln = min(max(in_ - d0, 0), NDW - 1), lp = min(max(ip_ - d0, 0), NDW - 1);  // local (clamped) indices
    u32 X[NQ];
    // byte selectors built arithmetically from the packed index pair (compares + selects cost several times as much):
    // low half of every selector follows ln, high half lp
    const u32 lnp = (u32)ln | ((u32)lp << 16);
    {
      // bytes (2a, 2a+1) with a = index & 3:  0x0100 + a * 0x0202 per half
      const u32 sel = __umul24(lnp & 0x00030003u, 0x0202u) + 0x01000100u;
#pragma unroll
      for (int q = 0; q < NQ; q++) X[q] = __builtin_amdgcn_perm(S[2 * q + 1], S[2 * q], sel);
    }
    // S is dead from here on: fetch the leaving row now (its latency hides behind the rest of the tree, the
    // merge, the sub-pixel arithmetic and the stores) without raising the register peak of the S-heavy phase
    // (the two fetches of a row are issued at raised priority as well, so that a wavefront's loads do not wait behind a
    // neighbour's arithmetic: KITTI x64 -1 %; with cooperating wavefronts it costs 18 % -- 1080p 2.63 -> 3.10 ms -- hence the
    // condition)
    if constexpr (NWAVES == 1) __builtin_amdgcn_s_setprio(SBM_FAST_PRIO_XCH);
    RowRegs lv = fetch(y - W2);
    if constexpr (NWAVES == 1) __builtin_amdgcn_s_setprio(0);
    {
      int lvl = 2;
#pragma unroll
      for (int n = NQ; n > 1; n >>= 1) {
        // bit `lvl` of each index picks the odd entry (src0 = bytes 4..7) or the even one (src1 = bytes 0..3):
        // selector halves 0x0100 / 0x0504 (low, index ln) and 0x0302 / 0x0706 (high, index lp) = base + bit * 0x0404
        const u32 sel = __umul24((lnp >> lvl) & 0x00010001u, 0x0404u) + 0x03020100u;
#pragma unroll
        for (int m = 0; m < n / 2; m++) X[m] = __builtin_amdgcn_perm(X[2 * m + 1], X[2 * m], sel);
        lvl++;
      }
    }
    int nn = (int)(X[0] & 0xffffu), pp = (int)(X[0] >> 16);
    u32 acc_lo = acc & 0xffffu, acc_hi = acc >> 16;
    bool mine = true;  // does this wavefront finalise this row?
    if constexpr (NWAVES > 1) {
      xacc[mpar + wv * 64 + lane] = make_uint2(acc, X[0]);
      __syncthreads();
      mine = (y % NWAVES) == wv;
      if (mine) {
        acc_lo = acc_hi = 0;
#pragma unroll
        for (int w = 0; w < NWAVES; w++) {
          const u32 aw = xacc[mpar + w * 64 + lane].x;
          acc_lo += aw & 0xffffu;
          acc_hi += aw >> 16;
        }
        nn = (int)(xacc[mpar + (in_ / NDW) * 64 + lane].y & 0xffffu);   // owner wavefront of index in_
        pp = (int)(xacc[mpar + (ip_ / NDW) * 64 + lane].y >> 16);       // owner wavefront of index ip_
      }
    }

    if (mine) {
      int tsum = (int)Vt;
#pragma unroll
      for (int k = 1; k < NTERM; k++) tsum += (int)xt[lx + KS * k];
      bool ok = tsum >= a.tex;
      // ---- uniqueness (part 2): any d outside [mind-1, mind+1] with S[d] <= thresh rejects ---------------------
      if (a.uniq > 0) {
        // expected deficits of the three neighbourhood entries, per 16-bit half (even / odd buffer index)
        const u32 dm = T - (u32)minsad;                                     // >= 1
        const u32 dn = (mind > 0 && (u32)nn < T) ? T - (u32)nn : 0u;         // S[mind-1] exists
        const u32 dp = (mind < a.nd - 1 && (u32)pp < T) ? T - (u32)pp : 0u;  // S[mind+1] exists
        const u32 e_same = dm, e_other = dn + dp;                            // mind's parity half / the other half
        const u32 exp_lo = (mind & 1) ? e_other : e_same, exp_hi = (mind & 1) ? e_same : e_other;
        ok = ok && acc_lo == exp_lo && acc_hi == exp_hi;
      }
      if (produces) {
        int out = a.filtered, cst = 0xffff;   // (a filtered pixel's cost reads 0xffff: the LR kernel relies on it, sbm_post.hip)
        if (ok) {
          const int ad = pp > nn ? pp - nn : nn - pp;
          const int den = pp + nn - 2 * minsad + ad;
          int frac = 0;
          if (den != 0) {
            // den = (p + n - 2 minsad) + |p - n| >= |p - n|, so the quotient is at most 256: one reciprocal estimate
            // (relative error 2^-22) is within 1 of it and one exact remainder settles which way (24-bit products)
            const u32 num = (u32)ad << 8;
            u32 qv = (u32)((float)num * __builtin_amdgcn_rcpf((float)den));
            const int rem = (int)num - (int)__umul24(qv, (u32)den);
            qv = rem < 0 ? qv - 1 : (rem >= den ? qv + 1 : qv);
            frac = pp >= nn ? (int)qv : -(int)qv;          // C division truncates toward zero
          }
          out = ((a.nd - mind - 1 + a.mindisp) * 256 + frac + 15) >> 4;
          cst = minsad >> a.pfshift;
        }
        if (a.cost) __builtin_amdgcn_raw_buffer_store_b16((short)cst, rs_c, ocol, orow, 0);
        __builtin_amdgcn_raw_buffer_store_b16((short)out, rs_d, ocol, orow, 0);
      }
    }

    if (y + 1 < ye) {
      if constexpr (NWAVES == 1) __builtin_amdgcn_s_setprio(SBM_FAST_PRIO_XCH);
      g = fetch(y + 1 + W2);   // next entering row: latency hides behind the leaving row's mqsad + subtractions
      if constexpr (NWAVES == 1) __builtin_amdgcn_s_setprio(0);
      apply(lv, 1);
    }
  }
}

// The strip with LDS-DIRECT STAGING (round 4; every layout of the product build since round 5): the right row pieces go from
// HBM / L2 straight into the expanded LDS layout (buffer_load_dword ... lds: lane i's source bytes land in dword slot i), so a
// row in flight costs no registers (21 VGPRs per row before, two rows in flight at the kernel's pressure peak) and no
// ds_write_b128 (10 per wavefront-row at KITTI size). Two staged-row areas per wavefront: b0 takes the entering row and, once
// that is consumed, serves as exchange level 0; b1 takes the leaving row. Both rows are consumed at the TOP of an output row
// (leave, then enter), so both areas are free for the rest of the row and the next rows' loads have a whole row to arrive; one
// s_waitcnt vmcnt(0) per row. A row's results are stored one iteration late, behind that wait. KITTI x64 (same box,
// alternating, bit-exact): SAD stage 0.910 -> 0.801 ms with the 16-byte form, 0.777 with the dword form; no scratch.
// b0 / b1: the wavefront's LDS areas, `restrict` so that the scoped no-alias information lets LDS traffic of one area run
// while LDS-direct loads into the other are in flight (the compiler makes every LDS access that MAY alias a pending
// LDS-direct load wait for it).
template <int NDW, int NWAVES, int NTERM, int PW, bool EXACT_ND, int CS>
__device__ __forceinline__ void sad_fast_strip_dma(const FastArgs& a, unsigned char* __restrict__ const b0c, unsigned char* __restrict__ const b1c,
                                                   unsigned char* __restrict__ const xl1c,
                                                   u32* __restrict__ const xkey,
                                                   const int cbase, const int segi, const int pair) {
  using L = FastLds<NDW, NTERM, PW, CS>;
  using D = DmaLds<NDW, NWAVES, NTERM, PW, CS>;
  using P = HPlan<NTERM, PW>;
  typedef __attribute__((address_space(3))) void* lds_vptr;
  constexpr int NQ = NDW / 4;           // disparity quads of this wavefront (one u64 accumulator each)
  constexpr int NR = NDW / 2;           // packed pair registers
  constexpr int NSLOT = L::NSLOT;
  constexpr int NIT = NSLOT / 64;
  constexpr int WSZ = PW * NTERM, W2 = WSZ / 2;
  constexpr int KS = L::KS;
  constexpr int NV = 64 - KS * (NTERM - 1);   // lanes that produce an output
  constexpr int XCH = D::XCH, XS = L::XS;

  const int lane = threadIdx.x & 63;
  const int wv = NWAVES > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
  const int d0 = wv * NDW;                              // first buffer index of this wavefront
  const int ndl = EXACT_ND ? NDW : a.nd - d0;           // disparities of this wavefront that exist (a multiple of 16; <= 0: none)
  const int c = cbase + CS * lane;                      // this lane's column (relative to lofs): V covers c..c+2
  const int xc = c + W2;                                // centre column this lane produces
  const bool produces = lane < NV && xc >= a.xc0 && xc < a.xc1;
  const int ys = a.segrow[segi];
  const int ye = a.segrow[segi + 1];
  // wavefront-uniform bases (scalar registers; the per-row step is scalar arithmetic) + this lane's 32-bit offset
  const uint8_t* pl = a.pf_l + (size_t)pair * a.plane + a.padl + a.lofs + cbase;  // left bytes: + CS * lane
  const uint8_t* pr = a.pf_r + (size_t)pair * a.plane + a.padl + a.rofs + cbase + d0;  // right piece: window of buffer index d starts at c + d
  uint2* const xacc = reinterpret_cast<uint2*>(xkey + 2 * NWAVES * 64);   // merge area [2][NWAVES][64] keys, then [2][NWAVES][64] (deficits, neighbours)
  const unsigned lane_u = (unsigned)lane;
  // raw buffer descriptors over the rest of this pair's planes (rows of one pair are < 2^31 bytes apart)
  const __amdgpu_buffer_rsrc_t rs_l = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(pl), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(pr), 0, 0x7fffffff, 0x00020000);
  uint4* const b0 = reinterpret_cast<uint4*>(b0c);
  uint4* const b1 = reinterpret_cast<uint4*>(b1c);
  // exchange areas of the horizontal window, one per level of the plan: [XCH/2 quad pairs][XS lanes] 8 x u16, then [XS]
  // texture column sums. Level 0 is b0 after its staged row has been consumed.
  uint4* const xq0 = b0;
  uint4* const xq1 = reinterpret_cast<uint4*>(xl1c);
  auto xt_of = [](uint4* const xq) { return reinterpret_cast<u32*>(xq + (XCH / 2) * XS); };
  const u32 capw = (u32)a.capb * 0x01010101u;

  // vertical sums, packed 4 x u16 per quad (low dword = indices 4q, 4q+1, high dword = 4q+2, 4q+3), accumulated in place
  // (v_mqsad_pk_u16_u8 with vdst == src2, see sad_fast_strip)
  u64 VB[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) VB[q] = 0ull;
  u32 Vt = 0;  // texture: window-row sum of the 3-column |L - cap|

  // buffer_load_dword ... lds: lane i of load `it` writes its 4 source bytes (row piece bytes 64 it + i .. + 3: a byte-granular
  // source address is fine, tools/ubench/lds_dma.hip) to LDS dword slot 64 it + i -- the 4x-expanded layout without a staging
  // register or a ds_write. The lanes' left patterns follow as 64 dwords at D::PAT_OFS (pat_of()).
  auto pat_of = [](uint4* const buf) { return reinterpret_cast<u32*>(reinterpret_cast<unsigned char*>(buf) + D::PAT_OFS); };
  auto stage = [&](const int y, uint4* const buf) {
    const int rowoff = __builtin_amdgcn_readfirstlane(y * a.pitch);
#pragma unroll
    for (int it = 0; it < NIT; it++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_r, (lds_vptr)(reinterpret_cast<u32*>(buf) + 64 * it), 4, (int)lane_u, rowoff + 64 * it, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_l, (lds_vptr)pat_of(buf), 4, (int)(CS * lane_u), rowoff, 0, 0);
  };
  auto landed = [] {        // everything this wavefront has in flight has landed (LDS-direct loads count in vmcnt)
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  auto reads_done = [] {    // every LDS read of this wavefront has returned (lgkmcnt = 0): an area may be overwritten
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  };
  auto published = [] {     // this wavefront's LDS writes are ordered before its following reads of other lanes' entries
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  // the staged row in `buf` enters (leave == false) or leaves the vertical sums
  auto apply = [&](uint4* const buf, const bool leave) {
    constexpr u32 PMASK = PW == 3 ? 0x00ffffffu : 0x000000ffu;
    const u32 pat = pat_of(buf)[lane] & PMASK;  // remaining bytes = 0 -> masked by mqsad
    const u32 tv = __builtin_amdgcn_sad_u8(pat | (capw & ~PMASK), capw, 0u);
    auto quad = [&](const int q, const u32 lo, const u32 hi) {
      const u64 win = ((u64)hi << 32) | lo;
      if (!leave) {
        asm("v_mqsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(VB[q]) : "v"(win), "v"(pat));
      } else {
        const uint2 tt = __builtin_bit_cast(uint2, __builtin_amdgcn_mqsad_pk_u16_u8(win, pat, 0ull));
        uint2 vb = __builtin_bit_cast(uint2, VB[q]);
        vb.x -= tt.x;                                         // no u16 lane borrows: every partial sum is exact
        vb.y -= tt.y;
        // (opaque: with the entering row's in-place accumulate right behind it the compiler otherwise turns the two
        // subtractions into a 64-bit subtract with a carry chain -- three slow instructions instead of two fast ones)
        asm("" : "+v"(vb.x), "+v"(vb.y));
        VB[q] = __builtin_bit_cast(u64, vb);
      }
    };
    {
      // 4x-expanded staging (dword slot p = bytes p..p+3): the window of quad q is the dword pair (4q, 4q + 4) behind the lane's
      // slot -- one ds_read2_b32 each (lane stride CS dwords: conflict-free for CS = 1 and 3), 16 issued before the first use
      const u32* const win4 = reinterpret_cast<const u32*>(buf) + CS * lane;
      if constexpr (EXACT_ND) {
  #pragma unroll
        for (int q0 = 0; q0 < NQ; q0 += 16) {
          u32 lo[16], hi[16];
  #pragma unroll
          for (int qq = 0; qq < 16 && q0 + qq < NQ; qq++) { lo[qq] = win4[4 * (q0 + qq)]; hi[qq] = win4[4 * (q0 + qq) + 4]; }
  #pragma unroll
          for (int qq = 0; qq < 16 && q0 + qq < NQ; qq++) quad(q0 + qq, lo[qq], hi[qq]);
        }
      } else {
  #pragma unroll
        for (int g = 0; g < NQ / 4; g++) {
          if (16 * g < ndl) {
            u32 lo[4], hi[4];
  #pragma unroll
            for (int k = 0; k < 4; k++) { lo[k] = win4[4 * (4 * g + k)]; hi[k] = win4[4 * (4 * g + k) + 4]; }
  #pragma unroll
            for (int k = 0; k < 4; k++) quad(4 * g + k, lo[k], hi[k]);
          }
        }
      }
    }
    Vt = leave ? Vt - tv : Vt + tv;
  };

  // prime: rows ys-W2 .. ys+W2-1 alternate between the two areas, the next one arriving while one is consumed; the last one
  // staged (into b0) is row ys+W2, the first output row's entering row
  stage(ys - W2, b0);
  for (int i = 0; i < 2 * W2; i += 2) {
    landed();
    stage(ys - W2 + i + 1, b1);
    apply(b0, false);
    landed();
    stage(ys - W2 + i + 2, b0);
    apply(b1, false);
  }
  // outputs through buffer stores: per-pair descriptors, this lane's byte offset, the row in a scalar register. A row's
  // results leave one iteration late, behind the wait at the top of the next row -- that wait covers everything this wavefront
  // has in flight, and stores issued at the end of a row would put their whole latency there.
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(a.disp + (size_t)pair * a.W * a.H, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(a.cost + (size_t)pair * a.W * a.H, 0, 0x7fffffff, 0x00020000);
  // (the previous row's result waits in ONE register -- disparity in the low half, cost in the high half, 0xffff for a filtered
  // pixel: that is also what the cost plane holds there, the LR kernel relies on it -- and
  // the lane's byte offset is rebuilt from CS * lane, which the staging loads keep live anyway, behind an opaque copy: two
  // registers fewer across the row loop, which is what kept <128,2> from fitting its 168 without scratch)
  const int ocol_u = __builtin_amdgcn_readfirstlane(2 * (a.lofs + cbase + W2));
  u32 res_prev = 0xffff0000u;
  auto flush = [&](const int yrow) {
    if (produces && (NWAVES == 1 || (yrow % NWAVES) == wv)) {   // (this wavefront finished that row)
      const int orow_prev = __builtin_amdgcn_readfirstlane(2 * yrow * a.W) + ocol_u;
      u32 l3 = CS * lane_u;
      asm volatile("" : "+v"(l3));
      const int ocol = (int)(2 * l3);
      if (a.cost) __builtin_amdgcn_raw_buffer_store_b16((short)(res_prev >> 16), rs_c, ocol, orow_prev, 0);
      __builtin_amdgcn_raw_buffer_store_b16((short)res_prev, rs_d, ocol, orow_prev, 0);
    }
  };
  for (int y = ys; y < ye; y++) {
    // b0: entering row y+W2; b1 (y > ys): leaving row y-W2-1. Both are consumed here, so both areas are free for the rest of
    // the row and the next rows' loads have a whole row to arrive.
    landed();
    if (y > ys) {
      flush(y - 1);
      apply(b1, true);
    }
    // (unconditional, like the entering row below: a branch here would let the compiler sink the exchange's 4 NR adds
    // below it -- and spill the 4 NR registers they read; the last iteration stages rows nobody consumes)
    reads_done();
    stage(y - W2, b1);
    apply(b0, false);

    // issue priority while this wavefront is in its exchange (a chain of LDS round trips with a few adds behind each): see
    // sad_fast_strip
    __builtin_amdgcn_s_setprio(SBM_FAST_PRIO_XCH);
    // ---- horizontal window across lanes: S(c + w/2) = sum_k V(c + PW k) through LDS, level by level (HPlan) ---------------
    // A chunk of XCH quads at a time: publish V, read the S1 - 1 partners of T, publish T, add the other T's and the remaining V's. Lanes >= NV read entries nobody wrote (halo): their sums are garbage and never stored. The
    // texture column sum takes the same route as a 32-bit column of its own, with the first chunk.
    u32 S[NR];
    unsigned long long tex_ok = 0;
    // Partner entries are addressed through an OPAQUE copy of the lane index. A lane publishes entry [lane] and reads entries
    // [lane + k] that only other lanes write; to the optimiser, which sees one thread, a load of [lane + 3] can never be changed
    // by a store to [lane], and it merged such loads across the chunks of the exchange (a wavefront-scope release fence does
    // not stop it): with the two-level sums every chunk's T partners came back as the first chunk's (caught by the parity tests
    // on the cooperating 64-disparity wavefronts). With an index it cannot relate to `lane` every load may alias every store
    // and stays where it was written; LDS operations of one wavefront execute in order, so nothing else is needed.
    u32 lx = lane_u;
    asm volatile("" : "+v"(lx));
    auto add4 = [](u32 (&acc)[4], const uint4 r) { acc[0] += r.x; acc[1] += r.y; acc[2] += r.z; acc[3] += r.w; };   // packed u16 pairs: no carries, every sum stays below 65535
#pragma unroll
    for (int q0 = 0; q0 < NQ; q0 += XCH) {
      const bool tex_now = q0 == 0;          // (the first chunk always exists: ndl >= 16)
      if (!EXACT_ND && 4 * q0 >= ndl) {      // (a chunk of disparities that do not exist: they never win)
#pragma unroll
        for (int j = 2 * q0; j < 2 * (q0 + XCH); j++) S[j] = 0xffffffffu;
        continue;
      }
      u32 A[XCH / 2][4];
      u32 tA = Vt;
      // level 0: two quads (16 bytes) per LDS entry: ds_write_b128 / ds_read_b128 at lane stride 16 B
#pragma unroll
      for (int e = 0; e < XCH / 2; e++) {
        const uint2 v0 = __builtin_bit_cast(uint2, VB[q0 + 2 * e]), v1 = __builtin_bit_cast(uint2, VB[q0 + 2 * e + 1]);
        A[e][0] = v0.x; A[e][1] = v0.y; A[e][2] = v1.x; A[e][3] = v1.y;
        xq0[e * XS + lane] = make_uint4(v0.x, v0.y, v1.x, v1.y);
      }
      if (tex_now) xt_of(xq0)[lane] = Vt;
      published();
      if constexpr (P::S1 > 1) {            // T = S1 consecutive V
#pragma unroll
        for (int e = 0; e < XCH / 2; e++)
#pragma unroll
          for (int k = 1; k < P::S1; k++) add4(A[e], xq0[e * XS + lx + KS * k]);
        if (tex_now)
#pragma unroll
          for (int k = 1; k < P::S1; k++) tA += xt_of(xq0)[lx + KS * k];
        if constexpr (P::PUB1) {
#pragma unroll
          for (int e = 0; e < XCH / 2; e++) xq1[e * XS + lane] = make_uint4(A[e][0], A[e][1], A[e][2], A[e][3]);
          if (tex_now) xt_of(xq1)[lane] = tA;
          published();
        }
      }
      // the window: NTT T's, then NVV V's, left to right; the first term is this lane's own (in A)
      {
        constexpr int OV = P::NTT * P::S1;   // first V behind the T's
#pragma unroll
        for (int e = 0; e < XCH / 2; e++) {
#pragma unroll
          for (int t = 1; t < P::NTT; t++) add4(A[e], xq1[e * XS + lx + KS * P::S1 * t]);
#pragma unroll
          for (int v = (P::NTT > 0 ? 0 : 1); v < P::NVV; v++) add4(A[e], xq0[e * XS + lx + KS * (OV + v)]);
        }
        if (tex_now) {
#pragma unroll
          for (int t = 1; t < P::NTT; t++) tA += xt_of(xq1)[lx + KS * P::S1 * t];
#pragma unroll
          for (int v = (P::NTT > 0 ? 0 : 1); v < P::NVV; v++) tA += xt_of(xq0)[lx + KS * (OV + v)];
          // the verdict crosses the winner search as a wavefront-uniform mask, not in a vector register
          tex_ok = __ballot((int)tA >= a.tex);
        }
      }
#pragma unroll
      for (int e = 0; e < XCH / 2; e++)
#pragma unroll
        for (int i = 0; i < 4; i++) S[2 * (q0 + 2 * e) + i] = A[e][i];
      __builtin_amdgcn_wave_barrier();
    }
    // level 0 of the exchange is about to receive the next entering row
    reads_done();
    stage(min(y + 1 + W2, a.H - 1), b0);
    __builtin_amdgcn_s_setprio(0);

    // ---- WTA: first index attaining the minimum (see sad_fast_strip for the three variants) --------------------------------
    u32 best = 0xffffffffu;
    constexpr int TSMAX = WSZ <= 15 ? 2 : 1;
    if (NR >= 16 && TSMAX == 2 && a.pfshift == 2) {
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
    } else if (NR >= 16 && TSMAX == 1 && a.pfshift == 1) {
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
    best += (u32)d0;
    const int par = y & 1, mpar = par * NWAVES * 64;   // the merge arrays alternate by row parity
    if constexpr (NWAVES > 1) {
      xkey[mpar + wv * 64 + lane] = best;
      __syncthreads();
#pragma unroll
      for (int w = 0; w < NWAVES; w++) best = min(best, xkey[mpar + w * 64 + lane]);
    }
    const int minsad = (int)(best >> 16), mind = (int)(best & 0xffffu);

    // ---- uniqueness (part 1): saturating sum of the deficits max(T - S[d], 0), per 16-bit half --------------
    u32 acc = 0, T = 0;
    if (a.uniq > 0) {
      const int ms = minsad >> a.pfshift;                     // the threshold is defined on the unscaled sum
      const int thresh = ms + (ms * a.uniq / 100);
      T = (u32)min((thresh + 1) << a.pfshift, 65535);
      const u32 T2 = T | (T << 16);
      constexpr int NACC = NR >= 32 ? NR / 8 : 4;
      u32 ac[NACC];
#pragma unroll
      for (int k = 0; k < NACC; k++) ac[k] = 0u;
      if (a.uniq_plain) {
#pragma unroll
        for (int j = 0; j < NR; j++) ac[j % NACC] += pk_sub_sat(T2, S[j]);
      } else {
        // (an opaque copy of the threshold: otherwise the compiler hoists the NR subtractions both paths share above
        // the branch and keeps all of them live at once)
        u32 T2s = T2;
        asm("" : "+v"(T2s));
#pragma unroll
        for (int j = 0; j < NR; j++) ac[j % NACC] = pk_add_sat(ac[j % NACC], pk_sub_sat(T2s, S[j]));
      }
#pragma unroll
      for (int n = NACC; n > 1; n >>= 1)
#pragma unroll
        for (int k = 0; k < n / 2; k++) ac[k] = pk_add_sat(ac[k], ac[k + n / 2]);
      acc = ac[0];
    }

    // ---- neighbours S[mind-1], S[mind+1] (mirrored at the ends) ----------------------------------------------------------
    const int in_ = mind > 0 ? mind - 1 : 1;
    const int ip_ = mind < a.nd - 1 ? mind + 1 : a.nd - 2;
    const int ln = min(max(in_ - d0, 0), NDW - 1), lp = min(max(ip_ - d0, 0), NDW - 1);  // local (clamped) indices
    u32 X0;
    {
      // via a byte-permute selection tree
      u32 X[NQ];
      const u32 lnp = (u32)ln | ((u32)lp << 16);
      {
        // bytes (2a, 2a+1) with a = index & 3:  0x0100 + a * 0x0202 per half
        const u32 sel = __umul24(lnp & 0x00030003u, 0x0202u) + 0x01000100u;
#pragma unroll
        for (int q = 0; q < NQ; q++) X[q] = __builtin_amdgcn_perm(S[2 * q + 1], S[2 * q], sel);
      }
      {
        int lvl = 2;
#pragma unroll
        for (int n = NQ; n > 1; n >>= 1) {
          const u32 sel = __umul24((lnp >> lvl) & 0x00010001u, 0x0404u) + 0x03020100u;
#pragma unroll
          for (int m = 0; m < n / 2; m++) X[m] = __builtin_amdgcn_perm(X[2 * m + 1], X[2 * m], sel);
          lvl++;
        }
      }
      X0 = X[0];
    }
    int nn = (int)(X0 & 0xffffu), pp = (int)(X0 >> 16);
    u32 acc_lo = acc & 0xffffu, acc_hi = acc >> 16;
    bool mine = true;  // does this wavefront finalise this row?
    if constexpr (NWAVES > 1) {
      xacc[mpar + wv * 64 + lane] = make_uint2(acc, X0);
      __syncthreads();
      mine = (y % NWAVES) == wv;
      if (mine) {
        acc_lo = acc_hi = 0;
#pragma unroll
        for (int w = 0; w < NWAVES; w++) {
          const u32 aw = xacc[mpar + w * 64 + lane].x;
          acc_lo += aw & 0xffffu;
          acc_hi += aw >> 16;
        }
        nn = (int)(xacc[mpar + (in_ / NDW) * 64 + lane].y & 0xffffu);   // owner wavefront of index in_
        pp = (int)(xacc[mpar + (ip_ / NDW) * 64 + lane].y >> 16);       // owner wavefront of index ip_
      }
    }

    bool ok = __builtin_amdgcn_inverse_ballot_w64(tex_ok);
    // ---- uniqueness (part 2): any d outside [mind-1, mind+1] with S[d] <= thresh rejects ---------------------
    if (a.uniq > 0) {
      const u32 dm = T - (u32)minsad;                                     // >= 1
      const u32 dn = (mind > 0 && (u32)nn < T) ? T - (u32)nn : 0u;         // S[mind-1] exists
      const u32 dp = (mind < a.nd - 1 && (u32)pp < T) ? T - (u32)pp : 0u;  // S[mind+1] exists
      const u32 e_same = dm, e_other = dn + dp;                            // mind's parity half / the other half
      const u32 exp_lo = (mind & 1) ? e_other : e_same, exp_hi = (mind & 1) ? e_same : e_other;
      ok = ok && acc_lo == exp_lo && acc_hi == exp_hi;
    }
    if (mine && produces) {
      int out = a.filtered, cst = -1;
      if (ok) {
        const int ad = pp > nn ? pp - nn : nn - pp;
        const int den = pp + nn - 2 * minsad + ad;
        int frac = 0;
        if (den != 0) {
          // the quotient is at most 256: one reciprocal estimate is within 1 of it, one exact remainder settles which way
          const u32 num = (u32)ad << 8;
          u32 qv = (u32)((float)num * __builtin_amdgcn_rcpf((float)den));
          const int rem = (int)num - (int)__umul24(qv, (u32)den);
          qv = rem < 0 ? qv - 1 : (rem >= den ? qv + 1 : qv);
          frac = pp >= nn ? (int)qv : -(int)qv;          // C division truncates toward zero
        }
        out = ((a.nd - mind - 1 + a.mindisp) * 256 + frac + 15) >> 4;
        if (a.cost) cst = minsad >> a.pfshift;
      }
      res_prev = ((u32)out & 0xffffu) | ((u32)cst << 16);   // (a cost is at most 65534; filtered = 0xffff)
    }
  }
  flush(ye - 1);   // the segment's last row
}

// DUAL (windows that are multiples of 3): strips [0, strips3) are column-stride-3 strips in triples, the rest plain ones
template <int NDW, int NWAVES, int NTERM, int PW, bool EXACT_ND, bool DUAL>
__global__ void __launch_bounds__(64 * NWAVES) SBM_FAST_WAVES_PER_EU sad_fast_kernel(FastArgs a) {
  // XCD-aware decode of the 1-D workgroup id: consecutive ids go round-robin over the 8 XCDs (each with its own
  // 4 MiB L2), so give XCD k the pairs k, k+8, ...: all strips and row segments of a pair then share one L2.
  // (Placement only affects speed; any mapping is correct.)
  // Row segments are the slowest-varying index and get shorter towards the end of the grid: every segment pays w-1
  // priming rows, so few long segments keep that overhead low while the short last ones keep the tail of the launch
  // (CUs idling while the last workgroups finish) short.
  const int bpp = a.strips;                             // workgroups per pair and segment
  int strip, segi, pair;
  {
    // The grid starts with the border jobs (a.bord workgroups for each of their a.nbseg row segments, sbm_sad_border_wave.h):
    // they are long serial chains, so they are dispatched before any strip and finish under the strips instead of behind them.
    if constexpr (NDW * NWAVES <= 256) {   // (a border wavefront holds a disparity quad per lane: up to 256; launch_t never asks beyond)
    if ((int)blockIdx.x < a.bord * a.nbseg) {
      // wavefront wv of border workgroup b takes border wavefront b * NWAVES + wv of its segment; no barriers in there
      using BL = BorderLds<(PW * NTERM) / 2, NDW * NWAVES>;
      const int bseg = blockIdx.x / a.bord, b = blockIdx.x - bseg * a.bord;
      const int wv = NWAVES > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
      const int wi = b * NWAVES + wv;
      if (wi < a.bnw) sad_border_wave<(PW * NTERM) / 2, NDW * NWAVES>(a, reinterpret_cast<unsigned char*>(fast_lds) + wv * BL::BYTES, bseg, wi);
      return;
    }
    }
    const int sid = blockIdx.x - a.bord * a.nbseg;
    const int per_seg = a.strips * a.npairs;
    segi = sid / per_seg;
    const int b = sid - segi * per_seg;
    const int full = (a.npairs / 8) * 8 * bpp;          // ids covered by complete groups of 8 pairs
    int p, inner;
    if (b < full) {
      const int xcd = b & 7, k = b >> 3;
      p = (k / bpp) * 8 + xcd;
      inner = k % bpp;
    } else {
      const int r = b - full;
      p = (a.npairs / 8) * 8 + r / bpp;
      inner = r % bpp;
    }
    pair = p;
    strip = inner;
  }
  // LDS of the workgroup: per wavefront one area of WSLOT slots (staged row / exchange) + the merge area of the workgroup;
  // LDS-direct strips: per wavefront the areas of DmaLds (two staged rows, further exchange levels), then the merge area
  const int wvk = NWAVES > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
  unsigned char* const ldsb = reinterpret_cast<unsigned char*>(fast_lds);
  auto dma_strip = [&](auto cs_tag, const int cb) {
    constexpr int CSV = decltype(cs_tag)::value;
    using D = DmaLds<NDW, NWAVES, NTERM, PW, CSV>;
    unsigned char* const wb = ldsb + wvk * D::WAVE_B;
    unsigned char* const xl = wb + 2 * D::AREA_B;
    sad_fast_strip_dma<NDW, NWAVES, NTERM, PW, EXACT_ND, CSV>(a, wb, wb + D::AREA_B, xl,
                                                            reinterpret_cast<u32*>(ldsb + NWAVES * D::WAVE_B), cb, segi, pair);
  };
  if constexpr (DUAL) {
    constexpr int NV3 = 64 - (NTERM - 1), NV1 = 64 - PW * (NTERM - 1);
    if (strip < a.strips3) {
      const int t = strip / 3;
      if constexpr (fast_dma(NDW, NWAVES)) dma_strip(std::integral_constant<int, 3>{}, t * (3 * NV3) + (strip - 3 * t));
      else sad_fast_strip<NDW, NWAVES, NTERM, PW, EXACT_ND, 3>(a, t * (3 * NV3) + (strip - 3 * t), segi, pair);
    } else {
      const int cb1 = (a.strips3 / 3) * (3 * NV3) + (strip - a.strips3) * NV1;
      if constexpr (fast_dma(NDW, NWAVES)) dma_strip(std::integral_constant<int, 1>{}, cb1);
      else sad_fast_strip<NDW, NWAVES, NTERM, PW, EXACT_ND, 1>(a, cb1, segi, pair);
    }
  } else {
    constexpr int NV1 = 64 - PW * (NTERM - 1);
    if constexpr (fast_dma(NDW, NWAVES)) dma_strip(std::integral_constant<int, 1>{}, strip * NV1);
    else sad_fast_strip<NDW, NWAVES, NTERM, PW, EXACT_ND, 1>(a, strip * NV1, segi, pair);
  }
}

#if defined(SBM_DEV_PROF) && !SBM_FAST_PINGPONG && SBM_FAST_TU == 0
// profiling builds: read and clear the border row-loop cycle counters (tools/exp/r04_bwprof.py)
extern "C" int sbm_dev_bw_prof(unsigned long long* out8) {
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_bw_prof), sizeof(z)) != hipSuccess) return -2;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_bw_prof), z, sizeof(z)) == hipSuccess ? 0 : -3;
}
#endif

#if !SBM_FAST_PINGPONG && SBM_FAST_TU == 0
hipError_t launch_sad_fast_pp(const uint8_t* pf_l, const uint8_t* pf_r, int16_t* disp, int32_t* cost, const Geom& g,
                              int* xa, int* xb, bool border, hipStream_t s);

// Device check behind the in-place accumulate: v_mqsad_pk_u16_u8 with vdst == src2 against the compiler's
// non-aliased form, pseudo-random operands, single instructions and dependent chains (tools/ubench/mqsad_alias.hip
// is the long version); three short launches, once per device and process.
__global__ void __launch_bounds__(256) mqsad_inplace_selftest_kernel(unsigned* bad) {
  unsigned long long st = (unsigned long long)(blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull + 88172645463325252ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
  unsigned nbad = 0;
  for (int it = 0; it < 64; it++) {
    unsigned long long accA[4], accR[4], win[4];
    const unsigned pat = (unsigned)rnd() & ((it & 1) ? 0x00ffffffu : 0x000000ffu);
#pragma unroll
    for (int q = 0; q < 4; q++) { accA[q] = accR[q] = rnd() & 0x3fff3fff3fff3fffull; win[q] = rnd(); }
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        accR[q] = __builtin_amdgcn_mqsad_pk_u16_u8(win[(q + r) & 3], pat, accR[q]);
        asm volatile("v_mqsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(accA[q]) : "v"(win[(q + r) & 3]), "v"(pat));
      }
#pragma unroll
    for (int q = 0; q < 4; q++) nbad += accA[q] != accR[q];
  }
  if (nbad) atomicAdd(bad, nbad);
  atomicAdd(bad + 1, 1u);   // "this workgroup's thread ran": a launch that never executed must not read as a pass
}

// Runs once per device and process (std::call_once: handles may be created from several threads). The check is launched at
// three occupancies -- 1, 4 and 8 wavefronts per SIMD chip-wide -- and passes only if every thread of every launch reported
// in and none saw a difference; a failed launch, copy or synchronisation counts as "not ok" (the ping-pong build then runs).
bool mqsad_inplace_ok(hipStream_t s) {
  static std::once_flag once[64];
  static bool ok[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  std::call_once(once[dev], [&] {
    ok[dev] = false;
    if (env_switch("SBM_FAST_INPLACE", 1) == 0) return;
    unsigned* d = nullptr;
    if (hipMalloc(&d, 8) != hipSuccess) return;
    bool pass = true;
    const unsigned grids[3] = {256u, 1024u, 2048u};   // x 256 threads = 4 wavefronts per workgroup
    for (int k = 0; k < 3 && pass; k++) {
      unsigned h[2] = {1u, 0u};
      pass = hipMemsetAsync(d, 0, 8, s) == hipSuccess;
      if (pass) {
        hipLaunchKernelGGL(mqsad_inplace_selftest_kernel, dim3(grids[k]), dim3(256), 0, s, d);
        pass = hipGetLastError() == hipSuccess && hipMemcpyAsync(h, d, 8, hipMemcpyDeviceToHost, s) == hipSuccess &&
               hipStreamSynchronize(s) == hipSuccess && h[0] == 0u && h[1] == grids[k] * 256u;
      }
    }
    (void)hipFree(d);
    ok[dev] = pass;
  });
  return ok[dev];
}

// Pre-scaled planes for the tagged winner search: (value << sh) + 1 must fit a byte and (maxS << sh) + tag a packed half;
// the uniqueness envelope is the one of sad_fast_supported() on the scaled sums. SBM_FAST_PFSHIFT=0 turns it off, =1 limits
// it to one tag bit.
int sad_fast_pfshift(const Geom& g) {
  static const int env = env_switch("SBM_FAST_PFSHIFT", 2);
  if (env <= 0 || !sad_fast_supported(g)) return 0;
  const long maxs = (long)g.wsz * g.wsz * 2 * g.cap;
  // the kernels hold one tagged variant each: two tag bits (4 v + 1) for windows up to 15, one (2 v + 1) above
  const int sh = std::min(env, g.wsz <= 15 ? 2 : 1);
  const long f = 1L << sh;
  if (f * 2 * g.cap + 1 > 255) return 0;
  if (f * maxs + (f - 1) > 65535) return 0;
  if (2 * (f * (maxs * g.uniq / 100 + 1)) >= 65535) return 0;
  if ((long)g.tex * f > 0x3fffffff) return 0;
  return sh == (g.wsz <= 15 ? 2 : 1) ? sh : 0;
}

bool sad_fast_borders_in_launch(const Geom& g) { return g.nd <= 256; }

bool sad_fast_supported(const Geom& g) {
  if (g.wsz < 5 || g.wsz > 31) return false;   // every odd window 5..31: multiples of 3 with 3-column sums, the rest 1-column
  // (mqsad_inplace_ok: cached per device; sbm_compute_device has primed it on the handle's stream before it asks here)
  if (g.wsz > 27 && !mqsad_inplace_ok(nullptr)) return false;   // 29 and 31 are not in the two-accumulator fallback build
  if (g.nd > kFastNdMax) return false;
  // beyond 256 disparities: three / four cooperating 128-disparity wavefronts (not in the two-accumulator fallback build);
  // their border columns come from the sliding-sum kernel (sbm_sad_wide.hip) in launches of their own
  if (g.nd > 256 && !mqsad_inplace_ok(nullptr)) return false;
  const long maxs = (long)g.wsz * g.wsz * 2 * g.cap;
  if (maxs > 65534) return false;
  if (2 * (maxs * g.uniq / 100 + 1) >= 65535) return false;
  if (g.row0 < g.w2 || g.row1 > g.H - g.w2 || g.row1 <= g.row0) return false;
  if ((long)g.plane * 4 >= (1L << 31)) return false;   // a border wavefront reaches up to 4 pairs through 32-bit offsets
  const int xhi = std::min(g.W - g.lofs - 1, g.W - g.rofs - g.nd);  // last unclamped window column
  if (xhi - g.w2 + 1 <= g.w2) return false;
  return true;
}
#endif  // !SBM_FAST_PINGPONG && SBM_FAST_TU == 0

// Tuning constants of the border jobs' row segments (chip- and kernel-version specific; they only move time, never results):
// the launch's expected duration is priced at kBorderModelRate pixel-disparities per second (the interior kernel's rate when
// the segments were tuned: 3.6e12, profiles/r04_border_bseg.txt), a border row at kBorderRowUs + kBorderColUs per output
// column (tools/exp/r04_bwprof.py), and a chain may last kBorderChainShare of the launch.
constexpr double kBorderModelRate = 3.6e12, kBorderRowUs = 2.0, kBorderColUs = 0.25, kBorderChainShare = 0.205;

template <int NDW, int NWAVES, int NTERM, int PW>
static hipError_t launch_t(FastArgs a, bool border, hipStream_t s) {
  constexpr bool DUAL = PW == 3;
  constexpr int WSLOT1 = FastLds<NDW, NTERM, PW, 1>::WSLOT, WSLOT3 = FastLds<NDW, NTERM, PW, DUAL ? 3 : 1>::WSLOT;
  constexpr int WSLOTM = WSLOT1 > WSLOT3 ? WSLOT1 : WSLOT3;
  // per wavefront the staged-row / exchange area; then the workgroup's merge area, or -- single-wavefront workgroups with
  // LDS-direct staging -- the second staged-row area
  size_t lds = (size_t)NWAVES * WSLOTM * 16 + (NWAVES > 1 ? (size_t)2 * NWAVES * 64 * (4 + 8) : 0);
  if (fast_dma(NDW, NWAVES)) {  // per wavefront the areas of DmaLds, then the merge area
    constexpr int WB1 = DmaLds<NDW, NWAVES, NTERM, PW, 1>::WAVE_B, WB3 = DmaLds<NDW, NWAVES, NTERM, PW, DUAL ? 3 : 1>::WAVE_B;
    lds = (size_t)NWAVES * (WB1 > WB3 ? WB1 : WB3) + (NWAVES > 1 ? (size_t)2 * NWAVES * 64 * (4 + 8) : 0);
  }
  a.bord = a.bnw = 0;
  a.bseg = a.row1 - a.row0;
  a.nbseg = 0;
  if constexpr (NDW * NWAVES <= 256) if (border) {
    // border wavefronts per segment: 2 sides x ceil(n / JW) groups of JW consecutive pairs, NWAVES of them per workgroup
    using BL = BorderLds<(PW * NTERM) / 2, NDW * NWAVES>;
    a.bnw = 2 * ((a.npairs + BL::JW - 1) / BL::JW);
    a.bord = (a.bnw + NWAVES - 1) / NWAVES;
    lds = std::max(lds, (size_t)NWAVES * BL::BYTES);
    // A border wavefront is a serial chain of rows (~2 us + 0.25 us per output column and row, a quarter of that for each of
    // its w-1 priming rows -- measured alone on the chip, tools/exp/r04_bwprof.py); it must end well inside the launch, so the
    // border jobs get their own, finer row segments: a chain of about a fifth of the launch's expected duration (640x480 nd 64
    // w 21 x 64 pairs, where the border columns weigh most: 10 rows per segment 0.520 ms per step, 13 rows 0.537, 6 rows 0.533 --
    // profiles/r04_border_bseg.txt; KITTI x 64 is flat between 32 and 96 rows).
    const int rows = a.row1 - a.row0, wsz = PW * NTERM;
    const double t_kernel_us = (double)a.npairs * a.W * rows * a.nd / kBorderModelRate * 1e6;
    const double t_row_us = kBorderRowUs + kBorderColUs * (wsz / 2);
    int bseg = (int)(kBorderChainShare * t_kernel_us / t_row_us - 0.25 * (wsz - 1));
    bseg = SBM_TUNE("SBM_DEV_BSEG", bseg);
    bseg = std::max(4, std::min(bseg, rows));
    a.nbseg = (rows + bseg - 1) / bseg;
    a.bseg = (rows + a.nbseg - 1) / a.nbseg;
    a.nbseg = (rows + a.bseg - 1) / a.bseg;
  }
  dim3 grid((unsigned)(a.bord * a.nbseg + a.strips * a.npairs * a.nseg));
  if (SBM_TUNE("SBM_DEV_PRINT", 0))   // development builds: the launch geometry
    fprintf(stderr, "[sbm] <%d,%d,%d,%d> strips %d (cs3 %d) nseg %d pairs %d bord %d x %d grid %u lds %zu\n", NDW, NWAVES, NTERM, PW, a.strips, a.strips3,
            a.nseg, a.npairs, a.bord, a.nbseg, grid.x, lds);
  // development builds: time the border wavefronts alone (results are wrong by construction)
  if (SBM_TUNE("SBM_DEV_BORDER_ONLY", 0)) grid.x = (unsigned)(a.bord * a.nbseg);
  // (the fallback build only carries the masked-count kernels: they are right for every count up to NDW * NWAVES)
  // ... and <64,4> only runs one-pair calls beyond 192 disparities: its masked kernel serves 256 as well
  // (three and four 128-disparity wavefronts, 257 .. 512 disparities: exact kernels for 384 and 512 -- 5-7 % over the masked ones,
  // profiles/r05_exact512.txt; SBM_FAST_EXACT512=0 drops them: 24 kernels, 0.6 MB, ~15 s of build)
  constexpr bool HAS_EXACT = !SBM_FAST_PINGPONG && !(NDW == 64 && NWAVES == 4) && !(NDW == 128 && NWAVES >= 3 && !SBM_FAST_EXACT512);
  const bool exact = HAS_EXACT && a.nd == NDW * NWAVES;
  snprintf(g_sad_kernel_name, sizeof(g_sad_kernel_name), "%s<%d,%d,%d,%d,%s,%s> pfshift=%d", SBM_FAST_PINGPONG ? "sad_fast_pp_kernel" : "sad_fast_kernel",
           NDW, NWAVES, NTERM, PW, exact ? "true" : "false", DUAL ? "true" : "false", a.pfshift);
  if constexpr (HAS_EXACT) {
    if (exact) {
      hipLaunchKernelGGL((sad_fast_kernel<NDW, NWAVES, NTERM, PW, true, DUAL>), grid, dim3(64 * NWAVES), lds, s, a);
      return hipGetLastError();
    }
  }
    hipLaunchKernelGGL((sad_fast_kernel<NDW, NWAVES, NTERM, PW, false, DUAL>), grid, dim3(64 * NWAVES), lds, s, a);
  return hipGetLastError();
}

// mode 1 (default): 128 disparities per wavefront wherever nd > 64 -- one wavefront holds a pixel's whole disparity range
// at nd <= 128 (no barriers, no merge, every per-row fixed cost paid once; 168 VGPRs = 3 wavefronts per SIMD with the
// in-place accumulate), two cooperate up to nd 256.  mode 0 (SBM_FAST_MODE=0, and the ping-pong build): 64 disparities
// per wavefront, nd/64 cooperating wavefronts -- the round-1/2 layout, kept for A/B measurements and as the fallback.
template <int NTERM, int PW>
static hipError_t launch_nd(const FastArgs& a, bool border, int mode, bool split, hipStream_t s) {
#if SBM_FAST_PINGPONG   // the fallback nobody should ever run: correct and reasonably fast, but a quarter of the kernels
  if (a.nd <= 64) return launch_t<64, 1, NTERM, PW>(a, border, s);
  if (a.nd <= 128) return launch_t<64, 2, NTERM, PW>(a, border, s);
  if (a.nd <= 192) return launch_t<64, 3, NTERM, PW>(a, border, s);
  return launch_t<64, 4, NTERM, PW>(a, border, s);
#else
  if (a.nd > 384) return launch_t<128, 4, NTERM, PW>(a, border, s);   // (whatever SBM_FAST_MODE says: the only layouts up there)
  if (a.nd > 256) return launch_t<128, 3, NTERM, PW>(a, border, s);
  if (a.nd <= 32) return launch_t<32, 1, NTERM, PW>(a, border, s);
  if (a.nd == 48 && SBM_TUNE("SBM_DEV_ND48", 0)) return launch_t<32, 2, NTERM, PW>(a, border, s);   // (rounds 1-3: two 32-disparity wavefronts)
  // one-pair calls: too few workgroups to fill the chip, so split the disparities over two wavefronts (half the serial work
  // per row; SBM_FAST_SPLIT=0 disables)
  if (a.nd <= 64 && a.nd > 32 && split) return launch_t<32, 2, NTERM, PW>(a, border, s);
  if (a.nd <= 64) return launch_t<64, 1, NTERM, PW>(a, border, s);
  if (mode >= 1 && !split && a.nd <= 128) return launch_t<128, 1, NTERM, PW>(a, border, s);
  // Beyond 128 disparities: two cooperating 128-disparity wavefronts with LDS-direct staging (round 4: 1080p nd 256 2.58 ->
  // 2.21 ms per step, 2160p 2.85 -> 2.43; in round 3, register-staged, this layout starved the border kernel) -- except at
  // exactly 192, where three 64-disparity wavefronts have no masked disparities to carry (1080p nd 192: 2.09 against 2.51 ms;
  // nd 160: 2.63 against 5.38 for the masked <64,3>; profiles/r04_dma_nd.txt). SBM_FAST_MODE=1: the 64-disparity cooperating
  // wavefronts of rounds 1-3 everywhere (kept for small launches and as the A/B reference).
  if (mode >= 2 && !split && a.nd > 128 && a.nd != 192) return launch_t<128, 2, NTERM, PW>(a, border, s);
  if (a.nd <= 128) return launch_t<64, 2, NTERM, PW>(a, border, s);
  if (a.nd <= 192) return launch_t<64, 3, NTERM, PW>(a, border, s);
  return launch_t<64, 4, NTERM, PW>(a, border, s);
#endif
}

// the windows that are not multiples of 3 (1-column sums): 5..13, 17..25 and 29 / 31, three translation units in the product build
#if SBM_FAST_TU == 3
hipError_t launch_sad_fast_pw3(const FastArgs& a, int wsz, bool border, int mode, bool split, hipStream_t s) {
  switch (wsz) {
    case 29: return launch_nd<29, 1>(a, border, mode, split, s);
    case 31: return launch_nd<31, 1>(a, border, mode, split, s);
    default: return hipErrorInvalidValue;
  }
}
#elif SBM_FAST_TU == 2
hipError_t launch_sad_fast_pw3(const FastArgs& a, int wsz, bool border, int mode, bool split, hipStream_t s);   // sbm_sad_fast_pw3.hip
#endif
#if SBM_FAST_TU == 2 || SBM_FAST_PINGPONG || defined(SBM_DEV_FEW19)
#if SBM_FAST_TU == 2
hipError_t launch_sad_fast_pw2(const FastArgs& a, int wsz, bool border, int mode, bool split, hipStream_t s) {
#else
static hipError_t launch_sad_fast_pw2(const FastArgs& a, int wsz, bool border, int mode, bool split, hipStream_t s) {
#endif
  switch (wsz) {
#if !defined(SBM_DEV_FEW19)   // (development builds, tools/exp: windows 19 and 23 only)
    case 17: return launch_nd<17, 1>(a, border, mode, split, s);
    case 25: return launch_nd<25, 1>(a, border, mode, split, s);
#endif
    case 19: return launch_nd<19, 1>(a, border, mode, split, s);
    case 23: return launch_nd<23, 1>(a, border, mode, split, s);
#if SBM_FAST_TU == 2
    default: return launch_sad_fast_pw3(a, wsz, border, mode, split, s);
#else
    default: return hipErrorInvalidValue;
#endif
  }
}
#else
hipError_t launch_sad_fast_pw2(const FastArgs& a, int wsz, bool border, int mode, bool split, hipStream_t s);   // sbm_sad_fast_pw2.hip
#endif
#if SBM_FAST_TU == 1 || SBM_FAST_PINGPONG || defined(SBM_DEV_FEW19)
#if SBM_FAST_TU == 1
hipError_t launch_sad_fast_pw1(const FastArgs& a, int wsz, bool border, int mode, bool split, hipStream_t s) {
#else
static hipError_t launch_sad_fast_pw1(const FastArgs& a, int wsz, bool border, int mode, bool split, hipStream_t s) {
#endif
  switch (wsz) {
#if !defined(SBM_DEV_FEW19)
    case 5: return launch_nd<5, 1>(a, border, mode, split, s);
    case 7: return launch_nd<7, 1>(a, border, mode, split, s);
    case 11: return launch_nd<11, 1>(a, border, mode, split, s);
    case 13: return launch_nd<13, 1>(a, border, mode, split, s);
#endif
    default: return launch_sad_fast_pw2(a, wsz, border, mode, split, s);
  }
}
#elif SBM_FAST_TU == 0
hipError_t launch_sad_fast_pw1(const FastArgs& a, int wsz, bool border, int mode, bool split, hipStream_t s);   // sbm_sad_fast_pw1.hip
#endif

#if SBM_FAST_TU == 0
hipError_t launch_sad_fast(const uint8_t* pf_l, const uint8_t* pf_r, int16_t* disp, int32_t* cost, const Geom& g,
                           int* xa, int* xb, bool border, hipStream_t s) {
  *xa = *xb = 0;
  if (!sad_fast_supported(g)) return hipSuccess;
#if !SBM_FAST_PINGPONG
  if (!mqsad_inplace_ok(s)) return launch_sad_fast_pp(pf_l, pf_r, disp, cost, g, xa, xb, border, s);
#endif
  static const int mode = env_switch("SBM_FAST_MODE", SBM_FAST_PINGPONG ? 0 : 2);
  FastArgs a;
  a.pf_l = pf_l; a.pf_r = pf_r; a.disp = disp; a.cost = g.want_cost ? reinterpret_cast<uint16_t*>(cost) : nullptr;
  a.W = g.W; a.H = g.H; a.pitch = g.pitch; a.padl = g.padl; a.plane = g.plane;
  a.nd = g.nd; a.mindisp = g.mindisp; a.lofs = g.lofs; a.rofs = g.rofs; a.tex = g.tex << g.pfshift; a.uniq = g.uniq;
  a.filtered = g.filtered; a.capb = (g.cap << g.pfshift) + kPfBias; a.pfshift = g.pfshift;
  a.row0 = g.row0; a.row1 = g.row1;
  const int xhi = std::min(g.W - g.lofs - 1, g.W - g.rofs - g.nd);
  a.xc0 = g.w2; a.xc1 = xhi - g.w2 + 1;
  const int pw = g.wsz % 3 == 0 ? 3 : 1;
  const int nv = 64 - (g.wsz - pw);
  // windows that are multiples of 3: triples of column-stride-3 strips (3 * nv3 columns each) as far as they pay, plain
  // strips for the rest (SBM_FAST_CS3=0: plain strips only)
  const int cs3_env = env_switch("SBM_FAST_CS3", 1);   // read per call (the GPU tests flip it)
  const int ncols = a.xc1 - a.xc0;
  const int nv3 = 64 - (g.wsz / 3 - 1);
  int triples = 0;
  if (pw == 3 && cs3_env) {
    triples = ncols / (3 * nv3);
    if (ncols - triples * 3 * nv3 > 2 * nv) triples++;   // a remainder worth three plain strips is one more triple
  }
  a.strips3 = 3 * triples;
  const int rem = std::max(0, ncols - triples * 3 * nv3);
  const int strips = a.strips3 + (rem + nv - 1) / nv;
  const int rows = g.row1 - g.row0;
  // Row segments. Every segment pays w - 1 priming rows (~0.2 of a full row each: mqsad only); few long segments keep that low,
  // many short ones keep the tail of the launch short (the last, partly empty round of workgroups). With R rounds of the chip
  // the two costs are ~ prime * nseg / rows and ~ c / R, R = strips * pairs * nseg / slots, so the optimum is
  //   nseg = sqrt(c * slots * rows / (strips * pairs * prime)),   slots = workgroups the chip holds at once for the layout
  // (4 096 / 3 072 / 1 536 at up to 64 / 128 / 256 disparities), c = 0.196 fitted on the forced-count sweeps of round 5
  // (profiles/r05_small_launch_segments.txt, last table: KITTI x64 best at 8 = the formula's 8.0; 640x480 nd 64 w 21 x64 best at
  // 10-14, formula 12, round 4's rule 6: 0.476 -> 0.469 ms; 1080p nd 256 x64 best 6-8, formula 6.6, round 4's 4: 8.28 -> 8.19;
  // 2160p x32 best 8-12, formula 9, round 4's 3: 17.64 -> 16.97; 1080p x16 and 2160p x4 unchanged at 13 and 25). Round 4's rule
  // (a workgroup target of 24 000 / 5 600 with segments of at least three window heights) was tuned at 64 / 16 / 4 pairs only.
  // (one round of the chip for the cooperating 128-disparity wavefronts: 12 wavefronts per CU in workgroups of 2 / 3 / 4)
  const int round1 = g.nd <= 256 ? 1536 : (g.nd <= 384 ? 1024 : 768);
  int nseg = 1;
  {
    const double prime = 0.2 * (g.wsz - 1);
    const double slots = g.nd <= 64 ? 4096.0 : (g.nd <= 128 ? 3072.0 : (double)round1);
    static const int c1000 = SBM_TUNE("SBM_DEV_SEG_C", 196);
    nseg = (int)(std::sqrt(c1000 * 1e-3 * slots * rows / ((double)strips * g.n * prime)) + 0.5);
    nseg = std::max(1, std::min(nseg, std::max(1, rows / g.wsz)));   // (at least one window height per segment here; see below)
  }
  // Launches that do not fill the chip (round 5, profiles/r05_small_launch_segments.txt): a wavefront's row segment is a serial
  // chain, and w - 1 priming rows at a third of a row's cost are cheaper than idle SIMDs -- segments go down to 8 rows (up to
  // 64 of them) until the launch has ~5 000 workgroups. Launches that cannot even reach ~1 800 wavefronts that way (one or two
  // pairs) split the disparities over more wavefronts per workgroup instead (launch_nd: `split`) and take as many segments
  // as keep them under 1 024 workgroups. One 640x480 nd 64 w 21 pair: SAD stage 0.054 -> 0.049 ms, one KITTI pair 0.052 ->
  // 0.036, 8 KITTI pairs 0.209 -> 0.14, 16 pairs 640x480 0.203 -> 0.12, 4 pairs 1080p nd 256 0.64 -> 0.57.
  static const int small_rows = SBM_TUNE("SBM_DEV_SMALL_ROWS", 8);
  static const long fill = SBM_TUNE("SBM_DEV_FILL", 5000);
  const int maxseg = std::max(nseg, std::min(64, rows / small_rows));
  const long per_seg = (long)strips * g.n;
  if (per_seg * maxseg * (g.nd > 128 ? (g.nd + 127) / 128 : 1) < 1800) {
    nseg = std::max(nseg, (int)std::min<long>(maxseg, 1023 / per_seg));
  } else {
    const int nseg1 = nseg;
    while (per_seg * nseg < fill && nseg < maxseg) nseg++;
    // (two cooperating 128-disparity wavefronts: 1 536 workgroups are one round of the chip; a launch that ends between 1 and
    // 1.5 rounds pays a second, mostly empty round -- one 1080p nd 256 pair: 64 segments 0.204 ms, 36..48 segments 0.189..0.196)
    if (g.nd > 128 && per_seg * nseg > round1 && 2 * per_seg * nseg < 3 * round1) nseg = std::max(nseg1, (int)((long)round1 * 1450 / 1536 / per_seg));
  }
  static const int nseg_env = SBM_TUNE("SBM_FAST_NSEG", 0);
  if (nseg_env > 0) nseg = std::min(nseg_env, std::max(1, rows / 2));
  // taper: the last third of the rows is cut into segments of 2/3, 1/2, 1/3 ... of the regular length
  static const int taper = SBM_TUNE("SBM_FAST_TAPER", 1);
  nseg = std::min(nseg, 64);
  int ns = 0;
  a.segrow[0] = g.row0;
  if (!taper || nseg < 3) {
    const int seg = (rows + nseg - 1) / nseg;
    for (int y = g.row0; y < g.row1; y += seg) a.segrow[++ns] = std::min(y + seg, g.row1);
  } else {
    // nseg segments with weights 1,...,1, 3/4, 1/2, 1/4 (sum nseg - 1.5): fewer rows to prime than nseg equal ones
    const double unit = rows / (nseg - 1.5);
    double acc = 0;
    for (int k = 0; k < nseg; k++) {
      const double wgt = k < nseg - 3 ? 1.0 : (k == nseg - 3 ? 0.75 : (k == nseg - 2 ? 0.5 : 0.25));
      acc += wgt * unit;
      const int y = k == nseg - 1 ? g.row1 : std::min(g.row1, g.row0 + (int)(acc + 0.5));
      if (y > a.segrow[ns]) a.segrow[++ns] = y;
    }
  }
  nseg = ns;
  a.strips = strips; a.nseg = nseg; a.npairs = g.n;
  {
    const long maxs = (long)g.wsz * g.wsz * 2 * g.cap;
    static const int plain_env = SBM_TUNE("SBM_FAST_UNIQ_PLAIN", 1);
    a.uniq_plain = plain_env && 8 * ((maxs * g.uniq / 100 + 1) << g.pfshift) <= 65535;   // (8 registers per accumulator)
  }
  static const int split_env = SBM_TUNE("SBM_FAST_SPLIT", 1);
  const bool split = (long)strips * nseg * g.n < 1024 && split_env;
  hipError_t e;
  switch (g.wsz) {
#if defined(SBM_DEV_FEW19)
    default: e = launch_sad_fast_pw1(a, g.wsz, border, mode, split, s); break;
#elif defined(SBM_DEV_FEW31)   // (experiment: windows 29 and 31 as 1-column sums)
    case 29: e = launch_nd<29, 1>(a, border, mode, split, s); break;
    case 31: e = launch_nd<31, 1>(a, border, mode, split, s); break;
    default: e = hipErrorInvalidValue; break;
#elif defined(SBM_DEV_FEW)   // development builds (tools/exp): only the bench workloads' windows are instantiated
    case 15: e = launch_nd<5, 3>(a, border, mode, split, s); break;
    case 21: e = launch_nd<7, 3>(a, border, mode, split, s); break;
    default: e = hipErrorInvalidValue; break;
#else
    case 9: e = launch_nd<3, 3>(a, border, mode, split, s); break;
    case 15: e = launch_nd<5, 3>(a, border, mode, split, s); break;
    case 21: e = launch_nd<7, 3>(a, border, mode, split, s); break;
    case 27: e = launch_nd<9, 3>(a, border, mode, split, s); break;
    default: e = launch_sad_fast_pw1(a, g.wsz, border, mode, split, s); break;
#endif
  }
  *xa = a.xc0; *xb = a.xc1;
  return e;
}
#endif  // SBM_FAST_TU == 0

}  // namespace sbm
