// sbm_fpga.hip -- the reference's OWN block matcher (FPGA RTL, "flavour B" of SURVEY.md) on gfx950:
// 6-bit x-Sobel inputs, 10-bit saturating column sums, 32-disparity phases with record merge, tournament minimum with
// runner-up, non-restoring-divider sub-pixel fraction, optional min1/min2 ratio filter, s11.4 output with 0xFFFF.
//
// Device counterpart of src/dvp/rtl/bm_calc_sad.v:78-149,350-612 (abs-diff, HSAD, SAD), bm_calc_det.v:121-438,
// bm_calc_frac.v:59-173 + diven.v, bm_calc_upd.v:107-209, bm_calc_uni.v:117-134 / bm_calc.v:312-328, bm_obuf2.v:119-154,
// but NOT scheduled like bm_ibuf.v:143-286 / bm_obuf.v (one pass over the frame per 32 disparities with 8-byte records
// in DRAM in between): up to four phases live in one wavefront's registers and the record is merged there. Index
// mappings (lane j of phase k <-> disparity 32k+j-1, HSAD column c <-> image column ndisp+c, output sample i at column
// ndisp+hwsz+1+i) are derived in DESIGN.md (section "FPGA flavour").
//
// Mapping: a wavefront owns 64 consecutive HSAD columns (lane = column) and marches down the rows of one segment.
//   rows    the wavefront's piece of a right row (63 + 4 NU bytes) is staged once by LDS-direct loads into a 4x-expanded
//           layout (dword slot p = bytes p..p+3; round 6 -- the per-lane register copies of rounds 2-5 cost 40 VGPRs and a
//           wavefront per SIMD), two areas: the entering and the leaving row, loaded a row ahead
//   AD      one v_mqsad_pk_u16_u8 with a single-byte pattern = |R[t..t+3] - L| for 4 consecutive disparities
//           (9 per row cover the 34 lanes of a phase); the leaving row is re-evaluated instead of stored
//   HSAD    18 VGPRs of packed u16: v_pk_add_u16 + v_pk_min_u16(.., 1023) on entry, v_pk_sub_u16 clamp on exit
//   SAD     inclusive prefix sum over the 64 lanes (4 DPP row shifts + 2 row broadcasts per register; 64 * 1023 still
//           fits 16 bits), window sum = P[l + 2*hwsz] - P[l - 1] through LDS
//   det     key minimum per quarter of the tournament, the top two rounds literally; with the uniqueness filter off and the
//           whole range in one launch (the firmware's configuration) a plain first minimum, the fraction once per pixel (LEAN)
// The 10-bit saturation makes HSAD history dependent, so row segments are only exact while nothing saturates: the
// segmented launch stamps a per-pair word with the call's generation number when a column sum passes 1023 and a second,
// one-segment launch (which exits immediately for unstamped pairs; not launched when wsz * 63 <= 1023) recomputes such
// pairs strictly top to bottom.
#include <algorithm>

#include "sbm_common.h"

namespace sbm {

typedef unsigned int u32;
typedef unsigned long long u64;
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

struct FpgaArgs {
  const uint8_t* xl;     // dense x-Sobel planes [pair][H][W] (read through range-checked buffer loads: no padded copies)
  const uint8_t* xr;
  uint2* rec;            // [pair][sad_hgt][sad_wdt] records {min1 | min2 << 16, disp1 | disp2 << 8 | frac << 16}: only
                         // between the launches of a disparity range that does not fit one launch (more than 4 phases)
  int16_t* disp;         // dense output; the kernel writes every pixel (0xFFFF where the RTL keeps the firmware's fill)
  int* flag;             // [pair] generation number of the last call in which a column sum saturated
  int gen;
  int W, H;
  int nd, wsz, hwsz, hsad_wdt, sad_wdt, sad_hgt;
  int phase0, first, last;   // first phase of this launch; first / last launch of the disparity range
  int uni_enb, uni_mode, uni_thr;
  int seg;               // output rows per segment
  int exact;             // 1: the one-segment launch that only runs for flagged pairs
  int track_sat;         // wsz * 63 > 1023: a column sum can saturate
};

__device__ __forceinline__ u32 pk_add(u32 a, u32 b) {
  u16x2 r = __builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b);
  return __builtin_bit_cast(u32, r);
}
__device__ __forceinline__ u32 pk_min(u32 a, u32 b) {
  u16x2 r = __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b));
  return __builtin_bit_cast(u32, r);
}
__device__ __forceinline__ u32 pk_subs(u32 a, u32 b) {
  u16x2 r = __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b));
  return __builtin_bit_cast(u32, r);
}

// diven.v #(DW, VW, QW, MSB_INV) with EXT_DIV == 0 (both instances of the matcher): RW = DW + 1, EVW = VW.
template <int DW, int QW>
__device__ __forceinline__ u32 rtl_diven(u32 dividend, u32 divisor) {
  constexpr int RW = DW + 1;
  constexpr u32 mRW = (1u << RW) - 1u, mQ = (1u << QW) - 1u;
  const u32 ediv = divisor & ((1u << DW) - 1u);
  u32 rem = dividend & ((1u << DW) - 1u);
  if ((rem >> (DW - 1)) & 1u) rem |= 1u << DW;                       // sign extension by EXT_REM = 1 bit
  const u32 sdiv = (ediv >> (DW - 1)) & 1u;
  const u32 d2 = (ediv << 1) & mRW;
  u32 quot = 0;
#pragma unroll
  for (int i = 0; i <= QW; i++) {                                    // diven.v:129-176
    const u32 op = sdiv ^ ((rem >> (RW - 1)) & 1u);                  // 1 = add, 0 = subtract
    const u32 a = ((rem << 1) | (op ^ 1u)) & mRW;
    rem = (a + (op ? d2 : (d2 ^ mRW))) & mRW;
    if (i > 0) quot = ((quot << 1) | (op ^ 1u)) & mQ;
  }
  return (quot + sdiv) & mQ;
}

// bm_calc_frac.v:59-173
__device__ __forceinline__ u32 rtl_frac(u32 c, u32 l, u32 r) {
  const u32 m17 = 0x1ffffu;
  const u32 dif_lr = (l - r) & m17, dif_lc = (l - c) & m17, dif_rc = (r - c) & m17;
  const bool cmp = l < r, neg = ((dif_lc | dif_rc) >> 16) & 1u;
  const u32 dividend = neg ? 0u : ((((dif_lr >> 16) & 1u) << 17) | dif_lr);
  const u32 divisor = ((cmp ? dif_rc : dif_lc) << 1) & 0x3ffffu;
  if (divisor == 0u) return cmp ? 0x40u : 0xC0u;
  return rtl_diven<18, 8>(dividend, divisor);
}

// bm_obuf2.v:119-154
__device__ __forceinline__ int rtl_pack(u32 disp, u32 frac) {
  const u32 fe = (frac & 0x80u) ? (0x1ff00u | frac) : frac;
  const u32 depth = ((disp << 8) + fe) & 0x1ffffu;
  if (((depth >> 16) & 1u) || depth == 0u) return -1;
  u32 v = (depth >> 4) & 0xfffu;
  if ((depth >> 15) & 1u) v |= 0xf000u;
  return (int)(short)v;
}

extern __shared__ __attribute__((aligned(16))) u32 fpga_lds[];

constexpr int FP_NR = 18;        // packed registers: reg r = (lane 34-2r | lane 33-2r << 16), lanes 34 and -1 are padding
constexpr int FP_XS = 64 + 34;   // LDS row: slot 0 = 0 (prefix left of lane 0), slot 1+l = prefix of lane l
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u32 pk_max(u32 a, u32 b) {
  u16x2 r = __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b));
  return __builtin_bit_cast(u32, r);
}

// One launch = NPH consecutive 32-disparity phases of the RTL's schedule, ALL held in registers by the same wavefront (the
// FPGA makes one pass over the frame per phase with 8-byte records in DRAM in between -- bm_ibuf.v:143-189, bm_obuf.v --
// because it has 34 abs-diff lanes; a wavefront has registers for 4 x 34). The record {min1, min2, disp1, disp2, frac}
// of a pixel is merged phase by phase in registers (bm_calc_upd.v:125-209); only a range of more than 4 phases goes
// through the record plane, once per 4 phases. The x-Sobel planes are read where they lie: range-checked buffer loads
// return 0 for the few window bytes of the padding lanes that fall outside the batch, and rows/columns the RTL never
// writes get the firmware's 0xFFFF from the same kernel.
// LEAN: the whole disparity range in this launch and the uniqueness filter off (what the firmware programs: fpga.c:150-160 never
// writes UniFiltCtrl) -- the second minimum of bm_calc_det.v / bm_calc_upd.v then feeds nothing: the search is a plain first
// minimum, the merge across phases "strictly smaller wins", and the sub-pixel fraction (nine divider steps) is taken once per
// pixel, for the final winner, from its two neighbour sums carried along.
#ifndef SBM_FPGA_WPE2   // wavefronts per SIMD the register allocation of the two-phase kernel aims at (development builds compare)
#define SBM_FPGA_WPE2 3
#endif
template <int NPH, bool LEAN>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NPH == 1 ? 4 : (NPH == 2 ? SBM_FPGA_WPE2 : 1)))) fpga_bm_kernel(FpgaArgs a) {
  const int lane = threadIdx.x;
  const int pair = blockIdx.z;
  if (a.exact && a.flag[pair] != a.gen) return;
  const int NV = 64 - 2 * a.hwsz;                 // outputs per strip
  const int c = blockIdx.x * NV + lane;           // HSAD column of this lane
  const bool col_ok = c < a.hsad_wdt;
  const int x = a.nd + min(c, a.hsad_wdt - 1);    // image column (clamped for the idle lanes: their sums are never used)
  const int r0 = blockIdx.y * a.seg, r1 = min(r0 + a.seg, a.sad_hgt);
  if (r0 >= r1) return;
  const size_t plane = (size_t)a.W * a.H;
  // range-checked descriptors over this pair's planes (+ what follows in the batch, at most 2 GB): the lowest window byte
  // of the leftmost column lies one byte in front of its row, the highest one of the rightmost column a few behind it
  const size_t rest = ((size_t)gridDim.z - pair) * plane;
  const unsigned nrec = (unsigned)(rest < 0x7fffffffull ? rest : 0x7fffffffull);
  const __amdgpu_buffer_rsrc_t rs_l = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.xl + pair * plane), 0, nrec, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.xr + pair * plane), 0, nrec, 0x00020000);
  // window byte t of phase kk = R[x - 32 kk - 33 + t], t = 34 - j for lane j (disparity 32 kk + j - 1). The phases of
  // this launch read one piece per lane: bytes x - 32 (phase0 + NPH - 1) - 33 ... + 32 (NPH - 1) + 39, phase k at dword
  // 8 (NPH-1-k). Adjacent lanes' pieces overlap in all but one byte, so the wavefront's piece -- 63 + 4 NU bytes from the
  // first lane's start -- is staged ONCE, by LDS-direct loads into a 4x-expanded layout (dword slot p = bytes p .. p+3 of the
  // piece: lane i of a load writes its 4 source bytes to slot i, a byte-granular source address is fine), and lane l reads
  // dword j of its window from slot l + 4 j. Round 6: before, every lane held its own 4 NU bytes of two rows in registers
  // (192 VGPRs = 2 wavefronts per SIMD at two phases; 243 register moves per row to rotate clamped pieces into place).
  constexpr int NU = 8 * (NPH - 1) + 10;
  constexpr int NS = 64 + 4 * (NU - 1);            // slots a wavefront reads
  constexpr int NLD = (NS + 63) / 64;              // LDS-direct loads per staged row
  constexpr int SROW = 64 * NLD;                   // dword slots of one staged-row area
  const int cb = blockIdx.x * NV;                  // HSAD column of lane 0
  const int pbase = a.nd + cb - 32 * (a.phase0 + NPH - 1) - 33;   // piece start within a row (-1 for strip 0 of the launch with the last phase)
  u32* const xrow = fpga_lds;                      // [FP_NR][FP_XS]
  u32* const stg0 = fpga_lds + FP_NR * FP_XS;      // staged rows: entering / leaving
  u32* const stg1 = stg0 + SROW;
  typedef __attribute__((address_space(3))) void* lds_vptr;

  u32 V[NPH][FP_NR];
#pragma unroll
  for (int k = 0; k < NPH; k++)
#pragma unroll
    for (int r = 0; r < FP_NR; r++) V[k][r] = 0u;
  u32 satmax = 0u;

  // stage row y into `buf`; returns this lane's left byte (a register load that lands with the row)
  auto stage = [&](const int y, u32* const buf) -> u32 {
    const int po = y * a.W + pbase;                // piece start as a byte offset into the pair's plane (uniform)
#pragma unroll
    for (int it = 0; it < NLD; it++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_r, (lds_vptr)(buf + 64 * it), 4, po + 64 * it + lane, 0, 0, 0);
    return (u32)__builtin_amdgcn_raw_buffer_load_b8(rs_l, y * a.W + x, 0, 0);
  };
  // everything in flight has landed. The hardware zeroes a WHOLE dword whose range check fails, and two kinds of slots
  // straddle the descriptor: slot 0 of the piece that starts one byte in front of its plane (row 0, strip 0, the launch that
  // holds the last phase -- only a padding lane's byte lies outside, but it shares the dword with real ones) and the slots
  // over the last three bytes of the batch (last rows of the LAST frame). Both are rebuilt from a neighbour slot that lies
  // inside (wave-uniform, rare branches).
  auto landed = [&](const int y, u32* const buf) {
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int po = y * a.W + pbase;
    if (po < 0) {                                   // (po == -1) slot 0 = bytes -1 .. 2: byte -1 reads as 0, the others sit in slot 1
      if (lane == 0) buf[0] = buf[1] << 8;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    if ((unsigned)(po + SROW + 3) > nrec) {         // slots p with p + 3 >= nrec > p: from the last dword that lies inside
      const int s4 = (int)nrec - 4 - po;            // its slot
#pragma unroll
      for (int it = 0; it < NLD; it++) {
        const int p = 64 * it + lane, over = p - s4;   // 1 .. 3: bytes that fall outside
        if (s4 >= 0 && over >= 1 && over <= 3) buf[p] = buf[s4] >> (8 * over);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  };
  auto reads_done = [] {    // every LDS read of this wavefront has returned: an area may be overwritten
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  };
  // |R[t] - L| for the 36 window bytes of one staged row and phase, packed like V
  auto row_ad = [&](const u32* const buf, const u32 lb, const int k, u32 (&ad)[FP_NR]) {
    // 6 significant bits (bm_calc_sad.v:375,380); bit 6 set on both sides cancels in the difference and keeps the pattern byte
    // from being zero (a zero pattern byte is masked by the instruction)
    const u32 l = (lb & 63u) | 64u;
    const u32* const w = buf + lane + 32 * (NPH - 1 - k);
    u32 u[10];
#pragma unroll
    for (int q = 0; q < 10; q++) u[q] = w[4 * q];
#pragma unroll
    for (int q = 0; q < 9; q++) {
      const u32 wq = (u[q] & 0x3f3f3f3fu) | 0x40404040u;
      const u64 win = ((u64)u[q + 1] << 32) | wq;   // only bytes 0..3 meet a non-masked pattern byte
      const u64 m = __builtin_amdgcn_mqsad_pk_u16_u8(win, l, 0ull);
      ad[2 * q] = (u32)m;
      ad[2 * q + 1] = (u32)(m >> 32);
    }
  };
  auto add_row = [&](const u32* const buf, const u32 lb) {   // bm_calc_sad.v:450-457: + |.|, upper limit 1023
#pragma unroll
    for (int k = 0; k < NPH; k++) {
      u32 ad[FP_NR];
      row_ad(buf, lb, k, ad);
#pragma unroll
      for (int r = 0; r < FP_NR; r++) {
        const u32 sum = pk_add(V[k][r], ad[r]);
        if (a.track_sat) satmax |= sum;     // a sum is at most 1023 + 63: bit 10 of a half <=> that column passed the limiter
        V[k][r] = pk_min(sum, 0x03ff03ffu);
      }
    }
  };
  auto sub_row = [&](const u32* const buf, const u32 lb) {   // bm_calc_sad.v:459-462: - |.|, lower limit 0
#pragma unroll
    for (int k = 0; k < NPH; k++) {
      u32 ad[FP_NR];
      row_ad(buf, lb, k, ad);
#pragma unroll
      for (int r = 0; r < FP_NR; r++) V[k][r] = pk_subs(V[k][r], ad[r]);
    }
  };

  // rows of the first window but the last, alternating between the two areas (the next one arrives while one is consumed);
  // the row staged last -- into whichever area is then due -- is the first output row's entering row
  u32 le = stage(r0, stg0), ll = 0u;
  u32* be = stg0;            // area of the row that enters next
  u32* bl = stg1;
  for (int y = r0; y < r0 + a.wsz - 1; y++) {
    landed(y, be);
    const u32 ln = stage(y + 1, bl);
    add_row(be, le);
    reads_done();
    le = ln;
    u32* const t = be; be = bl; bl = t;
  }

  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < FP_NR; r++) xrow[r * FP_XS] = 0u;
  }
  const int i = c;                                 // output sample of this lane
  const bool sample = lane < NV && i < a.sad_wdt;
  int16_t* const dpair = a.disp + (size_t)pair * plane;
  // columns left of the first sample and right of the last one, and the rows above / below the SAD rows: 0xFFFF
  // (fpga.c:105-106); written by the first launch of the segmented pass only
  const bool fill = a.first && !a.exact;
  const int col_s = a.nd + a.hwsz + 1;             // image column of sample 0
  auto fill_row = [&](int yy) {                    // this strip's share of a row that holds no sample at all
    int16_t* row = dpair + (size_t)yy * a.W;
    if (blockIdx.x == 0)
      for (int xx = lane; xx < col_s; xx += 64) row[xx] = (int16_t)-1;
    if (sample) row[col_s + i] = (int16_t)-1;
    if (blockIdx.x == gridDim.x - 1)
      for (int xx = col_s + a.sad_wdt + lane; xx < a.W; xx += 64) row[xx] = (int16_t)-1;
  };
  if (fill) {
    if (r0 == 0)
      for (int yy = 0; yy < a.hwsz; yy++) fill_row(yy);
    if (r1 == a.sad_hgt)
      for (int yy = a.hwsz + a.sad_hgt; yy < a.H; yy++) fill_row(yy);
  }

  for (int r = r0; r < r1; r++) {
    // `be` holds row r + wsz - 1 (entering); it is consumed here, then both areas take the rows this iteration's end and the
    // next one's start need -- the leaving row r and the next entering row -- which land under the phases below
    landed(r + a.wsz - 1, be);
    add_row(be, le);
    reads_done();
    ll = stage(r, bl);
    if (r + 1 < r1) le = stage(r + a.wsz, be);
    u32 min1 = 0, min2 = 0, disp1 = 0, disp2 = 0, frac = 0;
    uint2* rp = a.rec + ((size_t)pair * a.sad_hgt + r) * a.sad_wdt + i;
    if (!a.first && sample) {                      // a range of more than NPH phases: the record of the earlier launches
      const uint2 rc = *rp;
      min1 = rc.x & 0xffffu; min2 = rc.x >> 16; disp1 = rc.y & 0xffu; disp2 = (rc.y >> 8) & 0xffu; frac = (rc.y >> 16) & 0xffu;
    }
#pragma unroll
    for (int k = 0; k < NPH; k++) {
      // ---- SAD: prefix sums over the lanes, window = P[l + 2*hwsz] - P[l - 1] (bm_calc_sad.v:569-605) ---------------
      u32 S[FP_NR];
      // (issue priority up while the prefix rows go through LDS, as in the interior kernel of the cv flavour: 1 % here)
      __builtin_amdgcn_s_setprio(2);
#pragma unroll
      for (int q = 0; q < FP_NR; q++) {
        u32 p = V[k][q];
        p += (u32)__builtin_amdgcn_update_dpp(0, (int)p, 0x111, 0xf, 0xf, false);   // row_shr:1
        p += (u32)__builtin_amdgcn_update_dpp(0, (int)p, 0x112, 0xf, 0xf, false);   // row_shr:2
        p += (u32)__builtin_amdgcn_update_dpp(0, (int)p, 0x114, 0xf, 0xf, false);   // row_shr:4
        p += (u32)__builtin_amdgcn_update_dpp(0, (int)p, 0x118, 0xf, 0xf, false);   // row_shr:8
        p += (u32)__builtin_amdgcn_update_dpp(0, (int)p, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
        p += (u32)__builtin_amdgcn_update_dpp(0, (int)p, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
        xrow[q * FP_XS + 1 + lane] = p;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // (other lanes' entries are read through an opaque copy of the lane index: to the optimiser a load of [lane] cannot be
      // changed by this lane's store to [1 + lane], and it would be free to merge it with the previous phase's load of the
      // same address -- see sad_fast_strip_dma, round 5)
      int lxo = lane;
      asm volatile("" : "+v"(lxo));
#pragma unroll
      for (int q = 0; q < FP_NR; q++) S[q] = xrow[q * FP_XS + 1 + lxo + 2 * a.hwsz] - xrow[q * FP_XS + lxo];
      __builtin_amdgcn_s_setprio(0);

      // neighbours of a winner lane jw = idx + 1: lanes jw - 1 and jw + 1, looked up in the prefix rows
      auto sad_of_lane = [&](int j) -> u32 {
        const int q = (34 - j) >> 1, hi = (34 - j) & 1;   // lane 34-2q is the low half
        const u32 pa = xrow[q * FP_XS + 1 + lxo + 2 * a.hwsz], pb = xrow[q * FP_XS + lxo];
        const u32 d = pa - pb;
        return hi ? (d >> 16) : (d & 0xffffu);
      };
      if constexpr (LEAN) {
        if (sample) {
          // first minimum over the 32 lanes by keys (value << 16 | idx, idx = lane - 1): reg q holds lane 34-2q (low half,
          // idx 33-2q) and lane 33-2q (high half, idx 32-2q), q = 1 .. 16
          u32 b[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
#pragma unroll
          for (int q = 1; q <= 16; q++) {
            const u32 klo = (S[q] << 16) | (u32)(33 - 2 * q), khi = (S[q] & 0xffff0000u) | (u32)(32 - 2 * q);
            b[q & 3] = min(b[q & 3], min(klo, khi));
          }
          const u32 win = min(min(b[0], b[1]), min(b[2], b[3]));
          const int i1 = (int)(win & 0xffffu);
          const u32 dmin1 = win >> 16;
          if (k == 0 || dmin1 < min1) {                // bm_calc_upd.v:125-209 reduced to its min1 / disp1 columns
            min1 = dmin1;
            disp1 = (u32)((((a.phase0 + k) & 7) << 5) | i1);
            min2 = sad_of_lane(i1);                    // (the dead second-minimum registers carry the winner's neighbour sums)
            disp2 = sad_of_lane(i1 + 2);
          }
        }
      } else
      if (sample) {
        // ---- bm_calc_det.v: first minimum of every quarter by keys (value << 16 | idx, idx = lane - 1); reg q holds
        // lane 34-2q (low half, idx 33-2q) and lane 33-2q (high half, idx 32-2q); quarter m = regs 16-4m-3 .. 16-4m
        u32 Q[4];
#pragma unroll
        for (int m = 0; m < 4; m++) {
          u32 best = 0xffffffffu;
#pragma unroll
          for (int q = 13 - 4 * m; q <= 16 - 4 * m; q++) {
            best = min(best, (S[q] << 16) | (u32)(33 - 2 * q));
            best = min(best, (S[q] & 0xffff0000u) | (u32)(32 - 2 * q));
          }
          Q[m] = best;
        }
        const u32 h0 = min(Q[0], Q[1]), l0 = max(Q[0], Q[1]);      // stage 4: half winners and their victims
        const u32 h1 = min(Q[2], Q[3]), l1 = max(Q[2], Q[3]);
        const u32 win = min(h0, h1), c0 = max(h0, h1);             // stage 5: winner, loser of the final
        const u32 c1 = (l1 >> 16) < (l0 >> 16) ? l1 : l0;          //          better of the two semi-final victims
        const int i1 = (int)(win & 0xffffu), ia = (int)(c0 & 0xffffu), ib = (int)(c1 & 0xffffu);
        const bool adj0 = ia == i1 + 1 || i1 == ia + 1, adj1 = ib == i1 + 1 || i1 == ib + 1;
        const bool pick1 = (((c1 >> 16) < (c0 >> 16)) && !adj1) || adj0;   // stage 6
        const u32 m2k = pick1 ? c1 : c0;
        const u32 dmin1 = win >> 16, dmin2 = m2k >> 16;
        const int k32 = ((a.phase0 + k) & 7) << 5;
        const u32 ddisp1 = (u32)(k32 | i1), ddisp2 = (u32)(k32 | (int)(m2k & 0xffffu));
        const u32 dl = sad_of_lane(i1), dr = sad_of_lane(i1 + 2);
        const u32 frac_new = rtl_frac(dmin1, dl, dr);

        if (a.first && k == 0) {                     // bm_calc_upd.v:147-154
          min1 = dmin1; min2 = dmin2; disp1 = ddisp1; disp2 = ddisp2; frac = frac_new;
        } else {                                     // bm_calc_upd.v:125-209
          const u32 s1 = min1, s2 = min2, sd1 = disp1, sd2 = disp2;
          const bool d1s1 = dmin1 < s1, d2s1 = dmin2 < s1, d1s2 = dmin1 < s2, d2s2 = dmin2 < s2;
          const bool adj = ddisp1 == ((sd1 + 1u) & 0xffu);
          if (d1s1 && d2s1) {
            min1 = dmin1; disp1 = ddisp1; min2 = dmin2; disp2 = ddisp2; frac = frac_new;
          } else if (d1s1 && d2s2) {
            min1 = dmin1; disp1 = ddisp1; frac = frac_new;
            min2 = !adj ? s1 : dmin2; disp2 = !adj ? sd1 : ddisp2;
          } else if (d1s1) {
            min1 = dmin1; disp1 = ddisp1; frac = frac_new;
            min2 = !adj ? s1 : s2; disp2 = !adj ? sd1 : sd2;
          } else if (d1s2 && d2s2) {
            min2 = !adj ? dmin1 : dmin2; disp2 = !adj ? ddisp1 : ddisp2;
          } else if (d1s2) {
            min2 = !adj ? dmin1 : s2; disp2 = !adj ? ddisp1 : sd2;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();             // the prefix rows are rewritten by the next phase / output row
    }
    int16_t* orow = dpair + (size_t)(a.hwsz + r) * a.W;
    if (sample) {
      if constexpr (LEAN) {
        orow[col_s + i] = (int16_t)rtl_pack(disp1, rtl_frac(min1, min2, disp2));
      } else if (!a.last) {
        *rp = make_uint2(min1 | (min2 << 16), disp1 | (disp2 << 8) | (frac << 16));
      } else {
        u32 od = disp1, of = frac;
        if (a.uni_enb) {                           // bm_calc_uni.v:117-134, bm_calc.v:312-328
          const u32 ratio = rtl_diven<17, 11>(min1, min2) & 0x3ffu;
          if (ratio > (u32)(a.uni_thr & 0x3ff)) od = of = a.uni_mode ? 0xffu : 0x00u;
        }
        orow[col_s + i] = (int16_t)rtl_pack(od, of);
      }
    }
    if (fill) {                                    // the columns of this row that hold no sample
      if (blockIdx.x == 0)
        for (int xx = lane; xx < col_s; xx += 64) orow[xx] = (int16_t)-1;
      if (blockIdx.x == gridDim.x - 1)
        for (int xx = col_s + a.sad_wdt + lane; xx < a.W; xx += 64) orow[xx] = (int16_t)-1;
    }
    if (r + 1 < r1) {
      landed(r, bl);          // (covers the next entering row in `be` as well; its own fix-ups run at the top of the next iteration)
      sub_row(bl, ll);
      reads_done();
    }
  }
  if (a.track_sat && !a.exact && col_ok && (satmax & 0x04000400u)) atomicMax(a.flag + pair, a.gen);
}

// xl/xr: dense n*H*W x-Sobel planes on the device. rec: n*sad_hgt*sad_wdt uint2 (used beyond 128 disparities only),
// flag: n ints, zero when allocated; gen: a number that grows with every call on these buffers (> 0).
hipError_t launch_fpga_bm(const uint8_t* xl, const uint8_t* xr, void* rec, int* flag, int gen, int16_t* disp, int n,
                          const sbm_fpga_params& p, hipStream_t s) {
  FpgaArgs a;
  const int W = p.width, H = p.height;
  a.W = W; a.H = H;
  a.nd = p.num_disparities; a.wsz = p.block_size; a.hwsz = p.block_size >> 1;
  a.hsad_wdt = W - a.nd - 1; a.sad_wdt = a.hsad_wdt - 2 * a.hwsz; a.sad_hgt = H - 2 * a.hwsz;
  a.uni_enb = p.uni_enable; a.uni_mode = p.uni_mode; a.uni_thr = p.uni_threshold;
  a.xl = xl; a.xr = xr; a.rec = static_cast<uint2*>(rec); a.disp = disp; a.flag = flag; a.gen = gen;
  a.track_sat = a.wsz * 63 > 1023;                 // 6-bit abs-diffs: a column sum of up to 16 rows cannot reach the limiter
  const int NV = 64 - 2 * a.hwsz;
  const int strips = (a.sad_wdt + NV - 1) / NV;
  const int nphase = a.nd >> 5;
  // segments: enough wavefronts for the chip, each paying wsz-1 priming rows
  int nseg = 1;
  while ((long)strips * nseg * n < 4096 && a.sad_hgt / (nseg + 1) >= 2 * a.wsz) nseg++;
  // prefix rows + the two staged-row areas of the widest launch (4 phases: 64 + 4 * 33 = 196 slots -> 4 loads of 64)
  const size_t lds = (size_t)(FP_NR * FP_XS + 2 * 256) * sizeof(u32);
  const int nph = nphase >= 4 ? 4 : nphase;        // phases per launch: up to 4 live in one wavefront's registers
  for (int pass = 0; pass < 2; pass++) {
    // pass 0: segmented (exact unless a column sum saturates); pass 1: flagged pairs only, one segment top to bottom
    a.exact = pass;
    const int ns = pass ? 1 : nseg;
    a.seg = (a.sad_hgt + ns - 1) / ns;
    if (pass == 1 && (nseg == 1 || !a.track_sat)) break;   // the first pass already ran top to bottom / nothing can saturate
    const dim3 grid(strips, (a.sad_hgt + a.seg - 1) / a.seg, n);
    for (int k = 0; k < nphase;) {
      // the remaining phases in groups of 4, 2 or 1 (3 phases = 2 + 1, 6 = 4 + 2, ...)
      const int left = nphase - k, grp = left >= 4 && nph == 4 ? 4 : (left >= 2 ? 2 : 1);
      a.phase0 = k; a.first = k == 0; a.last = k + grp == nphase;
      const bool lean = a.first && a.last && !a.uni_enb;   // the whole range in one launch, no uniqueness filter
      if (grp == 4) {
        if (lean) hipLaunchKernelGGL((fpga_bm_kernel<4, true>), grid, dim3(64), lds, s, a);
        else hipLaunchKernelGGL((fpga_bm_kernel<4, false>), grid, dim3(64), lds, s, a);
      } else if (grp == 2) {
        if (lean) hipLaunchKernelGGL((fpga_bm_kernel<2, true>), grid, dim3(64), lds, s, a);
        else hipLaunchKernelGGL((fpga_bm_kernel<2, false>), grid, dim3(64), lds, s, a);
      } else {
        if (lean) hipLaunchKernelGGL((fpga_bm_kernel<1, true>), grid, dim3(64), lds, s, a);
        else hipLaunchKernelGGL((fpga_bm_kernel<1, false>), grid, dim3(64), lds, s, a);
      }
      k += grp;
    }
  }
  return hipGetLastError();
}

}  // namespace sbm
