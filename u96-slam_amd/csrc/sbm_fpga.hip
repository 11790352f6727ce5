// sbm_fpga.hip -- the reference's OWN block matcher (FPGA RTL, "flavour B" of SURVEY.md) on gfx950:
// 6-bit x-Sobel inputs, 10-bit saturating column sums, 32-disparity phases with record merge, tournament minimum with
// runner-up, non-restoring-divider sub-pixel fraction, optional min1/min2 ratio filter, s11.4 output with 0xFFFF.
//
// Device counterpart of src/dvp/rtl/bm_calc_sad.v:78-149,350-612 (abs-diff, HSAD, SAD), bm_calc_det.v:121-438,
// bm_calc_frac.v:59-173 + diven.v, bm_calc_upd.v:107-209, bm_calc_uni.v:117-134 / bm_calc.v:312-328, bm_obuf2.v:119-154,
// scheduled like bm_ibuf.v:143-286 (one pass over the frame per 32 disparities, 8-byte records in between). Index
// mappings (lane j of phase k <-> disparity 32k+j-1, HSAD column c <-> image column ndisp+c, output sample i at column
// ndisp+hwsz+1+i) are derived in DESIGN.md (section "FPGA flavour").
//
// Mapping: a wavefront owns 64 consecutive HSAD columns (lane = column) and marches down the rows of one segment.
//   AD      one v_mqsad_pk_u16_u8 with a single-byte pattern = |R[t..t+3] - L| for 4 consecutive disparities
//           (9 per row cover the 34 lanes of a phase); the leaving row is re-evaluated instead of stored
//   HSAD    18 VGPRs of packed u16: v_pk_add_u16 + v_pk_min_u16(.., 1023) on entry, v_pk_sub_u16 clamp on exit
//   SAD     inclusive prefix sum over the 64 lanes (4 DPP row shifts + 2 row broadcasts per register; 64 * 1023 still
//           fits 16 bits), window sum = P[l + 2*hwsz] - P[l - 1] through LDS
//   det     key minimum per quarter of the tournament, the top two rounds literally
// The 10-bit saturation makes HSAD history dependent, so row segments are only exact while nothing saturates: the
// segmented launch raises a per-pair flag when a column sum passes 1023 and a second, one-segment launch (which exits
// immediately when the flag is clear) recomputes such pairs strictly top to bottom.
#include <algorithm>

#include "sbm_common.h"

namespace sbm {

typedef unsigned int u32;
typedef unsigned long long u64;
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

struct FpgaArgs {
  const uint8_t* xl;     // padded x-Sobel planes: row pitch `pitch`, column 0 at byte `padl`
  const uint8_t* xr;
  uint2* rec;            // [pair][sad_hgt][sad_wdt] records {min1 | min2 << 16, disp1 | disp2 << 8 | frac << 16}
  int16_t* disp;         // dense output, pre-filled with 0xFFFF
  int* flag;             // [pair] saturation seen
  int W, H, pitch, padl, plane;
  int nd, wsz, hwsz, hsad_wdt, sad_wdt, sad_hgt;
  int phase, last;
  int uni_enb, uni_mode, uni_thr;
  int seg;               // output rows per segment
  int exact;             // 1: the one-segment launch that only runs for flagged pairs
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

__global__ void __launch_bounds__(64) fpga_bm_kernel(FpgaArgs a) {
  const int lane = threadIdx.x;
  const int pair = blockIdx.z;
  if (a.exact && a.flag[pair] == 0) return;
  const int NV = 64 - 2 * a.hwsz;                 // outputs per strip
  const int c = blockIdx.x * NV + lane;           // HSAD column of this lane
  const bool col_ok = c < a.hsad_wdt;
  const int x = a.nd + min(c, a.hsad_wdt - 1);    // image column (clamped for the idle lanes: their sums are never used)
  const int r0 = blockIdx.y * a.seg, r1 = min(r0 + a.seg, a.sad_hgt);
  if (r0 >= r1) return;
  const uint8_t* pl = a.xl + (size_t)pair * a.plane + a.padl + x;
  // window byte t of a lane = R[x - 32k - 33 + t]: t = 34 - j for lane j (disparity 32k + j - 1)
  const uint8_t* pr = a.xr + (size_t)pair * a.plane + a.padl + x - 32 * a.phase - 33;
  u32* const xrow = fpga_lds;                     // [FP_NR][FP_XS]

  u32 V[FP_NR];
#pragma unroll
  for (int r = 0; r < FP_NR; r++) V[r] = 0u;
  u32 satacc = 0u;

  // |R[t] - L| for the 36 window bytes of row y, packed like V
  auto row_ad = [&](int y, u32 (&ad)[FP_NR]) {
    const uint8_t* rr = pr + (size_t)y * a.pitch;
    u32 w[10];
    uint4 q0, q1;
    uint2 q2;
    __builtin_memcpy(&q0, rr, 16);
    __builtin_memcpy(&q1, rr + 16, 16);
    __builtin_memcpy(&q2, rr + 32, 8);
    w[0] = q0.x; w[1] = q0.y; w[2] = q0.z; w[3] = q0.w; w[4] = q1.x; w[5] = q1.y; w[6] = q1.z; w[7] = q1.w; w[8] = q2.x; w[9] = q2.y;
    const u32 l = (u32)(pl[(size_t)y * a.pitch] & 63) + 1u;   // +1: a zero pattern byte would be masked by the instruction
#pragma unroll
    for (int q = 0; q < 9; q++) {
      // 6 significant bits (bm_calc_sad.v:375,380); the +1 bias cancels in the difference
      const u32 wq = (w[q] & 0x3f3f3f3fu) + 0x01010101u;
      const u64 win = ((u64)w[q + 1] << 32) | wq;               // only bytes 0..3 meet a non-masked pattern byte
      const u64 m = __builtin_amdgcn_mqsad_pk_u16_u8(win, l, 0ull);
      ad[2 * q] = (u32)m;
      ad[2 * q + 1] = (u32)(m >> 32);
    }
  };
  auto add_row = [&](int y) {                     // bm_calc_sad.v:450-457: + |.|, upper limit 1023
    u32 ad[FP_NR];
    row_ad(y, ad);
#pragma unroll
    for (int r = 0; r < FP_NR; r++) {
      const u32 s = pk_add(V[r], ad[r]);
      const u32 m = pk_min(s, 0x03ff03ffu);
      satacc |= s ^ m;
      V[r] = m;
    }
  };
  auto sub_row = [&](int y) {                     // bm_calc_sad.v:459-462: - |.|, lower limit 0
    u32 ad[FP_NR];
    row_ad(y, ad);
#pragma unroll
    for (int r = 0; r < FP_NR; r++) V[r] = pk_subs(V[r], ad[r]);
  };

  for (int y = r0; y < r0 + a.wsz - 1; y++) add_row(y);          // rows of the first window but the last

  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < FP_NR; r++) xrow[r * FP_XS] = 0u;
  }
  const int k32 = (a.phase & 7) << 5;
  for (int r = r0; r < r1; r++) {
    add_row(r + a.wsz - 1);
    // ---- SAD: prefix sums over the lanes, window = P[l + 2*hwsz] - P[l - 1] (bm_calc_sad.v:569-605) -----------------
    u32 S[FP_NR];
#pragma unroll
    for (int q = 0; q < FP_NR; q++) {
      u32 p = V[q];
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
#pragma unroll
    for (int q = 0; q < FP_NR; q++) S[q] = xrow[q * FP_XS + 1 + lane + 2 * a.hwsz] - xrow[q * FP_XS + lane];

    const int i = c;                               // output sample of this lane
    if (lane < NV && i < a.sad_wdt) {
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
      const u32 ddisp1 = (u32)(k32 | i1), ddisp2 = (u32)(k32 | (int)(m2k & 0xffffu));
      // neighbours of the winner lane jw = i1 + 1: lanes jw - 1 and jw + 1, looked up in the prefix rows
      auto sad_of_lane = [&](int j) -> u32 {
        const int q = (34 - j) >> 1, hi = (34 - j) & 1;   // lane 34-2q is the low half
        const u32 pa = xrow[q * FP_XS + 1 + lane + 2 * a.hwsz], pb = xrow[q * FP_XS + lane];
        const u32 d = pa - pb;
        return hi ? (d >> 16) : (d & 0xffffu);
      };
      const u32 dl = sad_of_lane(i1), dr = sad_of_lane(i1 + 2);
      const u32 frac_new = rtl_frac(dmin1, dl, dr);

      uint2* rp = a.rec + ((size_t)pair * a.sad_hgt + r) * a.sad_wdt + i;
      u32 min1, min2, disp1, disp2, frac;
      if (a.phase == 0) {                          // bm_calc_upd.v:147-154
        min1 = dmin1; min2 = dmin2; disp1 = ddisp1; disp2 = ddisp2; frac = frac_new;
      } else {                                     // bm_calc_upd.v:125-209
        const uint2 rc = *rp;
        const u32 s1 = rc.x & 0xffffu, s2 = rc.x >> 16, sd1 = rc.y & 0xffu, sd2 = (rc.y >> 8) & 0xffu;
        frac = (rc.y >> 16) & 0xffu;
        const bool d1s1 = dmin1 < s1, d2s1 = dmin2 < s1, d1s2 = dmin1 < s2, d2s2 = dmin2 < s2;
        const bool adj = ddisp1 == ((sd1 + 1u) & 0xffu);
        min1 = s1; disp1 = sd1; min2 = s2; disp2 = sd2;
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
      if (!a.last) {
        *rp = make_uint2(min1 | (min2 << 16), disp1 | (disp2 << 8) | (frac << 16));
      } else {
        u32 od = disp1, of = frac;
        if (a.uni_enb) {                           // bm_calc_uni.v:117-134, bm_calc.v:312-328
          const u32 ratio = rtl_diven<17, 11>(min1, min2) & 0x3ffu;
          if (ratio > (u32)(a.uni_thr & 0x3ff)) od = of = a.uni_mode ? 0xffu : 0x00u;
        }
        a.disp[((size_t)pair * a.H + a.hwsz + r) * a.W + a.nd + a.hwsz + 1 + i] = (int16_t)rtl_pack(od, of);
      }
    }
    __builtin_amdgcn_wave_barrier();               // the prefix rows are rewritten by the next output row
    if (r + 1 < r1) sub_row(r);
  }
  if (!a.exact && col_ok && satacc) atomicOr(a.flag + pair, 1);
}

// ---- dense plane -> padded plane (row pitch, left pad) so that the window loads never leave the allocation -----------
__global__ void __launch_bounds__(256) fpga_pad_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int W, int H,
                                                       int pitch, int padl, int plane) {
  const int y = blockIdx.x, img = blockIdx.y;
  const uint8_t* s = src + ((size_t)img * H + y) * W;
  uint8_t* d = dst + (size_t)img * plane + (size_t)y * pitch;
  for (int xx = threadIdx.x; xx < pitch; xx += 256) {
    const int sx = xx - padl;
    d[xx] = (sx >= 0 && sx < W) ? s[sx] : 0;
  }
}

int fpga_pitch(int W) { return ((W + 64 + 64 + 63) / 64) * 64; }

// xl/xr: dense n*H*W x-Sobel planes on the device. pad_l/pad_r: scratch of n * fpga_pitch(W) * H + 64 bytes each,
// rec: n*sad_hgt*sad_wdt uint2, flag: n ints.
hipError_t launch_fpga_bm(const uint8_t* xl, const uint8_t* xr, uint8_t* pad_l, uint8_t* pad_r, void* rec, int* flag,
                          int16_t* disp, int n, const sbm_fpga_params& p, hipStream_t s) {
  FpgaArgs a;
  const int W = p.width, H = p.height;
  a.W = W; a.H = H; a.pitch = fpga_pitch(W); a.padl = 64; a.plane = a.pitch * H;
  a.nd = p.num_disparities; a.wsz = p.block_size; a.hwsz = p.block_size >> 1;
  a.hsad_wdt = W - a.nd - 1; a.sad_wdt = a.hsad_wdt - 2 * a.hwsz; a.sad_hgt = H - 2 * a.hwsz;
  a.uni_enb = p.uni_enable; a.uni_mode = p.uni_mode; a.uni_thr = p.uni_threshold;
  a.xl = pad_l; a.xr = pad_r; a.rec = static_cast<uint2*>(rec); a.disp = disp; a.flag = flag;
  hipError_t e;
  if ((e = hipMemsetAsync(disp, 0xFF, (size_t)n * W * H * sizeof(int16_t), s)) != hipSuccess) return e;   // fpga.c:105-106
  if ((e = hipMemsetAsync(flag, 0, (size_t)n * sizeof(int), s)) != hipSuccess) return e;
  hipLaunchKernelGGL(fpga_pad_kernel, dim3(H, n), dim3(256), 0, s, xl, pad_l, W, H, a.pitch, a.padl, a.plane);
  hipLaunchKernelGGL(fpga_pad_kernel, dim3(H, n), dim3(256), 0, s, xr, pad_r, W, H, a.pitch, a.padl, a.plane);
  const int NV = 64 - 2 * a.hwsz;
  const int strips = (a.sad_wdt + NV - 1) / NV;
  const int nphase = a.nd >> 5;
  // segments: enough wavefronts for the chip, each paying wsz-1 priming rows
  int nseg = 1;
  while ((long)strips * nseg * n < 4096 && a.sad_hgt / (nseg + 1) >= 2 * a.wsz) nseg++;
  const size_t lds = (size_t)FP_NR * FP_XS * sizeof(u32);
  for (int pass = 0; pass < 2; pass++) {         // 0: segmented (exact unless a column sum saturates), 1: flagged pairs, one segment
    a.exact = pass;
    const int ns = pass ? 1 : nseg;
    a.seg = (a.sad_hgt + ns - 1) / ns;
    if (pass == 1 && nseg == 1) break;            // the first pass already ran top to bottom
    for (int k = 0; k < nphase; k++) {
      a.phase = k; a.last = k == nphase - 1;
      hipLaunchKernelGGL(fpga_bm_kernel, dim3(strips, (a.sad_hgt + a.seg - 1) / a.seg, n), dim3(64), lds, s, a);
    }
  }
  return hipGetLastError();
}

}  // namespace sbm
