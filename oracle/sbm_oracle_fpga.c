/*
 * sbm_oracle_fpga.c -- CPU restatement of the reference's OWN block matcher, the FPGA RTL under
 * src/dvp/rtl/bm*.v ("flavour B" of SURVEY.md section 0 / Appendix B). TEST INFRASTRUCTURE ONLY (see sbm_oracle.h).
 *
 * PARITY UNPINNED: the reference ships the RTL's stimulus (data/ref_xsbl_{l,r}) but no disparity output of it, and no
 * Verilog simulator exists in this environment. Every function below follows the cited RTL statement by statement;
 * index mappings that the RTL only implies through pipeline timing were derived by tracing the registers cycle by
 * cycle (all block RAMs have a 1-cycle read latency: src/dvp/ip/{dpram_32_512,tdpram_340_1024}/ *.xci have no
 * output registers; the delay FIFO bm_calc_dly is a Standard_FIFO). The derivations are written next to the code.
 *
 * What the RTL computes (wdt x hgt 6-bit x-Sobel images L, R; wsz odd; ndisp a multiple of 32; hwsz = wsz >> 1):
 *   AD(y,x,D)    = |R[y][x-D] - L[y][x]|                       x in [ndisp, wdt-2], D in [-1, ndisp]
 *   HSAD(x,D)    = per-column vertical running sum over wsz rows, SATURATING at 10 bits on every add and clamped at 0
 *                  on every subtract (so it is history dependent)                 bm_calc_sad.v:103-125,450-470
 *   SAD(i,D)     = sum of HSAD over the 2*hwsz+1 columns ndisp+i .. ndisp+i+2*hwsz (16-bit limiter)  :569-605
 *   per 32-disparity phase ("dphase" k: D = 32k-1 .. 32k+32 on 34 lanes): tournament minimum with runner-up
 *                  (bm_calc_det.v), sub-pixel fraction (bm_calc_frac.v + diven.v), merge with the record of the
 *                  previous phases (bm_calc_upd.v), on the last phase the min1/min2 ratio filter (bm_calc_uni.v,
 *                  bm_calc.v:312-328) and the s11.4 output with 0xFFFF for "no disparity" (bm_obuf2.v:119-154)
 *   placement    = output sample i of SAD row r lands at row hwsz + r, column ndisp + hwsz + 1 + i (bm_obuf2.v:125,
 *                  state machine 232-262); everything else keeps the firmware's 0xFF fill (fpga.c:105-106).
 *                  NOTE the window of sample i is centred on column ndisp + hwsz + i: the RTL places the map one pixel
 *                  to the right of the window centre. Restated as is.
 */
#include <stdlib.h>
#include <string.h>

#include "sbm_oracle.h"

/* ---- diven.v: pipelined non-restoring signed divider, parameters (DW, VW, QW, MSB_INV) ------------------------ */
static uint64_t diven_update(uint64_t div, uint64_t rem, int op, int RW, int EVW) {
  /* diven.v:84-91   update = {rem[RW-2:0], ~op} + ({EVW+1{~op}} ^ {div[EVW-1:0], 1'b0}), truncated to RW bits */
  const uint64_t mRW = (1ull << RW) - 1ull, mE1 = (1ull << (EVW + 1)) - 1ull;
  const uint64_t a = ((rem << 1) | (uint64_t)(!op)) & mRW;
  uint64_t b = (div << 1) & mE1;
  if (!op) b ^= mE1;
  return (a + b) & mRW;
}

uint32_t sbmo_rtl_diven(int DW, int VW, int QW, int MSB_INV, uint32_t dividend, uint32_t divisor) {
  /* diven.v:34-44 */
  const int EXT_REM = VW - MSB_INV, EXT_DIV = DW - MSB_INV - 1;
  const int RW = EXT_DIV < 0 ? DW - EXT_DIV : DW + EXT_REM;
  const int EVW = EXT_DIV < 0 ? VW : VW + EXT_DIV;
  const uint64_t mDW = (1ull << DW) - 1ull, mVW = (1ull << VW) - 1ull, mRW = (1ull << RW) - 1ull, mQ = (1ull << QW) - 1ull;
  /* diven.v:113-123 */
  uint64_t ediv = (uint64_t)divisor & mVW;
  if (EXT_DIV > 0) ediv <<= EXT_DIV;
  uint64_t edvd = (uint64_t)dividend & mDW;
  if (EXT_DIV < 0) {
    edvd <<= -EXT_DIV;
  } else if ((edvd >> (DW - 1)) & 1ull) {
    edvd |= mRW & ~mDW; /* {EXT_REM{dividend[DW-1]}} */
  }
  const int sdiv = (int)((ediv >> (EVW - 1)) & 1ull);
  /* diven.v:129-150: op = sign(divisor) ^ sign(remainder), 1 = add, 0 = subtract */
  int op = sdiv ^ (int)((edvd >> (RW - 1)) & 1ull);
  uint64_t rem = diven_update(ediv, edvd, op, RW, EVW);
  uint64_t quot = 0;
  for (int i = 1; i <= QW; i++) { /* diven.v:153-176 */
    op = sdiv ^ (int)((rem >> (RW - 1)) & 1ull);
    rem = diven_update(ediv, rem, op, RW, EVW);
    quot = ((quot << 1) | (uint64_t)(!op)) & mQ;
  }
  return (uint32_t)((quot + (uint64_t)sdiv) & mQ); /* diven.v:178-181 */
}

/* ---- bm_calc_det.v:121-438: minimum over lanes 1..32 by a binary tournament ('<' : the lower lane wins ties), the
 * (left, centre, right) SADs of the winner travelling with it, and a "second minimum" taken from the losers of the
 * last two rounds, preferring one that is not adjacent to the winner. ------------------------------------------- */
void sbmo_rtl_det(const uint16_t sad[34], uint16_t* min1, uint16_t* min2, int* idx1, int* idx2, uint16_t* det_l,
                  uint16_t* det_r) {
  uint16_t L[32], C[32], R[32];
  int idx[32];
  int n = 16;
  for (int j = 0; j < 16; j++) { /* stages 1-2, first half: pairs (sad[2j+1], sad[2j+2]), :121-160 */
    const int c = sad[2 * j + 2] < sad[2 * j + 1];
    L[j] = c ? sad[2 * j + 1] : sad[2 * j];
    C[j] = c ? sad[2 * j + 2] : sad[2 * j + 1];
    R[j] = c ? sad[2 * j + 3] : sad[2 * j + 2];
    idx[j] = 2 * j + c;
  }
  /* stage 2 (16 -> 8) and stage 3 (8 -> 4): :162-247; the index arithmetic of :170-181 is "index of the survivor" */
  while (n > 4) {
    for (int i = 0; i < n / 2; i++) {
      const int c = C[2 * i + 1] < C[2 * i];
      const int s = 2 * i + c;
      L[i] = L[s]; C[i] = C[s]; R[i] = R[s]; idx[i] = idx[s];
    }
    n /= 2;
  }
  /* stage 4 (:249-296): 4 -> 2, the loser of each pair becomes a second-minimum candidate */
  uint16_t l4[2], c4[2], r4[2], m2_4[2];
  int i1_4[2], i2_4[2];
  for (int i = 0; i < 2; i++) {
    const int c = C[2 * i + 1] < C[2 * i];
    const int s = 2 * i + c, o = 2 * i + (1 - c);
    l4[i] = L[s]; c4[i] = C[s]; r4[i] = R[s]; i1_4[i] = idx[s];
    m2_4[i] = C[o]; i2_4[i] = idx[o];
  }
  /* stage 5 (:298-356) */
  const int c5 = c4[1] < c4[0];
  const uint16_t l5 = l4[c5], cc5 = c4[c5], r5 = r4[c5];
  const int i1_5 = i1_4[c5];
  uint16_t m2_5[2];
  int i2_5[2];
  m2_5[0] = c4[1 - c5]; i2_5[0] = i1_4[1 - c5];
  const int c52 = m2_4[1] < m2_4[0];
  m2_5[1] = m2_4[c52]; i2_5[1] = i2_4[c52];
  /* stage 6 (:358-416): 6-bit "+1" so index 31 has no upper neighbour */
  const int adj0 = (i2_5[0] == i1_5 + 1) || (i1_5 == i2_5[0] + 1);
  const int adj1 = (i2_5[1] == i1_5 + 1) || (i1_5 == i2_5[1] + 1);
  const int pick1 = ((m2_5[1] < m2_5[0]) && !adj1) || adj0;
  *min1 = cc5; *idx1 = i1_5; *det_l = l5; *det_r = r5;
  *min2 = m2_5[pick1]; *idx2 = i2_5[pick1];
}

/* ---- bm_calc_frac.v:59-173: frac = (L - R) / (2 * (max(L,R) - C)) as a signed 8-bit fraction through diven
 * #(18,18,8,17); 0 when a neighbour is below the centre; +-0.25 (0x40 / 0xC0) when the divisor is 0. ------------- */
uint8_t sbmo_rtl_frac(uint16_t c, uint16_t l, uint16_t r) {
  const uint32_t m17 = 0x1ffffu;
  const uint32_t dif_lr = ((uint32_t)l - (uint32_t)r) & m17, dif_lc = ((uint32_t)l - (uint32_t)c) & m17,
                 dif_rc = ((uint32_t)r - (uint32_t)c) & m17;
  const int cmp = l < r, neg = (int)(((dif_lc | dif_rc) >> 16) & 1u);
  const uint32_t dividend = neg ? 0u : ((((dif_lr >> 16) & 1u) << 17) | dif_lr);       /* :82-97  */
  const uint32_t divisor = ((!cmp ? dif_lc : dif_rc) << 1) & 0x3ffffu;                 /* :99-114 */
  if (divisor == 0) return !cmp ? 0xC0 : 0x40;                                          /* :151-160 */
  return (uint8_t)sbmo_rtl_diven(18, 18, 8, 17, dividend, divisor);
}

/* record kept per pixel between disparity phases: bm_calc.v:387-400 / bm_calc_upd.v:100-104 */
typedef struct {
  uint16_t min1, min2;
  uint8_t disp1, disp2, frac;
} fpga_rec;

/* ---- bm_calc_upd.v:107-209: merge the winner pair of this phase (d1 <= d2) with the stored pair (s1, s2) ------- */
static int rtl_merge(fpga_rec* s, uint16_t dmin1, uint16_t dmin2, uint8_t ddisp1, uint8_t ddisp2) {
  const int d1_lt_s1 = dmin1 < s->min1, d2_lt_s1 = dmin2 < s->min1, d1_lt_s2 = dmin1 < s->min2, d2_lt_s2 = dmin2 < s->min2;
  const int d1_adj_s1 = ddisp1 == (uint8_t)(s->disp1 + 1);
  fpga_rec o = *s;
  int upd;
  if (d1_lt_s1 && d2_lt_s1) { /* 11xx */
    o.min1 = dmin1; o.disp1 = ddisp1; o.min2 = dmin2; o.disp2 = ddisp2; upd = 1;
  } else if (d1_lt_s1 && d2_lt_s2) { /* 10x1 */
    o.min1 = dmin1; o.disp1 = ddisp1;
    o.min2 = !d1_adj_s1 ? s->min1 : dmin2; o.disp2 = !d1_adj_s1 ? s->disp1 : ddisp2; upd = 1;
  } else if (d1_lt_s1) { /* 10x0 */
    o.min1 = dmin1; o.disp1 = ddisp1;
    o.min2 = !d1_adj_s1 ? s->min1 : s->min2; o.disp2 = !d1_adj_s1 ? s->disp1 : s->disp2; upd = 1;
  } else if (d1_lt_s2 && d2_lt_s2) { /* 0x11 */
    o.min2 = !d1_adj_s1 ? dmin1 : dmin2; o.disp2 = !d1_adj_s1 ? ddisp1 : ddisp2; upd = 0;
  } else if (d1_lt_s2) { /* 0x10 */
    o.min2 = !d1_adj_s1 ? dmin1 : s->min2; o.disp2 = !d1_adj_s1 ? ddisp1 : s->disp2; upd = 0;
  } else {
    upd = 0;
  }
  *s = o;
  return upd;
}

/* ---- bm_obuf2.v:119-154: (disp, frac) -> s11.4 int16, negative or exactly zero -> 0xFFFF ------------------------ */
int16_t sbmo_rtl_pack_disparity(uint8_t disp, uint8_t frac) {
  const uint32_t disp_ext = (uint32_t)disp << 8;
  const uint32_t frac_ext = (frac & 0x80u) ? (0x1ff00u | frac) : frac;      /* 17-bit sign extension of s-1.8 */
  const uint32_t depth = (disp_ext + frac_ext) & 0x1ffffu;
  if ((depth >> 16) & 1u) return (int16_t)0xFFFF;
  if (depth == 0) return (int16_t)0xFFFF;
  const uint32_t v = ((depth >> 15) & 1u) ? (0xf000u | ((depth >> 4) & 0xfffu)) : ((depth >> 4) & 0xfffu);
  return (int16_t)(uint16_t)v;
}

/* ---- register-level configuration: struct FPGA_REG_BM, src/StereoBM/src/fpga.h:154-169, programmed at
 * fpga.c:150-160 (ImageSize = hgt<<16 | wdt, BmSetting = 0x00150040), decoded at src/dvp/rtl/bm.v:172-193. -------- */
int sbmo_fpga_regs_decode(uint32_t image_size, uint32_t bm_setting, uint32_t uni_filt_ctrl, int32_t out[7]) {
  out[0] = (int32_t)(image_size & 0x3ffu);          /* wdt   [9:0]   */
  out[1] = (int32_t)((image_size >> 16) & 0x1ffu);  /* hgt   [24:16] */
  out[2] = (int32_t)((bm_setting >> 16) & 0x1fu);   /* wsz   [20:16] */
  out[3] = (int32_t)(bm_setting & 0x1ffu);          /* ndisp [8:0]   */
  out[4] = (int32_t)((uni_filt_ctrl >> 31) & 1u);   /* uni_enb  [31] */
  out[5] = (int32_t)((uni_filt_ctrl >> 16) & 1u);   /* uni_mode [16] */
  out[6] = (int32_t)(uni_filt_ctrl & 0x3ffu);       /* uni_thr [9:0] */
  return SBM_OK;
}

int sbmo_fpga_check(int width, int height, int wsz, int ndisp) {
  if (width <= 0 || height <= 0 || width > 1023 || height > 511) return SBM_ERR_SIZE;
  if (wsz < 3 || wsz > 31 || (wsz & 1) == 0) return SBM_ERR_BLOCK_SIZE;
  if (ndisp < 32 || ndisp > 256 || (ndisp & 31)) return SBM_ERR_NUM_DISPARITIES;
  const int hwsz = wsz >> 1;
  if (width - ndisp - 1 - 2 * hwsz < 1 || height - 2 * hwsz < 1) return SBM_ERR_SIZE;
  if (((ndisp + hwsz + 1) & 31) == 0) return SBM_ERR_UNSUPPORTED; /* bm_obuf2.v:239 would never leave state 6 */
  return SBM_OK;
}

/* Block matcher on x-Sobel planes (6 significant bits: bm_calc_sad.v:375,380 take lr_din[13:8] / [5:0]).
 * xl/xr dense width*height; disp dense width*height int16 (fully written). */
int sbmo_fpga_bm(const uint8_t* xl, const uint8_t* xr, int width, int height, int wsz, int ndisp, int uni_enb,
                 int uni_mode, int uni_thr, int16_t* disp) {
  if (!xl || !xr || !disp) return SBM_ERR_NULL;
  const int st = sbmo_fpga_check(width, height, wsz, ndisp);
  if (st != SBM_OK) return st;
  const int hwsz = wsz >> 1;
  const int hsad_wdt = width - ndisp - 1;         /* bm.v:249 */
  const int sad_wdt = hsad_wdt - 2 * hwsz;        /* bm.v:252 */
  const int sad_hgt = height - 2 * hwsz;          /* bm.v:255 */
  const int nphase = ndisp >> 5;
  const size_t npix = (size_t)width * height;
  for (size_t i = 0; i < npix; i++) disp[i] = (int16_t)0xFFFF; /* fpga.c:105-106 */

  uint16_t* hsad = (uint16_t*)malloc((size_t)hsad_wdt * 34 * sizeof(uint16_t));
  fpga_rec* rec = (fpga_rec*)malloc((size_t)sad_hgt * sad_wdt * sizeof(fpga_rec));
  if (!hsad || !rec) { free(hsad); free(rec); return SBM_ERR_NOMEM; }

  for (int k = 0; k < nphase; k++) { /* bm_ibuf.v:143-189: one pass over the frame per 32 disparities */
    const int last = k == nphase - 1;
    /* Lane j of phase k compares L[x] with R[x - D], D = 32k + j - 1 (lane 0 and lane 33 are the guard lanes that
     * feed the sub-pixel neighbours). Derivation (bm_calc_sad.v:226-419): lr_din(c) = line[lr_rdaddr(c-1)];
     * l_buf(c) = L[lr_rdaddr(c-2)]; r_din(c) = R[lr_rdaddr(c-1)] in phase 0, the delay FIFO re-aligns it by 32k
     * columns in phase k; r_buf[i](c) = r_din(c-1-i); dif[0] = r_din - l_buf, dif[j] = r_buf[j-1] - l_buf.
     * HSAD column c holds L column x = ndisp + c: the first abs_r that is written (hsad_on_r[1]) was formed from
     * l_buf = L[lr_rdaddr at the end of state 2] = L[ndisp]; the last from L[wdt-2]. */
#define AD(y, c, j) abs((int)(xr[(size_t)(y) * width + (ndisp + (c)) - (32 * k + (j) - 1)] & 63) - \
                        (int)(xl[(size_t)(y) * width + (ndisp + (c))] & 63))     /* dif6/abs7, :78-101 */
    for (int out_r = 0; out_r < sad_hgt; out_r++) {
      /* bm_ibuf.v:195-286: rows 0..wsz-2 are added (op_type 0, first_line resets), then per output row
       * "add row r+wsz-1, emit" (op_type 2 / 0+3) preceded, from the second output row on, by "subtract row r-1". */
      if (out_r == 0) {
        for (int y = 0; y < wsz; y++)
          for (int c = 0; c < hsad_wdt; c++)
            for (int j = 0; j < 34; j++) {
              const int a = AD(y, c, j);
              const int v = y == 0 ? a : hsad[c * 34 + j] + a;       /* first_line: :455 */
              hsad[c * 34 + j] = (uint16_t)(v > 1023 ? 1023 : v);    /* upper_lim10: :103-113,457 */
            }
      } else {
        for (int c = 0; c < hsad_wdt; c++)
          for (int j = 0; j < 34; j++) {
            int v = hsad[c * 34 + j] - AD(out_r - 1, c, j);          /* :459-462 lower_lim10 */
            if (v < 0) v = 0;
            v += AD(out_r + wsz - 1, c, j);
            hsad[c * 34 + j] = (uint16_t)(v > 1023 ? 1023 : v);
          }
      }
      /* horizontal sliding (bm_calc_sad.v:497-605): sad += hsad[new] - hsad[old] through limit16; the first output
       * is the sum of columns 0..2*hwsz (sad_state 2), one more per cycle until column hsad_wdt-1 */
      int32_t sad[34];
      for (int j = 0; j < 34; j++) sad[j] = 0;
      for (int c = 0; c < hsad_wdt; c++) {
        for (int j = 0; j < 34; j++) {
          int32_t v = sad[j] + hsad[c * 34 + j] - (c >= wsz ? hsad[(c - wsz) * 34 + j] : 0);
          sad[j] = v < 0 ? 0 : (v > 65535 ? 65535 : v);              /* limit16: :127-142 */
        }
        if (c < 2 * hwsz) continue;
        const int i = c - 2 * hwsz;
        uint16_t s16[34];
        for (int j = 0; j < 34; j++) s16[j] = (uint16_t)sad[j];
        uint16_t min1, min2, dl, dr;
        int idx1, idx2;
        sbmo_rtl_det(s16, &min1, &min2, &idx1, &idx2, &dl, &dr);
        const uint8_t frac_new = sbmo_rtl_frac(min1, dl, dr);
        const uint8_t ddisp1 = (uint8_t)(((k & 7) << 5) | idx1), ddisp2 = (uint8_t)(((k & 7) << 5) | idx2); /* upd.v:121-124 */
        fpga_rec* rc = &rec[(size_t)out_r * sad_wdt + i];
        if (k == 0) { /* mode 0: "initial SAD", bm_calc_upd.v:147-154 */
          rc->min1 = min1; rc->min2 = min2; rc->disp1 = ddisp1; rc->disp2 = ddisp2; rc->frac = frac_new;
        } else if (rtl_merge(rc, min1, min2, ddisp1, ddisp2)) {
          rc->frac = frac_new; /* bm_calc.v:309 uni_frac_upd */
        }
        if (last) {
          /* bm_calc_uni.v:117-134: ratio = min1 / min2 through diven #(17,17,11,16), low 10 bits; bm_calc.v:312-328 */
          uint8_t od = rc->disp1, of = rc->frac;
          if (uni_enb) {
            const uint32_t ratio = sbmo_rtl_diven(17, 17, 11, 16, rc->min1, rc->min2) & 0x3ffu;
            if (ratio > (uint32_t)(uni_thr & 0x3ff)) od = of = uni_mode ? 0xFF : 0x00;
          }
          disp[(size_t)(hwsz + out_r) * width + (ndisp + hwsz + 1 + i)] = sbmo_rtl_pack_disparity(od, of);
        }
      }
    }
#undef AD
  }
  free(hsad);
  free(rec);
  return SBM_OK;
}

/* x-Sobel (RTL flavour, rows 0 and H-1 = 0 as in the testbench) of both images, then the matcher. */
int sbmo_fpga_compute(const uint8_t* left, const uint8_t* right, int width, int height, int wsz, int ndisp, int uni_enb,
                      int uni_mode, int uni_thr, int16_t* disp) {
  if (!left || !right || !disp) return SBM_ERR_NULL;
  const int st = sbmo_fpga_check(width, height, wsz, ndisp);
  if (st != SBM_OK) return st;
  const size_t npix = (size_t)width * height;
  uint8_t* xl = (uint8_t*)malloc(npix);
  uint8_t* xr = (uint8_t*)malloc(npix);
  if (!xl || !xr) { free(xl); free(xr); return SBM_ERR_NOMEM; }
  sbmo_prefilter_xsobel_fpga(left, (size_t)width, xl, (size_t)width, width, height, 0);
  sbmo_prefilter_xsobel_fpga(right, (size_t)width, xr, (size_t)width, width, height, 0);
  const int r = sbmo_fpga_bm(xl, xr, width, height, wsz, ndisp, uni_enb, uni_mode, uni_thr, disp);
  free(xl);
  free(xr);
  return r;
}

/* ==== GFTT minimum-eigenvalue map of the PL: src/dvp/rtl/gftt_sbl.v, gftt_box.v, gftt_eig.v,
 * placement and `max` register gftt_obuf.v, consumer src/slam/src/core/GFTT.cpp:41-170 (reads the map as CV_16UC1 plus the
 * register value, FPGA.cpp:283-291). PARITY UNPINNED (no golden map in the reference, no simulator), and the square root is
 * a Xilinx CORDIC core (src/dvp/ip/gftt_sqrt/gftt_sqrt.xci: Square_Root, UnsignedFraction, 32 -> 17 bits, Truncate) whose
 * last bit is not specified: this restatement takes the exact floor and the GPU test allows +-1 LSB on the map.
 *
 * Derived from the pipeline timing (all line RAMs have 1-cycle reads):
 *   Sobel (gftt_sbl.v:113-204), defined for image rows 1..H-2: sample t of a row is centred on column t,
 *     dx = sum_k w_k (p[y+k][x+1] - p[y+k][x-1]), dy = sum_j w_j (p[y+1][x+j] - p[y-1][x+j]), w = (1,2,1); both forced to 0
 *     at x = 0 and x = W-1 (first_r / last_r).
 *   products (gftt_eig.v:122-143): |dx|^2 >> 6, |dy|^2 >> 6, |dx|*|dy| >> 6  (absolute values: the sign of dx*dy is dropped)
 *   box (gftt_box.v): horizontal 3-sum centred on the sample, forced to 0 in the first and last column, then the sum
 *     of three consecutive Sobel rows centred on the middle one, saturated to 16 bits -> a, c, b; defined for rows 2..H-3
 *   eig (gftt_eig.v:226-362): (a + c) - sqrt(((a-c)^2 >> 10) + (b^2 >> 8)), the radicand saturated to 22 bits; the CORDIC
 *     result is floor(sqrt(radicand << 10)) in the units of a + c; negative -> 0, above 16 bits -> 0xFFFF
 *   output (gftt_obuf.v:295-305): rows 2..H-3 of a dense H x W uint16 map, every other row keeps the firmware's 0 fill
 *     (fpga.c:107-108); max = largest value written (gftt_obuf.v:101-130). */
static uint32_t isqrt64(uint64_t n) {
  uint64_t r = 0, bit = 1ull << 62;
  while (bit > n) bit >>= 2;
  while (bit) {
    if (n >= r + bit) { n -= r + bit; r = (r >> 1) + bit; } else { r >>= 1; }
    bit >>= 2;
  }
  return (uint32_t)r;
}

int sbmo_gftt_eig(const uint8_t* img, int width, int height, uint16_t* eig, uint32_t* max_out) {
  if (!img || !eig) return SBM_ERR_NULL;
  if (width < 3 || height < 5 || width > 1023 || height > 511) return SBM_ERR_SIZE;
  const int W = width, H = height;
  memset(eig, 0, (size_t)W * H * sizeof(uint16_t));
  uint32_t* hs = (uint32_t*)calloc((size_t)3 * W * H, sizeof(uint32_t));   /* horizontal 3-sums of the three products */
  if (!hs) return SBM_ERR_NOMEM;
  uint32_t* v = (uint32_t*)calloc((size_t)3 * W, sizeof(uint32_t));
  if (!v) { free(hs); return SBM_ERR_NOMEM; }
  for (int y = 1; y <= H - 2; y++) {
    const uint8_t *r0 = img + (size_t)(y - 1) * W, *r1 = img + (size_t)y * W, *r2 = img + (size_t)(y + 1) * W;
    for (int x = 0; x < W; x++) {
      int dx = 0, dy = 0;
      if (x > 0 && x < W - 1) {
        dx = (r0[x + 1] - r0[x - 1]) + 2 * (r1[x + 1] - r1[x - 1]) + (r2[x + 1] - r2[x - 1]);
        dy = (r2[x - 1] - r0[x - 1]) + 2 * (r2[x] - r0[x]) + (r2[x + 1] - r0[x + 1]);
      }
      const uint32_t ax = (uint32_t)abs(dx), ay = (uint32_t)abs(dy);
      v[x] = ((ax * ax) >> 6) & 0xffffu; v[W + x] = ((ay * ay) >> 6) & 0xffffu; v[2 * W + x] = ((ax * ay) >> 6) & 0xffffu;
    }
    for (int k = 0; k < 3; k++)
      for (int x = 1; x < W - 1; x++) hs[((size_t)k * H + y) * W + x] = v[k * W + x - 1] + v[k * W + x] + v[k * W + x + 1];
  }
  uint32_t mx = 0;
  for (int y = 2; y <= H - 3; y++)
    for (int x = 0; x < W; x++) {
      uint32_t abc[3];
      for (int k = 0; k < 3; k++) {
        const uint32_t s = hs[((size_t)k * H + y - 1) * W + x] + hs[((size_t)k * H + y) * W + x] + hs[((size_t)k * H + y + 1) * W + x];
        abc[k] = s > 0xffffu ? 0xffffu : s;                                   /* gftt_box.v: lim */
      }
      const uint32_t a = abc[0], c = abc[1], b = abc[2];
      const uint32_t apc = a + c, amc = a > c ? a - c : c - a;
      const uint32_t amc2 = (uint32_t)(((uint64_t)amc * amc) >> 10) & 0x3fffffu;   /* [31:10] */
      const uint32_t b2 = (uint32_t)(((uint64_t)b * b) >> 8) & 0xffffffu;           /* [31:8]  */
      uint32_t s = amc2 + b2;
      if (s > 0x3fffffu) s = 0x3fffffu;
      const uint32_t root = isqrt64((uint64_t)s << 10) & 0xffffu;                 /* CORDIC 17-bit result, low 16 bits used */
      const int32_t e = (int32_t)apc - (int32_t)root;
      const uint32_t out = e < 0 ? 0u : (e > 0xffff ? 0xffffu : (uint32_t)e);
      eig[(size_t)y * W + x] = (uint16_t)out;
      if (out > mx) mx = out;
    }
  if (max_out) *max_out = mx;
  free(v);
  free(hs);
  return SBM_OK;
}
