/*
 * sbm_oracle.c -- CPU restatement of cv::StereoBM (OpenCV calib3d 4.x, version unpinned by the reference)
 * and of the reference's FPGA x-Sobel. TEST INFRASTRUCTURE ONLY: see sbm_oracle.h for who may call this
 * and for the pinning status ("parity unpinned" for the block-matching output, pinned for the prefilter).
 *
 * Plain C11, integer arithmetic only, no dependencies. Compiled by oracle/Makefile into
 * oracle/libsbm_oracle.so. Nothing here is derived from /root/reference source text; the OpenCV routines
 * are restated from their published algorithm (SURVEY.md Appendix A records the recall and its risks).
 */
#include "sbm_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int iabs(int a) { return a < 0 ? -a : a; }
static inline int iclamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* Alternative readings of the OpenCV behaviours this restatement recalls from memory and nothing in the reference can pin
 * (SURVEY.md A.7 + getValidDisparityROI): one bit each, 0 = the reading the oracle and the engine default to. Process-global,
 * set by the tests / tools/pin_kit.py only (sbm_oracle.h: SBMO_READ_*). The engine has the same bits (SBM_CV_READING). */
static int g_reading = 0;
void sbmo_set_reading(int mask) { g_reading = mask; }
int sbmo_get_reading(void) { return g_reading; }

/* 1: sbmo_compute / sbmo_compute_batch take the u16-vectorised correspondence of sbm_oracle_simd.c where it applies (every window
 * sum fits 16 bits) -- bench.py's second cpu_baseline figure; the CHECKER stays the scalar restatement (default 0). */
static int g_simd = 0;
void sbmo_set_simd(int on) { g_simd = on; }
int sbmo_get_simd(void) { return g_simd; }

/* ------------------------------------------------------------------------------------------------
 * Prefilter, OpenCV flavour.  Follows prefilterXSobel() in OpenCV calib3d stereobm.cpp:
 *   tab(v) = v < -cap ? 0 : v > cap ? 2*cap : v + cap
 *   rows handled in pairs y, y+1 while y < H-1; the row above the first / below the last is mirrored
 *   (reflect-101); columns 0 and W-1 get tab(0) = cap; a leftover last row (odd H) is filled with cap.
 * Call site that fixes cap = 31: src/slam/src/core/main.cpp:204.
 * ------------------------------------------------------------------------------------------------ */
static inline uint8_t xsobel_tab(int v, int cap) { return (uint8_t)(v < -cap ? 0 : (v > cap ? 2 * cap : v + cap)); }

void sbmo_prefilter_xsobel(const uint8_t* src, size_t sstride, uint8_t* dst, size_t dstride, int width, int height,
                           int cap) {
  const uint8_t val0 = xsobel_tab(0, cap);
  int y = 0;
  for (; y < height - 1; y += 2) {
    const uint8_t* r1 = src + (size_t)y * sstride;
    const uint8_t* r0 = y > 0 ? r1 - sstride : (height > 1 ? r1 + sstride : r1);
    const uint8_t* r2 = y < height - 1 ? r1 + sstride : (height > 1 ? r1 - sstride : r1);
    const uint8_t* r3 = y < height - 2 ? r1 + 2 * sstride : r1;
    uint8_t* d0 = dst + (size_t)y * dstride;
    uint8_t* d1 = d0 + dstride;
    d0[0] = d0[width - 1] = d1[0] = d1[width - 1] = val0;
    for (int x = 1; x < width - 1; x++) {
      int g0 = r0[x + 1] - r0[x - 1], g1 = r1[x + 1] - r1[x - 1];
      int g2 = r2[x + 1] - r2[x - 1], g3 = r3[x + 1] - r3[x - 1];
      d0[x] = xsobel_tab(g0 + 2 * g1 + g2, cap);
      d1[x] = xsobel_tab(g1 + 2 * g2 + g3, cap);
    }
  }
  for (; y < height; y++) {
    if ((g_reading & SBMO_READ_ODD_ROW_COMPUTED) && height > 1) {   /* alternative: the leftover row like any other (reflect-101 below) */
      const uint8_t* r1 = src + (size_t)y * sstride;
      const uint8_t* r0 = r1 - sstride;
      uint8_t* d0 = dst + (size_t)y * dstride;
      d0[0] = d0[width - 1] = val0;
      for (int x = 1; x < width - 1; x++) {
        int g0 = r0[x + 1] - r0[x - 1], g1 = r1[x + 1] - r1[x - 1];
        d0[x] = xsobel_tab(g0 + 2 * g1 + g0, cap);
      }
      continue;
    }
    memset(dst + (size_t)y * dstride, val0, (size_t)width);
  }
}

/* ------------------------------------------------------------------------------------------------
 * Prefilter, FPGA flavour.  Follows src/dvp/rtl/xsbl2.v:
 *   horizontal difference p[x+1]-p[x-1]            (684-698)
 *   s = dif(y-1) + 2*dif(y) + dif(y+1)              (two line buffers 787-809, add_a/add_b 826-857)
 *   limit(): clip s to [-32,31], emit offset binary (+32)                     (185-198)
 *   first and last pixel of each line forced to 0x20                            (869-872)
 *   output row index is hcnt-1, so rows 0 and H-1 are never written             (1020-1025)
 * ------------------------------------------------------------------------------------------------ */
void sbmo_prefilter_xsobel_fpga(const uint8_t* src, size_t sstride, uint8_t* dst, size_t dstride, int width,
                                int height, uint8_t fill) {
  for (int y = 0; y < height; y++) {
    uint8_t* d = dst + (size_t)y * dstride;
    if (y == 0 || y == height - 1) {
      memset(d, fill, (size_t)width);
      continue;
    }
    const uint8_t* a = src + (size_t)(y - 1) * sstride;
    const uint8_t* b = src + (size_t)y * sstride;
    const uint8_t* c = src + (size_t)(y + 1) * sstride;
    d[0] = d[width - 1] = 0x20;
    for (int x = 1; x < width - 1; x++) {
      int s = (a[x + 1] - a[x - 1]) + 2 * (b[x + 1] - b[x - 1]) + (c[x + 1] - c[x - 1]);
      d[x] = (uint8_t)(iclamp(s, -32, 31) + 32);
    }
  }
}

/* ------------------------------------------------------------------------------------------------
 * getValidDisparityROI (OpenCV calib3d stereosgbm.cpp). The reference passes empty rects
 * (main.cpp:200-203); cv::StereoBM::compute substitutes the whole image for an empty rect.
 * ------------------------------------------------------------------------------------------------ */
void sbmo_valid_roi(const int32_t roi1[4], const int32_t roi2[4], int min_disparity, int num_disparities,
                    int block_size, int32_t out[4]) {
  int sw2 = block_size / 2;
  int maxd = min_disparity + num_disparities - 1;
  int xmin = imax(roi1[0], roi2[0] + maxd) + sw2;
  int xmax = imin(roi1[0] + roi1[2], roi2[0] + roi2[2] - ((g_reading & SBMO_READ_ROI_MINUS_MIND) ? min_disparity : 0)) - sw2;
  int ymin = imax(roi1[1], roi2[1]) + sw2;
  int ymax = imin(roi1[1] + roi1[3], roi2[1] + roi2[3]) - sw2;
  int w = xmax - xmin, h = ymax - ymin;
  if (w > 0 && h > 0) {
    out[0] = xmin; out[1] = ymin; out[2] = w; out[3] = h;
  } else {
    out[0] = out[1] = out[2] = out[3] = 0;
  }
}

/* dispDescale<short>() of stereobm.cpp: ((v1*256 + (d ? v2*256/d : 0) + 15) >> 4), C truncating division. */
static inline int16_t disp_descale(int v1, int v2, int d) { return (int16_t)((v1 * 256 + (d != 0 ? v2 * 256 / d : 0) + 15) >> 4); }

/* ------------------------------------------------------------------------------------------------
 * findStereoCorrespondenceBM (OpenCV calib3d stereobm.cpp), scalar formulation, for the stripe
 * [row0,row1) exactly as FindStereoCorrespInvoker hands it over: the stripe sees dy0 = min(row0, w/2+1)
 * real rows above it and dy1 = min(H-row1, w/2+1) below; rows further out are replaced by the nearest
 * available row of column sums (never reached for rows inside the valid ROI).
 *
 *   buffer index d in [0,nd)  <->  true disparity D = nd-1-d+mindisp
 *   left  sample column for window column x': lofs + clamp(x', -lofs, W-lofs-1)
 *   right sample column for window column x' and index d: rofs + clamp(x', -rofs, W-rofs-nd) + d
 *   hsad[y][d]   = sum over the w window columns of |L - R|   (kept per stripe row, slides along x)
 *   sad[d]       = sum over the w window rows of hsad         (slides along y)
 *   htext[y]     = sum over the w window columns of |L - cap| ; tsum slides along y
 *   WTA: first d attaining the minimum (strict '<' scan)  => ties pick the largest true disparity
 *   texture: tsum < textureThreshold -> FILTERED
 *   uniqueness: exists d outside [mind-1,mind+1] with sad[d] <= minsad + minsad*ratio/100 -> FILTERED
 *   sub-pixel: p = sad[mind+1], n = sad[mind-1] (mirrored at the ends), den = p+n-2c+|p-n|
 * ------------------------------------------------------------------------------------------------ */
void sbmo_find_correspondence(const uint8_t* left_full, const uint8_t* right_full, size_t stride, int width,
                              int height_full, int row0, int row1, const sbm_params* p, int16_t* disp_full,
                              size_t dstride, int32_t* cost_full, size_t cstride) {
  const int wsz = p->block_size, wsz2 = wsz / 2;
  const int ndisp = p->num_disparities, mindisp = p->min_disparity;
  const int height = row1 - row0;
  if (height <= 0) return;
  const int dy0 = imin(row0, wsz2 + 1), dy1 = imin(height_full - row1, wsz2 + 1);
  const int lofs = imax(ndisp - 1 + mindisp, 0);
  const int rofs = -imin(ndisp - 1 + mindisp, 0);
  const int width1 = width - rofs - ndisp + 1;
  const int ftzero = p->prefilter_cap;
  const int16_t FILTERED = (int16_t)((mindisp - 1) * 16);   /* (cv writes (minDisparity - 1) << 4; a negative left shift is UB in C) */

  const uint8_t* lbase = left_full + (size_t)row0 * stride + lofs;
  const uint8_t* rbase = right_full + (size_t)row0 * stride + rofs;
  int16_t* dptr = disp_full + (size_t)row0 * dstride;
  int32_t* cptr = cost_full ? cost_full + (size_t)row0 * cstride : NULL;

  const int nrows = height + dy0 + dy1;                 /* stripe rows incl. the real halo rows   */
  int* sad = (int*)malloc(sizeof(int) * (size_t)(ndisp + 2));
  int* hsad_store = (int*)calloc((size_t)nrows * ndisp, sizeof(int));
  int* htext_store = (int*)calloc((size_t)(height + wsz + 2), sizeof(int));
  uint8_t* cbuf_store = (uint8_t*)malloc((size_t)(wsz + 1) * nrows * ndisp);
  uint8_t tab[256];
  int* hsad0 = hsad_store + (size_t)dy0 * ndisp;        /* hsad0[y*ndisp + d], y in [-dy0, height+dy1) */
  int* htext = htext_store + wsz2 + 1;                  /* htext[y], y in [-wsz2-1, height+wsz2]        */
  int* sadp = sad + 1;                                  /* sadp[-1] and sadp[ndisp] are the mirrors     */
  const size_t cstep = (size_t)nrows * ndisp;
  uint8_t* cbuf0 = cbuf_store + (size_t)dy0 * ndisp;    /* column ring: cbuf0[slot*cstep + y*ndisp + d] */

  for (int x = 0; x < 256; x++) tab[x] = (uint8_t)iabs(x - ftzero);

  /* prime the ring with window columns x' = -wsz2-1 .. wsz2-1 */
  for (int x = -wsz2 - 1; x < wsz2; x++) {
    int* hs = hsad0 - dy0 * ndisp;
    uint8_t* cb = cbuf0 + (size_t)(x + wsz2 + 1) * cstep - (size_t)dy0 * ndisp;
    const uint8_t* lp = lbase + iclamp(x, -lofs, width - lofs - 1) - (ptrdiff_t)dy0 * (ptrdiff_t)stride;
    const uint8_t* rp = rbase + iclamp(x, -rofs, width - rofs - ndisp) - (ptrdiff_t)dy0 * (ptrdiff_t)stride;
    for (int y = -dy0; y < height + dy1; y++, hs += ndisp, cb += ndisp, lp += stride, rp += stride) {
      int lval = lp[0];
      for (int d = 0; d < ndisp; d++) {
        int diff = iabs(lval - rp[d]);
        cb[d] = (uint8_t)diff;
        hs[d] += diff;
      }
      htext[y] += tab[lval];
    }
  }

  /* columns that can never be matched */
  for (int y = 0; y < height; y++) {
    for (int x = 0; x < lofs; x++) dptr[(size_t)y * dstride + x] = FILTERED;
    for (int x = lofs + width1; x < width; x++) dptr[(size_t)y * dstride + x] = FILTERED;
  }

  /* For minDisparity > 0 OpenCV's loop bound width1 = W-rofs-nd+1 runs lofs+x past the last column (its
   * stores spill into the never-valid first columns of the next row and are erased by the ROI fill); the
   * defined part of that behaviour is "columns >= W are not produced", which is what is restated here. */
  const int xend = imin(width1, width - lofs);
  for (int x = 0; x < xend; x++) {
    int16_t* dcol = dptr + lofs + x;
    int32_t* ccol = cptr ? cptr + lofs + x : NULL;
    int x0 = x - wsz2 - 1, x1 = x + wsz2;
    const uint8_t* cb_sub = cbuf0 + (size_t)((x0 + wsz2 + 1) % (wsz + 1)) * cstep - (size_t)dy0 * ndisp;
    uint8_t* cb = cbuf0 + (size_t)((x1 + wsz2 + 1) % (wsz + 1)) * cstep - (size_t)dy0 * ndisp;
    int* hs = hsad0 - dy0 * ndisp;
    const uint8_t* lp_sub = lbase + iclamp(x0, -lofs, width - 1 - lofs) - (ptrdiff_t)dy0 * (ptrdiff_t)stride;
    const uint8_t* lp = lbase + iclamp(x1, -lofs, width - 1 - lofs) - (ptrdiff_t)dy0 * (ptrdiff_t)stride;
    const uint8_t* rp = rbase + iclamp(x1, -rofs, width - ndisp - rofs) - (ptrdiff_t)dy0 * (ptrdiff_t)stride;

    /* slide the per-row horizontal sums one column to the right */
    for (int y = -dy0; y < height + dy1; y++, cb += ndisp, cb_sub += ndisp, hs += ndisp, lp += stride,
             lp_sub += stride, rp += stride) {
      int lval = lp[0];
      for (int d = 0; d < ndisp; d++) {
        int diff = iabs(lval - rp[d]);
        cb[d] = (uint8_t)diff;
        hs[d] = hs[d] + diff - cb_sub[d];
      }
      htext[y] += tab[lval] - tab[lp_sub[0]];
    }

    /* replicate the texture sums beyond the available rows */
    for (int y = dy1; y <= wsz2; y++) htext[height + y] = htext[height + dy1 - 1];
    for (int y = -wsz2 - 1; y < -dy0; y++) htext[y] = htext[-dy0];

    /* vertical sum primed for output row 0 (minus its bottom row, plus one extra top row) */
    for (int d = 0; d < ndisp; d++) sadp[d] = hsad0[d - ndisp * dy0] * (wsz2 + 2 - dy0);
    hs = hsad0 + (1 - dy0) * ndisp;
    for (int y = 1 - dy0; y < wsz2; y++, hs += ndisp)
      for (int d = 0; d < ndisp; d++) sadp[d] += hs[d];
    int tsum = 0;
    for (int y = -wsz2 - 1; y < wsz2; y++) tsum += htext[y];

    for (int y = 0; y < height; y++) {
      int minsad = INT_MAX, mind = -1;
      const int* hadd = hsad0 + (size_t)imin(y + wsz2, height + dy1 - 1) * ndisp;
      const int* hsub = hsad0 + (ptrdiff_t)imax(y - wsz2 - 1, -dy0) * ndisp;
      for (int d = 0; d < ndisp; d++) {
        int cur = sadp[d] + hadd[d] - hsub[d];
        sadp[d] = cur;
        if (cur < minsad) {
          minsad = cur;
          mind = d;
        }
      }
      tsum += htext[y + wsz2] - htext[y - wsz2 - 1];
      if (tsum < p->texture_threshold) {
        dcol[(size_t)y * dstride] = FILTERED;
        continue;
      }
      if (p->uniqueness_ratio > 0) {
        int thresh = minsad + (minsad * p->uniqueness_ratio / 100);
        int d;
        for (d = 0; d < ndisp; d++)
          if ((d < mind - 1 || d > mind + 1) && sadp[d] <= thresh) break;
        if (d < ndisp) {
          dcol[(size_t)y * dstride] = FILTERED;
          continue;
        }
      }
      sadp[-1] = sadp[1];
      sadp[ndisp] = sadp[ndisp - 2];
      int pp = sadp[mind + 1], nn = sadp[mind - 1];
      int den = pp + nn - 2 * sadp[mind] + iabs(pp - nn);
      dcol[(size_t)y * dstride] = disp_descale(ndisp - mind - 1 + mindisp, pp - nn, den);
      if (ccol) ccol[(size_t)y * cstride] = sadp[mind];
    }
  }
  free(sad);
  free(hsad_store);
  free(htext_store);
  free(cbuf_store);
}

/* ------------------------------------------------------------------------------------------------
 * Brute force: evaluates SURVEY.md Appendix A.3 directly from its definition for every pixel, with the
 * vertical window clamped to the rows the stripe can see (same replicate rule as above, expressed on the
 * row index). Independent of the sliding-sum bookkeeping above; used to cross-check it.
 * ------------------------------------------------------------------------------------------------ */
void sbmo_find_correspondence_bruteforce(const uint8_t* left_full, const uint8_t* right_full, size_t stride,
                                         int width, int height_full, int row0, int row1, const sbm_params* p,
                                         int16_t* disp_full, size_t dstride, int32_t* cost_full, size_t cstride) {
  const int wsz = p->block_size, wsz2 = wsz / 2;
  const int ndisp = p->num_disparities, mindisp = p->min_disparity;
  const int lofs = imax(ndisp - 1 + mindisp, 0), rofs = -imin(ndisp - 1 + mindisp, 0);
  const int width1 = width - rofs - ndisp + 1;
  const int cap = p->prefilter_cap;
  const int16_t FILTERED = (int16_t)((mindisp - 1) * 16);   /* (cv writes (minDisparity - 1) << 4; a negative left shift is UB in C) */
  const int dy0 = imin(row0, wsz2 + 1), dy1 = imin(height_full - row1, wsz2 + 1);
  const int ylo = row0 - dy0, yhi = row1 + dy1 - 1; /* rows the stripe may read */
  int* sad = (int*)malloc(sizeof(int) * (size_t)(ndisp + 2));
  int* s = sad + 1;
  for (int y = row0; y < row1; y++) {
    int16_t* drow = disp_full + (size_t)y * dstride;
    for (int X = 0; X < width; X++) {
      int x = X - lofs;
      if (x < 0 || x >= width1) {
        drow[X] = FILTERED;
        continue;
      }
      int tsum = 0;
      for (int d = 0; d < ndisp; d++) s[d] = 0;
      for (int dy = -wsz2; dy <= wsz2; dy++) {
        int yy = iclamp(y + dy, ylo, yhi);
        const uint8_t* lrow = left_full + (size_t)yy * stride;
        const uint8_t* rrow = right_full + (size_t)yy * stride;
        for (int dx = -wsz2; dx <= wsz2; dx++) {
          int xp = x + dx;
          int lv = lrow[lofs + iclamp(xp, -lofs, width - lofs - 1)];
          const uint8_t* rp = rrow + rofs + iclamp(xp, -rofs, width - rofs - ndisp);
          tsum += iabs(lv - cap);
          for (int d = 0; d < ndisp; d++) s[d] += iabs(lv - rp[d]);
        }
      }
      int minsad = INT_MAX, mind = -1;
      for (int d = 0; d < ndisp; d++)
        if (s[d] < minsad) {
          minsad = s[d];
          mind = d;
        }
      if (tsum < p->texture_threshold) {
        drow[X] = FILTERED;
        continue;
      }
      int reject = 0;
      if (p->uniqueness_ratio > 0) {
        int thresh = minsad + (minsad * p->uniqueness_ratio / 100);
        for (int d = 0; d < ndisp && !reject; d++)
          if ((d < mind - 1 || d > mind + 1) && s[d] <= thresh) reject = 1;
      }
      if (reject) {
        drow[X] = FILTERED;
        continue;
      }
      s[-1] = s[1];
      s[ndisp] = s[ndisp - 2];
      int pp = s[mind + 1], nn = s[mind - 1];
      int den = pp + nn - 2 * s[mind] + iabs(pp - nn);
      drow[X] = disp_descale(ndisp - mind - 1 + mindisp, pp - nn, den);
      if (cost_full) cost_full[(size_t)y * cstride + X] = s[mind];
    }
  }
  free(sad);
}

/* ------------------------------------------------------------------------------------------------
 * validateDisparity (OpenCV calib3d stereosgbm.cpp), enabled by setDisp12MaxDiff(1) at main.cpp:212.
 * Pass 1 projects every valid left disparity into the right view keeping the cheapest claimant (strict
 * '>' : on equal cost the earlier x keeps the slot); pass 2 invalidates a left pixel only if BOTH the
 * floor- and the ceil-rounded target disagree by more than disp12MaxDiff*16.
 * ------------------------------------------------------------------------------------------------ */
void sbmo_validate_disparity(int16_t* disp, size_t dstride, const int32_t* cost, size_t cstride, int width, int rows,
                             int min_disparity, int num_disparities, int disp12_max_diff) {
  const int minD = min_disparity, maxD = min_disparity + num_disparities;
  const int minX1 = imax(maxD, 0), maxX1 = width + imin(minD, 0);
  const int DISP_SHIFT = 4, DISP_SCALE = 1 << DISP_SHIFT;
  const int INVALID = (minD - 1) * DISP_SCALE;
  const int tol = disp12_max_diff * DISP_SCALE;
  int* disp2 = (int*)malloc(sizeof(int) * (size_t)width * 2);
  int* cost2 = disp2 + width;
  for (int y = 0; y < rows; y++) {
    int16_t* dp = disp + (size_t)y * dstride;
    const int32_t* cp = cost + (size_t)y * cstride;
    for (int x = 0; x < width; x++) {
      disp2[x] = INVALID;
      cost2[x] = INT_MAX;
    }
    for (int x = minX1; x < maxX1; x++) {
      int d = dp[x];
      if (d == INVALID) continue;
      int c = (g_reading & SBMO_READ_COST_SHORT) ? (int)(short)cp[x] : cp[x];
      int x2 = x - ((d + DISP_SCALE / 2) >> DISP_SHIFT);
      if ((g_reading & SBMO_READ_LR_TIE_LATER) ? cost2[x2] >= c : cost2[x2] > c) {
        cost2[x2] = c;
        disp2[x2] = d;
      }
    }
    for (int x = minX1; x < maxX1; x++) {
      int d = dp[x];
      if (d == INVALID) continue;
      int d0 = d >> DISP_SHIFT, d1 = (d + DISP_SCALE - 1) >> DISP_SHIFT;
      int xa = x - d0, xb = x - d1;
      if ((0 <= xa && xa < width && disp2[xa] > INVALID && iabs(disp2[xa] - d) > tol) &&
          (0 <= xb && xb < width && disp2[xb] > INVALID && iabs(disp2[xb] - d) > tol))
        dp[x] = (int16_t)INVALID;
    }
  }
  free(disp2);
}

/* ------------------------------------------------------------------------------------------------
 * filterSpeckles (OpenCV calib3d stereosgbm.cpp, filterSpecklesImpl<short>), main.cpp:210-211.
 * Raster scan; an unlabelled valid pixel seeds a flood fill over 4-neighbours that are valid, unlabelled
 * and within maxDiff of the CURRENT pixel; regions with <= maxSpeckleSize pixels are flagged and their
 * pixels set to newVal as the scan reaches them. cv::StereoBM passes speckleRange UNSCALED (raw 1/16 px).
 * ------------------------------------------------------------------------------------------------ */
void sbmo_filter_speckles(int16_t* img, size_t stride, int width, int height, int new_val, int max_speckle_size,
                          int max_diff) {
  const size_t npix = (size_t)width * height;
  int* labels = (int*)calloc(npix, sizeof(int));
  int* stack = (int*)malloc(sizeof(int) * npix);
  uint8_t* small = (uint8_t*)calloc(npix + 1, 1);
  int cur = 0;
  for (int i = 0; i < height; i++) {
    int16_t* ds = img + (size_t)i * stride;
    int* ls = labels + (size_t)i * width;
    for (int j = 0; j < width; j++) {
      if (ds[j] == new_val) continue;
      if (ls[j]) {
        if (small[ls[j]]) ds[j] = (int16_t)new_val;
        continue;
      }
      int sp = 0, count = 0;
      int py = i, px = j;
      cur++;
      ls[j] = cur;
      for (;;) {
        count++;
        const int16_t* dpp = img + (size_t)py * stride + px;
        int dp = *dpp;
        int* lpp = labels + (size_t)py * width + px;
        if (py < height - 1 && !lpp[width] && dpp[stride] != new_val && iabs(dp - dpp[stride]) <= max_diff) {
          lpp[width] = cur;
          stack[sp++] = (py + 1) * width + px;
        }
        if (py > 0 && !lpp[-width] && dpp[-(ptrdiff_t)stride] != new_val && iabs(dp - dpp[-(ptrdiff_t)stride]) <= max_diff) {
          lpp[-width] = cur;
          stack[sp++] = (py - 1) * width + px;
        }
        if (px < width - 1 && !lpp[1] && dpp[1] != new_val && iabs(dp - dpp[1]) <= max_diff) {
          lpp[1] = cur;
          stack[sp++] = py * width + px + 1;
        }
        if (px > 0 && !lpp[-1] && dpp[-1] != new_val && iabs(dp - dpp[-1]) <= max_diff) {
          lpp[-1] = cur;
          stack[sp++] = py * width + px - 1;
        }
        if (sp == 0) break;
        int q = stack[--sp];
        py = q / width;
        px = q % width;
      }
      if (count <= max_speckle_size) {
        small[cur] = 1;
        ds[j] = (int16_t)new_val;
      } else {
        small[cur] = 0;
      }
    }
  }
  free(labels);
  free(stack);
  free(small);
}

/* ------------------------------------------------------------------------------------------------
 * Prefilter, PREFILTER_NORMALIZED_RESPONSE.  Follows prefilterNorm() in OpenCV calib3d stereobm.cpp
 * (not used by the reference, restated for the completeness of the cv::StereoBM surface):
 *   box(x,y)  = sum of src over the winsize x winsize window centred on (x,y), rows and columns
 *               replicated at the image border (OpenCV keeps the column sums in ushort; winsize*255
 *               fits, so nothing wraps)
 *   g0 = winsize*winsize/8, ss = (1024 + g0)/(2*g0), sg = g0*ss         (integer divisions)
 *   centre    = 4*src(x,y) + src(x-1,y) + src(x+1,y) + src(x,y-1) + src(x,y+1), rows replicated;
 *               in columns 0 and W-1 the missing horizontal neighbour is replaced by the centre
 *               pixel (5*src + the one existing neighbour + up + down)
 *   val       = (centre*sg - box*ss) >> 10      (arithmetic shift)
 *   dst       = val < -cap ? 0 : val > cap ? 2*cap : val + cap
 * ------------------------------------------------------------------------------------------------ */
void sbmo_prefilter_norm(const uint8_t* src, size_t sstride, uint8_t* dst, size_t dstride, int width, int height,
                         int winsize, int cap) {
  const int wsz2 = winsize / 2;
  const int g0 = winsize * winsize / 8;
  const int ss = g0 > 0 ? (1024 + g0) / (g0 * 2) : 0;
  const int sg = g0 * ss;
  int* vsum = (int*)malloc(sizeof(int) * (size_t)width);
  for (int y = 0; y < height; y++) {
    for (int x = 0; x < width; x++) {
      int v = 0;
      for (int dy = -wsz2; dy <= wsz2; dy++) v += src[(size_t)iclamp(y + dy, 0, height - 1) * sstride + x];
      vsum[x] = v;
    }
    const uint8_t* prev = src + (size_t)imax(y - 1, 0) * sstride;
    const uint8_t* curr = src + (size_t)y * sstride;
    const uint8_t* next = src + (size_t)imin(y + 1, height - 1) * sstride;
    for (int x = 0; x < width; x++) {
      int box = 0;
      for (int dx = -wsz2; dx <= wsz2; dx++) box += vsum[iclamp(x + dx, 0, width - 1)];
      int centre;
      if (width == 1) centre = curr[x] * 6 + prev[x] + next[x];   /* degenerate: OpenCV's first-column formula reads curr[1] */
      else if (x == 0) centre = curr[x] * 5 + curr[x + 1] + prev[x] + next[x];
      else if (x == width - 1) centre = curr[x] * 5 + curr[x - 1] + prev[x] + next[x];
      else centre = curr[x] * 4 + curr[x - 1] + curr[x + 1] + prev[x] + next[x];
      const int val = (centre * sg - box * ss) >> 10;
      dst[(size_t)y * dstride + x] = (uint8_t)(val < -cap ? 0 : (val > cap ? 2 * cap : val + cap));
    }
  }
  free(vsum);
}

/* ------------------------------------------------------------------------------------------------
 * Parameter checks of cv::StereoBM::compute (stereobm.cpp), one status code per CV_Error.
 * (The product library implements the same table in its own source; the two are compared by tests.)
 * ------------------------------------------------------------------------------------------------ */
static int oracle_validate(const sbm_params* p, int width, int height) {
  if (!p) return SBM_ERR_NULL;
  if (width <= 0 || height <= 0) return SBM_ERR_SIZE;
  if (p->prefilter_type != SBM_PREFILTER_NORMALIZED_RESPONSE && p->prefilter_type != SBM_PREFILTER_XSOBEL)
    return SBM_ERR_PREFILTER_TYPE;
  if (p->prefilter_size < 5 || p->prefilter_size > 255 || p->prefilter_size % 2 == 0) return SBM_ERR_PREFILTER_SIZE;
  if (p->prefilter_cap < 1 || p->prefilter_cap > 63) return SBM_ERR_PREFILTER_CAP;
  if (p->block_size < 5 || p->block_size > 255 || p->block_size % 2 == 0 || p->block_size >= imin(width, height))
    return SBM_ERR_BLOCK_SIZE;
  if (p->num_disparities <= 0 || p->num_disparities % 16 != 0) return SBM_ERR_NUM_DISPARITIES;
  if (p->texture_threshold < 0) return SBM_ERR_TEXTURE;
  if (p->uniqueness_ratio < 0) return SBM_ERR_UNIQUENESS;
  return SBM_OK;
}

/* ------------------------------------------------------------------------------------------------
 * cv::StereoBM::compute (stereobm.cpp) with a CV_16SC1 destination.
 * Order: checks -> FILTERED fill if the disparity range cannot fit -> prefilter both images -> valid ROI
 * -> correspondence on the ROI rows (rows outside = FILTERED) -> validateDisparity (if disp12MaxDiff >= 0)
 * -> columns outside the ROI = FILTERED -> filterSpeckles (if speckleRange >= 0 && speckleWindowSize > 0).
 * An empty valid ROI leaves OpenCV's output unwritten; here it is filled with FILTERED.
 * ------------------------------------------------------------------------------------------------ */
int sbmo_compute(const sbm_params* p, const uint8_t* left, size_t lstride, const uint8_t* right, size_t rstride,
                 int width, int height, int16_t* disp, size_t dstride_bytes, uint8_t* pf_l_out, uint8_t* pf_r_out,
                 int16_t* pre_lr, int32_t* cost_out) {
  int st = oracle_validate(p, width, height);
  if (st != SBM_OK) return st;
  if (!left || !right || !disp) return SBM_ERR_NULL;
  if (lstride < (size_t)width || rstride < (size_t)width || dstride_bytes < (size_t)width * 2 || (dstride_bytes & 1))
    return SBM_ERR_SIZE;
  const size_t dstride = dstride_bytes / 2;
  const int mindisp = p->min_disparity, ndisp = p->num_disparities;
  const int16_t FILTERED = (int16_t)((mindisp - 1) * 16);   /* (cv writes (minDisparity - 1) << 4; a negative left shift is UB in C) */
  const int lofs = imax(ndisp - 1 + mindisp, 0), rofs = -imin(ndisp - 1 + mindisp, 0);
  const int width1 = width - rofs - ndisp + 1;
  const size_t npix = (size_t)width * height;

  for (int y = 0; y < height; y++)
    for (int x = 0; x < width; x++) disp[(size_t)y * dstride + x] = FILTERED;
  if (pre_lr)
    for (size_t i = 0; i < npix; i++) pre_lr[i] = FILTERED;
  if (cost_out) memset(cost_out, 0, npix * sizeof(int32_t));
  if (lofs >= width || rofs >= width || width1 < 1) {
    if (pf_l_out) memset(pf_l_out, 0, npix);
    if (pf_r_out) memset(pf_r_out, 0, npix);
    return SBM_OK;
  }

  uint8_t* pl = (uint8_t*)malloc(npix);
  uint8_t* pr = (uint8_t*)malloc(npix);
  int32_t* cost = (int32_t*)calloc(npix, sizeof(int32_t));
  if (!pl || !pr || !cost) {
    free(pl); free(pr); free(cost);
    return SBM_ERR_NOMEM;
  }
  if (p->prefilter_type == SBM_PREFILTER_XSOBEL) {
    sbmo_prefilter_xsobel(left, lstride, pl, (size_t)width, width, height, p->prefilter_cap);
    sbmo_prefilter_xsobel(right, rstride, pr, (size_t)width, width, height, p->prefilter_cap);
  } else {
    sbmo_prefilter_norm(left, lstride, pl, (size_t)width, width, height, p->prefilter_size, p->prefilter_cap);
    sbmo_prefilter_norm(right, rstride, pr, (size_t)width, width, height, p->prefilter_size, p->prefilter_cap);
  }
  if (pf_l_out) memcpy(pf_l_out, pl, npix);
  if (pf_r_out) memcpy(pf_r_out, pr, npix);

  int32_t full[4] = {0, 0, width, height}, r1[4], r2[4], roi[4];
  memcpy(r1, (p->roi1[2] > 0 && p->roi1[3] > 0) ? p->roi1 : full, sizeof r1);
  memcpy(r2, (p->roi2[2] > 0 && p->roi2[3] > 0) ? p->roi2 : full, sizeof r2);
  sbmo_valid_roi(r1, r2, mindisp, ndisp, p->block_size, roi);
  /* intersect with the image rows, as FindStereoCorrespInvoker does with its stripe rectangle */
  int row0 = imax(roi[1], 0), row1 = imin(roi[1] + roi[3], height);
  if (roi[2] > 0 && roi[3] > 0 && row1 > row0) {
    if (g_simd && sbmo_simd_ok(p))
      sbmo_find_correspondence_u16(pl, pr, (size_t)width, width, height, row0, row1, p, disp, dstride,
                                   p->disp12_max_diff >= 0 ? cost : NULL, (size_t)width);
    else
      sbmo_find_correspondence(pl, pr, (size_t)width, width, height, row0, row1, p, disp, dstride,
                               p->disp12_max_diff >= 0 ? cost : NULL, (size_t)width);
    if (pre_lr)
      for (int y = row0; y < row1; y++) memcpy(pre_lr + (size_t)y * width, disp + (size_t)y * dstride, (size_t)width * 2);
    if (cost_out && p->disp12_max_diff >= 0) memcpy(cost_out, cost, npix * sizeof(int32_t));
    if (p->disp12_max_diff >= 0)
      sbmo_validate_disparity(disp + (size_t)row0 * dstride, dstride, cost + (size_t)row0 * width, (size_t)width, width,
                              row1 - row0, mindisp, ndisp, p->disp12_max_diff);
    int c0 = imax(imin(roi[0], width), 0), c1 = imax(imin(roi[0] + roi[2], width), 0);
    for (int y = row0; y < row1; y++) {
      int16_t* dr = disp + (size_t)y * dstride;
      for (int x = 0; x < c0; x++) dr[x] = FILTERED;
      for (int x = c1; x < width; x++) dr[x] = FILTERED;
    }
  }
  if (p->speckle_range >= 0 && p->speckle_window_size > 0)
    sbmo_filter_speckles(disp, dstride, width, height, FILTERED, p->speckle_window_size,
                         (g_reading & SBMO_READ_SPECKLE_X16) ? p->speckle_range * 16 : p->speckle_range);
  free(pl);
  free(pr);
  free(cost);
  return SBM_OK;
}

/* ------------------------------------------------------------------------------------------------
 * Rectification, FPGA flavour (the only flavour of the reference that rectifies).
 *
 * sbmo_rect_map follows rect_remap() of src/StereoBM/src/fpga.c:303-366: destination pixel -> normalised
 * rectified ray -> rotate back -> perspective divide -> source pixel, all in s1.24 fixed point with the
 * firmware's shifts and its UNSIGNED 2^48 / lw division; the stored coordinate is s10.5 (1/32 px), rounded
 * from s10.6 by (v + 1) >> 1, truncated to 16 bits.
 * ------------------------------------------------------------------------------------------------ */
static inline int64_t rect_coord(int64_t num, int64_t lw_inv, int32_t f, int32_t c) {
  const int64_t n2 = (num * lw_inv) >> 24;       /* s1.24 * s1.24 -> s1.24 */
  const int64_t nf = (n2 * (int64_t)f) >> 34;    /* s1.24 * u10.16 -> s10.6 */
  const int64_t v = nf + ((int64_t)c << 6);      /* + principal point */
  return (v + 1) >> 1;                           /* s10.5 */
}

void sbmo_rect_map(const sbm_rect_cam* cam, int width, int height, int16_t* map) {
  for (int yd = 0; yd < height; yd++) {
    for (int xd = 0; xd < width; xd++) {
      /* normalised coordinates in the rectified camera: dst / f2 - c2 / f2 */
      const int64_t xn = (((int64_t)xd * (int64_t)cam->f2inv[0]) >> 8) - (int64_t)cam->c2_f2[0];
      const int64_t yn = (((int64_t)yd * (int64_t)cam->f2inv[1]) >> 8) - (int64_t)cam->c2_f2[1];
      /* [lx ly lw] = [xn yn 1] * rot  (rot[2][*] enters unscaled: the homogeneous 1) */
      int64_t l[3];
      for (int k = 0; k < 3; k++)
        l[k] = (((int64_t)cam->rot[0][k] * xn) >> 24) + (((int64_t)cam->rot[1][k] * yn) >> 24) + (int64_t)cam->rot[2][k];
      /* the firmware divides (1ull << 48) by lw converted to unsigned */
      const uint64_t den = (uint64_t)l[2];
      const int64_t lw_inv = den ? (int64_t)(((uint64_t)1 << 48) / den) : 0;
      int16_t* m = map + ((size_t)yd * width + xd) * 2;
      m[0] = (int16_t)rect_coord(l[0], lw_inv, cam->f[0], cam->c[0]);
      m[1] = (int16_t)rect_coord(l[1], lw_inv, cam->f[1], cam->c[1]);
    }
  }
}

/* rect_intp.v:285-404: u0.5 fractions, u1.10 weight products, u8.10 sum, >> 9, + 1, >> 1. */
void sbmo_rect_remap(const uint8_t* src, const int16_t* map, int width, int height, uint8_t* dst) {
  for (int y = 0; y < height; y++) {
    for (int x = 0; x < width; x++) {
      const int mx = map[((size_t)y * width + x) * 2], my = map[((size_t)y * width + x) * 2 + 1];
      const int xi = mx >> 5, yi = my >> 5, xf = mx & 31, yf = my & 31;
      int tap[2][2];
      for (int dy = 0; dy < 2; dy++)
        for (int dx = 0; dx < 2; dx++) {
          const int sx = xi + dx, sy = yi + dy;
          tap[dy][dx] = (sx >= 0 && sx < width && sy >= 0 && sy < height) ? src[(size_t)sy * width + sx] : 0;
        }
      const int acc = tap[0][0] * ((32 - xf) * (32 - yf)) + tap[0][1] * (xf * (32 - yf)) + tap[1][0] * ((32 - xf) * yf) +
                      tap[1][1] * (xf * yf);
      dst[(size_t)y * width + x] = (uint8_t)(((acc >> 9) + 1) >> 1);
    }
  }
}

int sbmo_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

int sbmo_compute_batch(const sbm_params* p, int n, const uint8_t* left, const uint8_t* right, int width, int height,
                       int16_t* disp, int threads) {
  int status = SBM_OK;
  const size_t npix = (size_t)width * height;
  if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
#endif
  for (int i = 0; i < n; i++) {
    int st = sbmo_compute(p, left + i * npix, (size_t)width, right + i * npix, (size_t)width, width, height,
                          disp + i * npix, (size_t)width * 2, NULL, NULL, NULL, NULL);
    if (st != SBM_OK) {
#ifdef _OPENMP
#pragma omp critical
#endif
      status = st;
    }
  }
  return status;
}

/* ------------------------------------------------------------------------------------------------
 * Consumers of the disparity map, restated from the reference's own sources (these ARE in /root/reference,
 * so this part of the oracle follows in-tree code, operation by operation; compiled with -ffp-contract=off).
 * ------------------------------------------------------------------------------------------------ */
/* SensorData::setFeatures, src/slam/src/core/SensorData.cpp:50-58 */
void sbmo_decimate(const int16_t* disp, int width, int height, int scale, int16_t* out) {
  const int wd = width / scale, hd = height / scale;
  for (int r = 0; r < hd; r++)
    for (int c = 0; c < wd; c++) out[(size_t)r * wd + c] = disp[(size_t)(r * scale) * width + c * scale];
}

typedef struct { float x, y, z; } pt3;

/* projectDisparityTo3D, src/slam/src/core/Stereo.cpp:157-182 */
static pt3 project_disparity(float px, float py, float disp, const sbm_stereo_model* m) {
  pt3 p;
  if (disp > 0.0f) {
    float c = (float)(m->cx_r - m->cx_l);
    /* volatile: gcc -O3's SLP vectoriser otherwise keeps Wx/Wy in double across the products below (observed with
     * gcc 11, -march=x86-64-v3), which is not what the C++ source (float Wx, Wy) computes */
    volatile float Wx = (float)((m->Tx_l / m->fx_l - m->Tx_r / m->fx_r) / (disp + c));
    volatile float Wy = (float)((m->Tx_l / m->fy_l - m->Tx_r / m->fy_r) / (disp + c));
    p.x = (float)((px - m->cx_l) * Wx);
    p.y = (float)((py - m->cy_l) * Wy);
    p.z = (float)(m->fx_l * Wx);
  } else {
    p.x = p.y = p.z = NAN;
  }
  return p;
}

/* isFinite / transformPoint, src/slam/src/core/Stereo.cpp:184-199 */
static int finite3(pt3 p) { return isfinite(p.x) && isfinite(p.y) && isfinite(p.z); }
static pt3 transform_point(pt3 p, const float* t) {
  pt3 r;
  r.x = t[0] * p.x + t[1] * p.y + t[2] * p.z + t[3];
  r.y = t[4] * p.x + t[5] * p.y + t[6] * p.z + t[7];
  r.z = t[8] * p.x + t[9] * p.y + t[10] * p.z + t[11];
  return r;
}

/* the per-pixel part of buildOccupancyGridMap, src/slam/src/core/main.cpp:522-553 */
void sbmo_reproject(const int16_t* disp, int width, int height, int scale, const sbm_stereo_model* m, int apply_local,
                    float* xyz) {
  for (int r = 0; r < height; r++)
    for (int c = 0; c < width; c++) {
      const size_t o = (size_t)r * width + c;
      float d = (float)(disp[o] / 16.0f);
      pt3 p = {NAN, NAN, NAN};
      if (d > 0) {
        p = project_disparity((float)(c * scale), (float)(r * scale), d, m);
        if (finite3(p)) {
          if (apply_local && m->has_local) p = transform_point(p, m->local);
        } else {
          p.x = p.y = p.z = NAN;
        }
      }
      xyz[3 * o] = p.x; xyz[3 * o + 1] = p.y; xyz[3 * o + 2] = p.z;
    }
}

/* generateKeypoints3DStereo, dense-map branch, src/slam/src/core/Stereo.cpp:53-117 */
void sbmo_keypoints3d(const int16_t* disp, int width, int height, const float* kpts, int nk, const sbm_stereo_model* m,
                      float min_depth, float max_depth, float* xyz) {
  for (int i = 0; i < nk; i++) {
    pt3 pt = {NAN, NAN, NAN};
    const float kx = kpts[2 * i], ky = kpts[2 * i + 1];
    const int ix = (int)kx, iy = (int)ky;
    if (ix >= 0 && ix < width && iy >= 0 && iy < height) {
      short tmps = disp[(size_t)iy * width + ix];
      float disparity = (float)(tmps / 16.0f);
      if (disparity < 0) disparity = 0;
      if (disparity != 0.0f) {
        pt3 t = project_disparity(kx, ky, disparity, m);
        if (finite3(t) && (min_depth < 0.0f || t.z > min_depth) && (max_depth <= 0.0f || t.z <= max_depth)) {
          pt = t;
          if (m->has_local) pt = transform_point(pt, m->local);
        }
      }
    }
    xyz[3 * i] = pt.x; xyz[3 * i + 1] = pt.y; xyz[3 * i + 2] = pt.z;
  }
}
