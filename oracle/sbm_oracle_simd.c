/*
 * sbm_oracle_simd.c -- a u16-vectorised TIMING variant of the oracle's findStereoCorrespondenceBM. TEST INFRASTRUCTURE ONLY
 * (same rules as sbm_oracle.c: only tests/ and bench.py's cpu_baseline leg may use it; nothing under u96-slam_amd/ does).
 *
 * Why it exists: bench.py reports a CPU baseline next to the GPU number, and the scalar int32 restatement
 * in sbm_oracle.c understates what the reference's CPU path delivers -- cv::StereoBM switches to 16-bit SIMD sums ("useShorts":
 * preFilterCap <= 31 && SADWindowSize <= 21, 8 x u16 per SSE register, OpenCV calib3d stereobm.cpp) for exactly the parameters
 * the reference sets (src/slam/src/core/main.cpp:204-205). This file keeps the scalar restatement's structure -- columns outer,
 * rows inner, a ring of w+1 abs-diff columns, sliding sums in both directions -- but holds every sum as uint16_t and writes the
 * disparity loops so that gcc -O3 -march=x86-64-v3 vectorises them (16 x u16 per AVX2 register): abs-diff + slide, vertical
 * slide, a key minimum (sum << 16 | d) for "first d attaining the minimum", a counting uniqueness test. It is NOT the checker:
 * tests/test_oracle_properties.py pins it to the scalar restatement (bit-identical maps and costs); bench.py checks the same on
 * its sample and reports both rates, each under its own label.
 *
 * Valid when every window sum fits 16 bits: blockSize^2 * 2 * preFilterCap <= 65535 (sbmo_simd_ok).
 */
#include <limits.h>
#include <stdlib.h>
#include <string.h>

#include "sbm_oracle.h"

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int iabs(int a) { return a < 0 ? -a : a; }
static inline int iclamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int16_t disp_descale(int v1, int v2, int d) { return (int16_t)((v1 * 256 + (d != 0 ? v2 * 256 / d : 0) + 15) >> 4); }

int sbmo_simd_ok(const sbm_params* p) {
  return p && (long)p->block_size * p->block_size * 2 * p->prefilter_cap <= 65535 && p->num_disparities % 16 == 0;
}

/* Same contract as sbmo_find_correspondence (sbm_oracle.c): stripe [row0,row1) of the prefiltered pair, disparities and (optional)
 * costs of the columns [lofs, lofs + xend), FILTERED for the columns that can never be matched. */
void sbmo_find_correspondence_u16(const uint8_t* left_full, const uint8_t* right_full, size_t stride, int width, int height_full,
                                  int row0, int row1, const sbm_params* p, int16_t* disp_full, size_t dstride, int32_t* cost_full,
                                  size_t cstride) {
  const int wsz = p->block_size, wsz2 = wsz / 2;
  const int ndisp = p->num_disparities, mindisp = p->min_disparity;
  const int height = row1 - row0;
  if (height <= 0) return;
  const int dy0 = imin(row0, wsz2 + 1), dy1 = imin(height_full - row1, wsz2 + 1);
  const int lofs = imax(ndisp - 1 + mindisp, 0);
  const int rofs = -imin(ndisp - 1 + mindisp, 0);
  const int width1 = width - rofs - ndisp + 1;
  const int ftzero = p->prefilter_cap;
  const int16_t FILTERED = (int16_t)((mindisp - 1) * 16);

  const uint8_t* lbase = left_full + (size_t)row0 * stride + lofs;
  const uint8_t* rbase = right_full + (size_t)row0 * stride + rofs;
  int16_t* dptr = disp_full + (size_t)row0 * dstride;
  int32_t* cptr = cost_full ? cost_full + (size_t)row0 * cstride : NULL;

  const int nrows = height + dy0 + dy1;
  uint16_t* sad = (uint16_t*)aligned_alloc(64, (((size_t)ndisp + 32) * sizeof(uint16_t) + 63) / 64 * 64);
  uint16_t* hsad_store = (uint16_t*)calloc((size_t)nrows * ndisp, sizeof(uint16_t));
  int* htext_store = (int*)calloc((size_t)(height + wsz + 2), sizeof(int));
  uint8_t* cbuf_store = (uint8_t*)malloc((size_t)(wsz + 1) * nrows * ndisp);
  uint8_t tab[256];
  uint16_t* hsad0 = hsad_store + (size_t)dy0 * ndisp;
  int* htext = htext_store + wsz2 + 1;
  uint16_t* sadp = sad + 16;                            /* sadp[-1] and sadp[ndisp] are the mirrors */
  const size_t cstep = (size_t)nrows * ndisp;
  uint8_t* cbuf0 = cbuf_store + (size_t)dy0 * ndisp;

  for (int x = 0; x < 256; x++) tab[x] = (uint8_t)iabs(x - ftzero);

  for (int x = -wsz2 - 1; x < wsz2; x++) {
    uint16_t* hs = hsad0 - dy0 * ndisp;
    uint8_t* cb = cbuf0 + (size_t)(x + wsz2 + 1) * cstep - (size_t)dy0 * ndisp;
    const uint8_t* lp = lbase + iclamp(x, -lofs, width - lofs - 1) - (ptrdiff_t)dy0 * (ptrdiff_t)stride;
    const uint8_t* rp = rbase + iclamp(x, -rofs, width - rofs - ndisp) - (ptrdiff_t)dy0 * (ptrdiff_t)stride;
    for (int y = -dy0; y < height + dy1; y++, hs += ndisp, cb += ndisp, lp += stride, rp += stride) {
      const uint8_t lval = lp[0];
      for (int d = 0; d < ndisp; d++) {
        const uint8_t r = rp[d];
        const uint8_t diff = (uint8_t)(lval > r ? lval - r : r - lval);
        cb[d] = diff;
        hs[d] = (uint16_t)(hs[d] + diff);
      }
      htext[y] += tab[lval];
    }
  }

  for (int y = 0; y < height; y++) {
    for (int x = 0; x < lofs; x++) dptr[(size_t)y * dstride + x] = FILTERED;
    for (int x = lofs + width1; x < width; x++) dptr[(size_t)y * dstride + x] = FILTERED;
  }

  const int xend = imin(width1, width - lofs);
  for (int x = 0; x < xend; x++) {
    int16_t* dcol = dptr + lofs + x;
    int32_t* ccol = cptr ? cptr + lofs + x : NULL;
    const int x0 = x - wsz2 - 1, x1 = x + wsz2;
    const uint8_t* cb_sub = cbuf0 + (size_t)((x0 + wsz2 + 1) % (wsz + 1)) * cstep - (size_t)dy0 * ndisp;
    uint8_t* cb = cbuf0 + (size_t)((x1 + wsz2 + 1) % (wsz + 1)) * cstep - (size_t)dy0 * ndisp;
    uint16_t* hs = hsad0 - dy0 * ndisp;
    const uint8_t* lp_sub = lbase + iclamp(x0, -lofs, width - 1 - lofs) - (ptrdiff_t)dy0 * (ptrdiff_t)stride;
    const uint8_t* lp = lbase + iclamp(x1, -lofs, width - 1 - lofs) - (ptrdiff_t)dy0 * (ptrdiff_t)stride;
    const uint8_t* rp = rbase + iclamp(x1, -rofs, width - ndisp - rofs) - (ptrdiff_t)dy0 * (ptrdiff_t)stride;

    for (int y = -dy0; y < height + dy1; y++, cb += ndisp, cb_sub += ndisp, hs += ndisp, lp += stride, lp_sub += stride, rp += stride) {
      const uint8_t lval = lp[0];
      for (int d = 0; d < ndisp; d++) {
        const uint8_t r = rp[d];
        const uint8_t diff = (uint8_t)(lval > r ? lval - r : r - lval);
        cb[d] = diff;
        hs[d] = (uint16_t)(hs[d] + diff - cb_sub[d]);
      }
      htext[y] += tab[lval] - tab[lp_sub[0]];
    }

    for (int y = dy1; y <= wsz2; y++) htext[height + y] = htext[height + dy1 - 1];
    for (int y = -wsz2 - 1; y < -dy0; y++) htext[y] = htext[-dy0];

    {
      const uint16_t* h0 = hsad0 - ndisp * dy0;
      const uint16_t mul = (uint16_t)(wsz2 + 2 - dy0);
      for (int d = 0; d < ndisp; d++) sadp[d] = (uint16_t)(h0[d] * mul);
    }
    hs = hsad0 + (1 - dy0) * ndisp;
    for (int y = 1 - dy0; y < wsz2; y++, hs += ndisp)
      for (int d = 0; d < ndisp; d++) sadp[d] = (uint16_t)(sadp[d] + hs[d]);
    int tsum = 0;
    for (int y = -wsz2 - 1; y < wsz2; y++) tsum += htext[y];

    for (int y = 0; y < height; y++) {
      const uint16_t* hadd = hsad0 + (size_t)imin(y + wsz2, height + dy1 - 1) * ndisp;
      const uint16_t* hsub = hsad0 + (ptrdiff_t)imax(y - wsz2 - 1, -dy0) * ndisp;
      /* slide + first minimum: min over keys (sum << 16 | d) -- the smaller key is the smaller sum, then the smaller d */
      uint32_t best = 0xffffffffu;
      for (int d = 0; d < ndisp; d++) {
        const uint16_t cur = (uint16_t)(sadp[d] + hadd[d] - hsub[d]);
        sadp[d] = cur;
        const uint32_t key = ((uint32_t)cur << 16) | (uint32_t)d;
        best = key < best ? key : best;
      }
      const int minsad = (int)(best >> 16), mind = (int)(best & 0xffffu);
      tsum += htext[y + wsz2] - htext[y - wsz2 - 1];
      if (tsum < p->texture_threshold) {
        dcol[(size_t)y * dstride] = FILTERED;
        continue;
      }
      if (p->uniqueness_ratio > 0) {
        const int thresh = minsad + (minsad * p->uniqueness_ratio / 100);
        /* sums <= thresh, all of them, then without the three around the winner */
        const uint16_t t16 = (uint16_t)imin(thresh, 65535);
        int below = 0;
        for (int d = 0; d < ndisp; d++) below += sadp[d] <= t16;
        for (int d = imax(mind - 1, 0); d <= imin(mind + 1, ndisp - 1); d++) below -= sadp[d] <= t16;
        if (below > 0) {
          dcol[(size_t)y * dstride] = FILTERED;
          continue;
        }
      }
      sadp[-1] = sadp[1];
      sadp[ndisp] = sadp[ndisp - 2];
      const int pp = sadp[mind + 1], nn = sadp[mind - 1];
      const int den = pp + nn - 2 * sadp[mind] + iabs(pp - nn);
      dcol[(size_t)y * dstride] = disp_descale(ndisp - mind - 1 + mindisp, pp - nn, den);
      if (ccol) ccol[(size_t)y * cstride] = sadp[mind];
    }
  }
  free(sad);
  free(hsad_store);
  free(htext_store);
  free(cbuf_store);
}
