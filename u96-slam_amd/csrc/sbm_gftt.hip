// sbm_gftt.hip -- the PL's GFTT minimum-eigenvalue map on gfx950.
//
// Device counterpart of src/dvp/rtl/gftt_sbl.v:113-204 (3x3 Sobel, first / last column forced to 0), gftt_eig.v:122-143
// (|dx|^2 >> 6, |dy|^2 >> 6, |dx||dy| >> 6), gftt_box.v (3x3 box sums, edge columns forced to 0, 16-bit limiter),
// gftt_eig.v:226-362 ((a + c) - sqrt(((a-c)^2 >> 10) + (b^2 >> 8)) with its limiters) and gftt_obuf.v:101-130,295-305
// (rows 2..H-3 of a dense uint16 map, `Max` register); consumer: src/slam/src/core/GFTT.cpp:41-170 via FPGA.cpp:283-291.
// The RTL takes the square root in a Xilinx CORDIC core whose last bit is unspecified; this kernel takes the exact floor.
//
// 1 B read + 2 B written per pixel, but about 50 vector operations per pixel: VALU-bound. Lane = image
// column, a wavefront marches down a row segment like the RTL's line buffers do. Per pixel row one unaligned 4-byte load
// gives (left, centre, right); only the row's horizontal difference r - l and its 1-2-1 sum are kept (three rows each,
// rolling), so a Sobel pair costs three more operations. dx^2 >> 6 and dy^2 >> 6 travel in the two halves of one register
// through the horizontal 3-sum (two DPP wave shifts + one v_add3_u32 serve both; a 3-sum is at most 3 * 16256 < 2^16) and
// through the vertical one, where v_pk_add_u16 clamp IS gftt_box.v's 16-bit limiter. Lanes 0 and 63 only serve as
// neighbours (62 outputs per wavefront).
#include "sbm_common.h"

namespace sbm {

constexpr int GF_NV = 62;    // output columns per wavefront

typedef unsigned short gu16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned gf_pk_add_sat(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_add_sat(__builtin_bit_cast(gu16x2, a), __builtin_bit_cast(gu16x2, b)));
}
__device__ __forceinline__ unsigned gf_shr1(unsigned v) {   // value of lane-1 (lane 0: 0)
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ unsigned gf_shl1(unsigned v) {   // value of lane+1 (lane 63: 0)
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}

template <bool WIDE>   // WIDE: W >= 4 (every real image); narrower ones read byte by byte
__global__ void __launch_bounds__(64) gftt_eig_kernel(const uint8_t* __restrict__ img, uint16_t* __restrict__ eig,
                                                      unsigned* __restrict__ maxv, int W, int H, int seg) {
  const int lane = threadIdx.x;
  const int n = blockIdx.z;
  const int x = blockIdx.x * GF_NV + lane - 1;            // lane 0 is the left neighbour of the first output column
  const int y0 = blockIdx.y * seg, y1 = min(y0 + seg, H);
  const uint8_t* src = img + (size_t)n * W * H;
  uint16_t* dst = eig + (size_t)n * W * H;
  const bool in_img = x >= 0 && x < W;
  const bool sob_ok = x >= 1 && x <= W - 2;               // gftt_sbl.v: first_r / last_r force the edge samples to 0
  const unsigned colmask = sob_ok ? 0xffffffffu : 0u;     // gftt_box.v: col_start_r | col_end_r -> 0 (and no Sobel sample)
  const bool writes = lane >= 1 && lane <= GF_NV && in_img;
  const int xl = min(max(x - 1, 0), W - 1), xc = min(max(x, 0), W - 1), xr = min(max(x + 1, 0), W - 1);

  // one pixel row -> the raw bytes (left | centre << 8 | right << 16); unpacked into (r - l, l + 2c + r) only when the row
  // is consumed, three loop iterations after the load was issued (the row loop is one load round trip per row otherwise)
  // Branch-free where the image is at least 4 pixels wide: every lane loads 4 bytes that lie inside its row -- from
  // column x-1, or from W-4 with the value shifted down one byte for the last Sobel column x = W-2 (whose fourth byte would
  // be the next row's, or past the batch); lanes without a Sobel sample load from a clamped column and their value is never
  // used. A divergent load would make the row loop wait for every load at the branch join.
  const int xa = min(max(x - 1, 0), max(W - 4, 0));
  const unsigned sh = sob_ok ? 8u * (unsigned)((x - 1) - xa) : 0u;
  auto load_raw = [&](int y) -> unsigned {
    const int yy = min(max(y, 0), H - 1);
    const uint8_t* p = src + (size_t)yy * W;
    if constexpr (WIDE) {
      unsigned v;
      __builtin_memcpy(&v, p + xa, 4);
      return v;                                             // (shifted by `sh` when it is unpacked: no use right behind the load)
    }
    return (unsigned)p[xl] | ((unsigned)p[xc] << 8) | ((unsigned)p[xr] << 16);
  };
  auto unpack_ds = [&](unsigned v, int& d, int& sm) {
    if constexpr (WIDE) v >>= sh;
    const int l = (int)(v & 0xffu), c = (int)((v >> 8) & 0xffu), r = (int)((v >> 16) & 0xffu);
    d = r - l;
    sm = l + r + 2 * c;
  };
  auto load_ds = [&](int y, int& d, int& sm) { unpack_ds(load_raw(y), d, sm); };
  // horizontal 3-sums of the three products of Sobel row ys (0 outside rows 1..H-2 and in the edge columns):
  // hac = sum(dx^2 >> 6) | sum(dy^2 >> 6) << 16, hb = sum(|dx dy| >> 6)
  auto hsums = [&](int ys, int d0, int d1, int d2, int s0, int s2, unsigned& hac, unsigned& hb) {
    unsigned pac = 0, pb = 0;
    if (ys >= 1 && ys <= H - 2) {                          // uniform
      const int dx = d0 + 2 * d1 + d2;                     // (r0-l0) + 2 (r1-l1) + (r2-l2)
      const int dy = s2 - s0;                              // (l2+2c2+r2) - (l0+2c0+r0)
      const unsigned vxx = (unsigned)__mul24(dx, dx) >> 6, vyy = (unsigned)__mul24(dy, dy) >> 6;
      const int xy = __mul24(dx, dy);
      pb = ((unsigned)(xy < 0 ? -xy : xy) >> 6) & colmask;
      pac = (vxx | (vyy << 16)) & colmask;
    }
    // (two-input adds: each folds its wave shift into one v_add_u32_dpp; three terms of at most 16256 per half: no carry
    // between the halves)
    hac = ((pac + gf_shr1(pac)) + gf_shl1(pac)) & colmask;
    hb = ((pb + gf_shr1(pb)) + gf_shl1(pb)) & colmask;
  };

  int d0, d1, d2, s0, s1, s2;                              // pixel rows ys-1, ys, ys+1 (rolling)
  unsigned hac[3], hb[3];                                  // horizontal sums of Sobel rows y-1, y, y+1 (rolling)
  // prime: Sobel rows y0-1 and y0
  load_ds(y0 - 2, d0, s0);
  load_ds(y0 - 1, d1, s1);
  load_ds(y0, d2, s2);
  hsums(y0 - 1, d0, d1, d2, s0, s2, hac[0], hb[0]);
  d0 = d1; s0 = s1; d1 = d2; s1 = s2;
  load_ds(y0 + 1, d2, s2);
  hsums(y0, d0, d1, d2, s0, s2, hac[1], hb[1]);
  unsigned mx = 0;
  unsigned q0 = load_raw(y0 + 2), q1 = load_raw(y0 + 3), q2 = load_raw(y0 + 4);   // three rows in flight
  // rolling state passed by name (three rows per trip): d/s of pixel rows (ya, yb -> new yc), horizontal sums of Sobel rows
  // (ha, hb_ -> new hc); nothing is copied between rows
  auto row = [&](int y, unsigned q, int da, int db, int& dc, int sa, int& sc, unsigned hac_a, unsigned hac_b, unsigned& hac_c,
                 unsigned hb_a, unsigned hb_b, unsigned& hb_c) {
    // Sobel row y+1 from pixel rows y, y+1, y+2
    unpack_ds(q, dc, sc);
    hsums(y + 1, da, db, dc, sa, sc, hac_c, hb_c);
    unsigned out = 0;
    if (y >= 2 && y <= H - 3) {                            // gftt_obuf.v:295-305: four border lines are never written
      const unsigned ac = gf_pk_add_sat(gf_pk_add_sat(hac_a, hac_b), hac_c);      // both 16-bit limiters at once
      const unsigned a = ac & 0xffffu, c = ac >> 16, b = min(hb_a + hb_b + hb_c, 0xffffu);
      const unsigned apc = a + c, amc = __builtin_amdgcn_sad_u16(a, c, 0u);        // |a - c| (the high halves are 0)
      const unsigned amc2 = (amc * amc) >> 10, b2 = (b * b) >> 8;   // both operands < 2^16: the squares fit 32 bits
      const unsigned s = min(amc2 + b2, 0x3fffffu);
      // floor(sqrt(s << 10)): s < 2^22 and the factor 1024 are exact in float; the hardware root (v_sqrt_f32, 1 ulp) is
      // within 0.01 of the true one, so its floor is off by at most one either way and two exact comparisons settle it
      const unsigned rad = s << 10;
      unsigned r = min((unsigned)__builtin_amdgcn_sqrtf((float)s * 1024.0f), 65535u);
      r -= (r * r > rad) ? 1u : 0u;
      r += (r < 65535u && (r + 1u) * (r + 1u) <= rad) ? 1u : 0u;
      const int e = (int)apc - (int)r;
      out = (unsigned)min(max(e, 0), 0xffff);
    }
    if (writes) dst[(size_t)y * W + x] = (unsigned short)out;
    mx = max(mx, writes ? out : 0u);
  };
  // three rows per trip so that the registers of the loads in flight and of the rolling state are re-used by name, never
  // copied (a copy of a register that a load has not filled yet is a wait for that load); rows beyond the segment / image
  // are clamped loads. On entry: (d1, s1), (d2, s2) = pixel rows y, y+1; hac/hb[0], [1] = Sobel rows y-1, y.
  for (int y = y0; y < y1; y += 3) {
    row(y, q0, d1, d2, d0, s1, s0, hac[0], hac[1], hac[2], hb[0], hb[1], hb[2]);              // new pixel row -> (d0, s0), Sobel -> [2]
    q0 = load_raw(y + 5);
    if (y + 1 < y1) { row(y + 1, q1, d2, d0, d1, s2, s1, hac[1], hac[2], hac[0], hb[1], hb[2], hb[0]); q1 = load_raw(y + 6); }
    if (y + 2 < y1) { row(y + 2, q2, d0, d1, d2, s0, s2, hac[2], hac[0], hac[1], hb[2], hb[0], hb[1]); q2 = load_raw(y + 7); }
  }
  for (int off = 32; off > 0; off >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, off, 64));
  if (lane == 0 && mx) atomicMax(maxv + n, mx);
}

// Two image columns per lane (round 4; images at least 8 columns wide). Lane l of a wavefront owns columns cA = b - 1 + 2 l and
// cB = cA + 1 (b = 126 * blockIdx.x, even): ONE unaligned 4-byte load per pixel row covers both Sobel neighbourhoods
// (cA-1 .. cB+1), the row's (r - l, l + 2 c + r) pairs and the vertical Sobel combinations are packed 16-bit operations that
// serve both columns, and the horizontal 3-sums share their middle term: hA = (pA + pB) + pB(lane - 1), hB = (pA + pB) +
// pA(lane + 1) -- three operations per plane for two columns instead of four. Lane 0's column A and lane 63's column B only
// serve as neighbours: 126 outputs per wavefront (62 with one column per lane). The products and the eigenvalue arithmetic stay
// 32-bit per column (dx^2 needs 21 bits), which is most of the work: 0.100 -> see profiles/r04_frontend_kernels.json.
constexpr int GF2_NV = 126;

typedef short gi16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned gf_pk_add(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_bit_cast(gu16x2, a) + __builtin_bit_cast(gu16x2, b));
}
__device__ __forceinline__ unsigned gf_pk_sub(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_bit_cast(gu16x2, a) - __builtin_bit_cast(gu16x2, b));
}

__global__ void __launch_bounds__(64) gftt_eig2_kernel(const uint8_t* __restrict__ img, uint16_t* __restrict__ eig,
                                                       unsigned* __restrict__ maxv, int W, int H, int seg) {
  const int lane = threadIdx.x;
  const int n = blockIdx.z;
  const int cA = blockIdx.x * GF2_NV - 1 + 2 * lane, cB = cA + 1;
  const int y0 = blockIdx.y * seg, y1 = min(y0 + seg, H);
  // one image through raw buffer descriptors: the lane's byte offset in a vector register, the row in a scalar one (an image is
  // less than 2^31 bytes: sbm_gftt_eig_device checks)
  const __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img) + (size_t)n * W * H, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(eig + (size_t)n * W * H, 0, 0x7fffffff, 0x00020000);
  // gftt_sbl.v: first_r / last_r force the edge samples to 0; gftt_box.v: col_start_r | col_end_r -> 0. (Opaque: the compiler
  // otherwise turns every `& mask` into a select on the comparison it came from -- a slow-class instruction for a fast one.)
  unsigned maskA = (cA >= 1 && cA <= W - 2) ? 0xffffffffu : 0u, maskB = (cB >= 1 && cB <= W - 2) ? 0xffffffffu : 0u;
  asm volatile("" : "+v"(maskA), "+v"(maskB));
  const bool writesA = lane >= 1 && cA >= 0 && cA < W, writesB = lane <= 62 && cB >= 0 && cB < W;
  unsigned wmaskA = writesA ? 0xffffu : 0u, wmaskB = writesB ? 0xffffu : 0u;
  asm volatile("" : "+v"(wmaskA), "+v"(wmaskB));
  const int voffA = writesA ? 2 * cA : (int)0xfffffff0u, voffB = writesB ? 2 * cB : (int)0xfffffff0u;
  // every lane loads 4 bytes that lie inside its row: from column cA - 1, or from W - 4 with the value shifted down one byte
  // when column A is the last Sobel column W - 2 (its fourth byte would be the next row's, or past the batch). cA is odd, so
  // cA - 1 >= 0 wherever a column of the lane has a Sobel sample; the other lanes load from a clamped column and mask.
  const int xa = min(max(cA - 1, 0), W - 4);
  const unsigned sh = maskA ? 8u * (unsigned)((cA - 1) - xa) : 0u;
  auto load_raw = [&](int y) -> unsigned {
    const int yy = min(max(y, 0), H - 1);
    return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs_i, xa, __builtin_amdgcn_readfirstlane(yy * W), 0);
  };
  // raw bytes (b0 b1 b2 b3) = columns cA-1, cA, cB, cB+1 -> packed (A | B << 16): d = right - left, sm = left + 2 centre + right
  auto unpack_ds = [&](unsigned v, unsigned& d, unsigned& sm) {
    v >>= sh;
    const unsigned l = __builtin_amdgcn_perm(0u, v, 0x0c010c00u), c = __builtin_amdgcn_perm(0u, v, 0x0c020c01u),
                   r = __builtin_amdgcn_perm(0u, v, 0x0c030c02u);
    d = gf_pk_sub(r, l);
    sm = gf_pk_add(gf_pk_add(l, r), gf_pk_add(c, c));
  };
  // horizontal 3-sums of the products of Sobel row ys, per column: hac = sum(dx^2 >> 6) | sum(dy^2 >> 6) << 16, hb = sum(|dx dy| >> 6)
  // (Sobel rows outside 1..H-2 are 0: a wavefront-uniform mask in a scalar register, no branch and no zero moves)
  auto hsums = [&](int ys, unsigned d0, unsigned d1, unsigned d2, unsigned s0, unsigned s2, unsigned& hacA, unsigned& hacB,
                   unsigned& hbA, unsigned& hbB) {
    const unsigned rowm = (ys >= 1 && ys <= H - 2) ? 0xffffffffu : 0u;        // uniform
    const unsigned mA = maskA & rowm, mB = maskB & rowm;
    const unsigned dx2 = gf_pk_add(gf_pk_add(d0, d2), gf_pk_add(d1, d1));   // packed i16: (r0-l0) + 2 (r1-l1) + (r2-l2)
    const unsigned dy2 = gf_pk_sub(s2, s0);                                 // (l2+2c2+r2) - (l0+2c0+r0)
    const int dxA = (int)(short)(dx2 & 0xffffu), dxB = (int)dx2 >> 16, dyA = (int)(short)(dy2 & 0xffffu), dyB = (int)dy2 >> 16;
    const int xyA = __mul24(dxA, dyA), xyB = __mul24(dxB, dyB);
    const unsigned pbA = ((unsigned)(xyA < 0 ? -xyA : xyA) >> 6) & mA;
    const unsigned pbB = ((unsigned)(xyB < 0 ? -xyB : xyB) >> 6) & mB;
    const unsigned pacA = (((unsigned)__mul24(dxA, dxA) >> 6) | (((unsigned)__mul24(dyA, dyA) >> 6) << 16)) & mA;
    const unsigned pacB = (((unsigned)__mul24(dxB, dxB) >> 6) | (((unsigned)__mul24(dyB, dyB) >> 6) << 16)) & mB;
    // (three terms of at most 16256 per half: no carry between the halves; each neighbour add folds its wave shift)
    const unsigned mac = pacA + pacB, mb = pbA + pbB;
    hacA = (mac + gf_shr1(pacB)) & maskA;
    hacB = (mac + gf_shl1(pacA)) & maskB;
    hbA = (mb + gf_shr1(pbB)) & maskA;
    hbB = (mb + gf_shl1(pbA)) & maskB;
  };
  // gftt_eig.v:226-362 for one column: (a + c) - floor(sqrt(((a-c)^2 >> 10) + (b^2 >> 8)) << 10) with its limiters
  auto eigen = [&](unsigned ac, unsigned b) -> unsigned {
    const unsigned a = ac & 0xffffu, c = ac >> 16;
    const unsigned apc = a + c, amc = __builtin_amdgcn_sad_u16(a, c, 0u);        // |a - c| (the high halves are 0)
    // (plain 32-bit multiplies, operands < 2^16: v_mul_lo_u32 issues like any other slow-class instruction on gfx950, and the
    // 24-bit form measured WRONG here -- 17 of 307 200 noise pixels -- for a reason not tracked down)
    const unsigned amc2 = (amc * amc) >> 10, b2 = (b * b) >> 8;
    const unsigned s = min(amc2 + b2, 0x3fffffu);
    // floor(sqrt(s << 10)): s < 2^22 is exact in float and the hardware root (1 ulp) times 32 is within 0.015 of the true one;
    // taken 0.03 low, its floor r is the true floor or one below (and at most 65535), and ONE exact comparison settles which:
    // (r + 1)^2 <= rad  <=>  r (r + 2) < rad, whose left side fits 32 bits even for r = 65535
    const unsigned rad = s << 10;
    unsigned r = (unsigned)(__builtin_amdgcn_sqrtf((float)s) * 32.0f - 0.03f);   // (negative -> 0 in the conversion)
    r += (r * (r + 2u) < rad) ? 1u : 0u;
    const int e = (int)apc - (int)r;
    return (unsigned)min(max(e, 0), 0xffff);
  };

  unsigned d0, d1, d2, s0, s1, s2;                         // pixel rows ys-1, ys, ys+1 (rolling), both columns packed
  unsigned hacA[3], hacB[3], hbA[3], hbB[3];               // horizontal sums of Sobel rows y-1, y, y+1 (rolling)
  unpack_ds(load_raw(y0 - 2), d0, s0);
  unpack_ds(load_raw(y0 - 1), d1, s1);
  unpack_ds(load_raw(y0), d2, s2);
  hsums(y0 - 1, d0, d1, d2, s0, s2, hacA[0], hacB[0], hbA[0], hbB[0]);
  d0 = d1; s0 = s1; d1 = d2; s1 = s2;
  unpack_ds(load_raw(y0 + 1), d2, s2);
  hsums(y0, d0, d1, d2, s0, s2, hacA[1], hacB[1], hbA[1], hbB[1]);
  unsigned mx = 0;
  unsigned q0 = load_raw(y0 + 2), q1 = load_raw(y0 + 3), q2 = load_raw(y0 + 4);   // three rows in flight
  auto row = [&](int y, unsigned q, unsigned da, unsigned db, unsigned& dc, unsigned sa, unsigned& sc, int ia, int ib, int ic) {
    unpack_ds(q, dc, sc);                                  // Sobel row y+1 from pixel rows y, y+1, y+2
    hsums(y + 1, da, db, dc, sa, sc, hacA[ic], hacB[ic], hbA[ic], hbB[ic]);
    // gftt_obuf.v:295-305: four border lines are never written (0 here; uniform mask). v_pk_add_u16 clamp is gftt_box.v's
    // 16-bit limiter (both planes of a column at once)
    const unsigned om = (y >= 2 && y <= H - 3) ? 0xffffffffu : 0u;
    const unsigned outA = eigen(gf_pk_add_sat(gf_pk_add_sat(hacA[ia], hacA[ib]), hacA[ic]), min(hbA[ia] + hbA[ib] + hbA[ic], 0xffffu)) & om;
    const unsigned outB = eigen(gf_pk_add_sat(gf_pk_add_sat(hacB[ia], hacB[ib]), hacB[ic]), min(hbB[ia] + hbB[ib] + hbB[ic], 0xffffu)) & om;
    const int orow = __builtin_amdgcn_readfirstlane(2 * y * W);
    // (no branch around a store: a lane that does not write has an offset beyond the descriptor's range, and the hardware
    // drops out-of-range buffer stores -- the row stays one basic block, so the wave shifts above fold into their adds)
    __builtin_amdgcn_raw_buffer_store_b16((short)outA, rs_o, voffA, orow, 0);
    __builtin_amdgcn_raw_buffer_store_b16((short)outB, rs_o, voffB, orow, 0);
    mx = max(mx, max(outA & wmaskA, outB & wmaskB));
  };
  // three rows per trip (state re-used by name, never copied: see gftt_eig_kernel)
  for (int y = y0; y < y1; y += 3) {
    row(y, q0, d1, d2, d0, s1, s0, 0, 1, 2);
    q0 = load_raw(y + 5);
    if (y + 1 < y1) { row(y + 1, q1, d2, d0, d1, s2, s1, 1, 2, 0); q1 = load_raw(y + 6); }
    if (y + 2 < y1) { row(y + 2, q2, d0, d1, d2, s0, s2, 2, 0, 1); q2 = load_raw(y + 7); }
  }
  for (int off = 32; off > 0; off >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, off, 64));
  if (lane == 0 && mx) atomicMax(maxv + n, mx);
}

hipError_t launch_gftt_eig(const uint8_t* img, uint16_t* eig, unsigned* maxv, int n, int W, int H, hipStream_t s) {
  hipError_t e = hipMemsetAsync(maxv, 0, (size_t)n * sizeof(unsigned), s);
  if (e != hipSuccess) return e;
  // rows per wavefront: every segment re-reads 4 rows; long segments once the batch fills the chip anyway
  const bool two = W >= 8 && !SBM_TUNE("SBM_DEV_GFTT_ONE", 0);
  const int nv = two ? GF2_NV : GF_NV;
  const int strips = (W + nv - 1) / nv;
  int seg = 64;
  while (seg < 256 && (long)strips * ((H + 2 * seg - 1) / (2 * seg)) * n >= (two ? 4096 : 8192)) seg *= 2;
  const dim3 grid(strips, (H + seg - 1) / seg, n);
  if (two) hipLaunchKernelGGL(gftt_eig2_kernel, grid, dim3(64), 0, s, img, eig, maxv, W, H, seg);
  else if (W >= 4) hipLaunchKernelGGL(gftt_eig_kernel<true>, grid, dim3(64), 0, s, img, eig, maxv, W, H, seg);
  else hipLaunchKernelGGL(gftt_eig_kernel<false>, grid, dim3(64), 0, s, img, eig, maxv, W, H, seg);
  return hipGetLastError();
}

}  // namespace sbm
