// sbm_sad_fast_strip.h -- one strip of the interior SAD kernel: 64 lanes x NDW disparities marching down a row segment, rows
// staged by LDS-direct loads. Included by sbm_sad_fast_kernel.h (every build but the two-accumulator fallback). gfx950 only.
#pragma once
#include "sbm_sad_fast_core.h"

namespace sbm {

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
  // (v_mqsad_pk_u16_u8 with vdst == src2: right on gfx950 although LLVM marks vdst early-clobber -- tools/ubench/mqsad_alias, 9.4e9
  // results, and the device self-test mqsad_inplace_ok(); sbm_sad_fast_pp.hip is the two-array fallback)
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

    // issue priority while this wavefront is in its exchange (kFastPrioExchange, sbm_sad_fast_core.h)
    __builtin_amdgcn_s_setprio(kFastPrioExchange);
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

    // ---- WTA: first index attaining the minimum (fast_first_min: plain keys or the tagged packed search) ----------------------
    u32 best = fast_first_min<NR, WSZ>(S, a.pfshift);
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
      T = fast_uniq_threshold(minsad, a.uniq, a.pfshift);
      acc = fast_deficits<NR>(S, T, a.uniq_plain);
    }

    // ---- neighbours S[mind-1], S[mind+1] (mirrored at the ends) ----------------------------------------------------------
    const int in_ = mind > 0 ? mind - 1 : 1;
    const int ip_ = mind < a.nd - 1 ? mind + 1 : a.nd - 2;
    const int ln = min(max(in_ - d0, 0), NDW - 1), lp = min(max(ip_ - d0, 0), NDW - 1);  // local (clamped) indices
    u32 X0;
    {
      // via a byte-permute selection tree (low half of every selector follows ln, high half lp)
      u32 X[NQ];
      const u32 lnp = (u32)ln | ((u32)lp << 16);
      fast_neighbours_quads<NQ>(S, lnp, X);
      X0 = fast_neighbours_tree<NQ>(X, lnp);
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
    if (a.uniq > 0) ok = ok && fast_unique(acc_lo, acc_hi, T, minsad, mind, nn, pp, a.nd);
    if (mine && produces) {
      int out = a.filtered, cst = -1;
      if (ok) {
        out = fast_subpixel(nn, pp, minsad, mind, a.nd, a.mindisp);
        if (a.cost) cst = minsad >> a.pfshift;
      }
      res_prev = ((u32)out & 0xffffu) | ((u32)cst << 16);   // (a cost is at most 65534; filtered = 0xffff)
    }
  }
  flush(ye - 1);   // the segment's last row
}

}  // namespace sbm
