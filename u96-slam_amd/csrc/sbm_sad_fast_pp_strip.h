// sbm_sad_fast_pp_strip.h -- the strip of the two-accumulator FALLBACK build (sbm_sad_fast_pp.hip): what runs when the device
// self-test of the in-place v_mqsad accumulate fails (mqsad_inplace_ok(), sbm_sad_fast.hip) or with SBM_FAST_INPLACE=0. Two
// accumulator arrays in ping-pong -- the compiler never lets v_mqsad_pk_u16_u8 write a register it reads (vdst is early-clobber
// against every source in LLVM), so an entering row maps VA -> VB through the instruction's free accumulate and the leaving row
// maps VB -> VA with plain subtractions -- rows staged through registers into a 16x-expanded LDS layout (the round-2/3 strip),
// direct horizontal sums. 64-disparity layouts and windows up to 27 only; 8-25x slower layouts take over beyond (include/sbm.h).
// Included by sbm_sad_fast_kernel.h when SBM_FAST_PINGPONG is set. gfx950 only.
#pragma once
#include "sbm_sad_fast_core.h"

namespace sbm {

// One strip of one row segment of one pair: lane i works on column cbase + CS * i (relative to lofs).
template <int NDW, int NWAVES, int NTERM, int PW, bool EXACT_ND, int CS>
__device__ __forceinline__ void sad_fast_pp_strip(const FastArgs& a, const int cbase, const int segi, const int pair) {
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
  // Two arrays in ping-pong: the compiler never lets v_mqsad_pk_u16_u8 write a register it reads (vdst is early-clobber
  // against every source in LLVM), so an entering row maps VA -> VB through the instruction's free accumulate and the
  // leaving row maps VB -> VA with plain subtractions.
  uint2 VA[NQ];
  u64 VB[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) VA[q] = make_uint2(0u, 0u);
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

    // issue priority while this wavefront is in its exchange (kFastPrioExchange, sbm_sad_fast_core.h)
    __builtin_amdgcn_s_setprio(kFastPrioExchange);
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
    // ---- WTA: first index attaining the minimum (fast_first_min) ---------------------------------------------------------
    u32 best = fast_first_min<NR, WSZ>(S, a.pfshift);
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
      T = fast_uniq_threshold(minsad, a.uniq, a.pfshift);
      acc = fast_deficits<NR>(S, T, a.uniq_plain);
    }

    // ---- neighbours S[mind-1], S[mind+1] (mirrored at the ends) via a byte-permute selection tree -------------
    const int in_ = mind > 0 ? mind - 1 : 1;
    const int ip_ = mind < a.nd - 1 ? mind + 1 : a.nd - 2;
    const int ln = min(max(in_ - d0, 0), NDW - 1), lp = min(max(ip_ - d0, 0), NDW - 1);  // local (clamped) indices
    u32 X[NQ];
    const u32 lnp = (u32)ln | ((u32)lp << 16);   // low half of every selector follows ln, high half lp
    fast_neighbours_quads<NQ>(S, lnp, X);
    // S is dead from here on: fetch the leaving row now (its latency hides behind the rest of the tree, the
    // merge, the sub-pixel arithmetic and the stores) without raising the register peak of the S-heavy phase
    // (the two fetches of a row are issued at raised priority as well, so that a wavefront's loads do not wait behind a
    // neighbour's arithmetic: KITTI x64 -1 %; with cooperating wavefronts it costs 18 % -- 1080p 2.63 -> 3.10 ms -- hence the
    // condition)
    if constexpr (NWAVES == 1) __builtin_amdgcn_s_setprio(kFastPrioExchange);
    RowRegs lv = fetch(y - W2);
    if constexpr (NWAVES == 1) __builtin_amdgcn_s_setprio(0);
    fast_neighbours_tree<NQ>(X, lnp);
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
      if (a.uniq > 0) ok = ok && fast_unique(acc_lo, acc_hi, T, minsad, mind, nn, pp, a.nd);
      if (produces) {
        int out = a.filtered, cst = 0xffff;   // (a filtered pixel's cost reads 0xffff: the LR kernel relies on it, sbm_lrcheck.hip)
        if (ok) {
          out = fast_subpixel(nn, pp, minsad, mind, a.nd, a.mindisp);
          cst = minsad >> a.pfshift;
        }
        if (a.cost) __builtin_amdgcn_raw_buffer_store_b16((short)cst, rs_c, ocol, orow, 0);
        __builtin_amdgcn_raw_buffer_store_b16((short)out, rs_d, ocol, orow, 0);
      }
    }

    if (y + 1 < ye) {
      if constexpr (NWAVES == 1) __builtin_amdgcn_s_setprio(kFastPrioExchange);
      g = fetch(y + 1 + W2);   // next entering row: latency hides behind the leaving row's mqsad + subtractions
      if constexpr (NWAVES == 1) __builtin_amdgcn_s_setprio(0);
      apply(lv, 1);
    }
  }
}

}  // namespace sbm
