// sbm_sad_border_wave.h -- the w/2 clamped border columns of each side, as extra wavefronts of the interior SAD launch.
// Included by sbm_sad_fast_core.h (inside namespace sbm, after FastArgs).  gfx950 only.
//
// cv::StereoBM computes these columns with CLAMPED windows, lets them take part in
// validateDisparity and only then overwrites them with FILTERED, so they must be bit-exact whenever the LR check is on
// (setDisp12MaxDiff(1), src/slam/src/core/main.cpp:212).  A window column x' maps to the pair
//   (left column lofs + clamp(x', -lofs, W-lofs-1), right base rofs + clamp(x', -rofs, W-rofs-nd));
// a side has 3*(w/2) such "virtual columns" v and its w/2 outputs are the sums of w consecutive ones.
//
// Rounds 1-3 ran them in a kernel of their own on a side stream (sbm_sad_border.hip in the history): its workgroups were
// only dispatched when the interior grid was exhausted, so every speed-up of the interior kernel turned into a longer
// border tail, and at the reference's own parameters (640x480, 21x21, 64 disparities) it was as long as the interior
// kernel itself.  Now a border job is a WAVEFRONT of the interior grid: the grid starts with the border workgroups, the strips
// follow -- one launch, no second stream, no dispatch-order dependence.  A border wavefront is a serial chain of rows
// (2-5 us per output row alone on the chip, a quarter of that per priming row), so the jobs have their own row segments
// (FastArgs::bseg), sized to end well inside the launch: about 64 rows at KITTI x 64 pairs, 13 at 640x480 x 64.
// Measured (profiles/r04_border_fused_ab.txt, same box, alternating): KITTI x64 1.110 -> 1.095 ms per step, 640x480 nd 64
// w 21 0.566 -> 0.535, 1080p nd 256 2.73 -> 2.58, 2160p 2.86 -> 2.84; the border columns cost 4 % (KITTI) to 11 % (640x480)
// of the SAD stage inside the launch (profiles/r04_border_fused_sweep2.txt: with / without the LR check).
//
// Mapping (no workgroup barrier anywhere: every wavefront is independent):
//   wavefront = one SIDE (wavefront-uniform, so every clamp and every LDS offset of a virtual column is a scalar) of JW
//               consecutive pairs; lane = (job, disparity quad): GL = 64 / JW lanes per job
//               (JW = 4 up to 64 disparities, 2 up to 128, 1 beyond).
//   registers = S[j]: the finished window sums of the w/2 outputs for the lane's 4 disparities, packed 4 x u16 -- the
//               only state carried from row to row.  Horizontal first: per entering / leaving row the row sums
//               H_j = sum_{v=j}^{j+w-1} |L(v) - R(v)+d| slide over v (one accumulating v_mqsad_pk_u16_u8 with a one-byte
//               pattern in, one out per step) and go into S[j] with two 32-bit adds / subs.  That costs w + 2 (w/2 - 1)
//               mqsad per row and phase instead of 3 (w/2), but 2 (w/2) registers instead of 6 (w/2) + the per-column
//               offsets: the body fits under the interior kernel's register budget at every window (the round-3 border
//               kernel needed 184-250 VGPRs).
//   LDS       = the jobs' right row pieces (entering + leaving row) expanded 8x: entry p = bytes p..p+7, laid out
//               [p & 3][p >> 2] so that the quads of a job read consecutive 8-byte entries (conflict-free ds_read_b64).
//               Rows arrive through LDS-direct loads (global_load_lds_dword, scattered per-lane source addresses, linear
//               LDS destination): no staging registers, no ds_write -- the wavefront's VGPRs hold S and little else, and
//               the next rows' loads fly under the winner search.
//   WTA       = per output column a register butterfly over the job's lanes (DPP quad_perm / row_half_mirror / row_mirror,
//               v_permlane16_swap, v_permlane32_swap): first the packed key (sum << 16 | d), then -- with the winner known
//               to every lane -- the two neighbour sums (the owning lanes contribute, OR-butterfly) and the count of sums
//               <= the uniqueness threshold (sum-butterfly, several outputs per register).  One lane per (job, output)
//               finishes: texture, uniqueness verdict, sub-pixel, stores.
#pragma once

template <int GL>
__device__ __forceinline__ u32 bw_bfly_min(u32 b) {
  b = min(b, (u32)__builtin_amdgcn_mov_dpp((int)b, 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
  b = min(b, (u32)__builtin_amdgcn_mov_dpp((int)b, 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
  b = min(b, (u32)__builtin_amdgcn_mov_dpp((int)b, 0x141, 0xf, 0xf, true));   // row_half_mirror: 8 lanes
  b = min(b, (u32)__builtin_amdgcn_mov_dpp((int)b, 0x140, 0xf, 0xf, true));   // row_mirror: 16 lanes
  if constexpr (GL >= 32) {
    const auto r = __builtin_amdgcn_permlane16_swap(b, b, false, false);       // odd rows of one copy <-> even rows of the other
    b = min(r[0], r[1]);
  }
  if constexpr (GL >= 64) {
    const auto r = __builtin_amdgcn_permlane32_swap(b, b, false, false);
    b = min(r[0], r[1]);
  }
  return b;
}
// the same butterfly with + (fields of one register never carry into each other: host-checked widths) and with |
template <int GL, bool OR>
__device__ __forceinline__ u32 bw_bfly_acc(u32 b) {
  auto op = [](u32 x, u32 y) { return OR ? (x | y) : (x + y); };
  b = op(b, (u32)__builtin_amdgcn_mov_dpp((int)b, 0xB1, 0xf, 0xf, true));
  b = op(b, (u32)__builtin_amdgcn_mov_dpp((int)b, 0x4E, 0xf, 0xf, true));
  b = op(b, (u32)__builtin_amdgcn_mov_dpp((int)b, 0x141, 0xf, 0xf, true));
  b = op(b, (u32)__builtin_amdgcn_mov_dpp((int)b, 0x140, 0xf, 0xf, true));
  if constexpr (GL >= 32) {
    const auto r = __builtin_amdgcn_permlane16_swap(b, b, false, false);
    b = op(r[0], r[1]);
  }
  if constexpr (GL >= 64) {
    const auto r = __builtin_amdgcn_permlane32_swap(b, b, false, false);
    b = op(r[0], r[1]);
  }
  return b;
}

// LDS of one border wavefront in bytes (compile-time upper bound for its template tuple; the launcher sizes the workgroup's
// dynamic LDS as max(interior, NWAVES * this)). Rows arrive through LDS-direct loads (global_load_lds_dword: lane i of a
// load writes dword i of a 256-byte block), so every area is a whole number of such blocks.
template <int W2, int NDMAX>
struct BorderLds {
  static constexpr int NVC = 3 * W2;
  static constexpr int JW = NDMAX <= 64 ? 4 : (NDMAX <= 128 ? 2 : 1);
  static constexpr int NSL = (NVC + NDMAX) / 4 + 2;                 // 8-byte entries per residue class of one staged piece
  static constexpr int NRL = (JW * 4 * NSL * 2 + 63) / 64;          // loads (of 64 dwords) per staged right row: [job][p & 3][p >> 2] x 8 B
  static constexpr int NLL = (JW * NVC + 63) / 64;                  // loads per staged left row: [job][virtual column] x 4 B (byte 0 is the pixel)
  static constexpr int RB = NRL * 256, LB = NLL * 256;              // one staged row
  static constexpr int EP = JW * W2 * 16;                           // [job][output] {key, neighbours, count, -}
  static constexpr int TC = JW * NVC * 4;                           // [job][virtual column] texture column sums
  static constexpr int ZB = (NVC * 4 + 15) / 16 * 16;                 // zero patterns for idle lanes
  static constexpr int BYTES = 2 * (RB + LB) + ZB + EP + (TC + 15) / 16 * 16;   // two staged rows + the rest
};

// One border wavefront: side (gi & 1) of the pairs pair0 + k * pstride (k < JW), rows of segment segi.
// The LDS of the wavefront comes in as three `restrict` areas -- staged rows (+ zero patterns), finishing entries, texture
// sums: the rows arrive by LDS-direct loads, and the compiler makes every LDS access that MAY alias a pending one wait for
// it (vmcnt); the scoped no-alias information of the three parameters is what lets the winner search's own LDS traffic run
// while the next rows are still in flight.
template <int W2, int NDMAX>
__device__ __forceinline__ void sad_border_wave_body(const FastArgs& a, unsigned char* __restrict__ const wl, uint4* __restrict__ const Ep,
                                                     int* __restrict__ const Tc, const int segi, const int wi) {
  using BL = BorderLds<W2, NDMAX>;
  constexpr int NVC = 3 * W2, WSZ = 2 * W2 + 1;
  constexpr int JW = BL::JW, GL = 64 / JW;
  constexpr int NRL = BL::NRL, NLL = BL::NLL;
  constexpr int FB = NDMAX > 128 ? 16 : 8;              // field width of the packed uniqueness counts (a job's total is <= nd)
  constexpr int FPR = 32 / FB;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  const int lane = threadIdx.x & 63;
  const int side = wi & 1;
  const int job = lane / GL, q = lane & (GL - 1);
  // pairs of this wavefront: JW consecutive ones (giving a wavefront the pairs of one XCD -- stride 8 -- measured no different:
  // KITTI 0.966 / 0.968, 640x480 0.520 / 0.519, 1080p 2.226 / 2.225 ms per step)
  constexpr int pstride = 1;
  const int pair0 = JW * (wi >> 1);
  // latency-bound guest among the interior kernel's wavefronts: a serial chain per row that should never wait for the issue port
  __builtin_amdgcn_s_setprio(3);
  const int nq = a.nd >> 2;
  const int npiece = NVC + a.nd;                 // 8-byte entries of one staged piece (entry p = bytes p .. p+7)
  const int nsl = npiece / 4 + 2;
  const bool act = q < nq && pair0 + job * pstride < a.npairs;
  const int ys = a.row0 + segi * a.bseg, ye = min(ys + a.bseg, a.row1);   // the border jobs' own (finer) row segments

  // wavefront-uniform geometry of the side
  const int xo = side ? a.xc1 : 0;               // first output column (relative to lofs)
  const int xfirst = xo - W2;                    // window column of virtual column 0
  const int rlo = -a.rofs, rhi = a.W - a.rofs - a.nd, llo = -a.lofs, lhi = a.W - a.lofs - 1;
  auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
  const int rb0 = a.rofs + clampi(xfirst, rlo, rhi);
  // byte offset of virtual column v's window entry inside a staged piece, for quad 0 (scalar)
  auto boff = [&](int v) {
    const int ov = a.rofs + clampi(xfirst + v, rlo, rhi) - rb0;   // 0 .. NVC-1
    return ((ov & 3) * nsl + (ov >> 2)) * 8;
  };

  // LDS carve-up: staged row `which` = right entries [job][p & 3][p >> 2] (8 B each), then left dwords [job][v]
  constexpr int ROWB = BL::RB + BL::LB;
  unsigned char* const Zb = wl + 2 * ROWB;                        // NVC zero dwords
  if (lane < NVC) reinterpret_cast<u32*>(Zb)[lane] = 0u;

  const size_t jobstride = (size_t)pstride * a.plane;             // bytes between the planes of two jobs (< 2^31 / 4: host-checked)
  const uint8_t* const pl0 = a.pf_l + (size_t)pair0 * a.plane + a.padl;
  const uint8_t* const pr0 = a.pf_r + (size_t)pair0 * a.plane + a.padl;

  // Staging plan of this lane, fixed for the whole segment: LDS-direct dword loads, lane i of load k fills dword 64 k + i of
  // the staged row -- half (dword & 1) of the 8-byte entry in slot (dword >> 1) = [job][p & 3][p >> 2], i.e. source bytes
  // p + 4 half .. + 3 of the job's right row piece. Slots beyond the pieces (and pairs beyond the batch) load from the
  // first piece: valid memory, never read.
  u32 roff[NRL];   // byte offset from the row start of pair0's right plane
#pragma unroll
  for (int k = 0; k < NRL; k++) {
    const int dw = lane + 64 * k, sl = dw >> 1;
    const int j = sl / (4 * nsl), rem = sl - j * 4 * nsl;
    const int r = rem / nsl, c = rem - r * nsl;
    const bool valid = j < JW && pair0 + j * pstride < a.npairs;
    roff[k] = (u32)((valid ? j * (int)jobstride + 4 * c + r : 0) + rb0 + 4 * (dw & 1));
  }
  u32 loff[NLL];   // left pixels: dword 64 k + i = (job, virtual column); byte 0 of the dword is the pixel
#pragma unroll
  for (int k = 0; k < NLL; k++) {
    const int e = lane + 64 * k;
    const int j = e / NVC, v = e - j * NVC;
    const bool valid = j < JW && pair0 + j * pstride < a.npairs;
    loff[k] = (u32)((valid ? j * (int)jobstride : 0) + a.lofs + clampi(xfirst + (valid ? v : 0), llo, lhi));
  }
  // one staged row: NRL + NLL LDS-direct loads (no registers, no ds_write); `which` is wavefront-uniform
  auto stage = [&](const int y, const int which) {
    const uint8_t* const lrow = pl0 + (size_t)y * a.pitch;
    const uint8_t* const rrow = pr0 + (size_t)y * a.pitch;
    unsigned char* const dst = wl + which * ROWB;
#pragma unroll
    for (int k = 0; k < NRL; k++)
      __builtin_amdgcn_global_load_lds((gptr_t)(rrow + roff[k]), (lptr_t)(dst + 256 * k), 4, 0, 0);
#pragma unroll
    for (int k = 0; k < NLL; k++)
      __builtin_amdgcn_global_load_lds((gptr_t)(lrow + loff[k]), (lptr_t)(dst + BL::RB + 256 * k), 4, 0, 0);
  };
  // everything this wavefront has in flight has landed (LDS-direct loads count in vmcnt)
  auto landed = [] {
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  // every LDS read of this wavefront has returned (lgkmcnt = 0; vmcnt, expcnt untouched): the staged rows may be overwritten
  auto landed_lds = [] {
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  };

  // idle lanes (quads beyond nd, pairs beyond the batch) read zero patterns -- v_mqsad masks a zero byte, so their sums stay at
  // the all-ones they start from and lose every comparison below without a single select
  u64 S[W2];
#pragma unroll
  for (int j = 0; j < W2; j++) S[j] = act ? 0ull : ~0ull;
  int Ct[NLL];     // texture: window-row sums of |L - cap| of this lane's (job, virtual column) entries
#pragma unroll
  for (int k = 0; k < NLL; k++) Ct[k] = 0;

  // S[j] += / -= the row sums of the staged row `which`; the lanes that own a (job, virtual column) entry keep its texture sum
  const int qr = act ? q : 0;
  auto phase = [&](const int which, const int sign) {
    const unsigned char* const row = wl + which * ROWB;
    // (an opaque copy of the lane's base: otherwise the per-column window addresses -- loop invariants -- are hoisted out of the
    // row loop into 2 x NVC registers, which the allocator then spills)
    typedef const __attribute__((address_space(3))) unsigned char* lds_bytes;
    typedef const __attribute__((address_space(3))) u64* lds_u64;
    u32 rbase = (u32)(size_t)(lds_bytes)(row + (job * 4 * nsl + qr) * 8);   // the 32-bit LDS address
    asm volatile("" : "+v"(rbase));
    const lds_bytes rbb = (lds_bytes)(size_t)rbase;
    const unsigned char* const lb = act ? row + BL::RB + job * NVC * 4 : Zb;
#pragma unroll
    for (int k = 0; k < NLL; k++) {
      const int at = (int)__builtin_amdgcn_sad_u8((u32)row[BL::RB + 4 * (lane + 64 * k)], (u32)a.capb, 0u);
      Ct[k] += sign > 0 ? at : -at;
    }
    auto win = [&](int v) { return *(lds_u64)(rbb + boff(v)); };
    auto pat = [&](int v) { return (u32)lb[4 * v]; };
    auto acc = [&](u64& h, int v) {
      const u32 l = pat(v);
#if SBM_FAST_PINGPONG
      h = __builtin_amdgcn_mqsad_pk_u16_u8(win(v), l, h);
#else
      asm("v_mqsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(h) : "v"(win(v)), "v"(l));
#endif
    };
    auto apply = [&](u64& s, const u64 h) {
      uint2 sv = __builtin_bit_cast(uint2, s);
      const uint2 hv = __builtin_bit_cast(uint2, h);
      // 32-bit arithmetic on the packed halves: the result of (+ entering - leaving) has every half below 65536, so
      // transient carries between the halves cancel
      if (sign > 0) { sv.x += hv.x; sv.y += hv.y; } else { sv.x -= hv.x; sv.y -= hv.y; }
      s = __builtin_bit_cast(u64, sv);
    };
    u64 h = 0ull;
#pragma unroll
    for (int v = 0; v < WSZ; v++) acc(h, v);
    apply(S[0], h);
#pragma unroll
    for (int j = 0; j + 1 < W2; j++) {
      acc(h, j + WSZ);
      const uint2 t = __builtin_bit_cast(uint2, __builtin_amdgcn_mqsad_pk_u16_u8(win(j), pat(j), 0ull));
      uint2 hv = __builtin_bit_cast(uint2, h);
      hv.x -= t.x;   // no u16 borrows: every partial sum is exact
      hv.y -= t.y;
      h = __builtin_bit_cast(u64, hv);
      apply(S[j + 1], h);
    }
  };
  auto lds_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };

  // prime: rows ys-W2 .. ys+W2-1 alternate between the two staged-row areas, the next row arriving while one is consumed;
  // the last one fetched (into area 0) is row ys+W2, the first output row's entering row
  stage(ys - W2, 0);
  for (int i = 0; i < 2 * W2; i += 2) {
    landed();
    stage(ys - W2 + i + 1, 1);
    phase(0, +1);
    landed();
    stage(ys - W2 + i + 2, 0);
    phase(1, +1);
  }

  // the finishing lanes: one per (job, output column)
  const bool fin = lane < JW * W2;
  const int fjb = lane / W2, fj = lane - fjb * W2;
  const int fpair = pair0 + fjb * pstride;
  const bool fvalid = fin && fpair < a.npairs;
  const size_t fo = (size_t)(fvalid ? fpair : 0) * a.W * a.H + a.lofs + xo + fj;
  int16_t* fdisp = a.disp + fo + (size_t)ys * a.W;
  uint16_t* fcost = a.cost ? a.cost + fo + (size_t)ys * a.W : nullptr;
  const int* const ftc = Tc + (fin ? fjb * NVC + fj : 0);
  const u32 q2 = (u32)q * 0x00010001u;
  const int mind_base = a.nd - 1 + a.mindisp;
  // A row's results leave one iteration late, in front of the next rows' loads: landed() waits for everything this wavefront
  // has in flight, and stores issued at the end of an iteration would put their whole latency in front of the next one.
  int out_prev = 0, cost_prev = -1;
  auto flush = [&] {
    if (fvalid) {
      *fdisp = (int16_t)out_prev;
      if (fcost) *fcost = (uint16_t)cost_prev;   // (0xffff for a filtered pixel: the LR kernel relies on it, sbm_lrcheck.hip)
    }
    fdisp += a.W;
    if (fcost) fcost += a.W;
  };

  for (int y = ys; y < ye; y++) {
    landed();                    // area 0: entering row y+W2; area 1 (y > ys): leaving row y-W2-1
    phase(0, +1);
    if (y > ys) phase(1, -1);    // S = window rows y-W2 .. y+W2
    if (y > ys) flush();         // row y-1
    if (y + 1 < ye) {            // next iteration's rows arrive under the winner search below
      landed_lds();              // (every read of the two areas has returned)
      stage(y + 1 + W2, 0);
      stage(y - W2, 1);
    }
#pragma unroll
    for (int k = 0; k < NLL; k++)
      if (lane + 64 * k < JW * NVC) Tc[lane + 64 * k] = Ct[k];

    // ---- per output column: winner (key butterfly over the job's lanes), then -- with the winner known to every lane --
    // the neighbour sums S[mind-1], S[mind+1] (mirrored at the ends; the owning lanes contribute, OR-butterfly) and the
    // uniqueness count: "some disparity outside mind-1..mind+1 has a sum <= thresh" <=> the job holds more sums <= thresh
    // than mind's neighbourhood does. Every lane counts its four packed sums with saturating packed subtractions, the counts
    // of FPR output columns share a register and are summed over the job's lanes; the finishing lane compares.
    constexpr int NPK = (W2 + FPR - 1) / FPR;
    u32 pk[NPK];
#pragma unroll
    for (int i = 0; i < NPK; i++) pk[i] = 0;
    uint2* const epw = reinterpret_cast<uint2*>(Ep + job * W2);   // (key, neighbours) leave the registers as soon as they exist
#pragma unroll
    for (int j = 0; j < W2; j++) {
      const uint2 sv = __builtin_bit_cast(uint2, S[j]);
      const u32 d0 = 4u * (u32)q;
      const u32 k0 = (sv.x << 16) | d0, k1 = (sv.x & 0xffff0000u) | (d0 + 1);
      const u32 k2 = (sv.y << 16) | (d0 + 2), k3 = (sv.y & 0xffff0000u) | (d0 + 3);
      const u32 best = bw_bfly_min<GL>(min(umin3(k0, k1, k2), k3));
      const u32 mind = best & 0xffffu;
      // packed index pair (low half: mind-1, high half: mind+1, mirrored at the ends)
      const u32 in_ = mind > 0 ? mind - 1 : 1u, ip_ = (int)mind < a.nd - 1 ? mind + 1 : (u32)(a.nd - 2);
      const u32 lnp = in_ | (ip_ << 16);
      // field (index & 3) of this lane's four sums for both halves at once: bytes (2 i, 2 i + 1) of {sv.y : sv.x}
      const u32 sel = __umul24(lnp & 0x00030003u, 0x0202u) + 0x01000100u;
      const u32 call = __builtin_amdgcn_perm(sv.y, sv.x, sel);
      // ... kept only by the lane whose quad holds that index: per half 0xffff where (index >> 2) == q
      const u32 eq = ((lnp >> 2) & 0x3fff3fffu) ^ q2;
      const u32 own = __umul24(pk_sub_sat(0x00010001u, eq), 0xffffu);
      const u32 npj = bw_bfly_acc<GL, true>(call & own);
      if (q == 0) epw[2 * j] = make_uint2(best, npj);
      if (a.uniq > 0) {
        // minsad * uniq / 100 with a 24-bit multiply and one mulhi (x / 100 == mulhi(x, 0x51EB851F) >> 5); the envelope keeps
        // the product below 2^23. The threshold is defined on the unscaled sum.
        const u32 minsad = (best >> 16) >> a.pfshift;
        const u32 thresh = (minsad + (__umulhi(__umul24(minsad, (u32)a.uniq), 0x51EB851Fu) >> 5)) << a.pfshift;
        const u32 Tq = min(thresh + 1u, 65535u);                  // sums are <= 65534: sv <= thresh <=> Tq - sv > 0
        const u32 T2 = Tq | (Tq << 16);
        const u32 c2 = pk_min(pk_sub_sat(T2, sv.x), 0x00010001u) + pk_min(pk_sub_sat(T2, sv.y), 0x00010001u);   // 0..2 per half
        pk[j / FPR] += ((c2 & 0xffffu) + (c2 >> 16)) << (FB * (j % FPR));   // 0..4 (idle lanes: all-ones sums, 0)
      }
    }
    if (a.uniq > 0) {
#pragma unroll
      for (int i = 0; i < NPK; i++) pk[i] = bw_bfly_acc<GL, false>(pk[i]);
    }
    if (q == 0) {
#pragma unroll
      for (int j = 0; j < W2; j++) reinterpret_cast<u32*>(epw + 2 * j)[2] = (pk[j / FPR] >> (FB * (j % FPR))) & ((1u << FB) - 1u);
    }
    lds_sync();
    // ---- one lane per (job, output column) finishes -------------------------------------------------------------------
    if (fvalid) {
      const uint4 e = Ep[lane];
      const int minsad = (int)(e.x >> 16), mind = (int)(e.x & 0xffffu);
      const int n = (int)(e.y & 0xffffu), p = (int)(e.y >> 16);
      int ts = 0;
#pragma unroll
      for (int v = 0; v < WSZ; v++) ts += ftc[v];
      bool ok = ts >= a.tex;
      if (a.uniq > 0) {
        const u32 ms = (u32)minsad >> a.pfshift;
        const int tl = (int)min((ms + (__umulhi(__umul24(ms, (u32)a.uniq), 0x51EB851Fu) >> 5)) << a.pfshift, 65534u);
        const int expected = 1 + (mind + 1 < a.nd && p <= tl) + (mind - 1 >= 0 && n <= tl);   // of mind-1, mind, mind+1
        ok = ok && (int)e.z == expected;
      }
      int out = a.filtered, cst = -1;
      if (ok) {
        const int ad = p > n ? p - n : n - p;
        const int den = p + n - 2 * minsad + ad;
        int frac = 0;
        if (den != 0) {
          // den >= |p - n|, so the quotient is at most 256: one reciprocal estimate is within 1 of it and one exact
          // remainder settles which way (24-bit products); C division truncates toward zero
          const u32 num = (u32)ad << 8;
          u32 qv = (u32)((float)num * __builtin_amdgcn_rcpf((float)den));
          const int rem = (int)num - (int)__umul24(qv, (u32)den);
          qv = rem < 0 ? qv - 1 : (rem >= den ? qv + 1 : qv);
          frac = p >= n ? (int)qv : -(int)qv;
        }
        out = ((mind_base - mind) * 256 + frac + 15) >> 4;
        cst = minsad >> a.pfshift;
      }
      out_prev = out;
      cost_prev = cst;
    }
    lds_sync();   // Ep / Tc are rewritten by the next iteration
  }
  flush();        // the segment's last row
}

template <int W2, int NDMAX>
__device__ __forceinline__ void sad_border_wave(const FastArgs& a, unsigned char* const wl, const int segi, const int wi) {
  using BL = BorderLds<W2, NDMAX>;
  unsigned char* const rest = wl + 2 * (BL::RB + BL::LB) + BL::ZB;
  sad_border_wave_body<W2, NDMAX>(a, wl, reinterpret_cast<uint4*>(rest), reinterpret_cast<int*>(rest + BL::EP), segi, wi);
}
