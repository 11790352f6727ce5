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
//               dword pair (4q, 4q+4) behind the lane's slot, one conflict-free ds_read2_b32 (sad_fast_strip_dma,
//               sbm_sad_fast_strip.h). The register-staged strip with its 16x-expanded layout (sbm_sad_fast_pp_strip.h) remains
//               for the fallback build.
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
// Envelope (sad_fast_supported(); everything else takes the sliding-sum kernel, sbm_sad_wide.hip): odd w in 5..31, nd <= 512,
// w*w*2*cap <= 65534 (16-bit sums), 2*(maxS*uniq/100+1) < 65535, valid-ROI rows inside [w/2, H-w/2).
//
// Source layout: sbm_sad_fast_core.h (arguments, LDS carve-up, exchange plan, the per-row tail: winner search / uniqueness /
// neighbours / sub-pixel), sbm_sad_fast_strip.h (the strip), sbm_sad_border_wave.h (the clamped border columns, extra wavefronts
// of the same launch), sbm_sad_fast_kernel.h (kernel + layout choice), sbm_sad_fast_dev.h (development knobs), and this file:
// the host side -- envelope, device self-test of the in-place accumulate, strips / row segments of a launch -- plus the kernels
// of the windows 15 and 21. The ~270 instantiations compile as four translation units side by side (this one, sbm_sad_fast_pw1 /
// _pw2 / _pw3.hip: the other windows) + the fallback build sbm_sad_fast_pp.hip.
#include "sbm_sad_fast_kernel.h"

namespace sbm {

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

hipError_t launch_sad_fast(const uint8_t* pf_l, const uint8_t* pf_r, int16_t* disp, int32_t* cost, const Geom& g,
                           int* xa, int* xb, bool border, hipStream_t s) {
  *xa = *xb = 0;
  if (!sad_fast_supported(g)) return hipSuccess;
  const bool inplace = mqsad_inplace_ok(s);   // else: the two-accumulator build (sbm_sad_fast_pp.hip)
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
    nseg = (int)(std::sqrt(fast_tune().seg_c1000 * 1e-3 * slots * rows / ((double)strips * g.n * prime)) + 0.5);
    nseg = std::max(1, std::min(nseg, std::max(1, rows / g.wsz)));   // (at least one window height per segment here; see below)
  }
  // Launches that do not fill the chip (round 5, profiles/r05_small_launch_segments.txt): a wavefront's row segment is a serial
  // chain, and w - 1 priming rows at a third of a row's cost are cheaper than idle SIMDs -- segments go down to 8 rows (up to
  // 64 of them) until the launch has ~5 000 workgroups. Launches that cannot even reach ~1 800 wavefronts that way (one or two
  // pairs) split the disparities over more wavefronts per workgroup instead (launch_nd: `split`) and take as many segments
  // as keep them under 1 024 workgroups. One 640x480 nd 64 w 21 pair: SAD stage 0.054 -> 0.049 ms, one KITTI pair 0.052 ->
  // 0.036, 8 KITTI pairs 0.209 -> 0.14, 16 pairs 640x480 0.203 -> 0.12, 4 pairs 1080p nd 256 0.64 -> 0.57.
  const int small_rows = fast_tune().small_rows;
  const long fill = fast_tune().fill;
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
  if (fast_tune().nseg > 0) nseg = std::min(fast_tune().nseg, std::max(1, rows / 2));
  // taper: the last third of the rows is cut into segments of 2/3, 1/2, 1/3 ... of the regular length
  const int taper = fast_tune().taper;
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
    a.uniq_plain = fast_tune().uniq_plain && 8 * ((maxs * g.uniq / 100 + 1) << g.pfshift) <= 65535;   // (8 registers per accumulator)
  }
  const bool split = (long)strips * nseg * g.n < 1024 && fast_tune().split;
  hipError_t e;
  if (!inplace) {
    e = launch_sad_fast_pp(a, g.wsz, border, split, s);
  } else {
    switch (g.wsz) {
      case 15: e = launch_nd<5, 3>(a, border, split, s); break;
      case 21: e = launch_nd<7, 3>(a, border, split, s); break;
#ifdef SBM_DEV_FEW   // (development builds, sbm_sad_fast_dev.h: only the bench workloads' windows)
      default: e = hipErrorInvalidValue; break;
#else
      default: e = launch_sad_fast_pw1(a, g.wsz, border, split, s); break;   // the other windows: sbm_sad_fast_pw1 / _pw2 / _pw3.hip
#endif
    }
  }
  *xa = a.xc0; *xb = a.xc1;
  return e;
}

}  // namespace sbm
