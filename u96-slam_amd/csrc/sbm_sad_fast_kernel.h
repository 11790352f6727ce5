// sbm_sad_fast_kernel.h -- the interior SAD kernel (grid decode + strip dispatch) and its launchers, instantiated by each of the
// kernel's translation units for its share of the windows (sbm_sad_fast.hip: 15 and 21; _pw1 / _pw2 / _pw3: the other
// windows; _pp: the two-accumulator fallback build of all of them). gfx950 only.
#pragma once
#include "sbm_sad_fast_core.h"
#include "sbm_sad_fast_dev.h"
#if SBM_FAST_PINGPONG
#include "sbm_sad_fast_pp_strip.h"
#define SBM_FAST_KERNEL sad_fast_pp_kernel
#define SBM_FAST_WAVES_PER_EU
#else
#include "sbm_sad_fast_strip.h"
#define SBM_FAST_KERNEL sad_fast_kernel
// wavefronts per SIMD the register allocation aims at: 3 for a 128-disparity wavefront (168 VGPRs with the in-place accumulate),
// 5 for three or four cooperating 64-disparity wavefronts (their two barriers per row want the extra wavefront to cover the
// waits), 4 otherwise (a lone wavefront: measured slower at 5; two cooperating wavefronts of <= 64 disparities only run small
// launches -- one pair per call -- and 128 VGPRs keep their border wavefronts free of scratch: at 96 VGPRs their spill reloads
// were memory round trips inside the serial chain that IS the length of a one-pair SAD stage, 0.089 -> 0.071 ms)
#define SBM_FAST_WAVES_PER_EU __attribute__((amdgpu_waves_per_eu(NDW > 64 ? 3 : (NWAVES > 2 ? 5 : 4))))
#endif

namespace sbm {

// DUAL (windows that are multiples of 3): strips [0, strips3) are column-stride-3 strips in triples, the rest plain ones
template <int NDW, int NWAVES, int NTERM, int PW, bool EXACT_ND, bool DUAL>
__global__ void __launch_bounds__(64 * NWAVES) SBM_FAST_WAVES_PER_EU SBM_FAST_KERNEL(FastArgs a) {
  // XCD-aware decode of the 1-D workgroup id: consecutive ids go round-robin over the 8 XCDs (each with its own
  // 4 MiB L2), so give XCD k the pairs k, k+8, ...: all strips and row segments of a pair then share one L2.
  // (Placement only affects speed; any mapping is correct.)
  // Row segments are the slowest-varying index and get shorter towards the end of the grid: every segment pays w-1
  // priming rows, so few long segments keep that overhead low while the short last ones keep the tail of the launch
  // (CUs idling while the last workgroups finish) short.
  const int bpp = a.strips;                             // workgroups per pair and segment
  int strip, segi, pair;
  {
    // The grid starts with the border jobs (a.bord workgroups for each of their a.nbseg row segments, sbm_sad_border_wave.h):
    // they are long serial chains, so they are dispatched before any strip and finish under the strips instead of behind them.
    if constexpr (NDW * NWAVES <= 256) {   // (a border wavefront holds a disparity quad per lane: up to 256; launch_t never asks beyond)
    if ((int)blockIdx.x < a.bord * a.nbseg) {
      // wavefront wv of border workgroup b takes border wavefront b * NWAVES + wv of its segment; no barriers in there
      using BL = BorderLds<(PW * NTERM) / 2, NDW * NWAVES>;
      const int bseg = blockIdx.x / a.bord, b = blockIdx.x - bseg * a.bord;
      const int wv = NWAVES > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
      const int wi = b * NWAVES + wv;
      if (wi < a.bnw) sad_border_wave<(PW * NTERM) / 2, NDW * NWAVES>(a, reinterpret_cast<unsigned char*>(fast_lds) + wv * BL::BYTES, bseg, wi);
      return;
    }
    }
    const int sid = blockIdx.x - a.bord * a.nbseg;
    const int per_seg = a.strips * a.npairs;
    segi = sid / per_seg;
    const int b = sid - segi * per_seg;
    const int full = (a.npairs / 8) * 8 * bpp;          // ids covered by complete groups of 8 pairs
    int p, inner;
    if (b < full) {
      const int xcd = b & 7, k = b >> 3;
      p = (k / bpp) * 8 + xcd;
      inner = k % bpp;
    } else {
      const int r = b - full;
      p = (a.npairs / 8) * 8 + r / bpp;
      inner = r % bpp;
    }
    pair = p;
    strip = inner;
  }
#if SBM_FAST_PINGPONG
  // LDS of the workgroup: per wavefront one area of WSLOT slots (staged row / exchange), then the merge area
  auto strip_at = [&](auto cs_tag, const int cb) { sad_fast_pp_strip<NDW, NWAVES, NTERM, PW, EXACT_ND, decltype(cs_tag)::value>(a, cb, segi, pair); };
#else
  // LDS of the workgroup: per wavefront the areas of DmaLds (two staged rows, further exchange levels), then the merge area
  const int wvk = NWAVES > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
  unsigned char* const ldsb = reinterpret_cast<unsigned char*>(fast_lds);
  auto strip_at = [&](auto cs_tag, const int cb) {
    constexpr int CSV = decltype(cs_tag)::value;
    using D = DmaLds<NDW, NWAVES, NTERM, PW, CSV>;
    unsigned char* const wb = ldsb + wvk * D::WAVE_B;
    unsigned char* const xl = wb + 2 * D::AREA_B;
    sad_fast_strip_dma<NDW, NWAVES, NTERM, PW, EXACT_ND, CSV>(a, wb, wb + D::AREA_B, xl,
                                                            reinterpret_cast<u32*>(ldsb + NWAVES * D::WAVE_B), cb, segi, pair);
  };
#endif
  if constexpr (DUAL) {
    constexpr int NV3 = 64 - (NTERM - 1), NV1 = 64 - PW * (NTERM - 1);
    if (strip < a.strips3) {
      const int t = strip / 3;
      strip_at(std::integral_constant<int, 3>{}, t * (3 * NV3) + (strip - 3 * t));
    } else {
      strip_at(std::integral_constant<int, 1>{}, (a.strips3 / 3) * (3 * NV3) + (strip - a.strips3) * NV1);
    }
  } else {
    constexpr int NV1 = 64 - PW * (NTERM - 1);
    strip_at(std::integral_constant<int, 1>{}, strip * NV1);
  }
}

// Tuning constants of the border jobs' row segments (chip- and kernel-version specific; they only move time, never results):
// the launch's expected duration is priced at kBorderModelRate pixel-disparities per second (the interior kernel's rate when
// the segments were tuned: 3.6e12, profiles/r04_border_bseg.txt), a border row at kBorderRowUs + kBorderColUs per output
// column (tools/exp/r04_bwprof.py), and a chain may last kBorderChainShare of the launch.
constexpr double kBorderModelRate = 3.6e12, kBorderRowUs = 2.0, kBorderColUs = 0.25, kBorderChainShare = 0.205;

template <int NDW, int NWAVES, int NTERM, int PW>
static hipError_t launch_t(FastArgs a, bool border, hipStream_t s) {
  constexpr bool DUAL = PW == 3;
  constexpr int WSLOT1 = FastLds<NDW, NTERM, PW, 1>::WSLOT, WSLOT3 = FastLds<NDW, NTERM, PW, DUAL ? 3 : 1>::WSLOT;
  constexpr int WSLOTM = WSLOT1 > WSLOT3 ? WSLOT1 : WSLOT3;
  // fallback build: per wavefront the staged-row / exchange area, then the workgroup's merge area
  size_t lds = (size_t)NWAVES * WSLOTM * 16 + (NWAVES > 1 ? (size_t)2 * NWAVES * 64 * (4 + 8) : 0);
  if (!SBM_FAST_PINGPONG) {  // LDS-direct strips: per wavefront the areas of DmaLds, then the merge area
    constexpr int WB1 = DmaLds<NDW, NWAVES, NTERM, PW, 1>::WAVE_B, WB3 = DmaLds<NDW, NWAVES, NTERM, PW, DUAL ? 3 : 1>::WAVE_B;
    lds = (size_t)NWAVES * (WB1 > WB3 ? WB1 : WB3) + (NWAVES > 1 ? (size_t)2 * NWAVES * 64 * (4 + 8) : 0);
  }
  a.bord = a.bnw = 0;
  a.bseg = a.row1 - a.row0;
  a.nbseg = 0;
  if constexpr (NDW * NWAVES <= 256) if (border) {
    // border wavefronts per segment: 2 sides x ceil(n / JW) groups of JW consecutive pairs, NWAVES of them per workgroup
    using BL = BorderLds<(PW * NTERM) / 2, NDW * NWAVES>;
    a.bnw = 2 * ((a.npairs + BL::JW - 1) / BL::JW);
    a.bord = (a.bnw + NWAVES - 1) / NWAVES;
    lds = std::max(lds, (size_t)NWAVES * BL::BYTES);
    // A border wavefront is a serial chain of rows (~2 us + 0.25 us per output column and row, a quarter of that for each of
    // its w-1 priming rows -- measured alone on the chip, tools/exp/r04_bwprof.py); it must end well inside the launch, so the
    // border jobs get their own, finer row segments: a chain of about a fifth of the launch's expected duration (640x480 nd 64
    // w 21 x 64 pairs, where the border columns weigh most: 10 rows per segment 0.520 ms per step, 13 rows 0.537, 6 rows 0.533 --
    // profiles/r04_border_bseg.txt; KITTI x 64 is flat between 32 and 96 rows).
    const int rows = a.row1 - a.row0, wsz = PW * NTERM;
    const double t_kernel_us = (double)a.npairs * a.W * rows * a.nd / kBorderModelRate * 1e6;
    const double t_row_us = kBorderRowUs + kBorderColUs * (wsz / 2);
    int bseg = (int)(kBorderChainShare * t_kernel_us / t_row_us - 0.25 * (wsz - 1));
    if (fast_tune().bseg > 0) bseg = fast_tune().bseg;
    bseg = std::max(4, std::min(bseg, rows));
    a.nbseg = (rows + bseg - 1) / bseg;
    a.bseg = (rows + a.nbseg - 1) / a.nbseg;
    a.nbseg = (rows + a.bseg - 1) / a.bseg;
  }
  dim3 grid((unsigned)(a.bord * a.nbseg + a.strips * a.npairs * a.nseg));
  if (fast_tune().print)   // development builds: the launch geometry
    fprintf(stderr, "[sbm] <%d,%d,%d,%d> strips %d (cs3 %d) nseg %d pairs %d bord %d x %d grid %u lds %zu\n", NDW, NWAVES, NTERM, PW, a.strips, a.strips3,
            a.nseg, a.npairs, a.bord, a.nbseg, grid.x, lds);
  // development builds: time the border wavefronts alone (results are wrong by construction)
  if (fast_tune().border_only) grid.x = (unsigned)(a.bord * a.nbseg);
  // (the fallback build only carries the masked-count kernels: they are right for every count up to NDW * NWAVES)
  // ... and <64,4> only runs one-pair calls beyond 192 disparities: its masked kernel serves 256 as well
  // (three and four 128-disparity wavefronts, 257 .. 512 disparities: exact kernels for 384 and 512 -- 5-7 % over the masked ones,
  // profiles/r05_exact512.txt)
  constexpr bool HAS_EXACT = !SBM_FAST_PINGPONG && !(NDW == 64 && NWAVES == 4);
  const bool exact = HAS_EXACT && a.nd == NDW * NWAVES;
  snprintf(g_sad_kernel_name, sizeof(g_sad_kernel_name), "%s<%d,%d,%d,%d,%s,%s> pfshift=%d", SBM_FAST_PINGPONG ? "sad_fast_pp_kernel" : "sad_fast_kernel",
           NDW, NWAVES, NTERM, PW, exact ? "true" : "false", DUAL ? "true" : "false", a.pfshift);
  if constexpr (HAS_EXACT) {
    if (exact) {
      hipLaunchKernelGGL((SBM_FAST_KERNEL<NDW, NWAVES, NTERM, PW, true, DUAL>), grid, dim3(64 * NWAVES), lds, s, a);
      return hipGetLastError();
    }
  }
    hipLaunchKernelGGL((SBM_FAST_KERNEL<NDW, NWAVES, NTERM, PW, false, DUAL>), grid, dim3(64 * NWAVES), lds, s, a);
  return hipGetLastError();
}

// Layout by disparity count: one wavefront holds a pixel's whole range up to 128 disparities (32 / 64 / 128 per wavefront: no
// barriers, no merge, every per-row fixed cost paid once; 168 VGPRs = 3 wavefronts per SIMD at 128 with the in-place
// accumulate); two, three, four cooperating 128-disparity wavefronts up to 256 / 384 / 512 (round 4: 1080p nd 256 2.58 -> 2.21 ms
// per step against four 64-disparity wavefronts) -- except at exactly 192, where three 64-disparity wavefronts have no masked
// disparities to carry (1080p nd 192: 2.09 against 2.51 ms; profiles/r04_dma_nd.txt). `split`: launches that cannot fill the chip
// (one pair per call) spread the disparities over two to four narrower wavefronts instead -- half the serial work per row.
// The fallback build has the 64-disparity layouts only.
template <int NTERM, int PW>
static hipError_t launch_nd(const FastArgs& a, bool border, bool split, hipStream_t s) {
#if SBM_FAST_PINGPONG
  if (a.nd <= 64) return launch_t<64, 1, NTERM, PW>(a, border, s);
  if (a.nd <= 128) return launch_t<64, 2, NTERM, PW>(a, border, s);
  if (a.nd <= 192) return launch_t<64, 3, NTERM, PW>(a, border, s);
  return launch_t<64, 4, NTERM, PW>(a, border, s);
#else
  if (a.nd > 384) return launch_t<128, 4, NTERM, PW>(a, border, s);
  if (a.nd > 256) return launch_t<128, 3, NTERM, PW>(a, border, s);
  if (a.nd <= 32) return launch_t<32, 1, NTERM, PW>(a, border, s);
  if (a.nd <= 64) return split ? launch_t<32, 2, NTERM, PW>(a, border, s) : launch_t<64, 1, NTERM, PW>(a, border, s);
  if (!split && a.nd <= 128) return launch_t<128, 1, NTERM, PW>(a, border, s);
  if (!split && a.nd != 192) return launch_t<128, 2, NTERM, PW>(a, border, s);
  if (a.nd <= 128) return launch_t<64, 2, NTERM, PW>(a, border, s);
  if (a.nd <= 192) return launch_t<64, 3, NTERM, PW>(a, border, s);
  return launch_t<64, 4, NTERM, PW>(a, border, s);
#endif
}

// window dispatch of the other translation units (each holds the kernels of its windows; an unknown window falls through to the
// next unit and ends as hipErrorInvalidValue)
hipError_t launch_sad_fast_pw1(const FastArgs& a, int wsz, bool border, bool split, hipStream_t s);   // sbm_sad_fast_pw1.hip: 5, 7, 9, 11, 13
hipError_t launch_sad_fast_pw2(const FastArgs& a, int wsz, bool border, bool split, hipStream_t s);   // sbm_sad_fast_pw2.hip: 17, 19, 23, 25
hipError_t launch_sad_fast_pw3(const FastArgs& a, int wsz, bool border, bool split, hipStream_t s);   // sbm_sad_fast_pw3.hip: 27, 29, 31
hipError_t launch_sad_fast_pp(const FastArgs& a, int wsz, bool border, bool split, hipStream_t s);    // sbm_sad_fast_pp.hip: the fallback build, 5 .. 27

}  // namespace sbm
