// sbm_speckle.hip -- speckle filter of the disparity map (connected components of similar disparities, small ones erased). gfx950.
//
// Device counterpart of filterSpeckles (OpenCV calib3d stereosgbm.cpp), switched on by setSpeckleWindowSize(50),
// setSpeckleRange(32) at src/slam/src/core/main.cpp:210-211. No FPGA twin in the reference.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "sbm_common.h"

namespace sbm {

// ---------------------------------------------------------------------------------------------------------
// Speckle filter.  cv's raster-order flood fill yields plain 4-connected components of the relation
// "both != newVal and |a-b| <= maxDiff" (order-independent), so it is computed here as
// run-based union-find, in two implementations:
//  * default (further down: speckle_band_kernel, _seam_, _count_list_, _apply_list_): a wavefront walks a band of rows,
//    runs get compact 16-byte records, everything after the walk is driven by those records;
//  * row-walking kernels (right below; SBM_SPECKLE_BAND=0 / SBM_SPECKLE_LISTS=0, images of 2^27 pixels and more, rows wider
//    than 65535): ONE WAVEFRONT PER IMAGE ROW walks it in 64-pixel chunks; run membership is recomputed from the disparity
//    row with ballots (head = valid pixel not connected to its left neighbour), so only run HEADS own an entry in labels[]
//    (parent pointer) and counts[] (run length, later the component size at the root):
//   1. runs    heads: labels[head] = head, counts[head] = run length.
//   2. merge   rows y and y+1 together: the first pixel of every vertical contact between two runs unions them
//              (atomicMin hooks); later pixels of the same contact are skipped.
//   3. count   heads that are not the root of their component add their run length to the root, unless the root
//              is already known to exceed maxSpeckleSize (saturating: exact where it matters, no contention).
//   4. apply   heads look up their component size; the decision is broadcast along the run; small -> newVal.
// ---------------------------------------------------------------------------------------------------------
// find with path halving: a visited node is re-pointed at its grandparent with a fire-and-forget atomicMin (parents only
// ever move towards the root, so concurrent hooks are never undone)
// SCOPE: agent where wavefronts of other workgroups (possibly on another XCD, behind another L2) union into the same
// labels; workgroup where only the calling wavefront touches them during the kernel (band walk): those atomics are served by
// the XCD's own L2 instead of going out to the fabric -- several times shorter dependent round trips.
// STRIDE: ints between two parent words -- 1 for the per-pixel label plane of the row-walking kernels, 4 for the run records
// of the band walk (SpkRun::parent).
template <int SCOPE = __HIP_MEMORY_SCOPE_AGENT, int STRIDE = 1>
__device__ __forceinline__ int uf_find(int* L, int i) {
  int r = i;
  for (;;) {
    const int p = __hip_atomic_load(L + (size_t)r * STRIDE, __ATOMIC_RELAXED, SCOPE);
    if (p == r) return r;
    const int gp = __hip_atomic_load(L + (size_t)p * STRIDE, __ATOMIC_RELAXED, SCOPE);
    if (gp == p) return p;
    __hip_atomic_fetch_min(L + (size_t)r * STRIDE, gp, __ATOMIC_RELAXED, SCOPE);
    r = gp;
  }
}

template <int SCOPE = __HIP_MEMORY_SCOPE_AGENT, int STRIDE = 1>
__device__ __forceinline__ void uf_union(int* L, int a, int b) {
  for (;;) {
    a = uf_find<SCOPE, STRIDE>(L, a);
    b = uf_find<SCOPE, STRIDE>(L, b);
    if (a == b) return;
    if (a > b) { const int t = a; a = b; b = t; }
    const int old = __hip_atomic_fetch_min(L + (size_t)b * STRIDE, a, __ATOMIC_RELAXED, SCOPE);
    if (old == b) return;
    b = old;
  }
}

// root lookup once parents are final (after the merge kernel): plain, cacheable loads
template <int STRIDE = 1>
__device__ __forceinline__ int uf_root_final(const int* L, int i) {
  int r = i;
  for (int p = L[(size_t)r * STRIDE]; p != r; p = L[(size_t)r * STRIDE]) r = p;
  return r;
}

// Walks one row left to right in 64-pixel chunks. The per-chunk state lives in wavefront-uniform 64-bit masks (valid
// pixels, run heads) that cost a handful of compares plus scalar mask algebra; the per-lane "column where my run
// starts" -- the expensive part -- is only computed on demand (start()), i.e. in the few chunks where a kernel has
// something to do (a vertical contact, a run to erase). State carried across chunks: prev_last (value left of the
// chunk; invalid pixels hold newval), carry (column of the row's last run head so far).
// lane predicate / prefix count of a wavefront-uniform mask without per-lane 64-bit arithmetic (exec := mask; v_mbcnt)
__device__ __forceinline__ bool lane_in(unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
__device__ __forceinline__ int lanes_below(unsigned long long m) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
// element idx of a wavefront-uniform array through a 32-bit byte offset (scalar base + vector offset addressing); the
// callers guarantee idx * sizeof(T) < 4 GiB
template <typename T>
__device__ __forceinline__ T* at32(T* base, unsigned idx) {
  return (T*)((const char*)base + (size_t)(idx * (unsigned)sizeof(T)));
}

struct RowWalk {
  int prev_last, carry, cb;         // value left of the chunk; column of the last run head seen so far in the row
  unsigned long long valid, head;   // uniform
  __device__ __forceinline__ void init(int newval) { prev_last = newval; carry = -1; }
  // masks of the chunk at column cb_ (lanes beyond the row hold newval = invalid); maxdiff <= 2^17 (launcher clamps it)
  __device__ __forceinline__ void step(int v, int cb_, int lane, int newval, int maxdiff) {
    cb = cb_;
    // value of the left neighbour: full-rate DPP wave shift; lane 0 has no source lane and keeps `old` = prev_last
    // (__shfl_up would be a ds_bpermute_b32 -- an LDS crossbar round trip inside the serial chain of the walk)
    const int pv = __builtin_amdgcn_update_dpp(prev_last, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    valid = __ballot(v != newval);
    // distance to the left neighbour, "far" when that pixel is invalid (invalid pixels hold newval): one compare then
    // yields the lanes that are NOT joined to the left, instead of mask algebra on the scalar unit (these walks are bound
    // by scalar issue: one SALU instruction per SIMD every 4 cycles)
    const int dist = pv != newval ? abs(v - pv) : 0x40000000;
    head = valid & __ballot(dist > maxdiff);
  }
  // column where the lane's run starts (meaningful for valid lanes: a valid lane with no head at or below it in this chunk
  // is joined, through valid pixels only, to the last head of the earlier chunks)
  __device__ __forceinline__ int start(int lane) const {
    const unsigned long long t = head << (63 - lane);   // heads at or below this lane, the nearest one in bit 63
    return t ? cb + lane - __clzll((long long)t) : carry;
  }
  // advance the carried state to the next chunk (uniform arithmetic only)
  __device__ __forceinline__ void next(int v) {
    carry = head ? cb + (63 - __clzll((long long)head)) : carry;
    prev_last = __builtin_amdgcn_readlane(v, 63);
  }
};

// Rows are walked in groups of SPK_G chunks whose values are loaded up front (SPK_G independent loads in flight per
// lane) -- the walk itself is a serial chain, so without this every chunk would pay a full memory latency.
constexpr int SPK_G = 8;
__device__ __forceinline__ void spk_load_group(const int16_t* d, int cb0, int W, int lane, int newval, int (&v)[SPK_G]) {
#pragma unroll
  for (int g = 0; g < SPK_G; g++) {
    const int x = cb0 + 64 * g + lane;
    v[g] = x < W ? (int)d[x] : newval;
  }
}

#define SPK_ROW_SETUP                                          \
  const int lane = threadIdx.x & 63;                           \
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); /* uniform: keeps row pointers in SGPRs */ \
  const int y = blockIdx.x * 4 + wave;                         \
  const size_t plane_off = (size_t)blockIdx.y * W * H;         \
  (void)lane;

// grid: (ceil(H/4), n), block 256 = 4 wavefronts = 4 rows
__global__ void __launch_bounds__(256) speckle_runs_kernel(const int16_t* __restrict__ disp, int* __restrict__ labels,
                                                            int* __restrict__ counts, int W, int H, int newval,
                                                            int maxdiff) {
  SPK_ROW_SETUP
  if (y >= H) return;
  const int16_t* d = disp + plane_off + (size_t)y * W;
  int* L = labels + plane_off + (size_t)y * W;
  int* C = counts + plane_off + (size_t)y * W;
  // every head learns the distance to the next boundary (head, invalid pixel or row end). A run that leaves its chunk
  // stays "open" (uniform state) and is closed by the first boundary of a later chunk.
  RowWalk rw;
  rw.init(newval);
  int open_start = -1;
  for (int cb0 = 0; cb0 < W; cb0 += 64 * SPK_G) {
    int vs[SPK_G];
    spk_load_group(d, cb0, W, lane, newval, vs);
#pragma unroll
    for (int g = 0; g < SPK_G; g++) {
      const int cb = cb0 + 64 * g;
      if (cb >= W) break;
      rw.step(vs[g], cb, lane, newval, maxdiff);
      const unsigned long long hm = rw.head, bm = hm | ~rw.valid;   // lanes beyond W are invalid = boundary
      if (open_start >= 0 && bm) {
        if (lane == 0) C[open_start] = cb + (__ffsll((long long)bm) - 1) - open_start;
        open_start = -1;
      }
      if (hm) {                                                      // uniform: most chunks hold no run head
        if ((hm >> lane) & 1ull) {
          const unsigned long long above = lane == 63 ? 0ull : (bm >> (lane + 1));
          L[cb + lane] = y * W + cb + lane;
          if (above) C[cb + lane] = __ffsll((long long)above);      // else: closed by a later chunk (or the row end)
        }
        const int hb = 63 - __clzll((long long)hm);                  // the last head stays open if nothing bounds it
        if (hb == 63 || (bm >> (hb + 1)) == 0ull) open_start = cb + hb;
      }
      rw.next(vs[g]);
    }
  }
  if (lane == 0 && open_start >= 0) C[open_start] = W - open_start;
}

// The row walk only COLLECTS the contacts (pairs of run heads) into a wavefront-private LDS list; the unions -- chains
// of dependent L2 round trips -- then run 64 at a time. Doing them inside the walk made every 64-pixel chunk that holds
// a contact pay a full union latency with one or two lanes busy.
constexpr int SPK_CAP = 448;   // contacts buffered per wavefront (flushed when fewer than 64 free slots remain)
__global__ void __launch_bounds__(256) speckle_merge_kernel(const int16_t* __restrict__ disp, int* __restrict__ labels,
                                                             int W, int H, int newval, int maxdiff) {
  __shared__ int2 contact_lds[4][SPK_CAP];
  SPK_ROW_SETUP
  if (y >= H - 1) return;
  const int16_t* du = disp + plane_off + (size_t)y * W;
  const int16_t* dd = du + W;
  int* L = labels + plane_off;
  int2* const list = contact_lds[wave];
  int count = 0;   // uniform
  auto flush = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < count; i += 64) {
      const int2 c = list[i];
      uf_union(L, c.x, c.y);
    }
    __builtin_amdgcn_wave_barrier();
    count = 0;
  };
  RowWalk up, dn;
  up.init(newval);
  dn.init(newval);
  bool prev_cd = false;  // vertical contact at the pixel left of the chunk
  for (int cb0 = 0; cb0 < W; cb0 += 64 * SPK_G) {
    int vus[SPK_G], vds[SPK_G];
    spk_load_group(du, cb0, W, lane, newval, vus);
    spk_load_group(dd, cb0, W, lane, newval, vds);
#pragma unroll
    for (int g = 0; g < SPK_G; g++) {
      const int cb = cb0 + 64 * g;
      if (cb >= W) break;
      const int vu = vus[g], vd = vds[g];
      up.step(vu, cb, lane, newval, maxdiff);
      dn.step(vd, cb, lane, newval, maxdiff);
      const unsigned long long cdm = up.valid & dn.valid & __ballot(abs(vu - vd) <= maxdiff);
      const unsigned long long pcdm = (cdm << 1) | (prev_cd ? 1ull : 0ull);
      // same two runs as the pixel to the left and that pixel already made the contact -> nothing new
      const unsigned long long fm = cdm & ~(pcdm & ~up.head & ~dn.head);
      if (fm) {                                                // uniform: most chunks hold no new contact
        if ((fm >> lane) & 1ull)
          list[count + __popcll(fm & ((1ull << lane) - 1ull))] = make_int2(y * W + up.start(lane), (y + 1) * W + dn.start(lane));
        count += __popcll(fm);
        if (count > SPK_CAP - 64) flush();
      }
      prev_cd = (cdm >> 63) & 1ull;
      up.next(vu);
      dn.next(vd);
    }
  }
  if (count) flush();
}

// ---- band walk: runs + merge in one pass ---------------------------------------------------------------------------------
// G consecutive rows (a band) are walked together, chunk by chunk (64 columns), with the band's next row as a look-ahead -- by
// one wavefront, or by the S wavefronts of a workgroup with one column segment each (further down). Contacts between two rows
// of the band are unioned inside the kernel (nobody else touches the band's runs in it); contacts across the seam to the next
// band go to a list per band and segment (upper run | lower run << 16) and are unioned by speckle_seam_kernel once every band
// is done.
//
// Data: a RUN is the unit of everything downstream, so runs get compact 16-byte records -- run k of row y (and segment) lives in
// slot spk_slot(...) of the pair's record area: parent (union-find), size, first and last column. A row of a disparity map holds
// a handful of runs, so its records share one or two cache lines (labels and sizes indexed by the PIXEL of the run head, as the
// row-walking kernels above keep them, cost a cache line of HBM traffic per run and array). The k-th run of a row is the same
// run for the wavefront that owns the row and for the one that looks ahead into it, so contacts name runs by their index and
// the walk carries a run COUNT per row.
// Walk: built around what a disparity map looks like -- long runs. Per chunk, plain per-lane arithmetic decides whether ANYTHING
// happens in it: every pixel of every row valid and joined to its left neighbour, every vertical pair in contact (also at the
// pixel left of the chunk)  =>  no run starts, no run ends, no new contact, no state changes: the chunk costs ~45 vector
// instructions and the scalar unit nothing (86 % of the (row, chunk) cells of the bench frames; the general path --
// wavefront-uniform mask algebra, list appends -- runs on the rest). Chunks without a single valid pixel behind such a chunk are
// skipped likewise.
//  * Invalid pixels are replaced by a value that is far from every valid one AND from the substitutes of the four
//    neighbours (it alternates with lane and row parity), so "both valid and |a - b| <= maxDiff" is ONE unsigned compare,
//    t = a + maxDiff - b <= 2 maxDiff, and the all-quiet test is one v_max3 tree over the t's + one compare.
//  * The pixel left of the chunk comes from the previous chunk's registers (DPP wave_ror / wave_shr), not from carried
//    scalar state; run ends are detected at the pixel to their right, so a chunk never owes the next one anything.
// (What was measured on the way and not kept: profiles/r06_speckle.md.)
struct SpkRun { int parent, size, first, last; };
// first: bit 31 marks the root of an IN-BAND component (set by the band walk's last phase). size: at such a root the component's
// size -- in-band pixels first, then what the count kernel adds for roots hooked under it -- or >= kSpkBig once something
// larger than maxSpeckleSize is known to touch it; at every other run the in-band size of its component (a hint that lets the
// later kernels decide the bulk -- runs of components that are large inside their own band -- without leaving the record).
constexpr int kSpkRootFlag = (int)0x80000000u, kSpkBig = 0x40000000, kSpkMaxSize = 2048;
// Where the records live. A row of a disparity map holds a handful of runs, and a record plane indexed y * W + k would put every
// row's few records on a page of its own (19 KB apart at KITTI width): the kernels behind the band walk touch every row once and
// spent ~10 us each on 24 000 cold TLB entries (64 KITTI pairs). So the first kSpkDense runs of a row live in a dense block (256
// bytes per row: 6 MB for that batch), only the runs beyond them in the per-pixel plane behind it. Same for the seam lists: the
// first kSpkDenseSeam contacts of a band's seam in a dense block, the rest in the band's W slots.
constexpr unsigned kSpkDenseSeam = 32;
constexpr int kSpkSeg4 = 1000, kSpkSeg2 = 2500;   // launches of fewer pairs of rows than this are cut into 4 / 2 column segments (launch_speckle)
__device__ __forceinline__ unsigned spk_slot(unsigned y, unsigned k, unsigned W, unsigned H, unsigned dense) {
  return k < dense ? y * dense + k : H * dense + y * W + k;
}
__device__ __forceinline__ unsigned spk_seam_slot(unsigned band, unsigned i, unsigned nbands, unsigned HS) {
  return i < kSpkDenseSeam ? band * kSpkDenseSeam + i : nbands * kSpkDenseSeam + band * HS + i;
}
constexpr int SPK_BG = 4;   // chunks loaded up front per row
constexpr int kSpkFar = 0x20000000, kSpkFarStep = 0x01000000, kSpkFarMin = 0x10000000;

// In-band bookkeeping of the band walk. A band (G rows) is walked by S wavefronts of one workgroup, one column segment each
// (S = 1, 2 or 4: launch_speckle); for the books a segment of a row is a row of its own ("virtual row" y * S + segment: its own
// run numbering, record slots, run count and seam list), so the segments walk independently and only the runs that cross a
// segment boundary need joining afterwards -- in LDS, behind a workgroup barrier.
// LOCAL mode: the runs of the walk -- at most kSpkCapR per row and segment -- live in LDS while the band is walked (parent, first,
// last per run; id = wavefront * N + row * kSpkCapR + k), contacts inside the band are listed there and unioned 64 at a time
// (LDS atomics: ~100 ns a round trip instead of an L2's), and every record is written to memory ONCE, complete, when the band is
// done: parent = in-band root, in-band size, first (+ root flag), last. GLOBAL mode (the path of a band with more than kSpkCapR
// runs in some row of some segment; 18 us slower per band wavefront where it was the only mode): records created and closed in
// memory during the walk, contacts unioned by L2 atomics, two more sweeps of dependent L2 round trips for sizes and flags.
// (table sizes of 64 / 96 runs per row and workgroups of 1 / 2 wavefronts at one segment per band were measured: re-walks of the
// busiest bands, and bands that no longer share their look-ahead rows in the CU's cache -- profiles/r06_speckle.md)
template <int G>
constexpr int kSpkCapR = G == 2 ? 256 : 128;
constexpr int kSpkWaves = 4;   // wavefronts per workgroup of the band walk: 4 / S bands x S column segments
template <int G, int NW = kSpkWaves>
struct SpkWgLds {
  static constexpr int N = G * kSpkCapR<G>;   // run ids of one wavefront
  int par[NW * N];
  unsigned short first[NW * N], last[NW * N];   // (the band walk serves up to 65 535 columns)
  // walk: the wavefront's own N entries are its contact list (LOCAL: upper id | lower id << 16; GLOBAL: N / 2 pairs of record
  // slots); last phase: in-band size of a component, at its root
  unsigned cs[NW * N];
  int nh[NW][G];   // runs per owned row of every wavefront (the segment to the right joins its first run to this one's last)
  int ovf[NW];     // LOCAL mode ran out of run ids
};
static_assert(SpkWgLds<2>::N / 2 > 64 && SpkWgLds<4>::N / 2 > 64 * 3 && 4 * SpkWgLds<2>::N <= 65536 && 4 * SpkWgLds<4>::N <= 65536, "");

__device__ __forceinline__ int lds_find(int* par, int i) {
  for (;;) {
    const int p = __hip_atomic_load(par + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (p == i) return i;
    i = p;
  }
}
__device__ __forceinline__ void lds_union(int* par, int a, int b) {
  for (;;) {
    a = lds_find(par, a);
    b = lds_find(par, b);
    if (a == b) return;
    if (a > b) { const int t = a; a = b; b = t; }
    const int old = __hip_atomic_fetch_min(par + b, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (old == b) return;
    b = old;
  }
}

// Where everything of a pair lives: S column segments of SW columns, HV = H * S virtual rows, nbands * S virtual bands.
template <int S>
struct SpkLayout {
  static constexpr int D = S == 1 ? 16 : 8;   // records of a virtual row in the dense block
  int SW;
  __host__ __device__ size_t records(int H) const { return (size_t)H * S * ((size_t)D + SW); }
  __host__ __device__ size_t seam_slots(int nbands) const { return (size_t)nbands * S * ((size_t)kSpkDenseSeam + SW); }
};

// The walk of one column segment [cs, ce) of one band. Returns true when LOCAL mode ran out of run ids (nothing it wrote matters
// then: the band is walked again in GLOBAL mode). nh[r]: runs of row r in this segment; nsm: seam contacts listed.
template <int G, int S, bool LOCAL>
__device__ __forceinline__ bool speckle_band_walk(const int16_t* __restrict__ d, SpkRun* __restrict__ R, unsigned* __restrict__ sl,
                                                  const SpkLayout<S> lay, int W, int H, int newval, int maxdiff, SpkWgLds<G>& lds,
                                                  const int wave, const int band, const int seg, const int cs, const int ce,
                                                  int (&nh)[G + 1], int& nsm) {
  constexpr int CAPR = kSpkCapR<G>;
  constexpr int N = SpkWgLds<G>::N;
  const int lane = threadIdx.x & 63;
  const int nbandsv = ((H + G - 1) / G) * S, bandv = band * S + seg;
  const int SW = lay.SW;
  const int y0 = band * G;
  const int idbase = wave * N;
  int* const P = &R->parent;                                   // parent of slot i: P[4 i]
  unsigned* const clist = lds.cs + idbase;
  int2* const list = reinterpret_cast<int2*>(clist);
  auto slot = [&](int r, int k) -> unsigned { return spk_slot((unsigned)((y0 + r) * S + seg), (unsigned)k, (unsigned)SW, (unsigned)(H * S), (unsigned)lay.D); };
  int count = 0;   // uniform: buffered in-band contacts
  nsm = 0;
  auto flush = [&]() {
    if constexpr (LOCAL) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      for (int i = lane; i < count; i += 64) {
        const unsigned c = clist[i];
        lds_union(lds.par, (int)(c & 0xffffu), (int)(c >> 16));
      }
    } else {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // record stores of this wavefront have left before the unions start
      __builtin_amdgcn_wave_barrier();
      for (int i = lane; i < count; i += 64) {
        const int2 c = list[i];
        uf_union<__HIP_MEMORY_SCOPE_WORKGROUP, 4>(P, c.x, c.y);
      }
    }
    __builtin_amdgcn_wave_barrier();
    count = 0;
  };
  const unsigned tm = 2u * (unsigned)maxdiff;
  // substitutes of invalid pixels: rows of even / odd parity (bands start on even rows: G is even)
  const int far_e = kSpkFar + ((lane & 1) ? kSpkFarStep : 0), far_o = kSpkFar + ((lane & 1) ? 0 : kSpkFarStep);

  int ne[G];   // uniform: runs ended so far (nh: runs started, owned rows and the look-ahead row)
#pragma unroll
  for (int r = 0; r <= G; r++) nh[r] = 0;
#pragma unroll
  for (int r = 0; r < G; r++) ne[r] = 0;
  int vp[G + 1];              // the previous chunk's (substituted) values; its lane 63 is the pixel left of this chunk
#pragma unroll
  for (int r = 0; r <= G; r++) vp[r] = (r & 1) ? far_o : far_e;   // a row starts behind a boundary
  unsigned mvp = 0u;          // the previous chunk's vertical maximum (contact state at the pixel left of the chunk)
  bool prev_empty = true;     // uniform: the previous chunk held no valid pixel (nothing pending at its right edge)

  // Loads and stores share one counter (vmcnt) and may complete out of order, so a wait for a load with stores in flight
  // is a wait for ALL of them. Hence: a group's values are consumed (pinned) before the group's first store, and the next
  // group's loads are issued right then, so that they are in flight during this group's arithmetic and stores.
  // (buffer loads: row offset in an SGPR, chunk offset in the instruction, one VGPR of column offsets per group; columns
  // beyond the row read the next row or -- beyond the plane -- zero, rows are clamped into the image: such values become
  // newval when consumed)
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<int16_t*>(d), 0, 2 * W * H, 0x00020000);
  int vn[G + 1][SPK_BG];
  auto load_group = [&](int cb0) {
    const int vo = 2 * (cb0 + lane);
#pragma unroll
    for (int r = 0; r <= G; r++) {
      const int so = 2 * min(y0 + r, H - 1) * W;   // (columns beyond the segment are loaded and not used)
#pragma unroll
      for (int g = 0; g < SPK_BG; g++) vn[r][g] = (int)(short)__builtin_amdgcn_raw_buffer_load_b16(rs_d, vo + 128 * g, so, 0);
    }
  };
  load_group(cs);
  for (int cb0 = cs; cb0 < ce; cb0 += 64 * SPK_BG) {
    int vs[G + 1][SPK_BG];
    const bool inside = y0 + G < H && cb0 + 64 * SPK_BG <= ce;   // uniform: the whole group lies inside the image
#pragma unroll
    for (int r = 0; r <= G; r++) {
      const bool row_ok = y0 + r < H;   // uniform
#pragma unroll
      for (int g = 0; g < SPK_BG; g++) {
        vs[r][g] = (inside || (row_ok && cb0 + 64 * g + lane < ce)) ? vn[r][g] : newval;
        asm volatile("" : "+v"(vs[r][g]));   // consumed here, not at the first use further down
      }
    }
    if (cb0 + 64 * SPK_BG < ce) load_group(cb0 + 64 * SPK_BG);
#pragma unroll
    for (int g = 0; g < SPK_BG; g++) {
      const int cb = cb0 + 64 * g;
      if (cb >= ce) break;
      // ---- per-lane arithmetic of every chunk
      int vq[G + 1], pv[G + 1];
      unsigned th[G + 1], tv[G + 1];
      unsigned long long vm[G + 1];
#pragma unroll
      for (int r = 0; r <= G; r++) {
        const bool ok = vs[r][g] != newval;
        vm[r] = __ballot(ok);
        vq[r] = ok ? vs[r][g] : ((r & 1) ? far_o : far_e);
        const int a = vq[r] + maxdiff;
        const int t0 = __builtin_amdgcn_update_dpp(vp[r], vp[r], 0x13C /* wave_ror:1 */, 0xf, 0xf, false);   // lane 0 <- lane 63
        pv[r] = __builtin_amdgcn_update_dpp(t0, vq[r], 0x138 /* wave_shr:1 */, 0xf, 0xf, false);        // left neighbour
        th[r] = (unsigned)(a - pv[r]);                    // <= tm: joined to the left neighbour
        tv[r] = r ? (unsigned)(a - vq[r - 1]) : 0u;       // <= tm: in contact with the pixel above
      }
      unsigned mh = th[0], mv = tv[1];
#pragma unroll
      for (int r = 1; r <= G; r++) mh = max(mh, th[r]);
#pragma unroll
      for (int r = 2; r <= G; r++) mv = max(mv, tv[r]);
      // contact state left of the chunk: lanes 0..3 see the previous chunk's lanes 63, 0, 1, 2 (conservative: a chunk that
      // is declined here is simply walked in full)
      const unsigned ml = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mvp, 0x13C, 0x1, 0x1, false);
      const bool quiet = __ballot(max(max(mh, mv), ml) > tm) == 0ull;
      bool empty = false;
      if (!quiet) {
        unsigned long long anyv = vm[0];
#pragma unroll
        for (int r = 1; r <= G; r++) anyv |= vm[r];
        empty = anyv == 0ull;
      }
      if (!quiet && !(empty && prev_empty)) {
        // ---- general path. Run heads of every row; the vertical contacts first (a lane's run is the row's count of runs
        // before this chunk + the heads at or below the lane), then the records of the owned rows.
        unsigned long long head[G + 1];
        int nhd[G + 1], run[G + 1];   // heads of the row in this chunk (uniform); index of the lane's run within its row (valid lanes only)
#pragma unroll
        for (int r = 0; r <= G; r++) {
          head[r] = vm[r] & __ballot(th[r] > tm);
          nhd[r] = __popcll(head[r]);
          run[r] = nh[r] - 1 + lanes_below(head[r]) + (lane_in(head[r]) ? 1 : 0);
        }
        if constexpr (LOCAL) {
          // out of slots? (uniform) -> the band is walked again in GLOBAL mode
          bool full = false;
#pragma unroll
          for (int r = 0; r < G; r++) full |= nh[r] + nhd[r] > CAPR;
          if (full) return true;
        }
        // (the blocks below are predicated, not branched around: an empty mask skips the body by itself, and the uniform counts
        // are two scalar instructions -- cheaper than the compare-and-branch that would avoid them)
#pragma unroll
        for (int r = 1; r <= G; r++) {
          const unsigned long long cdm = __ballot(tv[r] <= tm);
          const unsigned long long pcdm = __ballot((unsigned)(pv[r] + maxdiff - pv[r - 1]) <= tm);   // the same, one pixel left
          // first pixel of a contact between two runs: not the same two runs as at the pixel to the left
          const unsigned long long fm = cdm & (head[r - 1] | head[r] | ~pcdm);
          if (lane_in(fm)) {
            const int k = lanes_below(fm);
            const int ku = run[r - 1], kd = run[r];
            if (r < G) {
              if constexpr (LOCAL) clist[count + k] = (unsigned)(idbase + (r - 1) * CAPR + ku) | ((unsigned)(idbase + r * CAPR + kd) << 16);
              else list[count + k] = make_int2((int)slot(r - 1, ku), (int)slot(r, kd));
            } else {
              *at32(sl, spk_seam_slot(bandv, (unsigned)(nsm + k), nbandsv, SW)) = (unsigned)ku | ((unsigned)kd << 16);
            }
          }
          if (r < G) count += __popcll(fm);
          else nsm += __popcll(fm);
        }
        // run records of the owned rows, in column order: a head opens the row's next record, a boundary pixel (a head or an
        // invalid pixel) right of a valid one closes the oldest open one -- the k-th start and the k-th end are the same run.
#pragma unroll
        for (int r = 0; r < G; r++) {
          const unsigned long long hm = head[r];
          const unsigned long long em = __ballot(pv[r] < kSpkFarMin) & (hm | ~vm[r]);
          if (lane_in(hm)) {   // (a head's run index is the row's count before the chunk + the heads left of it)
            if constexpr (LOCAL) {
              const int id = idbase + r * CAPR + run[r];
              lds.par[id] = id;
              lds.first[id] = (unsigned short)(cb + lane);
            } else {
              const unsigned self = slot(r, run[r]);
              int* q = &at32(R, self)->parent;
              // parent = self, size = 0 (summed in the last phase), first column; the last column follows when the run ends
              __hip_atomic_store(q, (int)self, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              q[1] = 0;
              q[2] = cb + lane;
            }
          }
          if (lane_in(em)) {
            const int k = ne[r] + lanes_below(em);
            if constexpr (LOCAL) lds.last[idbase + r * CAPR + k] = (unsigned short)(cb + lane - 1);
            else at32(R, slot(r, k))->last = cb + lane - 1;
          }
          ne[r] += __popcll(em);
        }
#pragma unroll
        for (int r = 0; r <= G; r++) nh[r] += nhd[r];
        if (count > (LOCAL ? N : N / 2) - 64 * (G - 1)) flush();   // a chunk adds at most 64 contacts per row pair
      }
      prev_empty = !quiet && empty;
#pragma unroll
      for (int r = 0; r <= G; r++) vp[r] = vq[r];
      mvp = mv;
    }
  }
  // runs that reach the segment's last column end there (a chunk-aligned end has no boundary lane behind it)
  unsigned tail = 0u;
  if (((ce - cs) & 63) == 0) {
#pragma unroll
    for (int r = 0; r < G; r++) tail |= (__builtin_amdgcn_readlane(vp[r], 63) < kSpkFarMin ? 1u : 0u) << r;
  }
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < G; r++) {
      if (y0 + r < H && ((tail >> r) & 1u)) {
        if constexpr (LOCAL) lds.last[idbase + r * CAPR + ne[r]] = (unsigned short)(ce - 1);
        else at32(R, slot(r, ne[r]))->last = ce - 1;
      }
    }
  }
  if (count) flush();
  return false;
}

// grid: (ceil(nbands * S / NW), n), block = NW wavefronts = NW / S bands x S column segments
template <int G, int S>
__global__ void __launch_bounds__(64 * kSpkWaves) speckle_band_kernel(const int16_t* __restrict__ disp, SpkRun* __restrict__ runs,
                                                            int* __restrict__ nheads, unsigned* __restrict__ seam,
                                                            int* __restrict__ nseam, const SpkLayout<S> lay, int W, int H, int newval,
                                                            int maxdiff) {
  constexpr int CAPR = kSpkCapR<G>;
  constexpr int N = SpkWgLds<G>::N;
  constexpr int NW = kSpkWaves;
  __shared__ SpkWgLds<G> lds;
  const int lane = threadIdx.x & 63;
  const int SW = lay.SW;
  // S = 1: the four wavefronts are four bands that have nothing to do with each other -- no workgroup barriers
  auto sync = [&](bool stores) {
    if constexpr (S > 1) {
      __syncthreads();
    } else {
      if (stores) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // (this wavefront's record stores have left)
      else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  };
  const int nbands = (H + G - 1) / G;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: keeps row pointers and counters in SGPRs
  const int bl = wave / S, seg = wave - bl * S;
  const int band = blockIdx.x * (NW / S) + bl;
  const int y0 = band * G;
  const int cs = seg * SW, ce = min(W, cs + SW);
  const bool walks = band < nbands && cs < W;        // uniform (every wavefront stays for the barriers)
  const int16_t* d = disp + (size_t)blockIdx.y * W * H;
  SpkRun* const R = runs + (size_t)blockIdx.y * lay.records(H);   // this pair's records (dense block + plane)
  int* const P = &R->parent;
  unsigned* const sl = seam + (size_t)blockIdx.y * lay.seam_slots(nbands);   // this pair's seam lists
  auto slot = [&](int r, int sg, int k) -> unsigned { return spk_slot((unsigned)((y0 + r) * S + sg), (unsigned)k, (unsigned)SW, (unsigned)(H * S), (unsigned)lay.D); };

  // the pixels either side of the segment's left edge (lane = row): joined or not, decided after the walks
  bool joined = false;
  if (S > 1 && walks && seg > 0 && lane < G && y0 + lane < H) {
    const int a = d[(size_t)(y0 + lane) * W + cs - 1], b = d[(size_t)(y0 + lane) * W + cs];
    joined = a != newval && b != newval && abs(a - b) <= maxdiff;
  }

  int nh[G + 1], nsm = 0;
#pragma unroll
  for (int r = 0; r <= G; r++) nh[r] = 0;
  bool over = false;
  if (walks) over = speckle_band_walk<G, S, true>(d, R, sl, lay, W, H, newval, maxdiff, lds, wave, band, seg, cs, ce, nh, nsm);
  if (S > 1 && lane == 0) lds.ovf[wave] = over ? 1 : 0;
  sync(false);
  bool global_mode = over;   // uniform per band
  if constexpr (S > 1) {
    for (int i = 0; i < S; i++) global_mode |= lds.ovf[bl * S + i] != 0;
  }
  if (global_mode && walks) speckle_band_walk<G, S, false>(d, R, sl, lay, W, H, newval, maxdiff, lds, wave, band, seg, cs, ce, nh, nsm);
  if (S == 1 && band >= nbands) return;
  if (band < nbands && lane == 0) {
#pragma unroll
    for (int r = 0; r < G; r++) {
      if constexpr (S > 1) lds.nh[wave][r] = nh[r];
      if (y0 + r < H) nheads[((size_t)blockIdx.y * H + y0 + r) * S + seg] = nh[r];
    }
    nseam[((size_t)blockIdx.y * nbands + band) * S + seg] = nsm;
  }
  sync(global_mode);

  // ---- runs that cross the segment's left edge: this segment's first run of the row continues the left segment's last one
  if (S > 1 && joined) {
    const int kl = lds.nh[wave - 1][lane] - 1;
    if (global_mode) uf_union<__HIP_MEMORY_SCOPE_WORKGROUP, 4>(P, (int)slot(lane, seg - 1, kl), (int)slot(lane, seg, 0));
    else lds_union(lds.par, (wave - 1) * N + lane * CAPR + kl, wave * N + lane * CAPR);
  }
  if (!global_mode) {
#pragma unroll
    for (int i = 0; i < N; i += 64) lds.cs[wave * N + i + lane] = 0u;   // (the contact lists are done with: sizes from here on)
  }
  sync(global_mode);

  // ---- last phase: the band's components are final INSIDE the band. Every run learns the in-band size of its component, every
  // in-band root is flagged: the seam / count / apply kernels then settle the bulk of the runs from the run's own record.
  // Lanes = (row of the band, run of that row in this segment), 64 / G runs of every row at a time; two sweeps with the band's
  // size sums between.
  constexpr int LPR = 64 / G;                      // lanes per row
  const int q = lane / LPR, li = lane - q * LPR;
  int nhq = 0;
#pragma unroll
  for (int r = 0; r < G; r++) nhq = q == r ? nh[r] : nhq;
  int nhmax = nh[0];
#pragma unroll
  for (int r = 1; r < G; r++) nhmax = max(nhmax, nh[r]);
  int* const lsize = reinterpret_cast<int*>(lds.cs);
  if (!global_mode) {
    for (int base = 0; base < nhmax; base += LPR) {
      const int k = base + li;
      if (k < nhq) {
        const int id = wave * N + q * CAPR + k;
        const int root = lds_find(lds.par, id);
        lds.par[id] = root;      // (a node's parent only ever moves towards its root: concurrent finds stay right)
        __hip_atomic_fetch_add(&lsize[root], (int)lds.last[id] - (int)lds.first[id] + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  } else {
    for (int base = 0; base < nhmax; base += LPR) {
      const int k = base + li;
      if (k < nhq) {
        const unsigned self = slot(q, seg, k);
        SpkRun* const rec = at32(R, self);
        const int fst = __hip_atomic_load(&rec->first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const int lst = __hip_atomic_load(&rec->last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const int root = uf_find<__HIP_MEMORY_SCOPE_WORKGROUP, 4>(P, (int)self);
        if (root != (int)self) __hip_atomic_store(&rec->parent, root, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // compressed: one step to the in-band root
        __hip_atomic_fetch_add(&at32(R, (unsigned)root)->size, lst - fst + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  if constexpr (S > 1) {
    __syncthreads();   // the size sums are complete
  } else if (global_mode) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
  } else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (!global_mode) {
    for (int base = 0; base < nhmax; base += LPR) {
      const int k = base + li;
      if (k < nhq) {
        const int id = wave * N + q * CAPR + k;
        const int root = lds.par[id];
        const int rw = root / N, rem = root - rw * N, rq = rem / CAPR, rk = rem - rq * CAPR;
        // the record, complete, in one store: parent = the in-band root's slot, the component's in-band size, first (+ root flag), last
        *reinterpret_cast<int4*>(at32(R, slot(q, seg, k))) = make_int4((int)slot(rq, rw - bl * S, rk), lsize[root],
                                                                        (int)lds.first[id] | (root == id ? kSpkRootFlag : 0), (int)lds.last[id]);
      }
    }
  } else {
    for (int base = 0; base < nhmax; base += LPR) {
      const int k = base + li;
      if (k < nhq) {
        const unsigned self = slot(q, seg, k);
        SpkRun* const rec = at32(R, self);
        const int root = __hip_atomic_load(&rec->parent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (root == (int)self) {
          const int fst = __hip_atomic_load(&rec->first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          rec->first = fst | kSpkRootFlag;
        } else {
          rec->size = __hip_atomic_load(&at32(R, (unsigned)root)->size, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
  }
}

// Seam contacts of the band walk: one wavefront per band, 64 contacts at a time. Both records of a contact are loaded at once and
// say how large the two components are INSIDE their bands: both larger than maxSpeckleSize -- the bulk -- and the contact does
// not matter (neither is a speckle, merged or not); one larger: the other one's in-band root is marked "touches something
// large" (a plain store; the count kernel carries the mark to the final root if that root gets hooked); both small: union of the
// two in-band roots.
template <int S>
__global__ void __launch_bounds__(256) speckle_seam_kernel(SpkRun* __restrict__ runs, const unsigned* __restrict__ seam,
                                                            const int* __restrict__ nseam, const SpkLayout<S> lay, int H, int G, int maxsize) {
  const int lane = threadIdx.x & 63;
  const int nbands = (H + G - 1) / G;
  const int bandv = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // virtual band: band * S + segment
  if (bandv >= nbands * S) return;
  const int band = bandv / S, seg = bandv - band * S;
  const int y = band * G + G - 1;   // upper row of the seam
  if (y + 1 >= H) return;
  SpkRun* const R = runs + (size_t)blockIdx.y * lay.records(H);
  int* const P = &R->parent;
  const unsigned* sl = seam + (size_t)blockIdx.y * lay.seam_slots(nbands);
  const int n = nseam[(size_t)blockIdx.y * nbands * S + bandv];
  const unsigned yv = (unsigned)(y * S + seg), HV = (unsigned)(H * S);
  for (int i = lane; i < n; i += 64) {
    const unsigned e = sl[spk_seam_slot((unsigned)bandv, (unsigned)i, (unsigned)(nbands * S), (unsigned)lay.SW)];
    const int a = (int)spk_slot(yv, e & 0xffffu, (unsigned)lay.SW, HV, (unsigned)lay.D), b = (int)spk_slot(yv + S, e >> 16, (unsigned)lay.SW, HV, (unsigned)lay.D);
    const int4 ra = *reinterpret_cast<const int4*>(R + a), rb = *reinterpret_cast<const int4*>(R + b);   // parent, size, first, last
    // (a record's size is its component's in-band size -- its own sum at a root; marks of this kernel only make it larger)
    const bool big_a = ra.y > maxsize, big_b = rb.y > maxsize;
    if (big_a && big_b) continue;
    if (big_a != big_b) {
      const int small_root = big_a ? rb.x : ra.x;
      __hip_atomic_fetch_max(&R[small_root].size, kSpkBig, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      continue;
    }
    uf_union<__HIP_MEMORY_SCOPE_AGENT, 4>(P, ra.x, rb.x);
  }
}

// count and apply work on one load group (SPK_G chunks = 512 pixels) at a time: the walk COLLECTS the group's run heads in
// a wavefront-private LDS list, then the root / size look-ups -- dependent L2 round trips -- run 64 heads at a time
// (one latency per group instead of one per chunk that holds a head).
constexpr int SPK_HEADS = 64 * SPK_G;   // a pixel is at most one head

__device__ __forceinline__ int spk_collect_heads(RowWalk& rw, const int (&vs)[SPK_G], int cb0, int W, int lane, int newval,
                                                 int maxdiff, int rowbase, int* list) {
  int nh = 0;   // uniform
#pragma unroll
  for (int g = 0; g < SPK_G; g++) {
    const int cb = cb0 + 64 * g;
    if (cb >= W) break;
    rw.step(vs[g], cb, lane, newval, maxdiff);
    const unsigned long long hm = rw.head;
    if (hm) {
      if ((hm >> lane) & 1ull) list[nh + __popcll(hm & ((1ull << lane) - 1ull))] = rowbase + cb + lane;
      nh += __popcll(hm);
    }
    rw.next(vs[g]);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  return nh;
}

__global__ void __launch_bounds__(256) speckle_count_kernel(const int16_t* __restrict__ disp, int* __restrict__ labels,
                                                             int* __restrict__ counts, int W, int H, int newval,
                                                             int maxdiff, int maxsize) {
  __shared__ int head_lds[4][SPK_HEADS];
  SPK_ROW_SETUP
  if (y >= H) return;
  const int16_t* d = disp + plane_off + (size_t)y * W;
  int* L = labels + plane_off;
  int* C = counts + plane_off;
  int* const list = head_lds[wave];
  RowWalk rw;
  rw.init(newval);
  for (int cb0 = 0; cb0 < W; cb0 += 64 * SPK_G) {
    int vs[SPK_G];
    spk_load_group(d, cb0, W, lane, newval, vs);
    const int nh = spk_collect_heads(rw, vs, cb0, W, lane, newval, maxdiff, y * W, list);
    for (int i = lane; i < nh; i += 64) {
      const int self = list[i];
      const int r = uf_root_final(L, self);
      if (r != self) {
        // parents are final: point straight at the root (any ancestor is a valid parent for concurrent readers),
        // so the apply kernel's lookup is one step
        L[self] = r;
        if (__hip_atomic_load(C + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= maxsize) atomicAdd(C + r, C[self]);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ void __launch_bounds__(256) speckle_apply_kernel(int16_t* __restrict__ disp, const int* __restrict__ labels,
                                                             const int* __restrict__ counts, int W, int H, int newval,
                                                             int maxdiff, int maxsize) {
  __shared__ int head_lds[4][SPK_HEADS];
  SPK_ROW_SETUP
  if (y >= H) return;
  int16_t* d = disp + plane_off + (size_t)y * W;
  const int* L = labels + plane_off;
  const int* C = counts + plane_off;
  int* const list = head_lds[wave];
  RowWalk rw;
  rw.init(newval);
  bool carry_kill = false;  // decision of the run that contains the pixel left of the chunk (uniform)
  for (int cb0 = 0; cb0 < W; cb0 += 64 * SPK_G) {
    int vs[SPK_G];
    spk_load_group(d, cb0, W, lane, newval, vs);   // original values: the stores below never feed a later load
    const RowWalk at_group_start = rw;
    const int nh = spk_collect_heads(rw, vs, cb0, W, lane, newval, maxdiff, y * W, list);
    // decisions of the group's heads, 64 at a time; list[i] becomes 1 (speckle: erase) or 0
    bool any = false;
    for (int i0 = 0; i0 < nh; i0 += 64) {
      const int i = i0 + lane;
      bool kill = false;
      if (i < nh) {
        kill = C[uf_root_final(L, list[i])] <= maxsize;
        list[i] = kill ? 1 : 0;
      }
      any |= __ballot(kill) != 0ull;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (any || carry_kill) {                                   // uniform: most groups have nothing to erase
      RowWalk r2 = at_group_start;
      int idx = 0;                                             // heads of the group before the current chunk
#pragma unroll
      for (int g = 0; g < SPK_G; g++) {
        const int cb = cb0 + 64 * g;
        if (cb >= W) break;
        r2.step(vs[g], cb, lane, newval, maxdiff);
        const unsigned long long hm = r2.head;
        const int below = __popcll(hm & ((2ull << lane) - 1ull));   // heads of this chunk at or left of the lane
        const bool mine = ((r2.valid >> lane) & 1ull) && (below ? list[idx + below - 1] != 0 : carry_kill);
        if (mine) d[cb + lane] = (int16_t)newval;
        const int nhc = __popcll(hm);
        carry_kill = ((r2.valid >> 63) & 1ull) ? (nhc ? list[idx + nhc - 1] != 0 : carry_kill) : false;
        idx += nhc;
        r2.next(vs[g]);
      }
    }
    // (nothing erased in this group and nothing carried in: the run reaching the next group is not a speckle either)
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- record-driven count / apply: one wavefront per image row, one lane per run ---------------------------------------------
// count: only in-band roots that the seam kernel hooked under another root have anything to do -- their size (or their "touches
// something large" mark) goes to the final root. Everything else leaves after one load.
template <int S>
__global__ void __launch_bounds__(256) speckle_count_list_kernel(SpkRun* __restrict__ runs, const int* __restrict__ nheads,
                                                                  const SpkLayout<S> lay, int H, int maxsize) {
  const int lane = threadIdx.x & 63;
  const int HV = H * S;
  const int yv = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // virtual row: y * S + segment
  if (yv >= HV) return;
  SpkRun* const R = runs + (size_t)blockIdx.y * lay.records(H);
  int* const P = &R->parent;
  const int nh = nheads[(size_t)blockIdx.y * HV + yv];
  for (int i = lane; i < nh; i += 64) {
    const int self = (int)spk_slot((unsigned)yv, (unsigned)i, (unsigned)lay.SW, (unsigned)HV, (unsigned)lay.D);
    const int4 rec = *reinterpret_cast<const int4*>(R + self);   // parent, size, first, last
    if (rec.z >= 0 || rec.x == self) continue;                   // not an in-band root / a root nobody hooked
    const int r = uf_root_final<4>(P, rec.x);
    if (r != rec.x) P[4 * (size_t)self] = r;    // parents are final: point straight at the root so the apply kernel's lookup is one step
    // What is large already stays a mark (idempotent); small sizes add up while the root is not known to be large. The check and
    // the add are not one operation -- lanes that read the root together all add -- so the sum may overshoot by what the racers
    // bring, at most maxsize each: with maxsize <= kSpkMaxSize (launch_speckle) that cannot reach the mark, let alone wrap.
    int* const sz = P + 4 * (size_t)r + 1;
    if (rec.y > maxsize) __hip_atomic_fetch_max(sz, kSpkBig, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (__hip_atomic_load(sz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= maxsize) atomicAdd(sz, rec.y);
  }
}

// apply: a run whose component is large inside its own band stays (one load: the bulk); otherwise its in-band root decides, or --
// if that root was hooked -- the final root.
template <int S>
__global__ void __launch_bounds__(256) speckle_apply_list_kernel(int16_t* __restrict__ disp, const SpkRun* __restrict__ runs,
                                                                  const int* __restrict__ nheads, const SpkLayout<S> lay, int W, int H,
                                                                  int newval, int maxsize) {
  const int lane = threadIdx.x & 63;
  const int HV = H * S;
  const int yv = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // virtual row: y * S + segment
  if (yv >= HV) return;
  int16_t* d = disp + (size_t)blockIdx.y * W * H + (size_t)(yv / S) * W;
  const SpkRun* const R = runs + (size_t)blockIdx.y * lay.records(H);
  const int nh = nheads[(size_t)blockIdx.y * HV + yv];
  for (int i = lane; i < nh; i += 64) {
    const int self = (int)spk_slot((unsigned)yv, (unsigned)i, (unsigned)lay.SW, (unsigned)HV, (unsigned)lay.D);
    const int4 rec = *reinterpret_cast<const int4*>(R + self);
    if (rec.y > maxsize) continue;
    int4 root = rec;
    int rid = self;
    if (rec.z >= 0) {                                   // not an in-band root: one step to it
      rid = rec.x;
      root = *reinterpret_cast<const int4*>(R + rid);
      if (root.y > maxsize) continue;
    }
    if (root.x != rid) {                                // the in-band root was hooked: the final root's size counts
      const int fin = uf_root_final<4>(&R->parent, root.x);
      if (R[fin].size > maxsize) continue;
    }
    const int x0 = rec.z & 0x7fffffff;                 // a speckle: its runs are at most maxsize long
    for (int j = x0; j <= rec.w; j++) d[j] = (int16_t)newval;
  }
}

hipError_t launch_speckle(int16_t* disp, void* runs, int32_t* nheads, uint32_t* seam, int32_t* nseam, const Geom& g, int max_size,
                          int max_diff, hipStream_t s) {
  dim3 grid((g.H + 3) / 4, g.n);
  if (g.reading & kReadSpeckleX16) max_diff = (int)std::min<long>((long)max_diff * 16, 1L << 17);
  max_diff = std::min(max_diff, 1 << 17);   // int16 values: any larger range joins everything alike
  // Two implementations. Default: band walk (runs + merge in one pass, G rows per wavefront) + seam unions,
  // then count and apply driven by the rows' run records (16-bit run indices: W <= 65535; 32-bit byte offsets within an image's
  // record plane). Fallback (SBM_SPECKLE_LISTS=0 or SBM_SPECKLE_BAND=0, or outside those limits): four kernels that each walk
  // the rows, per-pixel labels and sizes carved from the same scratch.
  // (read per call, ~0.1 us each: the GPU tests flip them between calls of one process to compare every variant with the CPU restatement)
  const int lists_env = env_switch("SBM_SPECKLE_LISTS", 1);
  const int band_env = env_switch("SBM_SPECKLE_BAND", -1);
  // (maxSpeckleSize beyond kSpkMaxSize: the row-walking kernels -- see the count kernel's size sums)
  const bool lists = nheads && seam && nseam && g.W <= 65535 && ((long)g.W + kSpkRecordPad) * g.H < (1L << 27) && max_size <= kSpkMaxSize &&
                     lists_env != 0 && band_env != 0;
  if (lists) {
    SpkRun* R = static_cast<SpkRun*>(runs);
    // 4 rows per wavefront once that still leaves ~6 000 band wavefronts (6 per SIMD), else 2: the walk of a band is a serial
    // chain and the look-ahead row costs 1/G (profiles/r06_speckle.md; 8 rows never won). SBM_SPECKLE_BAND=2/4 forces a height.
    int G = (long)g.n * g.H >= 24000 ? 4 : 2;
    if (band_env == 2 || band_env == 4) G = band_env;
    const int nbands = (g.H + G - 1) / G;
    // Column segments per band (wavefronts of one workgroup): the walk of a band is a serial chain, so launches that leave the
    // chip room are cut finer. SBM_SPECKLE_SEG=1/2/4 forces a count.
    const int nchunks = (g.W + 63) / 64;
    const long pairs_of_rows = (long)g.n * ((g.H + 1) / 2);
    int S = pairs_of_rows < kSpkSeg4 ? 4 : (pairs_of_rows < kSpkSeg2 ? 2 : 1);
    const int seg_env = env_switch("SBM_SPECKLE_SEG", 0);
    if (seg_env == 1 || seg_env == 2 || seg_env == 4) S = seg_env;
    while (S > 1 && nchunks < S) S >>= 1;
    dim3 bgrid((nbands * S + 3) / 4, g.n), vgrid((g.H * S + 3) / 4, g.n);
    const int SW = 64 * ((nchunks + S - 1) / S), newval = g.filtered;
    auto launch = [&](auto seg) {
      constexpr int SS = decltype(seg)::value;
      const SpkLayout<SS> lay{SW};
      const dim3 wgrid((nbands * SS + kSpkWaves - 1) / kSpkWaves, g.n), wblock(64 * kSpkWaves);
      if (G == 4) hipLaunchKernelGGL((speckle_band_kernel<4, SS>), wgrid, wblock, 0, s, disp, R, nheads, seam, nseam, lay, g.W, g.H, newval, max_diff);
      else hipLaunchKernelGGL((speckle_band_kernel<2, SS>), wgrid, wblock, 0, s, disp, R, nheads, seam, nseam, lay, g.W, g.H, newval, max_diff);
      hipLaunchKernelGGL(speckle_seam_kernel<SS>, bgrid, dim3(256), 0, s, R, seam, nseam, lay, g.H, G, max_size);
      hipLaunchKernelGGL(speckle_count_list_kernel<SS>, vgrid, dim3(256), 0, s, R, nheads, lay, g.H, max_size);
      hipLaunchKernelGGL(speckle_apply_list_kernel<SS>, vgrid, dim3(256), 0, s, disp, R, nheads, lay, g.W, g.H, newval, max_size);
    };
    if (S == 4) launch(std::integral_constant<int, 4>{});
    else if (S == 2) launch(std::integral_constant<int, 2>{});
    else launch(std::integral_constant<int, 1>{});
  } else {
    int* labels = static_cast<int*>(runs);
    int* counts = labels + (size_t)g.n * g.W * g.H;
    hipLaunchKernelGGL(speckle_runs_kernel, grid, dim3(256), 0, s, disp, labels, counts, g.W, g.H, g.filtered, max_diff);
    hipLaunchKernelGGL(speckle_merge_kernel, grid, dim3(256), 0, s, disp, labels, g.W, g.H, g.filtered, max_diff);
    hipLaunchKernelGGL(speckle_count_kernel, grid, dim3(256), 0, s, disp, labels, counts, g.W, g.H, g.filtered, max_diff,
                       max_size);
    hipLaunchKernelGGL(speckle_apply_kernel, grid, dim3(256), 0, s, disp, labels, counts, g.W, g.H, g.filtered, max_diff,
                       max_size);
  }
  return hipGetLastError();
}

}  // namespace sbm
