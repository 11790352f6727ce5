// isa_probe.hip -- gfx950 instruction-rate and semantics probe for the SAD kernel design.
// Not part of the product path; results are recorded in DESIGN.md.
// Build: hipcc --offload-arch=gfx950 -O3 -o isa_probe isa_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef unsigned long long u64;

// ---- semantics ---------------------------------------------------------------------------------
__global__ void sem_kernel(const u64* s0, const uint32_t* s1, const u64* s2, u64* out_qsad, u64* out_mqsad,
                           uint32_t* out_sad, uint32_t* out_msad, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64 a = s0[i]; uint32_t b = s1[i]; u64 c = s2[i];
  out_qsad[i] = __builtin_amdgcn_qsad_pk_u16_u8(a, b, c);
  out_mqsad[i] = __builtin_amdgcn_mqsad_pk_u16_u8(a, b, c);
  out_sad[i] = __builtin_amdgcn_sad_u8((uint32_t)a, b, (uint32_t)c);
  out_msad[i] = __builtin_amdgcn_msad_u8((uint32_t)a, b, (uint32_t)c);
}

static inline uint32_t absd(uint32_t a, uint32_t b) { return a > b ? a - b : b - a; }
static uint32_t ref_sad(uint32_t a, uint32_t b, uint32_t acc, bool masked) {
  for (int i = 0; i < 4; i++) {
    uint32_t x = (a >> (8 * i)) & 255, y = (b >> (8 * i)) & 255;
    if (masked && y == 0) continue;
    acc += absd(x, y);
  }
  return acc;
}
static u64 ref_qsad(u64 a, uint32_t b, u64 c, bool masked) {
  u64 r = 0;
  for (int k = 0; k < 4; k++) {
    uint32_t win = (uint32_t)(a >> (8 * k));
    uint32_t acc = (uint32_t)((c >> (16 * k)) & 0xFFFF);
    uint32_t v = ref_sad(win, b, acc, masked) & 0xFFFF;
    r |= (u64)v << (16 * k);
  }
  return r;
}

// ---- throughput --------------------------------------------------------------------------------
#define REP8(x) x x x x x x x x
template <int OP>
__global__ void __launch_bounds__(256) rate_kernel(uint32_t* out, int iters, uint32_t seed) {
  uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
  uint32_t a0 = t * 3 + seed, a1 = t * 5 + 1, a2 = t * 7 + 2, a3 = t * 11 + 3, a4 = t * 13, a5 = t * 17, a6 = t * 19, a7 = t * 23;
  uint32_t b0 = t ^ 0x12345678u, b1 = t ^ 0x9abcdef0u, p = t * 2654435761u;
  u64 q0 = ((u64)a0 << 32) | a1, q1 = ((u64)a2 << 32) | a3, q2 = ((u64)a4 << 32) | a5, q3 = ((u64)a6 << 32) | a7;
  u64 q4 = q0 ^ 1, q5 = q1 ^ 2, q6 = q2 ^ 3, q7 = q3 ^ 4;
  u64 w = ((u64)b0 << 32) | b1;
  for (int it = 0; it < iters; ++it) {
    if constexpr (OP == 0) {  // v_add_u32 baseline
      REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                        "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(p));)
    } else if constexpr (OP == 1) {  // v_qsad_pk_u16_u8
      REP8(asm volatile("v_qsad_pk_u16_u8 %0, %8, %9, %0\n v_qsad_pk_u16_u8 %1, %8, %9, %1\n v_qsad_pk_u16_u8 %2, %8, %9, %2\n v_qsad_pk_u16_u8 %3, %8, %9, %3\n"
                        "v_qsad_pk_u16_u8 %4, %8, %9, %4\n v_qsad_pk_u16_u8 %5, %8, %9, %5\n v_qsad_pk_u16_u8 %6, %8, %9, %6\n v_qsad_pk_u16_u8 %7, %8, %9, %7\n"
                        : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "v"(w), "v"(p));)
    } else if constexpr (OP == 2) {  // v_mqsad_pk_u16_u8
      REP8(asm volatile("v_mqsad_pk_u16_u8 %0, %8, %9, %0\n v_mqsad_pk_u16_u8 %1, %8, %9, %1\n v_mqsad_pk_u16_u8 %2, %8, %9, %2\n v_mqsad_pk_u16_u8 %3, %8, %9, %3\n"
                        "v_mqsad_pk_u16_u8 %4, %8, %9, %4\n v_mqsad_pk_u16_u8 %5, %8, %9, %5\n v_mqsad_pk_u16_u8 %6, %8, %9, %6\n v_mqsad_pk_u16_u8 %7, %8, %9, %7\n"
                        : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "v"(w), "v"(p));)
    } else if constexpr (OP == 3) {  // v_sad_u8
      REP8(asm volatile("v_sad_u8 %0, %8, %9, %0\n v_sad_u8 %1, %8, %9, %1\n v_sad_u8 %2, %8, %9, %2\n v_sad_u8 %3, %8, %9, %3\n"
                        "v_sad_u8 %4, %8, %9, %4\n v_sad_u8 %5, %8, %9, %5\n v_sad_u8 %6, %8, %9, %6\n v_sad_u8 %7, %8, %9, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(p));)
    } else if constexpr (OP == 4) {  // v_pk_add_u16
      REP8(asm volatile("v_pk_add_u16 %0, %0, %8\n v_pk_add_u16 %1, %1, %8\n v_pk_add_u16 %2, %2, %8\n v_pk_add_u16 %3, %3, %8\n"
                        "v_pk_add_u16 %4, %4, %8\n v_pk_add_u16 %5, %5, %8\n v_pk_add_u16 %6, %6, %8\n v_pk_add_u16 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(p));)
    } else if constexpr (OP == 5) {  // v_pk_sub_u16 clamp
      REP8(asm volatile("v_pk_sub_u16 %0, %8, %0 clamp\n v_pk_sub_u16 %1, %8, %1 clamp\n v_pk_sub_u16 %2, %8, %2 clamp\n v_pk_sub_u16 %3, %8, %3 clamp\n"
                        "v_pk_sub_u16 %4, %8, %4 clamp\n v_pk_sub_u16 %5, %8, %5 clamp\n v_pk_sub_u16 %6, %8, %6 clamp\n v_pk_sub_u16 %7, %8, %7 clamp\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(p));)
    } else if constexpr (OP == 6) {  // v_pk_min_u16
      REP8(asm volatile("v_pk_min_u16 %0, %0, %8\n v_pk_min_u16 %1, %1, %8\n v_pk_min_u16 %2, %2, %8\n v_pk_min_u16 %3, %3, %8\n"
                        "v_pk_min_u16 %4, %4, %8\n v_pk_min_u16 %5, %5, %8\n v_pk_min_u16 %6, %6, %8\n v_pk_min_u16 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(p));)
    } else if constexpr (OP == 7) {  // v_min3_u32
      REP8(asm volatile("v_min3_u32 %0, %0, %8, %9\n v_min3_u32 %1, %1, %8, %9\n v_min3_u32 %2, %2, %8, %9\n v_min3_u32 %3, %3, %8, %9\n"
                        "v_min3_u32 %4, %4, %8, %9\n v_min3_u32 %5, %5, %8, %9\n v_min3_u32 %6, %6, %8, %9\n v_min3_u32 %7, %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(p), "v"(b0));)
    } else if constexpr (OP == 8) {  // v_perm_b32
      REP8(asm volatile("v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n"
                        "v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(p), "v"(b0));)
    } else if constexpr (OP == 9) {  // v_and_or_b32
      REP8(asm volatile("v_and_or_b32 %0, %0, %8, %9\n v_and_or_b32 %1, %1, %8, %9\n v_and_or_b32 %2, %2, %8, %9\n v_and_or_b32 %3, %3, %8, %9\n"
                        "v_and_or_b32 %4, %4, %8, %9\n v_and_or_b32 %5, %5, %8, %9\n v_and_or_b32 %6, %6, %8, %9\n v_and_or_b32 %7, %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(p), "v"(b0));)
    } else if constexpr (OP == 10) {  // v_add_u32 dpp row_shr:1
      REP8(asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                        "v_add_u32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                        "v_add_u32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                        "v_add_u32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if constexpr (OP == 11) {  // v_cndmask_b32 (vcc)
      REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                        "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(p) : "vcc");)
    } else if constexpr (OP == 12) {  // v_alignbyte_b32
      REP8(asm volatile("v_alignbyte_b32 %0, %0, %8, 1\n v_alignbyte_b32 %1, %1, %8, 1\n v_alignbyte_b32 %2, %2, %8, 1\n v_alignbyte_b32 %3, %3, %8, 1\n"
                        "v_alignbyte_b32 %4, %4, %8, 1\n v_alignbyte_b32 %5, %5, %8, 1\n v_alignbyte_b32 %6, %6, %8, 1\n v_alignbyte_b32 %7, %7, %8, 1\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(p));)
    } else if constexpr (OP == 13) {  // v_pk_max_i16
      REP8(asm volatile("v_pk_max_i16 %0, %0, %8\n v_pk_max_i16 %1, %1, %8\n v_pk_max_i16 %2, %2, %8\n v_pk_max_i16 %3, %3, %8\n"
                        "v_pk_max_i16 %4, %4, %8\n v_pk_max_i16 %5, %5, %8\n v_pk_max_i16 %6, %6, %8\n v_pk_max_i16 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(p));)
    } else if constexpr (OP == 14) {  // v_pk_mad_u16
      REP8(asm volatile("v_pk_mad_u16 %0, %0, %8, %9\n v_pk_mad_u16 %1, %1, %8, %9\n v_pk_mad_u16 %2, %2, %8, %9\n v_pk_mad_u16 %3, %3, %8, %9\n"
                        "v_pk_mad_u16 %4, %4, %8, %9\n v_pk_mad_u16 %5, %5, %8, %9\n v_pk_mad_u16 %6, %6, %8, %9\n v_pk_mad_u16 %7, %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(p), "v"(b0));)
    } else if constexpr (OP == 15) {  // v_msad_u8
      REP8(asm volatile("v_msad_u8 %0, %8, %9, %0\n v_msad_u8 %1, %8, %9, %1\n v_msad_u8 %2, %8, %9, %2\n v_msad_u8 %3, %8, %9, %3\n"
                        "v_msad_u8 %4, %8, %9, %4\n v_msad_u8 %5, %8, %9, %5\n v_msad_u8 %6, %8, %9, %6\n v_msad_u8 %7, %8, %9, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(p));)
    } else if constexpr (OP == 16) {  // v_min_u32 with SDWA word select
      REP8(asm volatile("v_min_u32_sdwa %0, %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_min_u32_sdwa %1, %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                        "v_min_u32_sdwa %2, %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_min_u32_sdwa %3, %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                        "v_min_u32_sdwa %4, %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_min_u32_sdwa %5, %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                        "v_min_u32_sdwa %6, %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_min_u32_sdwa %7, %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(p));)
    } else if constexpr (OP == 17) {  // v_lshl_or_b32
      REP8(asm volatile("v_lshl_or_b32 %0, %0, 8, %8\n v_lshl_or_b32 %1, %1, 8, %8\n v_lshl_or_b32 %2, %2, 8, %8\n v_lshl_or_b32 %3, %3, 8, %8\n"
                        "v_lshl_or_b32 %4, %4, 8, %8\n v_lshl_or_b32 %5, %5, 8, %8\n v_lshl_or_b32 %6, %6, 8, %8\n v_lshl_or_b32 %7, %7, 8, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(p));)
    } else if constexpr (OP == 19) {  // ds_bpermute_b32 (LDS crossbar)
      REP8(a0 = __builtin_amdgcn_ds_bpermute((int)p, (int)a0); a1 = __builtin_amdgcn_ds_bpermute((int)p, (int)a1);
           a2 = __builtin_amdgcn_ds_bpermute((int)p, (int)a2); a3 = __builtin_amdgcn_ds_bpermute((int)p, (int)a3);
           a4 = __builtin_amdgcn_ds_bpermute((int)p, (int)a4); a5 = __builtin_amdgcn_ds_bpermute((int)p, (int)a5);
           a6 = __builtin_amdgcn_ds_bpermute((int)p, (int)a6); a7 = __builtin_amdgcn_ds_bpermute((int)p, (int)a7);)
    } else if constexpr (OP == 20) {  // bpermute + dependent v_add (the hsum pattern), 8 chains
      REP8(a0 += __builtin_amdgcn_ds_bpermute((int)p, (int)a0); a1 += __builtin_amdgcn_ds_bpermute((int)p, (int)a1);
           a2 += __builtin_amdgcn_ds_bpermute((int)p, (int)a2); a3 += __builtin_amdgcn_ds_bpermute((int)p, (int)a3);
           a4 += __builtin_amdgcn_ds_bpermute((int)p, (int)a4); a5 += __builtin_amdgcn_ds_bpermute((int)p, (int)a5);
           a6 += __builtin_amdgcn_ds_bpermute((int)p, (int)a6); a7 += __builtin_amdgcn_ds_bpermute((int)p, (int)a7);)
    } else if constexpr (OP == 21) {  // v_mov_b32 dpp wave_shl:1
      REP8(asm volatile("v_mov_b32_dpp %0, %0 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 wave_shl:1 row_mask:0xf bank_mask:0xf\n"
                        "v_mov_b32_dpp %2, %2 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 wave_shl:1 row_mask:0xf bank_mask:0xf\n"
                        "v_mov_b32_dpp %4, %4 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 wave_shl:1 row_mask:0xf bank_mask:0xf\n"
                        "v_mov_b32_dpp %6, %6 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 wave_shl:1 row_mask:0xf bank_mask:0xf\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if constexpr (OP == 22) {  // v_dot2_u32_u16
      REP8(asm volatile("v_dot2_u32_u16 %0, %8, %9, %0\n v_dot2_u32_u16 %1, %8, %9, %1\n v_dot2_u32_u16 %2, %8, %9, %2\n v_dot2_u32_u16 %3, %8, %9, %3\n"
                        "v_dot2_u32_u16 %4, %8, %9, %4\n v_dot2_u32_u16 %5, %8, %9, %5\n v_dot2_u32_u16 %6, %8, %9, %6\n v_dot2_u32_u16 %7, %8, %9, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(p), "v"(b0));)
    } else if constexpr (OP == 18) {  // v_mqsad_u32_u8 (128-bit acc)
      // needs 4-register tuples; use 2 chains
      typedef uint32_t v4u __attribute__((ext_vector_type(4)));
      v4u r0 = {a0, a1, a2, a3}, r1 = {a4, a5, a6, a7};
      REP8(asm volatile("v_mqsad_u32_u8 %0, %2, %3, %0\n v_mqsad_u32_u8 %1, %2, %3, %1\n v_mqsad_u32_u8 %0, %2, %3, %0\n v_mqsad_u32_u8 %1, %2, %3, %1\n"
                        "v_mqsad_u32_u8 %0, %2, %3, %0\n v_mqsad_u32_u8 %1, %2, %3, %1\n v_mqsad_u32_u8 %0, %2, %3, %0\n v_mqsad_u32_u8 %1, %2, %3, %1\n"
                        : "+v"(r0), "+v"(r1) : "v"(w), "v"(p));)
      a0 = r0.x ^ r0.y ^ r0.z ^ r0.w; a4 = r1.x ^ r1.y ^ r1.z ^ r1.w;
    }
  }
  uint32_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7) ^ (uint32_t)((q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7) >> 32);
  if (r == 0x12345) out[t] = r;
}

template <int OP>
static void run_rate(const char* name, uint32_t* dout, int waves_per_simd) {
  int iters = 2000;
  int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  rate_kernel<OP><<<blocks, 256>>>(dout, 10, 1); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    CK(hipEventRecord(e0));
    rate_kernel<OP><<<blocks, 256>>>(dout, iters, 1);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  double ninstr_per_wave = (double)iters * 64.0;
  double total_wave_instr = ninstr_per_wave * blocks * 4;
  double per_simd = total_wave_instr / (256.0 * 4.0);  // wave-instr per SIMD
  double ns_per = best * 1e6 / per_simd;
  printf("%-22s waves/SIMD=%d  %8.3f ms  %.3f ns/wave-instr/SIMD (= %.2f cyc @2.4GHz)  %.2f T lane-ops/s\n", name, waves_per_simd, best, ns_per,
         ns_per * 2.4, total_wave_instr * 64 / (best * 1e-3) / 1e12);
}

int main() {
  // semantics
  const int n = 4096;
  std::vector<u64> s0(n), s2(n), oq(n), om(n); std::vector<uint32_t> s1(n), os(n), oms(n);
  uint64_t st = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
  for (int i = 0; i < n; i++) {
    s0[i] = rnd(); s1[i] = (uint32_t)rnd(); s2[i] = rnd();
    if (i % 3 == 0) s1[i] &= 0x00FFFFFFu;       // zero top byte of the pattern
    if (i % 5 == 0) s1[i] &= 0xFFFF00FFu;       // zero byte 1
    if (i % 7 == 0) s0[i] &= 0xFFFFFFFFFF00FFFFull;  // zero a window byte
    if (i % 2 == 0) s2[i] &= 0x0FFF0FFF0FFF0FFFull;  // small accumulators
    if (i % 11 == 0) s2[i] = 0xFFF0FFF0FFF0FFF0ull;  // near-overflow accumulators
  }
  u64 *d0, *d2, *dq, *dm; uint32_t *d1, *ds, *dms;
  CK(hipMalloc(&d0, n * 8)); CK(hipMalloc(&d2, n * 8)); CK(hipMalloc(&dq, n * 8)); CK(hipMalloc(&dm, n * 8));
  CK(hipMalloc(&d1, n * 4)); CK(hipMalloc(&ds, n * 4)); CK(hipMalloc(&dms, n * 4));
  CK(hipMemcpy(d0, s0.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d1, s1.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d2, s2.data(), n * 8, hipMemcpyHostToDevice));
  sem_kernel<<<n / 256, 256>>>(d0, d1, d2, dq, dm, ds, dms, n); CK(hipDeviceSynchronize());
  CK(hipMemcpy(oq.data(), dq, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(om.data(), dm, n * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(os.data(), ds, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(oms.data(), dms, n * 4, hipMemcpyDeviceToHost));
  int bad_q = 0, bad_m = 0, bad_s = 0, bad_ms = 0, bad_m_alt = 0;
  for (int i = 0; i < n; i++) {
    if (oq[i] != ref_qsad(s0[i], s1[i], s2[i], false)) { if (bad_q < 3) printf("qsad mismatch i=%d s0=%016llx s1=%08x s2=%016llx got=%016llx exp=%016llx\n", i, s0[i], s1[i], s2[i], oq[i], ref_qsad(s0[i], s1[i], s2[i], false)); bad_q++; }
    if (om[i] != ref_qsad(s0[i], s1[i], s2[i], true)) { if (bad_m < 3) printf("mqsad mismatch i=%d s0=%016llx s1=%08x s2=%016llx got=%016llx exp=%016llx\n", i, s0[i], s1[i], s2[i], om[i], ref_qsad(s0[i], s1[i], s2[i], true)); bad_m++; }
    if (os[i] != ref_sad((uint32_t)s0[i], s1[i], (uint32_t)s2[i], false)) bad_s++;
    if (oms[i] != ref_sad((uint32_t)s0[i], s1[i], (uint32_t)s2[i], true)) bad_ms++;
  }
  printf("SEMANTICS: qsad mismatches %d/%d (model: window k = bytes k..k+3 of S0, pattern S1, wrap u16)\n", bad_q, n);
  printf("SEMANTICS: mqsad mismatches %d/%d (model: pattern byte==0 masks)\n", bad_m, n);
  printf("SEMANTICS: sad_u8 mismatches %d/%d, msad_u8 mismatches %d/%d\n", bad_s, n, bad_ms, n);
  (void)bad_m_alt;

  uint32_t* dout; CK(hipMalloc(&dout, 256 * 8 * 256 * 4 * 2));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s CUs=%d clock=%d kHz LDS/block=%zu\n", prop.name, prop.multiProcessorCount, prop.clockRate, prop.sharedMemPerBlock);
  for (int wps : {1, 4}) {
    run_rate<0>("v_add_u32", dout, wps);
    run_rate<1>("v_qsad_pk_u16_u8", dout, wps);
    run_rate<2>("v_mqsad_pk_u16_u8", dout, wps);
    run_rate<3>("v_sad_u8", dout, wps);
    run_rate<15>("v_msad_u8", dout, wps);
    run_rate<18>("v_mqsad_u32_u8", dout, wps);
    run_rate<4>("v_pk_add_u16", dout, wps);
    run_rate<5>("v_pk_sub_u16 clamp", dout, wps);
    run_rate<6>("v_pk_min_u16", dout, wps);
    run_rate<13>("v_pk_max_i16", dout, wps);
    run_rate<14>("v_pk_mad_u16", dout, wps);
    run_rate<7>("v_min3_u32", dout, wps);
    run_rate<8>("v_perm_b32", dout, wps);
    run_rate<9>("v_and_or_b32", dout, wps);
    run_rate<17>("v_lshl_or_b32", dout, wps);
    run_rate<10>("v_add_u32_dpp row_shr", dout, wps);
    run_rate<11>("v_cndmask_b32", dout, wps);
    run_rate<12>("v_alignbyte_b32", dout, wps);
    run_rate<16>("v_min_u32_sdwa", dout, wps);
    run_rate<19>("ds_bpermute_b32", dout, wps);
    run_rate<20>("ds_bpermute+v_add", dout, wps);
    run_rate<21>("v_mov_dpp wave_shl:1", dout, wps);
    run_rate<22>("v_dot2_u32_u16", dout, wps);
  }
  return 0;
}
