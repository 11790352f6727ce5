// mqsad_alias.hip -- gfx950: does v_mqsad_pk_u16_u8 give the right answer when its destination IS its accumulator
// (vdst == src2)?  LLVM marks vdst early-clobber against every source, which forces the SAD kernel to ping-pong two
// accumulator arrays (2 x 32 VGPRs).  This probe compares the aliased form (inline asm) with the builtin (non-aliased)
// on random operands, single instructions and dependent chains, every lane, many waves.  Round 3; not product code.
// Build: hipcc --offload-arch=gfx950 -O3 -o mqsad_alias mqsad_alias.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef unsigned long long u64;

__device__ __forceinline__ u64 rnd(u64& st) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; }

__global__ void __launch_bounds__(256) alias_kernel(unsigned* bad, int iters, u64 seed) {
  u64 st = seed + (u64)(blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull + 1;
  unsigned nbad = 0;
  for (int it = 0; it < iters; it++) {
    // 8 independent accumulators, a chain of 4 rows each (like one phase of the SAD kernel: same pattern, 8 windows)
    u64 accA[8], accR[8], win[4][8];
    unsigned pat[4];
    for (int q = 0; q < 8; q++) { accA[q] = accR[q] = rnd(st) & 0x0fff0fff0fff0fffull; }
    for (int r = 0; r < 4; r++) { pat[r] = (unsigned)rnd(st) & ((it & 1) ? 0x00ffffffu : 0xffffffffu); for (int q = 0; q < 8; q++) win[r][q] = rnd(st); }
    // reference: builtin, never aliased (the compiler ping-pongs)
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int q = 0; q < 8; q++) accR[q] = __builtin_amdgcn_mqsad_pk_u16_u8(win[r][q], pat[r], accR[q]);
    // aliased: vdst == src2
#pragma unroll
    for (int r = 0; r < 4; r++) {
      asm volatile(
          "v_mqsad_pk_u16_u8 %0, %8, %16, %0\n v_mqsad_pk_u16_u8 %1, %9, %16, %1\n v_mqsad_pk_u16_u8 %2, %10, %16, %2\n"
          "v_mqsad_pk_u16_u8 %3, %11, %16, %3\n v_mqsad_pk_u16_u8 %4, %12, %16, %4\n v_mqsad_pk_u16_u8 %5, %13, %16, %5\n"
          "v_mqsad_pk_u16_u8 %6, %14, %16, %6\n v_mqsad_pk_u16_u8 %7, %15, %16, %7\n"
          : "+v"(accA[0]), "+v"(accA[1]), "+v"(accA[2]), "+v"(accA[3]), "+v"(accA[4]), "+v"(accA[5]), "+v"(accA[6]), "+v"(accA[7])
          : "v"(win[r][0]), "v"(win[r][1]), "v"(win[r][2]), "v"(win[r][3]), "v"(win[r][4]), "v"(win[r][5]), "v"(win[r][6]),
            "v"(win[r][7]), "v"(pat[r]));
    }
    for (int q = 0; q < 8; q++) nbad += accA[q] != accR[q];
    // back-to-back dependent chain on ONE accumulator (worst case for any forwarding hazard)
    u64 cA = accR[0], cR = accR[0];
#pragma unroll
    for (int r = 0; r < 4; r++) cR = __builtin_amdgcn_mqsad_pk_u16_u8(win[r][1], pat[r], cR);
    asm volatile("v_mqsad_pk_u16_u8 %0, %1, %5, %0\n v_mqsad_pk_u16_u8 %0, %2, %6, %0\n v_mqsad_pk_u16_u8 %0, %3, %7, %0\n v_mqsad_pk_u16_u8 %0, %4, %8, %0\n"
                 : "+v"(cA) : "v"(win[0][1]), "v"(win[1][1]), "v"(win[2][1]), "v"(win[3][1]), "v"(pat[0]), "v"(pat[1]), "v"(pat[2]), "v"(pat[3]));
    nbad += cA != cR;
  }
  if (nbad) atomicAdd(bad, nbad);
}

int main() {
  unsigned* dbad; CK(hipMalloc(&dbad, 4)); CK(hipMemset(dbad, 0, 4));
  const int blocks = 256 * 8, iters = 2000;
  alias_kernel<<<blocks, 256>>>(dbad, iters, 12345); CK(hipDeviceSynchronize());
  unsigned bad = 0; CK(hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost));
  const double n = (double)blocks * 256 * iters * 9;
  printf("v_mqsad_pk_u16_u8 with vdst == src2: %u mismatching results of %.3g compared (%.3g instructions, 8 wavefronts/SIMD resident)\n", bad, n,
         (double)blocks * 256 * iters * 36 / 64);
  return bad != 0;
}
