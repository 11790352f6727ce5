// lds_write_width.hip -- gfx950: cost of writing 16 bytes per lane to LDS as one ds_write_b128, two ds_write_b64 (and
// ds_write2_b64), four ds_write_b32, at lane strides that keep the wavefront's bytes contiguous. Round 3: the SAD kernel's LDS
// pipe is ~90 % busy and 43 % of that are b128 writes at 13.8 cycles each against 8 for the bytes they move.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_write_width lds_write_width.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// MODE 0: b128 [lane]; 1: two b64, layout [half][lane] (each instruction contiguous 512 B); 2: two b64 at [lane][half] (16 B
// stride); 3: four b32 [word][lane]; 4: ds_write2_b64 (one instruction, two 8-byte slots [half][lane]); 5: ds_write2st64_b64
template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[4 * 2048];   // 8 KB per wavefront
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t* base = lds + wv * 2048;
  uint32_t a = lane, b = lane * 3, c = lane * 5, d = lane * 7;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int m = 0; m < 4; m++) {
      uint32_t* p = base + m * 256 * (MODE == 0 || MODE == 2 ? 1 : 1);
      const uint32_t ad16 = (uint32_t)(uintptr_t)(p + 4 * lane), ad8 = (uint32_t)(uintptr_t)(p + 2 * lane), ad4 = (uint32_t)(uintptr_t)(p + lane);
      const unsigned long long lo = ((unsigned long long)b << 32) | a, hi = ((unsigned long long)d << 32) | c;
      if (MODE == 0) {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = {a, b, c, d};
        asm volatile("ds_write_b128 %0, %1" :: "v"(ad16), "v"(v) : "memory");
      } else if (MODE == 1) {
        asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:512" :: "v"(ad8), "v"(lo), "v"(hi) : "memory");
      } else if (MODE == 2) {
        asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:8" :: "v"(ad16), "v"(lo), "v"(hi) : "memory");
      } else if (MODE == 3) {
        asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:256\n\tds_write_b32 %0, %3 offset:512\n\tds_write_b32 %0, %4 offset:768" :: "v"(ad4), "v"(a), "v"(b), "v"(c), "v"(d) : "memory");
      } else if (MODE == 4) {
        asm volatile("ds_write2_b64 %0, %1, %2 offset1:64" :: "v"(ad8), "v"(lo), "v"(hi) : "memory");
      } else {
        asm volatile("ds_write2st64_b64 %0, %1, %2 offset1:1" :: "v"(ad8), "v"(lo), "v"(hi) : "memory");
      }
      a += it; c ^= b;
    }
  }
  __syncthreads();
  if (base[lane] == 0x12345678u) out[0] = a + b + c + d;
}

template <int MODE>
static void run(const char* name, uint32_t* dout, int wps) {
  const int iters = 4000, blocks = 256 * wps;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<MODE><<<blocks, 256>>>(dout, 10); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    CK(hipEventRecord(e0));
    k<MODE><<<blocks, 256>>>(dout, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  const double n = (double)iters * 4 * wps * 4;   // 16-byte-per-lane stores per CU
  printf("%-58s wps=%d  %8.3f ms  %6.2f CU-cycles per 16 B/lane store @2.4GHz\n", name, wps, best, best * 1e6 * 2.4 / n);
}

int main() {
  uint32_t* dout; CK(hipMalloc(&dout, 4096));
  for (int wps = 1; wps <= 2; wps++) {
    run<0>("ds_write_b128 [lane]", dout, wps);
    run<1>("2 x ds_write_b64 [half][lane]", dout, wps);
    run<2>("2 x ds_write_b64 [lane][half]", dout, wps);
    run<3>("4 x ds_write_b32 [word][lane]", dout, wps);
    run<4>("ds_write2_b64 offset1:64 ([half][lane])", dout, wps);
    run<5>("ds_write2st64_b64 offset1:1 ([half][lane])", dout, wps);
  }
  return 0;
}
