// lds_dma_rate.hip -- throughput of LDS-direct loads on gfx950 (round 4): CU-cycles per wavefront-instruction for the 16-byte
// and the 4-byte form, with the interior SAD kernel's address pattern (lane i at byte offset i: overlapping, byte-misaligned),
// a lane-stride-3 variant, and non-overlapping 16-byte lanes for comparison; data from L2 (a few KB per workgroup, re-read).
// build: hipcc --offload-arch=gfx950 -O2 -o lds_dma_rate lds_dma_rate.hip ; run: ./lds_dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
extern __shared__ unsigned char lds[];

template <int SIZE>
__global__ void __launch_bounds__(256) rate(const unsigned char* src, int stride, int iters, unsigned* sink) {
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned char* area = lds + wv * 2 * 1024;              // two 1 KB areas per wavefront, alternating
  const unsigned char* p = src + (blockIdx.x & 63) * 4096 + wv * 512 + lane * stride;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if constexpr (SIZE == 16) __builtin_amdgcn_global_load_lds((gptr_t)(p + 64 * u), (lptr_t)(area + (u & 1) * 1024), 16, 0, 0);
      else __builtin_amdgcn_global_load_lds((gptr_t)(p + 64 * u), (lptr_t)(area + (u & 1) * 1024), 4, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);
  }
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = lds[0];
}

int main() {
  unsigned char* d; unsigned* s;
  hipMalloc(&d, 64 * 4096 + 8192); hipMemset(d, 7, 64 * 4096 + 8192); hipMalloc(&s, 65536 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  for (int size : {16, 4}) for (int stride : {1, 3, 16}) for (int wg : {1, 2, 4}) {    // workgroups (of 4 wavefronts) per CU
    const int blocks = 256 * wg;
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      if (size == 16) hipLaunchKernelGGL(rate<16>, dim3(blocks), dim3(256), 8192, 0, d, stride, iters, s);
      else hipLaunchKernelGGL(rate<4>, dim3(blocks), dim3(256), 8192, 0, d, stride, iters, s);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) {
        const double instr_per_cu = (double)wg * 4 * iters * 8;
        printf("size %2d lane stride %2d, %2d wavefronts per CU: %.3f ms, %.1f CU-cycles per wavefront-instruction at 2.4 GHz, %.1f LDS bytes per CU-cycle\n",
               size, stride, 4 * wg, ms, ms * 1e-3 * 2.4e9 / instr_per_cu, 64.0 * size * instr_per_cu / (ms * 1e-3 * 2.4e9));
      }
    }
  }
  return 0;
}
