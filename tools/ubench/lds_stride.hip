// lds_stride.hip -- gfx950: cost of aligned ds_read_b128 / ds_write_b128 as a function of the lane stride (in 16-byte
// slots). Round 3 probe for the column-stride-3 strips of the SAD kernel (window reads at a lane stride of 48 bytes).
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_stride lds_stride.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <int STRIDE, bool WRITE>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters) {
  __shared__ __attribute__((aligned(16))) uint4 lds[4 * 640];   // 10 KB per wavefront
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint4* base = lds + wv * 640;
  for (int i = lane; i < 640; i += 64) base[i] = make_uint4(i, i * 3, i * 5, i * 7);
  __syncthreads();
  uint32_t acc = 0;
  uint4 w = make_uint4(lane, 1, 2, 3);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int m = 0; m < 8; m++) {
      if (WRITE) {
        base[(STRIDE * lane + 16 * m) % 640] = w;
        w.x += acc;
      } else {
        const uint4 v = base[STRIDE * lane + 16 * m + (m & 1) * 4];
        acc += v.x ^ v.y ^ v.z ^ v.w;
      }
    }
    asm volatile("" : "+v"(acc));
  }
  if (acc == 0x12345) out[0] = acc + base[lane].x;
}

template <int STRIDE, bool WRITE>
static void run(uint32_t* dout, int wps) {
  const int iters = 2000, blocks = 256 * wps;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<STRIDE, WRITE><<<blocks, 256>>>(dout, 10); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    CK(hipEventRecord(e0));
    k<STRIDE, WRITE><<<blocks, 256>>>(dout, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  const double n = (double)iters * 8 * wps * 4;   // DS instructions per CU
  printf("%s b128, lane stride %d slots (%3d B)  wps=%d  %8.3f ms  %6.2f CU-cycles/DS-instr @2.4GHz\n", WRITE ? "write" : "read ", STRIDE, STRIDE * 16,
         wps, best, best * 1e6 * 2.4 / n);
}

int main() {
  uint32_t* dout; CK(hipMalloc(&dout, 4096));
  for (int wps = 1; wps <= 2; wps++) {
    run<1, false>(dout, wps); run<2, false>(dout, wps); run<3, false>(dout, wps); run<4, false>(dout, wps); run<5, false>(dout, wps);
    run<7, false>(dout, wps); run<9, false>(dout, wps);
    run<1, true>(dout, wps); run<3, true>(dout, wps);
  }
  return 0;
}
