// lds_unaligned.hip -- gfx950: are byte-misaligned ds_read_b64/b128 correct and how fast are they when 64 lanes read
// overlapping windows of one short row piece (lane stride 1 byte)?  Round 3 probe for the SAD kernel's staging.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_unaligned lds_unaligned.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// MODE 0: expanded layout (slot p = bytes p..p+15), aligned b128 at lane stride 16 B  (what the SAD kernel does today)
// MODE 1: raw layout, b128 at byte address lane + 16 m
// MODE 2: raw layout, b64 at byte address lane + 4 q
// MODE 3: raw layout, b128 at byte address 4*lane + 16 m (dword aligned, overlapping)
template <int MODE>
__global__ void __launch_bounds__(256) rd_kernel(const uint8_t* src, uint32_t* out, int iters, int check) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[4 * 8192];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint8_t* base = lds + wv * 8192;
  // fill: raw row piece of 512 bytes at base; expanded copy at base + 1024 (256 slots x 16 B)
  for (int i = lane; i < 512; i += 64) base[i] = src[i];
  for (int p = lane; p < 256; p += 64)
    for (int b = 0; b < 16; b++) base[1024 + p * 16 + b] = src[p + b];
  __syncthreads();
  uint32_t acc = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int m = 0; m < 8; m++) {
      if (MODE == 0) {
        uint4 v = *reinterpret_cast<const uint4*>(base + 1024 + (lane + 16 * m) * 16);
        acc += v.x ^ v.y ^ v.z ^ v.w;
      } else if (MODE == 1) {
        uint4 v; __builtin_memcpy(&v, base + lane + 16 * m, 16);
        acc += v.x ^ v.y ^ v.z ^ v.w;
      } else if (MODE == 2) {
        uint2 v; __builtin_memcpy(&v, base + lane + 4 * m, 8);
        uint2 w; __builtin_memcpy(&w, base + lane + 4 * m + 32, 8);
        acc += v.x ^ v.y ^ w.x ^ w.y;
      } else {
        uint4 v; __builtin_memcpy(&v, base + 4 * lane + 16 * m, 16);
        acc += v.x ^ v.y ^ v.z ^ v.w;
      }
    }
    asm volatile("" : "+v"(acc));
  }
  if (check) {
    // correctness: one window per lane compared on the host
    uint4 v; __builtin_memcpy(&v, base + lane + 16 * 3, 16);
    uint2 w; __builtin_memcpy(&w, base + lane + 4 * 5, 8);
    uint32_t* o = out + (blockIdx.x * 256 + threadIdx.x) * 8;
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; o[4] = w.x; o[5] = w.y; o[6] = acc; o[7] = 0;
  } else if (acc == 0x12345) out[0] = acc;
}

template <int MODE>
static void run(const char* name, const uint8_t* dsrc, uint32_t* dout, int wps) {
  const int iters = 2000, blocks = 256 * wps;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  rd_kernel<MODE><<<blocks, 256>>>(dsrc, dout, 10, 0); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    CK(hipEventRecord(e0));
    rd_kernel<MODE><<<blocks, 256>>>(dsrc, dout, iters, 0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  const double n = (double)iters * 8 * (MODE == 2 ? 2 : 1) * wps * 4;   // DS instructions per CU
  printf("%-44s wps=%d  %8.3f ms  %6.2f CU-cycles/DS-instr @2.4GHz\n", name, wps, best, best * 1e6 * 2.4 / n);
}

int main() {
  std::vector<uint8_t> h(1024);
  for (int i = 0; i < 1024; i++) h[i] = (uint8_t)(i * 37 + (i >> 3) * 11 + 5);
  uint8_t* dsrc; uint32_t* dout;
  CK(hipMalloc(&dsrc, 1024)); CK(hipMalloc(&dout, 256 * 8 * 256 * 8 * 4));
  CK(hipMemcpy(dsrc, h.data(), 1024, hipMemcpyHostToDevice));
  rd_kernel<1><<<1, 256>>>(dsrc, dout, 1, 1); CK(hipDeviceSynchronize());
  std::vector<uint32_t> o(256 * 8);
  CK(hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int t = 0; t < 256; t++) {
    const int lane = t & 63;
    uint32_t e[6];
    memcpy(e, &h[lane + 48], 16); memcpy(e + 4, &h[lane + 20], 8);
    for (int k = 0; k < 6; k++) if (o[t * 8 + k] != e[k]) bad++;
  }
  printf("unaligned ds_read_b128 / b64 correctness: %d mismatching dwords of %d\n", bad, 256 * 6);
  for (int wps : {1, 2, 4}) {
    run<0>("expanded layout, aligned b128 (today)", dsrc, dout, wps);
    run<1>("raw layout, b128 at lane + 16m (1-byte stride)", dsrc, dout, wps);
    run<2>("raw layout, b64 at lane + 4m", dsrc, dout, wps);
    run<3>("raw layout, b128 at 4*lane + 16m", dsrc, dout, wps);
  }
  return 0;
}
