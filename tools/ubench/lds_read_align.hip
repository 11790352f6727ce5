// lds_read_align.hip -- cost of ds_read_b128 / ds_read_b64 / ds_read2_b32 on gfx950 at dword-aligned (not 16- / 8-byte-aligned)
// addresses, with the window pattern of the SAD kernels (lane i at dword 3 i / 4 + ..., four "shifted copies" of a row), and the
// values they return (round 4: would a 4-copies staging layout keep the 16-byte window reads?).
// build: hipcc --offload-arch=gfx950 -O2 -o lds_read_align lds_read_align.hip ; run: ./lds_read_align
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

extern __shared__ unsigned lds[];

template <int MODE>   // 0: ds_read_b128, 1: ds_read_b64, 2: ds_read2_b32 offset1:1
__global__ void __launch_bounds__(256) rd(const int* addr_dw, int iters, unsigned* out, int check) {
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (unsigned)i * 2654435761u;
  __syncthreads();
  const unsigned a = (unsigned)addr_dw[threadIdx.x & 63] * 4u + (threadIdx.x >> 6) * 8192u;   // per wavefront an 8 KB area
  unsigned acc = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if constexpr (MODE == 0) {
        unsigned x, y, z, w;
        asm volatile("ds_read_b128 v[20:23], %4\n ds_read_b128 v[24:27], %4\n ds_read_b128 v[28:31], %4\n ds_read_b128 v[32:35], %4\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %0, v20\n v_mov_b32 %1, v21\n v_mov_b32 %2, v22\n v_mov_b32 %3, v23"
                     : "=v"(x), "=v"(y), "=v"(z), "=v"(w) : "v"(a) : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35");
        acc += x ^ y ^ z ^ w;
        if (check && it == 0 && u == 0) { out[threadIdx.x * 4] = x; out[threadIdx.x * 4 + 1] = y; out[threadIdx.x * 4 + 2] = z; out[threadIdx.x * 4 + 3] = w; }
      } else if constexpr (MODE == 1) {
        unsigned x, y;
        asm volatile("ds_read_b64 v[20:21], %2\n ds_read_b64 v[22:23], %2\n ds_read_b64 v[24:25], %2\n ds_read_b64 v[26:27], %2\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %0, v20\n v_mov_b32 %1, v21" : "=v"(x), "=v"(y) : "v"(a) : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
        acc += x ^ y;
        if (check && it == 0 && u == 0) { out[threadIdx.x * 4] = x; out[threadIdx.x * 4 + 1] = y; }
      } else {
        unsigned x, y;
        asm volatile("ds_read2_b32 v[20:21], %2 offset1:1\n ds_read2_b32 v[22:23], %2 offset1:1\n ds_read2_b32 v[24:25], %2 offset1:1\n ds_read2_b32 v[26:27], %2 offset1:1\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %0, v20\n v_mov_b32 %1, v21" : "=v"(x), "=v"(y) : "v"(a) : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
        acc += x ^ y;
        if (check && it == 0 && u == 0) { out[threadIdx.x * 4] = x; out[threadIdx.x * 4 + 1] = y; }
      }
    }
  }
  if (acc == 0x12345u) out[0] = acc;
}

int main() {
  int* d_addr; unsigned* d_out;
  hipMalloc(&d_addr, 64 * 4); hipMalloc(&d_out, 256 * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct Pat { const char* name; int (*f)(int); };
  Pat pats[] = {
    {"16-byte aligned, lane stride 4 dwords (the 16x layout's pattern)", [](int l) { return 4 * l; }},
    {"dword aligned, lane stride 3 dwords", [](int l) { return 3 * l; }},
    {"dword aligned, lane stride 1 dword", [](int l) { return l; }},
    {"four shifted copies (lane l: copy (3l)&3 at dword (3l)>>2, copies 528 dwords apart)", [](int l) { return ((3 * l) & 3) * 528 + ((3 * l) >> 2); }},
  };
  const int iters = 4000;
  for (int mode = 0; mode < 3; mode++)
    for (auto& p : pats) {
      std::vector<int> addr(64);
      for (int l = 0; l < 64; l++) addr[l] = p.f(l);
      hipMemcpy(d_addr, addr.data(), 256, hipMemcpyHostToDevice);
      float ms = 0;
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(rd<0>, dim3(256 * 2), dim3(256), 32768, 0, d_addr, iters, d_out, rep == 0);
        else if (mode == 1) hipLaunchKernelGGL(rd<1>, dim3(256 * 2), dim3(256), 32768, 0, d_addr, iters, d_out, rep == 0);
        else hipLaunchKernelGGL(rd<2>, dim3(256 * 2), dim3(256), 32768, 0, d_addr, iters, d_out, rep == 0);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        if (rep == 0) {   // values of wavefront 0
          std::vector<unsigned> o(256 * 4);
          hipMemcpy(o.data(), d_out, o.size() * 4, hipMemcpyDeviceToHost);
          int bad = 0;
          const int nw = mode == 0 ? 4 : 2;
          for (int l = 0; l < 64; l++)
            for (int k = 0; k < nw; k++)
              if (o[l * 4 + k] != (unsigned)(addr[l] + k) * 2654435761u) bad++;
          printf("%s, %s: values %s\n", mode == 0 ? "ds_read_b128" : mode == 1 ? "ds_read_b64" : "ds_read2_b32", p.name, bad ? "WRONG" : "right");
        }
      }
      // 2 workgroups of 4 wavefronts per CU, each wavefront iters * 8 * 4 reads, four in flight per wavefront
      printf("    %.3f ms: %.1f CU-cycles per wavefront-instruction (8 wavefronts per CU, four reads in flight each)\n", ms,
             ms * 1e-3 * 2.4e9 / (8.0 * iters * 8 * 4));
    }
  return 0;
}
