// lds_dma.hip -- what global_load_lds_{dword,ubyte,ushort} does on gfx950 (round 4, border wavefronts):
//   * is a byte-misaligned source address legal for the dword form, and does it return the right bytes?
//   * where do the sub-dword forms put their data (LDS stride per lane)?
// build: hipcc --offload-arch=gfx950 -O2 -o lds_dma lds_dma.hip ; run: ./lds_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
extern __shared__ unsigned char lds[];

template <int SIZE>
__global__ void k(const unsigned char* src, int stride, int mis, unsigned char* out) {
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 0xEE;
  __syncthreads();
  const unsigned char* p = src + mis + threadIdx.x * stride;
  if constexpr (SIZE == 16) __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)lds, 16, 0, 0);
  else if constexpr (SIZE == 4) __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)lds, 4, 0, 0);
  else if constexpr (SIZE == 2) __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)lds, 2, 0, 0);
  else __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)lds, 1, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}

int main() {
  std::vector<unsigned char> h(4096);
  for (int i = 0; i < 4096; i++) h[i] = (unsigned char)(i * 7 + 3);
  unsigned char *d, *o;
  hipMalloc(&d, 4096); hipMalloc(&o, 1024);
  hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
  std::vector<unsigned char> r(1024);
  for (int size : {4, 1, 2, 16}) {
    for (int mis = 0; mis < 4; mis++) {
      for (int stride : {4, 7, 1}) {
        if (size == 16) hipLaunchKernelGGL(k<16>, dim3(1), dim3(64), 1024, 0, d, stride, mis, o);
        else if (size == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 1024, 0, d, stride, mis, o);
        else if (size == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 1024, 0, d, stride, mis, o);
        else hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 1024, 0, d, stride, mis, o);
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("size %d mis %d stride %d: %s\n", size, mis, stride, hipGetErrorString(e)); return 1; }
        hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
        // find where lane L's bytes landed: try LDS stride 4 (dword slots) and packed (size bytes per lane)
        int ok4 = 1, okp = 1;
        for (int l = 0; l < 64; l++)
          for (int b = 0; b < size; b++) {
            const unsigned char want = h[mis + l * stride + b];
            if (size <= 4 && r[l * 4 + b] != want) ok4 = 0;
            if (r[l * size + b] != want) okp = 0;
          }
        if (size > 4) ok4 = 0;
        printf("size %d src misalign %d lane stride %d: LDS stride-4 layout %s, packed layout %s; lds[0..11] =", size, mis, stride, ok4 ? "MATCH" : "no", okp ? "MATCH" : "no");
        for (int i = 0; i < 12; i++) printf(" %02x", r[i]);
        printf("  (src[mis..] =");
        for (int i = 0; i < 6; i++) printf(" %02x", h[mis + i]);
        printf(")\n");
      }
    }
  }
  return 0;
}
