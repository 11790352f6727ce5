// hbm_calib.hip -- known-byte-count kernels to calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950
// (MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced reads by 2x on this stack; calibrate before trusting).
// copy16: every lane reads 16 B and writes 16 B (N bytes read, N written).  read4/write2: the access widths of the
// stereo kernels' plane traffic (dword reads, 2-byte stores).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void copy16(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) b[i] = a[i];
}
__global__ void read4_write2(const unsigned* __restrict__ a, unsigned short* __restrict__ b, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) b[i] = (unsigned short)(a[i] * 3u);
}
int main() {
  const size_t bytes = (size_t)1 << 30;  // 1 GiB > 256 MiB Infinity Cache
  void *a, *b;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
  size_t n16 = bytes / 16, n4 = bytes / 4;
  for (int r = 0; r < 3; r++) {
    copy16<<<(unsigned)((n16 + 255) / 256), 256>>>((const uint4*)a, (uint4*)b, n16);
    read4_write2<<<(unsigned)((n4 + 255) / 256), 256>>>((const unsigned*)a, (unsigned short*)b, n4);
  }
  CK(hipDeviceSynchronize());
  printf("copy16: read %zu B write %zu B per launch; read4_write2: read %zu B write %zu B per launch\n", bytes, bytes, bytes, n4 * 2);
  return 0;
}
