// sbm_consume.hip -- device-side consumers of the disparity map: 4x decimation,
// disparity -> 3-D reprojection, keypoint depth lookup.  gfx950.
//
// Reference: src/slam/src/core/SensorData.cpp:50-58 (decimation), Stereo.cpp:157-182 (projectDisparityTo3D),
// Stereo.cpp:184-199 (isFinite, transformPoint), Stereo.cpp:53-117 (generateKeypoints3DStereo, dense-map branch),
// main.cpp:522-553 (reprojection of the decimated map).  The arithmetic keeps the reference's types per operation
// (short/16.0f in float, disp + c in float, the W terms in double, the point products in double, the rigid transform
// in float) and never contracts a multiply-add, so the floats are bit-identical to the C++ expressions.
// All three kernels are trivially HBM-bound (2 B read per pixel, 12 B written per reprojected pixel).
#include "sbm_common.h"

namespace sbm {

#pragma clang fp contract(off)

struct Pt3 { float x, y, z; };

__device__ __forceinline__ Pt3 nan3() {
  const float q = __builtin_nanf("");
  return Pt3{q, q, q};
}

// Stereo.cpp:157-182
__device__ __forceinline__ Pt3 project_disparity(float px, float py, float disp, const sbm_stereo_model& m) {
#pragma clang fp contract(off)
  if (!(disp > 0.0f)) return nan3();
  const float c = (float)(m.cx_r - m.cx_l);
  const float dc = disp + c;                                                    // float + float
  const float Wx = (float)((m.Tx_l / m.fx_l - m.Tx_r / m.fx_r) / (double)dc);
  const float Wy = (float)((m.Tx_l / m.fy_l - m.Tx_r / m.fy_r) / (double)dc);
  Pt3 p;
  p.x = (float)(((double)px - m.cx_l) * (double)Wx);
  p.y = (float)(((double)py - m.cy_l) * (double)Wy);
  p.z = (float)(m.fx_l * (double)Wx);
  return p;
}

__device__ __forceinline__ bool finite3(const Pt3& p) { return isfinite(p.x) && isfinite(p.y) && isfinite(p.z); }

// Stereo.cpp:189-199 (float arithmetic, left-to-right sums)
__device__ __forceinline__ Pt3 transform_point(const Pt3& p, const float* t) {
#pragma clang fp contract(off)
  Pt3 r;
  r.x = t[0] * p.x + t[1] * p.y + t[2] * p.z + t[3];
  r.y = t[4] * p.x + t[5] * p.y + t[6] * p.z + t[7];
  r.z = t[8] * p.x + t[9] * p.y + t[10] * p.z + t[11];
  return r;
}

__global__ void __launch_bounds__(256) decimate_kernel(const int16_t* __restrict__ disp, int16_t* __restrict__ out, int W,
                                                        int H, int scale, int Wd, int Hd) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Wd * Hd) return;
  const int r = i / Wd, c = i % Wd;
  out[(size_t)blockIdx.y * Wd * Hd + i] = disp[(size_t)blockIdx.y * W * H + (size_t)(r * scale) * W + c * scale];
}

__global__ void __launch_bounds__(256) reproject_kernel(const int16_t* __restrict__ disp, float* __restrict__ xyz, int W,
                                                         int H, int scale, sbm_stereo_model m, int apply_local) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= W * H) return;
  const size_t o = (size_t)blockIdx.y * W * H + i;
  const int r = i / W, c = i % W;
  const float d = (float)disp[o] / 16.0f;       // main.cpp:529
  Pt3 p = nan3();
  if (d > 0) {
    p = project_disparity((float)(c * scale), (float)(r * scale), d, m);
    if (finite3(p)) {
      if (apply_local && m.has_local) p = transform_point(p, m.local);
    } else {
      p = nan3();
    }
  }
  xyz[3 * o + 0] = p.x;
  xyz[3 * o + 1] = p.y;
  xyz[3 * o + 2] = p.z;
}

// Stereo.cpp:66-112 (branch "from dense depth map")
__global__ void __launch_bounds__(256) keypoints3d_kernel(const int16_t* __restrict__ disp, const float* __restrict__ kp,
                                                           float* __restrict__ xyz, int W, int H, int nk,
                                                           sbm_stereo_model m, float min_depth, float max_depth) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nk) return;
  const float kx = kp[2 * i], ky = kp[2 * i + 1];
  Pt3 pt = nan3();
  const int ix = (int)kx, iy = (int)ky;
  if (ix >= 0 && ix < W && iy >= 0 && iy < H) {   // the reference indexes unchecked; out-of-image keypoints -> NaN here
    float d = (float)disp[(size_t)iy * W + ix] / 16.0f;
    if (d < 0) d = 0;
    if (d != 0.0f) {
      const Pt3 t = project_disparity(kx, ky, d, m);
      if (finite3(t) && (min_depth < 0.0f || t.z > min_depth) && (max_depth <= 0.0f || t.z <= max_depth)) {
        pt = t;
        if (m.has_local) pt = transform_point(pt, m.local);
      }
    }
  }
  xyz[3 * i + 0] = pt.x;
  xyz[3 * i + 1] = pt.y;
  xyz[3 * i + 2] = pt.z;
}

// cv::StereoBM::compute into a CV_32F destination: disp16.convertTo(dst, CV_32F, 1. / 16) -- exact in float
__global__ void __launch_bounds__(256) disp_to_float_kernel(const int16_t* __restrict__ disp, float* __restrict__ out, size_t n) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 4 <= n) {
    short v[4];
    __builtin_memcpy(v, disp + i, 8);
    float f[4];
#pragma unroll
    for (int k = 0; k < 4; k++) f[k] = (float)((double)v[k] * (1.0 / 16.0));
    __builtin_memcpy(out + i, f, 16);
  } else {
    for (size_t k = i; k < n; k++) out[k] = (float)((double)disp[k] * (1.0 / 16.0));
  }
}

hipError_t launch_disp_to_float(const int16_t* disp, float* out, size_t count, hipStream_t s) {
  if (count == 0) return hipSuccess;
  hipLaunchKernelGGL(disp_to_float_kernel, dim3((unsigned)((count + 1023) / 1024)), dim3(256), 0, s, disp, out, count);
  return hipGetLastError();
}

hipError_t launch_decimate(const int16_t* disp, int16_t* out, int n, int W, int H, int scale, hipStream_t s) {
  const int Wd = W / scale, Hd = H / scale;
  if (Wd <= 0 || Hd <= 0) return hipSuccess;
  hipLaunchKernelGGL(decimate_kernel, dim3((Wd * Hd + 255) / 256, n), dim3(256), 0, s, disp, out, W, H, scale, Wd, Hd);
  return hipGetLastError();
}

hipError_t launch_reproject(const int16_t* disp, float* xyz, int n, int W, int H, int scale, const sbm_stereo_model& m,
                            int apply_local, hipStream_t s) {
  hipLaunchKernelGGL(reproject_kernel, dim3((W * H + 255) / 256, n), dim3(256), 0, s, disp, xyz, W, H, scale, m, apply_local);
  return hipGetLastError();
}

hipError_t launch_keypoints3d(const int16_t* disp, const float* kp, float* xyz, int W, int H, int nk,
                              const sbm_stereo_model& m, float min_depth, float max_depth, hipStream_t s) {
  if (nk <= 0) return hipSuccess;
  hipLaunchKernelGGL(keypoints3d_kernel, dim3((nk + 255) / 256), dim3(256), 0, s, disp, kp, xyz, W, H, nk, m, min_depth,
                     max_depth);
  return hipGetLastError();
}

}  // namespace sbm
