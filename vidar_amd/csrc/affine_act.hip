// Fused frozen-BatchNorm epilogue for the ResNet backbone: y = act(x * scale[c] + shift[c] (+ res)).
//
// The reference backbone (mmdet ResNet, `norm_cfg=dict(type='BN2d', requires_grad=False)`,
// `norm_eval=True`, vidar_1_8_nusc_1future.py:93-95) runs BatchNorm with frozen statistics and
// frozen affine, followed by ReLU and -- at the end of a bottleneck -- the residual add.  In
// PyTorch these are 2-3 full passes over the activation tensor each (BN, add, ReLU); at
// 6x256x232x400 fp32 (570 MB) the backbone is bound by exactly these HBM passes, not by its
// convolutions (measured: convs ~100 TFLOP/s on MIOpen, 25 of 59 ms).  One pass here.
// Layout NCHW; a lane owns 4 consecutive pixels of one channel plane (float4) when HW % 4 == 0.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vidar_hip.h"
#include "vidar_common.h"


namespace {

template <bool VEC>
__global__ __launch_bounds__(256) void affine_act_fwd_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift,
                                                             const float* __restrict__ res,
                                                             float* __restrict__ y, int C, int HW,
                                                             int relu) {
  const int plane = blockIdx.y;                 // n * C + c
  const int c = plane % C;
  const float s = scale[c], b = shift[c];
  const size_t base = (size_t)plane * HW;
  if (VEC) {
    const int n4 = HW >> 2;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
      float4 v = reinterpret_cast<const float4*>(x + base)[i];
      v.x = v.x * s + b; v.y = v.y * s + b; v.z = v.z * s + b; v.w = v.w * s + b;
      if (res) {
        const float4 r = reinterpret_cast<const float4*>(res + base)[i];
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
      }
      if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      reinterpret_cast<float4*>(y + base)[i] = v;
    }
  } else {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
      float v = x[base + i] * s + b;
      if (res) v += res[base + i];
      y[base + i] = relu ? fmaxf(v, 0.f) : v;
    }
  }
}

template <bool VEC>
__global__ __launch_bounds__(256) void affine_act_bwd_kernel(const float* __restrict__ gy,
                                                             const float* __restrict__ y,
                                                             const float* __restrict__ scale,
                                                             float* __restrict__ gx,
                                                             float* __restrict__ gres, int C, int HW,
                                                             int relu) {
  const int plane = blockIdx.y;
  const float s = scale[plane % C];
  const size_t base = (size_t)plane * HW;
  if (VEC) {
    const int n4 = HW >> 2;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
      float4 g = reinterpret_cast<const float4*>(gy + base)[i];
      if (relu) {
        const float4 o = reinterpret_cast<const float4*>(y + base)[i];
        g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f;
        g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
      }
      if (gres) reinterpret_cast<float4*>(gres + base)[i] = g;
      reinterpret_cast<float4*>(gx + base)[i] = make_float4(g.x * s, g.y * s, g.z * s, g.w * s);
    }
  } else {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
      float g = gy[base + i];
      if (relu && !(y[base + i] > 0.f)) g = 0.f;
      if (gres) gres[base + i] = g;
      gx[base + i] = g * s;
    }
  }
}

// VEC forward: float4 elements a workgroup covers per pass (the backward kernel keeps one per thread)
inline dim3 aa_grid_fwd(int N, int C, int HW) {
  int bx = (HW / 4 + 255) / 256;
  if (bx < 1) bx = 1;
  if (bx > 64) bx = 64;
  return dim3(bx, N * C);
}
// Stem of the ResNet (conv1 -> frozen BN -> ReLU -> 3x3 / stride 2 / pad 1 max-pool; the stem is frozen in every
// ViDAR config, `frozen_stages=1`, so it is forward only): BN + ReLU + pool in ONE pass over the conv1 output instead
// of affine_act (read + write of the full-resolution map) followed by torch's pooling kernel (1.75 TB/s).  A thread
// produces a 2 x 2 block of outputs from 5 rows x (one float4 + the column left of it); the affine is the same fused
// multiply-add affine_act_fwd_kernel compiles to, and max / ReLU commute, so the result is bit-identical to the
// two-kernel path.  Needs W % 4 == 0.   grid: (ceil(W/4 / 64), ceil(ceil(Ho/2) / 4), N*C), 256 threads.
__global__ __launch_bounds__(256) void stem_pool_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, float* __restrict__ y,
                                                        int C, int H, int W, int Ho, int Wo) {
  const int plane = blockIdx.z;
  const float s = scale[plane % C], b = shift[plane % C];
  const int t = blockIdx.x * 64 + (threadIdx.x & 63);          // output columns 2t, 2t+1 <- input columns 4t-1 .. 4t+3
  const int r = blockIdx.y * 4 + (threadIdx.x >> 6);           // output rows 2r, 2r+1    <- input rows 4r-1 .. 4r+3
  if (t >= (W >> 2) || 2 * r >= Ho) return;
  const float* in = x + (size_t)plane * H * W;
  float* out = y + (size_t)plane * Ho * Wo;
  const float ninf = -__builtin_inff();
  float m0a = ninf, m0b = ninf, m1a = ninf, m1b = ninf;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int row = 4 * r - 1 + i;
    if (row < 0 || row >= H) continue;
    const float* p = in + (size_t)row * W + 4 * t;
    const float4 v = *reinterpret_cast<const float4*>(p);
    float l = ninf;
    if (t > 0) l = fmaxf(__builtin_fmaf(p[-1], s, b), 0.f);
    const float vx = fmaxf(__builtin_fmaf(v.x, s, b), 0.f), vy = fmaxf(__builtin_fmaf(v.y, s, b), 0.f);
    const float vz = fmaxf(__builtin_fmaf(v.z, s, b), 0.f), vw = fmaxf(__builtin_fmaf(v.w, s, b), 0.f);
    const float a = fmaxf(fmaxf(l, vx), vy), c2 = fmaxf(fmaxf(vy, vz), vw);
    if (i <= 2) { m0a = fmaxf(m0a, a); m0b = fmaxf(m0b, c2); }
    if (i >= 2) { m1a = fmaxf(m1a, a); m1b = fmaxf(m1b, c2); }
  }
  *reinterpret_cast<float2*>(out + (size_t)(2 * r) * Wo + 2 * t) = make_float2(m0a, m0b);
  if (2 * r + 1 < Ho) *reinterpret_cast<float2*>(out + (size_t)(2 * r + 1) * Wo + 2 * t) = make_float2(m1a, m1b);
}

inline dim3 aa_grid(int N, int C, int HW) {
  int bx = (HW / 4 + 255) / 256;
  if (bx < 1) bx = 1;
  if (bx > 64) bx = 64;
  return dim3(bx, N * C);
}
inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" {

int vidar_affine_act_fwd_f32(const float* x, const float* scale, const float* shift,
                             const float* residual, float* y, int N, int C, int HW, int relu,
                             void* stream) {
  VIDAR_ENTER();
  if (N < 0 || C <= 0 || HW <= 0) return VIDAR_ERR_BAD_ARG;
  if (N == 0) return 0;
  const bool vec = (HW % 4 == 0) && aligned16(x) && aligned16(y) && (!residual || aligned16(residual));
  if (vec)
    hipLaunchKernelGGL(affine_act_fwd_kernel<true>, aa_grid_fwd(N, C, HW), dim3(256), 0,
                       (hipStream_t)stream, x, scale, shift, residual, y, C, HW, relu);
  else
    hipLaunchKernelGGL(affine_act_fwd_kernel<false>, aa_grid(N, C, HW * 4), dim3(256), 0,
                       (hipStream_t)stream, x, scale, shift, residual, y, C, HW, relu);
  return vidar_last_error();
}

int vidar_stem_bn_relu_pool_f32(const float* x, const float* scale, const float* shift, float* y, int N, int C,
                                int H, int W, void* stream) {
  VIDAR_ENTER();
  if (N < 0 || C <= 0 || H <= 0 || W <= 0 || (W & 3) || !aligned16(x) || ((uintptr_t)y & 7)) return VIDAR_ERR_BAD_ARG;
  if ((int64_t)N * C > 65535) return VIDAR_ERR_BAD_ARG;          // grid.z
  if (N == 0) return 0;
  const int Ho = (H - 1) / 2 + 1, Wo = W / 2;
  const dim3 grid(((W >> 2) + 63) / 64, ((Ho + 1) / 2 + 3) / 4, N * C);
  hipLaunchKernelGGL(stem_pool_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, scale, shift, y, C, H, W, Ho, Wo);
  return vidar_last_error();
}

int vidar_affine_act_bwd_f32(const float* grad_y, const float* y, const float* scale, float* grad_x,
                             float* grad_residual, int N, int C, int HW, int relu, void* stream) {
  VIDAR_ENTER();
  if (N < 0 || C <= 0 || HW <= 0) return VIDAR_ERR_BAD_ARG;
  if (N == 0) return 0;
  const bool vec = (HW % 4 == 0) && aligned16(grad_y) && aligned16(y) && aligned16(grad_x) &&
                   (!grad_residual || aligned16(grad_residual));
  if (vec)
    hipLaunchKernelGGL(affine_act_bwd_kernel<true>, aa_grid(N, C, HW), dim3(256), 0,
                       (hipStream_t)stream, grad_y, y, scale, grad_x, grad_residual, C, HW, relu);
  else
    hipLaunchKernelGGL(affine_act_bwd_kernel<false>, aa_grid(N, C, HW * 4), dim3(256), 0,
                       (hipStream_t)stream, grad_y, y, scale, grad_x, grad_residual, C, HW, relu);
  return vidar_last_error();
}

}  // extern "C"
