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

#ifndef VIDAR_AA_ILP
#define VIDAR_AA_ILP 1           // float4 elements per thread of the forward kernel whose loads are issued together
#endif

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
#if VIDAR_AA_ILP > 1
    // staged (tools/staged_variants.sh): kIlp float4 per thread, all loads issued before the first use -- twice / four
    // times the bytes in flight per wave (aa_grid launches correspondingly fewer workgroups per plane)
    constexpr int kIlp = VIDAR_AA_ILP;
    for (int i0 = blockIdx.x * 256 * kIlp + threadIdx.x; i0 < n4; i0 += gridDim.x * 256 * kIlp) {
      float4 v[kIlp], r[kIlp];
#pragma unroll
      for (int u = 0; u < kIlp; ++u) {
        const int i = i0 + u * 256;
        if (i < n4) {
          v[u] = reinterpret_cast<const float4*>(x + base)[i];
          if (res) r[u] = reinterpret_cast<const float4*>(res + base)[i];
        }
      }
#pragma unroll
      for (int u = 0; u < kIlp; ++u) {
        const int i = i0 + u * 256;
        if (i < n4) {
          float4 o = v[u];
          o.x = o.x * s + b; o.y = o.y * s + b; o.z = o.z * s + b; o.w = o.w * s + b;
          if (res) { o.x += r[u].x; o.y += r[u].y; o.z += r[u].z; o.w += r[u].w; }
          if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
          reinterpret_cast<float4*>(y + base)[i] = o;
        }
      }
    }
#else
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
#endif
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
  int bx = (HW / 4 + 256 * VIDAR_AA_ILP - 1) / (256 * VIDAR_AA_ILP);
  if (bx < 1) bx = 1;
  if (bx > 64) bx = 64;
  return dim3(bx, N * C);
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
