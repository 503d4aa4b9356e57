// y = LayerNorm(dropout(x) + residual) in one pass, and its backward.
//
// Every attention / FFN block of the BEV encoder and the future decoder ends with
//     out = dropout(proj(...)) + identity ;  query = LayerNorm(out)
// (temporal_self_attention.py:270-271, spatial_cross_attention.py:172-174, vidar_decoder.py:515-516, mmcv FFN
//  + custom_base_transformer_layer.py norms), three torch kernels forward and about six backward over
// [bs*Q, 256] maps (41 MB each at 200x200).  Here one wave owns a row of 256 channels (lane = 4 channels):
// forward reads x and residual once and writes y plus the pre-norm sum the backward needs; backward produces
// both input gradients in one pass and accumulates the affine gradients per wave before one atomic per
// channel.  Dropout keeps an element iff hash(seed, element index) >= p -- the mask is recomputed in the
// backward, never stored (not torch's Philox stream: dropout noise has no parity contract; p = 0 / eval is
// exact).  fp32, mean / biased variance over the row, eps inside the square root (torch.nn.LayerNorm).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vidar_hip.h"
#include "vidar_common.h"

namespace {

constexpr int kC = 256;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

__device__ __forceinline__ float keep_scale(uint32_t seed, uint64_t idx, float p, float scale) {
  if (p <= 0.f) return 1.f;
  uint32_t h = (uint32_t)idx * 0x9E3779B9u ^ (uint32_t)(idx >> 32) * 0x85EBCA6Bu ^ seed;
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;       // murmur3 finaliser
  return ((h >> 8) * (1.0f / 16777216.0f) >= p) ? scale : 0.f;
}

__global__ __launch_bounds__(256) void drop_add_ln_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ y, float* __restrict__ sum_out,
    float* __restrict__ mean_out, float* __restrict__ rstd_out, int64_t rows, float p, float eps, uint32_t seed) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int64_t o = row * kC + lane * 4;
  const float4 v = *reinterpret_cast<const float4*>(x + o);
  const float4 r = *reinterpret_cast<const float4*>(res + o);
  const float sc = p > 0.f ? 1.f / (1.f - p) : 1.f;
  float4 s;
  s.x = v.x * keep_scale(seed, o, p, sc) + r.x;
  s.y = v.y * keep_scale(seed, o + 1, p, sc) + r.y;
  s.z = v.z * keep_scale(seed, o + 2, p, sc) + r.z;
  s.w = v.w * keep_scale(seed, o + 3, p, sc) + r.w;
  const float mean = wave_sum(s.x + s.y + s.z + s.w) * (1.f / kC);
  const float dx = s.x - mean, dy = s.y - mean, dz = s.z - mean, dw = s.w - mean;
  const float var = wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.f / kC);
  const float rstd = 1.f / sqrtf(var + eps);
  const float4 g = *reinterpret_cast<const float4*>(gamma + lane * 4);
  const float4 b = *reinterpret_cast<const float4*>(beta + lane * 4);
  float4 out;
  out.x = dx * rstd * g.x + b.x; out.y = dy * rstd * g.y + b.y;
  out.z = dz * rstd * g.z + b.z; out.w = dw * rstd * g.w + b.w;
  *reinterpret_cast<float4*>(y + o) = out;
  *reinterpret_cast<float4*>(sum_out + o) = s;
  if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

constexpr int kRowsPerWave = 16;
__global__ __launch_bounds__(256) void drop_add_ln_bwd_kernel(
    const float* __restrict__ gout, const float* __restrict__ sum_in, const float* __restrict__ gamma,
    const float* __restrict__ mean_in, const float* __restrict__ rstd_in, float* __restrict__ gx,
    float* __restrict__ gres, float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t rows, float p,
    uint32_t seed) {
  const int lane = threadIdx.x & 63;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * kRowsPerWave;
  const float4 g = *reinterpret_cast<const float4*>(gamma + lane * 4);
  const float sc = p > 0.f ? 1.f / (1.f - p) : 1.f;
  float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), ab = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = 0; k < kRowsPerWave; ++k) {
    const int64_t row = row0 + k;
    if (row >= rows) break;
    const int64_t o = row * kC + lane * 4;
    const float4 go = *reinterpret_cast<const float4*>(gout + o);
    const float4 s = *reinterpret_cast<const float4*>(sum_in + o);
    const float mean = mean_in[row], rstd = rstd_in[row];
    const float hx = (s.x - mean) * rstd, hy = (s.y - mean) * rstd, hz = (s.z - mean) * rstd, hw = (s.w - mean) * rstd;
    const float yx = go.x * g.x, yy = go.y * g.y, yz = go.z * g.z, yw = go.w * g.w;
    const float c1 = wave_sum(yx + yy + yz + yw) * (1.f / kC);
    const float c2 = wave_sum(yx * hx + yy * hy + yz * hz + yw * hw) * (1.f / kC);
    float4 gs;
    gs.x = rstd * (yx - c1 - hx * c2); gs.y = rstd * (yy - c1 - hy * c2);
    gs.z = rstd * (yz - c1 - hz * c2); gs.w = rstd * (yw - c1 - hw * c2);
    *reinterpret_cast<float4*>(gres + o) = gs;
    float4 gv;
    gv.x = gs.x * keep_scale(seed, o, p, sc); gv.y = gs.y * keep_scale(seed, o + 1, p, sc);
    gv.z = gs.z * keep_scale(seed, o + 2, p, sc); gv.w = gs.w * keep_scale(seed, o + 3, p, sc);
    *reinterpret_cast<float4*>(gx + o) = gv;
    ag.x += go.x * hx; ag.y += go.y * hy; ag.z += go.z * hz; ag.w += go.w * hw;
    ab.x += go.x; ab.y += go.y; ab.z += go.z; ab.w += go.w;
  }
  if (row0 < rows) {
    float* dg = dgamma + lane * 4;
    float* db = dbeta + lane * 4;
    unsafeAtomicAdd(dg, ag.x); unsafeAtomicAdd(dg + 1, ag.y); unsafeAtomicAdd(dg + 2, ag.z); unsafeAtomicAdd(dg + 3, ag.w);
    unsafeAtomicAdd(db, ab.x); unsafeAtomicAdd(db + 1, ab.y); unsafeAtomicAdd(db + 2, ab.z); unsafeAtomicAdd(db + 3, ab.w);
  }
}

}  // namespace

extern "C" {

int vidar_drop_add_ln_fwd_f32(const float* x, const float* residual, const float* gamma, const float* beta, float* y,
                              float* sum_out, float* mean_out, float* rstd_out, int64_t rows, int C, float p, float eps,
                              uint32_t seed, void* stream) {
  VIDAR_ENTER();
  if (rows < 0 || C != kC || p < 0.f || p >= 1.f) return VIDAR_ERR_BAD_ARG;
  if (rows == 0) return 0;
  hipLaunchKernelGGL(drop_add_ln_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x,
                     residual, gamma, beta, y, sum_out, mean_out, rstd_out, rows, p, eps, seed);
  return vidar_last_error();
}

int vidar_drop_add_ln_bwd_f32(const float* grad_y, const float* sum_in, const float* gamma, const float* mean_in,
                              const float* rstd_in, float* grad_x, float* grad_residual, float* grad_gamma,
                              float* grad_beta, int64_t rows, int C, float p, uint32_t seed, void* stream) {
  VIDAR_ENTER();
  if (rows < 0 || C != kC || p < 0.f || p >= 1.f) return VIDAR_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(grad_gamma, 0, sizeof(float) * kC, s);
  if (e == hipSuccess) e = hipMemsetAsync(grad_beta, 0, sizeof(float) * kC, s);
  if (e != hipSuccess) return (int)e;
  if (rows == 0) return 0;
  const int64_t waves = (rows + kRowsPerWave - 1) / kRowsPerWave;
  hipLaunchKernelGGL(drop_add_ln_bwd_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, grad_y, sum_in, gamma,
                     mean_in, rstd_in, grad_x, grad_residual, grad_gamma, grad_beta, rows, p, seed);
  return vidar_last_error();
}

}  // extern "C"
