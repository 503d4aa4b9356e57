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

constexpr int kRowsPerWave = 4;
__global__ __launch_bounds__(256) void drop_add_ln_bwd_kernel(
    const float* __restrict__ gout, const float* __restrict__ sum_in, const float* __restrict__ gamma,
    const float* __restrict__ mean_in, const float* __restrict__ rstd_in, float* __restrict__ gx,
    float* __restrict__ gres, float* __restrict__ partial, float* __restrict__ dgamma, float* __restrict__ dbeta,
    int64_t rows, float p, uint32_t seed) {
  if (blockIdx.x == 0) {                      // zero the second stage's accumulators (it runs after this kernel)
    dgamma[threadIdx.x] = 0.f;
    dbeta[threadIdx.x] = 0.f;
  }
  const int lane = threadIdx.x & 63;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * kRowsPerWave;
  const float4 g = *reinterpret_cast<const float4*>(gamma + lane * 4);
  const float sc = p > 0.f ? 1.f / (1.f - p) : 1.f;
  float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), ab = make_float4(0.f, 0.f, 0.f, 0.f);
  // all loads of the wave's rows are issued before the first reduction (the two wave reductions per row are a
  // dependent chain; with the loads inside the loop the kernel ran at 1.2 TB/s)
  float4 go[kRowsPerWave], sv[kRowsPerWave];
  float mean[kRowsPerWave], rstd[kRowsPerWave];
#pragma unroll
  for (int k = 0; k < kRowsPerWave; ++k) {
    const int64_t row = min(row0 + k, rows - 1);
    const int64_t o = row * kC + lane * 4;
    go[k] = *reinterpret_cast<const float4*>(gout + o);
    sv[k] = *reinterpret_cast<const float4*>(sum_in + o);
    mean[k] = mean_in[row]; rstd[k] = rstd_in[row];
  }
#pragma unroll
  for (int k = 0; k < kRowsPerWave; ++k) {
    const int64_t row = row0 + k;
    if (row >= rows) break;
    const int64_t o = row * kC + lane * 4;
    const float4 s = sv[k];
    const float hx = (s.x - mean[k]) * rstd[k], hy = (s.y - mean[k]) * rstd[k], hz = (s.z - mean[k]) * rstd[k],
                hw = (s.w - mean[k]) * rstd[k];
    const float yx = go[k].x * g.x, yy = go[k].y * g.y, yz = go[k].z * g.z, yw = go[k].w * g.w;
    const float c1 = wave_sum(yx + yy + yz + yw) * (1.f / kC);
    const float c2 = wave_sum(yx * hx + yy * hy + yz * hz + yw * hw) * (1.f / kC);
    float4 gs;
    gs.x = rstd[k] * (yx - c1 - hx * c2); gs.y = rstd[k] * (yy - c1 - hy * c2);
    gs.z = rstd[k] * (yz - c1 - hz * c2); gs.w = rstd[k] * (yw - c1 - hw * c2);
    *reinterpret_cast<float4*>(gres + o) = gs;
    float4 gv;
    gv.x = gs.x * keep_scale(seed, o, p, sc); gv.y = gs.y * keep_scale(seed, o + 1, p, sc);
    gv.z = gs.z * keep_scale(seed, o + 2, p, sc); gv.w = gs.w * keep_scale(seed, o + 3, p, sc);
    *reinterpret_cast<float4*>(gx + o) = gv;
    ag.x += go[k].x * hx; ag.y += go[k].y * hy; ag.z += go[k].z * hz; ag.w += go[k].w * hw;
    ab.x += go[k].x; ab.y += go[k].y; ab.z += go[k].z; ab.w += go[k].w;
  }
  // affine gradients: the 4 waves of the workgroup reduce in LDS, the workgroup writes ONE partial row
  // (atomics from every wave onto the same 256 addresses serialise: 0.98 ms for a 40000-row map)
  __shared__ float4 s_g[4][64], s_b[4][64];
  s_g[threadIdx.x >> 6][lane] = ag;
  s_b[threadIdx.x >> 6][lane] = ab;
  __syncthreads();
  if (threadIdx.x < 64) {
    float4 a = s_g[0][lane], b = s_b[0][lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float4 c = s_g[w][lane], d = s_b[w][lane];
      a.x += c.x; a.y += c.y; a.z += c.z; a.w += c.w;
      b.x += d.x; b.y += d.y; b.z += d.z; b.w += d.w;
    }
    float* part = partial + (size_t)blockIdx.x * 2 * kC;
    *reinterpret_cast<float4*>(part + lane * 4) = a;
    *reinterpret_cast<float4*>(part + kC + lane * 4) = b;
  }
}

// second stage: grid (16 column blocks, kSlices row slices); block = 32 of the 2*256 affine-gradient columns x 8 row
// lanes.  Every thread sums its share of the partial rows, the 8 lanes meet in LDS, one atomic per (block, column)
// onto the result -- which workgroup 0 of the first stage zeroed (stream order makes that safe, no memset launch).
constexpr int kSlices = 8;
__global__ __launch_bounds__(256) void affine_grad_reduce_kernel(const float* __restrict__ partial, int nparts,
                                                                 float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta) {
  __shared__ float s_acc[8][32];
  const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + c;                       // 0 .. 2*kC-1
  const int per = (nparts + kSlices - 1) / kSlices;
  const int p0 = blockIdx.y * per, p1 = min(nparts, p0 + per);
  float a0 = 0.f, a1 = 0.f;
  int i = p0 + r;
  for (; i + 8 < p1; i += 16) {
    a0 += partial[(size_t)i * 2 * kC + col];
    a1 += partial[(size_t)(i + 8) * 2 * kC + col];
  }
  if (i < p1) a0 += partial[(size_t)i * 2 * kC + col];
  s_acc[r][c] = a0 + a1;
  __syncthreads();
  if (r == 0 && p0 < p1) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += s_acc[k][c];
    unsafeAtomicAdd(col < kC ? dgamma + col : dbeta + (col - kC), t);
  }
}

// ---------------------------------------------------------------------------------------------
// Column sums of a row-major [rows, cols] matrix: the bias gradient of a Linear layer, grad_b = sum over rows of
// grad_y (torch's generic reduction runs this shape at ~1.2 TB/s: 134 calls, ~5 ms per training step).
// One resident grid (one 1024-thread workgroup per CU-sized slab): thread = (float4 column group, row lane), rows are
// read as whole coalesced lines, the row lanes of a workgroup meet in LDS and the workgroup adds ONE partial row to the
// zeroed output (256 workgroups x cols atomics).
// ---------------------------------------------------------------------------------------------
constexpr int kColsumThreads = 1024;
constexpr int kColsumIlp = 4;        // rows in flight per thread (16-byte loads); at most 256 workgroups: every workgroup ends
                                     // with `cols` atomics on the same addresses (2 048 small workgroups measured 4 x SLOWER)

__global__ __launch_bounds__(kColsumThreads) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                                int64_t rows, int cols) {
  __shared__ float4 s_acc[kColsumThreads];
  const int groups = cols >> 2;                                // float4 column groups, a power of two <= 1024
  const int g = threadIdx.x & (groups - 1), r0 = threadIdx.x / groups, rstep = kColsumThreads / groups;
  const int64_t per = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t row_lo = (int64_t)blockIdx.x * per, row_hi = min(rows, row_lo + per);
  float4 acc[kColsumIlp];
#pragma unroll
  for (int u = 0; u < kColsumIlp; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  int64_t r = row_lo + r0;
  for (; r + (int64_t)(kColsumIlp - 1) * rstep < row_hi; r += (int64_t)kColsumIlp * rstep) {
    float4 v[kColsumIlp];
#pragma unroll
    for (int u = 0; u < kColsumIlp; ++u) v[u] = reinterpret_cast<const float4*>(x + (r + (int64_t)u * rstep) * cols)[g];
#pragma unroll
    for (int u = 0; u < kColsumIlp; ++u) { acc[u].x += v[u].x; acc[u].y += v[u].y; acc[u].z += v[u].z; acc[u].w += v[u].w; }
  }
  for (; r < row_hi; r += rstep) {
    const float4 v0 = reinterpret_cast<const float4*>(x + r * cols)[g];
    acc[0].x += v0.x; acc[0].y += v0.y; acc[0].z += v0.z; acc[0].w += v0.w;
  }
  float4 a = acc[0];
#pragma unroll
  for (int u = 1; u < kColsumIlp; ++u) { a.x += acc[u].x; a.y += acc[u].y; a.z += acc[u].z; a.w += acc[u].w; }
  s_acc[threadIdx.x] = a;
  __syncthreads();
  for (int half = rstep >> 1; half >= 1; half >>= 1) {         // tree over the row lanes of a column group
    if (r0 < half) {
      const float4 o = s_acc[threadIdx.x + half * groups];
      float4& m = s_acc[threadIdx.x];
      m.x += o.x; m.y += o.y; m.z += o.z; m.w += o.w;
    }
    __syncthreads();
  }
  if (r0 == 0 && row_lo < row_hi) {
    const float4 t = s_acc[threadIdx.x];
    unsafeAtomicAdd(out + 4 * g, t.x); unsafeAtomicAdd(out + 4 * g + 1, t.y);
    unsafeAtomicAdd(out + 4 * g + 2, t.z); unsafeAtomicAdd(out + 4 * g + 3, t.w);
  }
}


// ---------------------------------------------------------------------------------------------
// y = dropout(relu(x)) in one pass, and its backward in one pass: the hidden activation of every FFN
// (mmcv FFN: Linear -> ReLU -> Dropout -> Linear, 12 blocks x 5 frames on [40 000, 512] maps).  torch runs relu
// (unless the GEMM epilogue did it), native_dropout (+ a byte mask) forward and native_dropout_backward +
// threshold_backward backward -- 2 + 2 passes over 82 MB; here 1 + 1, no mask: keep iff hash(seed, i) >= p like the
// fused LayerNorm tail above, and the backward needs neither the hash nor x: y > 0 <=> the element was positive AND
// kept, so grad_x = y > 0 ? grad_y / (1 - p) : 0.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void relu_drop_fwd_kernel(const float4* __restrict__ x, float4* __restrict__ y,
                                                            int64_t n4, float p, uint32_t seed) {
  const float sc = p > 0.f ? 1.f / (1.f - p) : 1.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = x[i];
    const uint64_t o = (uint64_t)i * 4;
    float4 r;
    r.x = fmaxf(v.x, 0.f) * keep_scale(seed, o, p, sc);     r.y = fmaxf(v.y, 0.f) * keep_scale(seed, o + 1, p, sc);
    r.z = fmaxf(v.z, 0.f) * keep_scale(seed, o + 2, p, sc); r.w = fmaxf(v.w, 0.f) * keep_scale(seed, o + 3, p, sc);
    y[i] = r;
  }
}

__global__ __launch_bounds__(256) void relu_drop_bwd_kernel(const float4* __restrict__ gy, const float4* __restrict__ y,
                                                            float4* __restrict__ gx, int64_t n4, float sc) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 g = gy[i], v = y[i];
    gx[i] = make_float4(v.x > 0.f ? g.x * sc : 0.f, v.y > 0.f ? g.y * sc : 0.f, v.z > 0.f ? g.z * sc : 0.f,
                        v.w > 0.f ? g.w * sc : 0.f);
  }
}

}  // namespace

extern "C" {

int vidar_relu_drop_fwd_f32(const float* x, float* y, int64_t n, float p, uint32_t seed, void* stream) {
  VIDAR_ENTER();
  if (n < 0 || n % 4 != 0 || !(p >= 0.f) || p >= 1.f) return VIDAR_ERR_BAD_ARG;
  if (n == 0) return 0;
  const int64_t n4 = n / 4;
  const unsigned grid = (unsigned)(n4 / 256 + 1 < 16384 ? n4 / 256 + 1 : 16384);
  hipLaunchKernelGGL(relu_drop_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(y), n4, p, seed);
  return vidar_last_error();
}

int vidar_relu_drop_bwd_f32(const float* grad_y, const float* y, float* grad_x, int64_t n, float p, void* stream) {
  VIDAR_ENTER();
  if (n < 0 || n % 4 != 0 || !(p >= 0.f) || p >= 1.f) return VIDAR_ERR_BAD_ARG;
  if (n == 0) return 0;
  const int64_t n4 = n / 4;
  const unsigned grid = (unsigned)(n4 / 256 + 1 < 16384 ? n4 / 256 + 1 : 16384);
  hipLaunchKernelGGL(relu_drop_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(grad_y), reinterpret_cast<const float4*>(y),
                     reinterpret_cast<float4*>(grad_x), n4, 1.f / (1.f - p));
  return vidar_last_error();
}

int vidar_colsum_f32(const float* x, float* out, int64_t rows, int cols, void* stream) {
  VIDAR_ENTER();
  // cols: a multiple of 4 whose float4 groups are a power of two and fit one workgroup row (4 .. 4096)
  const int groups = cols / 4;
  if (rows < 0 || cols <= 0 || cols % 4 != 0 || (groups & (groups - 1)) != 0 || groups > kColsumThreads)
    return VIDAR_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)cols, s);
  if (e != hipSuccess) return (int)e;
  if (rows == 0) return 0;
  const int64_t rows_per_pass = (int64_t)(kColsumThreads / groups) * kColsumIlp;
  const int grid = (int)min((int64_t)256, (rows + rows_per_pass - 1) / rows_per_pass);
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)grid), dim3(kColsumThreads), 0, s, x, out, rows, cols);
  return vidar_last_error();
}


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

size_t vidar_drop_add_ln_bwd_workspace_bytes(int64_t rows) {
  if (rows <= 0) return 0;
  const int64_t waves = (rows + kRowsPerWave - 1) / kRowsPerWave;
  return sizeof(float) * 2 * kC * (size_t)((waves + 3) / 4);
}

int vidar_drop_add_ln_bwd_f32(const float* grad_y, const float* sum_in, const float* gamma, const float* mean_in,
                              const float* rstd_in, float* grad_x, float* grad_residual, float* grad_gamma,
                              float* grad_beta, void* workspace, int64_t rows, int C, float p, uint32_t seed,
                              void* stream) {
  VIDAR_ENTER();
  if (rows < 0 || C != kC || p < 0.f || p >= 1.f) return VIDAR_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (rows == 0) {
    hipError_t e = hipMemsetAsync(grad_gamma, 0, sizeof(float) * kC, s);
    if (e == hipSuccess) e = hipMemsetAsync(grad_beta, 0, sizeof(float) * kC, s);
    return (int)e;
  }
  if (!workspace) return VIDAR_ERR_BAD_ARG;
  const int64_t waves = (rows + kRowsPerWave - 1) / kRowsPerWave;
  const int nparts = (int)((waves + 3) / 4);
  hipLaunchKernelGGL(drop_add_ln_bwd_kernel, dim3((unsigned)nparts), dim3(256), 0, s, grad_y, sum_in, gamma, mean_in,
                     rstd_in, grad_x, grad_residual, (float*)workspace, grad_gamma, grad_beta, rows, p, seed);
  hipLaunchKernelGGL(affine_grad_reduce_kernel, dim3(2 * kC / 32, kSlices), dim3(256), 0, s, (const float*)workspace, nparts,
                     grad_gamma, grad_beta);
  return vidar_last_error();
}

}  // extern "C"
