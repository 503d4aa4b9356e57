// BEV-encoder bookkeeping that the reference does with ~40 small torch ops and one host sync per camera
// per layer:
//   * BEVFormerEncoder.point_sampling (bevformer/modules/encoder.py:96-156): pillar anchors -> every
//     camera through lidar2img, perspective divide (eps 1e-5), normalise by the padded image shape,
//     strict in-image test;
//   * the visible-query rebatch index of SpatialCrossAttention.forward
//     (spatial_cross_attention.py:136-152, :164-171): per camera the list of BEV queries with at least
//     one valid anchor (taken from batch item 0, like the reference), and per query the number of
//     cameras that see it (clamped to >= 1).
// One call handles ALL frames of a training step (the lidar2img matrices of the whole queue are known
// when the step starts), so the host reads the per-camera list lengths ONCE per step.
// fp32 with the reference's operation order (this file is compiled with -ffp-contract=off): the mask
// is a set of strict comparisons, so the arithmetic must not be re-associated or fused.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vidar_hip.h"
#include "vidar_common.h"

namespace {

struct Range { float lo[3], span[3]; };

// thread = (frame f, batch b, query q); loops cameras and anchors
__global__ __launch_bounds__(256) void sca_project_kernel(
    const float* __restrict__ ref_3d, const float* __restrict__ lidar2img, float* __restrict__ ref_cam,
    uint8_t* __restrict__ bev_mask, float* __restrict__ count, uint8_t* __restrict__ vis0, Range r,
    float img_h, float img_w, int F, int B, int N, int Q, int D) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)F * B * Q) return;
  const int q = (int)(i % Q), b = (int)((i / Q) % B), f = (int)(i / Q / B);
  int cams = 0;
  for (int n = 0; n < N; ++n) {
    const float* M = lidar2img + (((int64_t)f * B + b) * N + n) * 16;
    bool any = false;
    for (int d = 0; d < D; ++d) {
      const float* p = ref_3d + ((int64_t)d * Q + q) * 3;
      const float x = p[0] * r.span[0] + r.lo[0];
      const float y = p[1] * r.span[1] + r.lo[1];
      const float z = p[2] * r.span[2] + r.lo[2];
      const float cx = ((M[0] * x + M[1] * y) + M[2] * z) + M[3];
      const float cy = ((M[4] * x + M[5] * y) + M[6] * z) + M[7];
      const float cz = ((M[8] * x + M[9] * y) + M[10] * z) + M[11];
      const float eps = 1e-5f;
      const float den = fmaxf(cz, eps);
      const float u = (cx / den) / img_w, v = (cy / den) / img_h;
      const bool ok = (cz > eps) && (v > 0.0f) && (v < 1.0f) && (u < 1.0f) && (u > 0.0f);
      const int64_t o = ((((int64_t)f * N + n) * B + b) * Q + q) * D + d;
      ref_cam[2 * o] = u;
      ref_cam[2 * o + 1] = v;
      bev_mask[o] = ok ? 1 : 0;
      any |= ok;
    }
    cams += any ? 1 : 0;
    if (b == 0) vis0[((int64_t)f * N + n) * Q + q] = any ? 1 : 0;
  }
  count[i] = (float)(cams < 1 ? 1 : cams);
}

// block = (frame, camera): stable compaction of the visible queries; `valid` holds the visibility flags
// on entry and (slot < length) on exit
constexpr int kCT = 1024;
__global__ __launch_bounds__(kCT) void sca_compact_kernel(uint8_t* __restrict__ valid, int64_t* __restrict__ idx,
                                                          int32_t* __restrict__ lens, int32_t* __restrict__ slot_of,
                                                          int Q) {
  __shared__ int s_wave[kCT / 64];
  __shared__ int s_base;
  uint8_t* v = valid + (int64_t)blockIdx.x * Q;
  int64_t* out = idx + (int64_t)blockIdx.x * Q;
  int32_t* inv = slot_of + (int64_t)blockIdx.x * Q;       // query -> slot, -1 = this camera does not see it
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int q0 = 0; q0 < Q; q0 += kCT) {
    const int q = q0 + threadIdx.x;
    const bool on = q < Q && v[q] != 0;
    const unsigned long long m = __ballot(on);
    const int rank = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int before = s_base, total = 0;
    for (int w = 0; w < kCT / 64; ++w) { if (w < wave) before += s_wave[w]; total += s_wave[w]; }
    if (on) out[before + rank] = q;
    if (q < Q) inv[q] = on ? before + rank : -1;
    __syncthreads();
    if (threadIdx.x == 0) s_base += total;
    __syncthreads();
  }
  const int len = s_base;
  if (threadIdx.x == 0) lens[blockIdx.x] = len;
  for (int q = threadIdx.x; q < Q; q += kCT) {
    v[q] = q < len ? 1 : 0;
    if (q >= len) out[q] = q;          // padded slots: in-range and DISTINCT (their contributions are masked, but
                                       // torch's sort-based index backward serialises over duplicate indices:
                                       // padding with one repeated index cost 2.5 ms per SCA backward)
  }
}

// ---------------------------------------------------------------------------------------------
// SpatialCrossAttention's rebatch / scatter-back as two gathers (spatial_cross_attention.py:141-152, :164-171).
// The reference copies the visible queries of every camera into a padded [bs, cams, max_len, C] batch with
// python loops and adds the attention output back with a loop of indexed `+=`; in torch that is an advanced
// index (whose backward is a sort-based index_put) and an index_add (atomics).  With the inverse map
// slot_of[cam][query] both directions of both ops are plain row gathers:
//   rows  : dst[b,n,s,:] = valid[n,s] ? src[b, idx[n,s], :] * (count ? 1/count[b,idx] : 1) : 0
//   combine: dst[b,q,:]  = (sum over cameras n that see q of src[b, n, slot_of[n,q], :]) / (count ? count[b,q] : 1)
// (rows is the forward of the rebatch and the backward of the scatter-back; combine the other two.)
// One wave per 256-float row quarter: thread = one float4.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sca_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx,
                                                       const uint8_t* __restrict__ valid, const float* __restrict__ count,
                                                       float* __restrict__ dst, int B, int N, int S, int stride, int Q,
                                                       int C4) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (int64_t)B * N * S * C4) return;
  const int c = (int)(t % C4);
  const int64_t row = t / C4;
  const int s = (int)(row % S), n = (int)((row / S) % N), b = (int)(row / S / N);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid[(int64_t)n * stride + s]) {
    const int64_t q = idx[(int64_t)n * stride + s];
    v = reinterpret_cast<const float4*>(src)[((int64_t)b * Q + q) * C4 + c];
    if (count != nullptr) {
      const float k = count[(int64_t)b * Q + q];
      v.x /= k; v.y /= k; v.z /= k; v.w /= k;
    }
  }
  reinterpret_cast<float4*>(dst)[t] = v;
}

__global__ __launch_bounds__(256) void sca_combine_kernel(const float* __restrict__ src, const int32_t* __restrict__ slot_of,
                                                          const float* __restrict__ count, float* __restrict__ dst,
                                                          int B, int N, int S, int Q, int C4) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (int64_t)B * Q * C4) return;
  const int c = (int)(t % C4);
  const int64_t row = t / C4;
  const int q = (int)(row % Q), b = (int)(row / Q);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int n = 0; n < N; ++n) {
    const int s = slot_of[(int64_t)n * Q + q];
    if (s < 0 || s >= S) continue;
    const float4 v = reinterpret_cast<const float4*>(src)[(((int64_t)b * N + n) * S + s) * C4 + c];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  if (count != nullptr) {
    const float k = count[(int64_t)b * Q + q];
    a.x /= k; a.y /= k; a.z /= k; a.w /= k;
  }
  reinterpret_cast<float4*>(dst)[t] = a;
}

}  // namespace

extern "C" {

int vidar_sca_rows_f32(const float* src, const int64_t* idx, const uint8_t* valid, const float* count, float* dst,
                       int B, int N, int S, int stride, int Q, int C, void* stream) {
  VIDAR_ENTER();
  if (B < 0 || N <= 0 || S < 0 || Q <= 0 || C <= 0 || C % 4 != 0 || stride < S) return VIDAR_ERR_BAD_ARG;
  const int64_t n = (int64_t)B * N * S * (C / 4);
  if (n == 0) return 0;
  hipLaunchKernelGGL(sca_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, idx,
                     valid, count, dst, B, N, S, stride, Q, C / 4);
  return vidar_last_error();
}

int vidar_sca_combine_f32(const float* src, const int32_t* slot_of, const float* count, float* dst, int B, int N,
                          int S, int Q, int C, void* stream) {
  VIDAR_ENTER();
  if (B < 0 || N <= 0 || S < 0 || Q <= 0 || C <= 0 || C % 4 != 0) return VIDAR_ERR_BAD_ARG;
  const int64_t n = (int64_t)B * Q * (C / 4);
  if (n == 0) return 0;
  hipLaunchKernelGGL(sca_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                     slot_of, count, dst, B, N, S, Q, C / 4);
  return vidar_last_error();
}

int vidar_sca_plan_f32(const float* ref_3d, const float* lidar2img, float* ref_cam, uint8_t* bev_mask,
                       float* count, int64_t* idx, uint8_t* valid, int32_t* lens, int32_t* slot_of,
                       const float* pc_range, float img_h, float img_w, int F, int B, int N, int Q, int D,
                       void* stream) {
  VIDAR_ENTER();
  if (F < 0 || B <= 0 || N <= 0 || Q <= 0 || D <= 0 || !pc_range) return VIDAR_ERR_BAD_ARG;
  if (F == 0) return 0;
  Range r;
  for (int a = 0; a < 3; ++a) { r.lo[a] = pc_range[a]; r.span[a] = pc_range[a + 3] - pc_range[a]; }
  const int64_t n = (int64_t)F * B * Q;
  hipLaunchKernelGGL(sca_project_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     ref_3d, lidar2img, ref_cam, bev_mask, count, valid, r, img_h, img_w, F, B, N, Q, D);
  hipLaunchKernelGGL(sca_compact_kernel, dim3(F * N), dim3(kCT), 0, (hipStream_t)stream, valid, idx, lens, slot_of, Q);
  return vidar_last_error();
}

}  // extern "C"
