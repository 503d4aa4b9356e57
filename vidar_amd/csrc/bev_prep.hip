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
                                                          int32_t* __restrict__ lens, int Q) {
  __shared__ int s_wave[kCT / 64];
  __shared__ int s_base;
  uint8_t* v = valid + (int64_t)blockIdx.x * Q;
  int64_t* out = idx + (int64_t)blockIdx.x * Q;
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
    __syncthreads();
    if (threadIdx.x == 0) s_base += total;
    __syncthreads();
  }
  const int len = s_base;
  if (threadIdx.x == 0) lens[blockIdx.x] = len;
  for (int q = threadIdx.x; q < Q; q += kCT) {
    v[q] = q < len ? 1 : 0;
    if (q >= len) out[q] = 0;          // padded slots: any in-range query (their contributions are masked)
  }
}

}  // namespace

extern "C" {

int vidar_sca_plan_f32(const float* ref_3d, const float* lidar2img, float* ref_cam, uint8_t* bev_mask,
                       float* count, int64_t* idx, uint8_t* valid, int32_t* lens, const float* pc_range,
                       float img_h, float img_w, int F, int B, int N, int Q, int D, void* stream) {
  VIDAR_ENTER();
  if (F < 0 || B <= 0 || N <= 0 || Q <= 0 || D <= 0 || !pc_range) return VIDAR_ERR_BAD_ARG;
  if (F == 0) return 0;
  Range r;
  for (int a = 0; a < 3; ++a) { r.lo[a] = pc_range[a]; r.span[a] = pc_range[a + 3] - pc_range[a]; }
  const int64_t n = (int64_t)F * B * Q;
  hipLaunchKernelGGL(sca_project_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     ref_3d, lidar2img, ref_cam, bev_mask, count, valid, r, img_h, img_w, F, B, N, Q, D);
  hipLaunchKernelGGL(sca_compact_kernel, dim3(F * N), dim3(kCT), 0, (hipStream_t)stream, valid, idx, lens, Q);
  return vidar_last_error();
}

}  // extern "C"
