// Fused occupancy-volume ray-march kernels of the ViDAR head for gfx950.
//
// Replace the PyTorch op chains of projects/mmdet3d_plugin/bevformer/dense_heads/vidar_head_base.py:
//   ray_ce      : _get_grid_features (:420-509) + F.cross_entropy(label 0) (:586-592)
//   ray_gumbel  : _get_grid_features on the dense voxel rays (:594-630) +
//                 _custom_gumbel_softmax_distance (:754-773)
//   ray_argmax  : test-time decode in get_point_cloud_prediction (:697-731)
// The reference materialises ~8 arrays of [rays, 513] floats per call (waypoints, lengths, masks,
// sampled logits, -inf masks, softmax); here one wave owns one ray, its 512 waypoints live in
// registers (8 per lane) and only O(1) values per ray ever reach HBM.
//
// Geometry (voxel units, fp32, same operation order as the reference):
//   rhat = (p - o)/|p - o| ;  s_k = o + rhat * (k + 0.5) * step, k = 0..K-1 (K = ray_grid_num = 512)
//   normalised g = s / (X,Y,Z) * 2 - 1 ; a waypoint is masked (-inf) iff any g <= -1 or g >= 1
//   trilinear sample of sigma[Z,Y,X] at pixel ((g+1)*size-1)/2, zero padding, align_corners=False.
// sigma layout: [F, Z, Y, X] f32 (x fastest) -- the head's own [bs,F,16,200,200] view.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "vidar_hip.h"
#include "vidar_common.h"

namespace {

constexpr int kWave = 64;
constexpr int kPerLane = 8;           // K = 512 waypoints
constexpr int kK = kWave * kPerLane;
constexpr int kThreads = 256;
constexpr int kRaysPerBlock = kThreads / kWave;

struct VolDims { int F, Z, Y, X; };

struct Ray {
  float ox, oy, oz, dx, dy, dz;   // origin, unit direction
  float px, py, pz;               // target point
  int f;                          // frame slot, -1 = skip
};

__device__ __forceinline__ Ray load_ray(const float* __restrict__ origin,
                                        const float* __restrict__ pts,
                                        const float* __restrict__ tindex, int r, const VolDims& v) {
  Ray ray;
  const float t = tindex[r];
  ray.f = (t >= 0.f && t < (float)v.F) ? (int)t : -1;   // NaN / -1 padding -> skip
  const int f = ray.f < 0 ? 0 : ray.f;
  ray.ox = origin[f * 3 + 0]; ray.oy = origin[f * 3 + 1]; ray.oz = origin[f * 3 + 2];
  ray.px = pts[(size_t)r * 3 + 0]; ray.py = pts[(size_t)r * 3 + 1]; ray.pz = pts[(size_t)r * 3 + 2];
  const float rx = ray.px - ray.ox, ry = ray.py - ray.oy, rz = ray.pz - ray.oz;
  const float n = sqrtf(rx * rx + ry * ry + rz * rz);
  ray.dx = rx / n; ray.dy = ry / n; ray.dz = rz / n;
  return ray;
}

struct Tri {
  int o[8];      // linear voxel offsets or -1
  float w[8];
  bool masked;   // waypoint outside the open volume -> logit -inf
};

template <bool USE_MASK = true>
__device__ __forceinline__ Tri make_tri(float sx, float sy, float sz, const VolDims& v) {
  Tri t;
  const float gx = sx / v.X * 2.f - 1.f, gy = sy / v.Y * 2.f - 1.f, gz = sz / v.Z * 2.f - 1.f;
  t.masked = USE_MASK && ((gx <= -1.f) || (gx >= 1.f) || (gy <= -1.f) || (gy >= 1.f) ||
                          (gz <= -1.f) || (gz >= 1.f) || (gx != gx) || (gy != gy) || (gz != gz));
  const float ix = ((gx + 1.f) * v.X - 1.f) / 2.f;
  const float iy = ((gy + 1.f) * v.Y - 1.f) / 2.f;
  const float iz = ((gz + 1.f) * v.Z - 1.f) / 2.f;
  const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
  const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
  const float ax = ix - fx, ay = iy - fy, az = iz - fz;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int cx = c & 1, cy = (c >> 1) & 1, cz = c >> 2;
    const int x = x0 + cx, y = y0 + cy, z = z0 + cz;
    const bool ok = !t.masked && x >= 0 && x < v.X && y >= 0 && y < v.Y && z >= 0 && z < v.Z;
    t.o[c] = ok ? (z * v.Y + y) * v.X + x : -1;
    t.w[c] = (cx ? ax : 1.f - ax) * (cy ? ay : 1.f - ay) * (cz ? az : 1.f - az);
  }
  return t;
}

__device__ __forceinline__ float tri_load(const float* __restrict__ vol, const Tri& t) {
  float acc = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c)
    if (t.o[c] >= 0) acc += t.w[c] * vol[t.o[c]];
  return acc;
}
__device__ __forceinline__ void tri_scatter(float* __restrict__ gvol, const Tri& t, float g) {
  if (g == 0.f) return;
#pragma unroll
  for (int c = 0; c < 8; ++c)
    if (t.o[c] >= 0) unsafeAtomicAdd(gvol + t.o[c], t.w[c] * g);
}
// the four corners with x-corner `cx` only.  fp32 global atomics cost per (instruction x 128-byte line touched)
// (tools/micro/atomic_bench.hip): the backward kernels pair adjacent lanes on ONE waypoint -- even lane x0, odd lane
// x0 + 1, neighbouring addresses of the x-fastest volume -- so an atomic instruction of 32 waypoints touches the lines
// of 32 corners instead of 64: half the requests of a lane-per-waypoint scatter for the same 8 adds per waypoint.
__device__ __forceinline__ void tri_scatter_x(float* __restrict__ gvol, const Tri& t, float g, int cx) {
  if (g == 0.f) return;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int o = cx ? t.o[2 * c + 1] : t.o[2 * c];
    const float w = cx ? t.w[2 * c + 1] : t.w[2 * c];
    if (o >= 0) unsafeAtomicAdd(gvol + o, w * g);
  }
}

__device__ __forceinline__ void waypoint(const Ray& r, int k, float step, float& sx, float& sy,
                                         float& sz) {
  const float d = (k + 0.5f) * step;
  sx = r.ox + r.dx * d; sy = r.oy + r.dy * d; sz = r.oz + r.dz * d;
}
__device__ __forceinline__ float dist_to(const Ray& r, float sx, float sy, float sz) {
  const float ex = sx - r.ox, ey = sy - r.oy, ez = sz - r.oz;
  return sqrtf(ex * ex + ey * ey + ez * ez);
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, kWave));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
  return v;
}

constexpr float kNegInf = -__builtin_inff();

// Every ray of a frame starts at the sensor origin, so the first waypoints of all rays of a frame scatter onto the same
// 8-27 voxels (210 000 rays x ~3 waypoints x 8 corners onto ~190 addresses in one ray_ce_bwd launch) and the atomics on
// those addresses serialise: the backward kernels add into kRayCopies private copies of the gradient volume
// (workgroup i -> copy i mod n) kept in the CALLER's workspace (vidar_ray_bwd_workspace_bytes; without one they add
// straight into grad_sigma) and a small kernel sums them.  Measured on MI355X
// (profiles/r04_staged_variants_kernel_times.log): ray_ce_bwd 0.69 -> 0.41 ms, ray_gumbel_bwd 0.41 -> 0.29 ms with 8
// copies, memset and sum included.  (Leaving the 512-waypoint loops after the run of live waypoints -- the waypoints
// inside the volume are ONE run of consecutive k, tests/test_ray_early_exit_cpu.py -- was measured too: no change,
// the masked passes cost almost nothing next to the atomics; removed.)
constexpr int kRayCopies = 8;


// logits of the K waypoints owned by this lane
__device__ __forceinline__ void lane_logits(const float* __restrict__ vol, const Ray& r,
                                            const VolDims& v, float step, int lane,
                                            float (&f)[kPerLane]) {
#pragma unroll
  for (int j = 0; j < kPerLane; ++j) {
    float sx, sy, sz;
    waypoint(r, lane + j * kWave, step, sx, sy, sz);
    const Tri t = make_tri(sx, sy, sz, v);
    f[j] = t.masked ? kNegInf : tri_load(vol, t);
  }
}

// ---------------------------------------------------------------------------------------------
// GT-ray march + cross entropy on the end-point sample
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void ray_ce_fwd_kernel(
    const float* __restrict__ sigma, const float* __restrict__ origin, const float* __restrict__ gt,
    const float* __restrict__ tindex, float* __restrict__ ce, float* __restrict__ lse_out,
    float* __restrict__ valid, int R, VolDims v, float step) {
  const int r = blockIdx.x * kRaysPerBlock + threadIdx.x / kWave;
  if (r >= R) return;
  const int lane = threadIdx.x % kWave;
  const Ray ray = load_ray(origin, gt, tindex, r, v);
  const Tri t0 = make_tri(ray.px, ray.py, ray.pz, v);
  const bool ok = ray.f >= 0 && !t0.masked;      // rays whose end point leaves the volume are dropped (:464-467)
  float out_ce = 0.f, out_lse = 0.f;
  if (ok) {
    const float* vol = sigma + (size_t)ray.f * v.Z * v.Y * v.X;
    const float f0 = tri_load(vol, t0);
    float f[kPerLane];
    lane_logits(vol, ray, v, step, lane, f);
    float m = f0;
#pragma unroll
    for (int j = 0; j < kPerLane; ++j) m = fmaxf(m, f[j]);
    m = wave_max(m);
    float s = (lane == 0) ? expf(f0 - m) : 0.f;
#pragma unroll
    for (int j = 0; j < kPerLane; ++j) s += expf(f[j] - m);   // exp(-inf) == 0
    s = wave_sum(s);
    out_lse = m + logf(s);
    out_ce = out_lse - f0;
  }
  if (lane == 0) {
    ce[r] = out_ce; lse_out[r] = out_lse; valid[r] = ok ? 1.f : 0.f;
  }
}

__global__ __launch_bounds__(kThreads) void ray_ce_bwd_kernel(
    const float* __restrict__ sigma, const float* __restrict__ origin, const float* __restrict__ gt,
    const float* __restrict__ tindex, const float* __restrict__ lse_in,
    const float* __restrict__ grad_ce, float* __restrict__ grad_sigma, int R, VolDims v,
    float step, int ncopies) {
  const int r = blockIdx.x * kRaysPerBlock + threadIdx.x / kWave;
  if (r >= R) return;
  const int lane = threadIdx.x % kWave;
  const float g = grad_ce[r];
  if (g == 0.f) return;
  const Ray ray = load_ray(origin, gt, tindex, r, v);
  const Tri t0 = make_tri(ray.px, ray.py, ray.pz, v);
  if (ray.f < 0 || t0.masked) return;
  const size_t slice = (size_t)ray.f * v.Z * v.Y * v.X;
  const float* vol = sigma + slice;
  float* gvol = grad_sigma + (size_t)(blockIdx.x % ncopies) * v.F * v.Z * v.Y * v.X + slice;
  const float lse = lse_in[r];
  if (lane == 0) tri_scatter(gvol, t0, g * (expf(tri_load(vol, t0) - lse) - 1.f));
  const int cx = lane & 1;
  for (int j = 0; j < 2 * kPerLane; ++j) {             // 32 waypoints per pass, a lane pair per waypoint
    float sx, sy, sz;
    waypoint(ray, (lane >> 1) + j * (kWave / 2), step, sx, sy, sz);
    const Tri t = make_tri(sx, sy, sz, v);
    if (!t.masked) tri_scatter_x(gvol, t, g * expf(tri_load(vol, t) - lse), cx);
  }
}

// ---------------------------------------------------------------------------------------------
// dense rays: hard gumbel sample of the hit waypoint + straight-through "mass beyond" factor
// aux[r] = {pred_dist, prob_next, lse}
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void ray_gumbel_fwd_kernel(
    const float* __restrict__ sigma, const float* __restrict__ origin, const float* __restrict__ pts,
    const float* __restrict__ tindex, const float* __restrict__ noise, float* __restrict__ dist,
    float* __restrict__ aux, int R, VolDims v, float step) {
  const int r = blockIdx.x * kRaysPerBlock + threadIdx.x / kWave;
  if (r >= R) return;
  const int lane = threadIdx.x % kWave;
  const Ray ray = load_ray(origin, pts, tindex, r, v);
  float o_dist = 0.f, o_pd = 0.f, o_pn = 0.f, o_lse = 0.f;
  if (ray.f >= 0) {
    const float* vol = sigma + (size_t)ray.f * v.Z * v.Y * v.X;
    float f[kPerLane], len[kPerLane];
    lane_logits(vol, ray, v, step, lane, f);
    // arg-max of logits + gumbel noise (first index wins ties, like torch.max)
    float best = kNegInf; int bi = kK; float blen = 0.f;
    float m = kNegInf;
#pragma unroll
    for (int j = 0; j < kPerLane; ++j) {
      const int k = lane + j * kWave;
      float sx, sy, sz;
      waypoint(ray, k, step, sx, sy, sz);
      len[j] = dist_to(ray, sx, sy, sz);
      const float z = f[j] + noise[(size_t)r * kK + k];
      if (z > best || (z == best && k < bi)) { best = z; bi = k; blen = len[j]; }
      m = fmaxf(m, f[j]);
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
      const float ob = __shfl_xor(best, s, kWave);
      const int oi = __shfl_xor(bi, s, kWave);
      const float ol = __shfl_xor(blen, s, kWave);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; blen = ol; }
    }
    m = wave_max(m);
    const float pd = blen;
    float se = 0.f, sn = 0.f;
#pragma unroll
    for (int j = 0; j < kPerLane; ++j) {
      const float e = expf(f[j] - m);
      se += e;
      sn += (len[j] > pd) ? e : 0.f;
    }
    se = wave_sum(se); sn = wave_sum(sn);
    const float pn = sn / se;
    o_pd = pd; o_pn = pn; o_lse = m + logf(se);
    o_dist = ((1.f - pn) + pn) * pd;
  }
  if (lane == 0) {
    dist[r] = o_dist;
    aux[(size_t)r * 3 + 0] = o_pd; aux[(size_t)r * 3 + 1] = o_pn; aux[(size_t)r * 3 + 2] = o_lse;
  }
}

__global__ __launch_bounds__(kThreads) void ray_gumbel_bwd_kernel(
    const float* __restrict__ sigma, const float* __restrict__ origin, const float* __restrict__ pts,
    const float* __restrict__ tindex, const float* __restrict__ aux,
    const float* __restrict__ grad_dist, float* __restrict__ grad_sigma, int R, VolDims v,
    float step, int ncopies) {
  const int r = blockIdx.x * kRaysPerBlock + threadIdx.x / kWave;
  if (r >= R) return;
  const int lane = threadIdx.x % kWave;
  const float g = grad_dist[r];
  if (g == 0.f) return;
  const Ray ray = load_ray(origin, pts, tindex, r, v);
  if (ray.f < 0) return;
  const size_t slice = (size_t)ray.f * v.Z * v.Y * v.X;
  const float* vol = sigma + slice;
  float* gvol = grad_sigma + (size_t)(blockIdx.x % ncopies) * v.F * v.Z * v.Y * v.X + slice;
  const float pd = aux[(size_t)r * 3 + 0], pn = aux[(size_t)r * 3 + 1], lse = aux[(size_t)r * 3 + 2];
  const int cx = lane & 1;
  for (int j = 0; j < 2 * kPerLane; ++j) {             // 32 waypoints per pass, a lane pair per waypoint
    float sx, sy, sz;
    waypoint(ray, (lane >> 1) + j * (kWave / 2), step, sx, sy, sz);
    const Tri t = make_tri(sx, sy, sz, v);
    if (t.masked) continue;
    const float p = expf(tri_load(vol, t) - lse);
    const float ind = dist_to(ray, sx, sy, sz) > pd ? 1.f : 0.f;
    tri_scatter_x(gvol, t, g * pd * p * (ind - pn), cx);
  }
}

// test-time decode: exact zeros are masked to -inf (:728), arg-max waypoint -> distance
__global__ __launch_bounds__(kThreads) void ray_argmax_kernel(
    const float* __restrict__ sigma, const float* __restrict__ origin, const float* __restrict__ pts,
    const float* __restrict__ tindex, float* __restrict__ pred_dist, float* __restrict__ gt_dist,
    int R, VolDims v, float step) {
  const int r = blockIdx.x * kRaysPerBlock + threadIdx.x / kWave;
  if (r >= R) return;
  const int lane = threadIdx.x % kWave;
  const Ray ray = load_ray(origin, pts, tindex, r, v);
  float o_pred = 0.f, o_gt = 0.f;
  if (ray.f >= 0) {
    const float* vol = sigma + (size_t)ray.f * v.Z * v.Y * v.X;
    float best = kNegInf; int bi = kK; float blen = 0.f;
#pragma unroll
    for (int j = 0; j < kPerLane; ++j) {
      const int k = lane + j * kWave;
      float sx, sy, sz;
      waypoint(ray, k, step, sx, sy, sz);
      // the reference samples with plain zero padding here (no open-volume mask): outside -> 0 -> -inf
      const Tri t = make_tri<false>(sx, sy, sz, v);
      const float val = (sx == sx) ? tri_load(vol, t) : 0.f;
      const float z = (val == 0.f) ? kNegInf : val;
      const float l = dist_to(ray, sx, sy, sz);
      if (z > best || (z == best && k < bi)) { best = z; bi = k; blen = l; }
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
      const float ob = __shfl_xor(best, s, kWave);
      const int oi = __shfl_xor(bi, s, kWave);
      const float ol = __shfl_xor(blen, s, kWave);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; blen = ol; }
    }
    o_pred = blen;
    o_gt = dist_to(ray, ray.px, ray.py, ray.pz);
  }
  if (lane == 0) { pred_dist[r] = o_pred; gt_dist[r] = o_gt; }
}

inline bool rm_bad(int F, int R, int Z, int Y, int X, int K) {
  return F <= 0 || R < 0 || Z <= 0 || Y <= 0 || X <= 0 || K != kK;
}
inline dim3 rm_grid(int R) { return dim3((R + kRaysPerBlock - 1) / kRaysPerBlock); }

}  // namespace

namespace {
__global__ __launch_bounds__(256) void ray_sum_copies_kernel(const float* __restrict__ copies, float* __restrict__ out,
                                                             size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float a = copies[i];
  for (int c = 1; c < kRayCopies; ++c) a += copies[(size_t)c * n + i];
  out[i] = a;
}
}  // namespace

extern "C" {

size_t vidar_ray_bwd_workspace_bytes(int F, int Z, int Y, int X) {
  if (F <= 0 || Z <= 0 || Y <= 0 || X <= 0) return 0;
  return sizeof(float) * (size_t)F * Z * Y * X * kRayCopies;
}

int vidar_ray_ce_fwd_f32(const float* sigma, const float* origin, const float* gt_pts,
                         const float* tindex, float* ce, float* lse, float* valid, int F, int R,
                         int Z, int Y, int X, int K, float step, void* stream) {
  VIDAR_ENTER();
  if (rm_bad(F, R, Z, Y, X, K)) return VIDAR_ERR_BAD_ARG;
  if (R == 0) return 0;
  VolDims v{F, Z, Y, X};
  hipLaunchKernelGGL(ray_ce_fwd_kernel, rm_grid(R), dim3(kThreads), 0, (hipStream_t)stream, sigma,
                     origin, gt_pts, tindex, ce, lse, valid, R, v, step);
  return vidar_last_error();
}

int vidar_ray_ce_bwd_f32(const float* sigma, const float* origin, const float* gt_pts,
                         const float* tindex, const float* lse, const float* grad_ce,
                         float* grad_sigma, int F, int R, int Z, int Y, int X, int K, float step,
                         void* workspace, size_t workspace_bytes, void* stream) {
  VIDAR_ENTER();
  if (rm_bad(F, R, Z, Y, X, K)) return VIDAR_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  const size_t n = (size_t)F * Z * Y * X;
  const bool copies = workspace != nullptr && workspace_bytes >= vidar_ray_bwd_workspace_bytes(F, Z, Y, X);
  float* acc = copies ? (float*)workspace : grad_sigma;
  hipError_t e = hipMemsetAsync(acc, 0, sizeof(float) * n * (copies ? kRayCopies : 1), s);
  if (e != hipSuccess) return (int)e;
  if (R == 0) return copies ? (int)hipMemsetAsync(grad_sigma, 0, sizeof(float) * n, s) : 0;
  VolDims v{F, Z, Y, X};
  hipLaunchKernelGGL(ray_ce_bwd_kernel, rm_grid(R), dim3(kThreads), 0, s, sigma, origin, gt_pts, tindex, lse, grad_ce,
                     acc, R, v, step, copies ? kRayCopies : 1);
  if (copies)
    hipLaunchKernelGGL(ray_sum_copies_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, acc, grad_sigma, n);
  return vidar_last_error();
}

int vidar_ray_gumbel_fwd_f32(const float* sigma, const float* origin, const float* pts,
                             const float* tindex, const float* noise, float* dist, float* aux, int F,
                             int R, int Z, int Y, int X, int K, float step, void* stream) {
  VIDAR_ENTER();
  if (rm_bad(F, R, Z, Y, X, K)) return VIDAR_ERR_BAD_ARG;
  if (R == 0) return 0;
  VolDims v{F, Z, Y, X};
  hipLaunchKernelGGL(ray_gumbel_fwd_kernel, rm_grid(R), dim3(kThreads), 0, (hipStream_t)stream,
                     sigma, origin, pts, tindex, noise, dist, aux, R, v, step);
  return vidar_last_error();
}

int vidar_ray_gumbel_bwd_f32(const float* sigma, const float* origin, const float* pts,
                             const float* tindex, const float* aux, const float* grad_dist,
                             float* grad_sigma, int F, int R, int Z, int Y, int X, int K, float step,
                             void* workspace, size_t workspace_bytes, void* stream) {
  VIDAR_ENTER();
  if (rm_bad(F, R, Z, Y, X, K)) return VIDAR_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  const size_t n = (size_t)F * Z * Y * X;
  const bool copies = workspace != nullptr && workspace_bytes >= vidar_ray_bwd_workspace_bytes(F, Z, Y, X);
  float* acc = copies ? (float*)workspace : grad_sigma;
  hipError_t e = hipMemsetAsync(acc, 0, sizeof(float) * n * (copies ? kRayCopies : 1), s);
  if (e != hipSuccess) return (int)e;
  if (R == 0) return copies ? (int)hipMemsetAsync(grad_sigma, 0, sizeof(float) * n, s) : 0;
  VolDims v{F, Z, Y, X};
  hipLaunchKernelGGL(ray_gumbel_bwd_kernel, rm_grid(R), dim3(kThreads), 0, s, sigma, origin, pts, tindex, aux,
                     grad_dist, acc, R, v, step, copies ? kRayCopies : 1);
  if (copies)
    hipLaunchKernelGGL(ray_sum_copies_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, acc, grad_sigma, n);
  return vidar_last_error();
}

int vidar_ray_argmax_f32(const float* sigma, const float* origin, const float* pts,
                         const float* tindex, float* pred_dist, float* gt_dist, int F, int R, int Z,
                         int Y, int X, int K, float step, void* stream) {
  VIDAR_ENTER();
  if (rm_bad(F, R, Z, Y, X, K)) return VIDAR_ERR_BAD_ARG;
  if (R == 0) return 0;
  VolDims v{F, Z, Y, X};
  hipLaunchKernelGGL(ray_argmax_kernel, rm_grid(R), dim3(kThreads), 0, (hipStream_t)stream, sigma,
                     origin, pts, tindex, pred_dist, gt_dist, R, v, step);
  return vidar_last_error();
}

}  // extern "C"
