// Fused LatentRendering ray-march for gfx950.
//
// Replaces the PyTorch op chain of
//   projects/mmdet3d_plugin/bevformer/modules/ray_operations/latent_rendering.py:96-150
// (3x F.grid_sample over [bs,16,Q,257] + sigmoid + cumprod + masked normalisation + reductions,
//  ~5 GB of intermediates per call at Q = 200x200) with two gather kernels and their adjoints that
// keep every intermediate in registers:
//   stage 1 (":102-129")  path_prob[b,q,z] = prod_{k<G, |n_k|<|n_q|} (1 - act(occ(n_k)))  * act(occ(n_q))
//   stage 2 (":131-150")  feat[b,q,z]      = sum_k a(n_k) m_k / (sum_k m_k + eps),  m_k = path_prob(n_k) [|n_k| < bound_q]
// where n_k = 2 * rhat_q * (k + 0.5) * step are the G waypoints of the ray from the BEV centre
// through cell q (normalised [-1,1] coordinates), sampled bilinearly with zero padding and
// align_corners=False, and z runs over the 16 height bins (= the 16 LoRA channels).
//
// Layout in HBM: all maps are channel-last [bs, h*w, 16] f32 -- exactly what the producing
// nn.Linear emits -- so one bilinear corner of all 16 bins is one 64-byte segment.
// Mapping: one wave per BEV cell; lane = (k mod 16, quarter of the 16 bins as a float4); the ray is
// walked 16 waypoints per iteration and reduced over the 16 k-lanes with xor shuffles.  Cells are
// assigned to workgroups in row-major order, 4 cells per 256-thread workgroup.
// Backward kernels recompute the forward samples and scatter with fp32 hardware atomics, with the
// lane map (waypoint k mod 4, height bin) so that one atomic instruction covers a corner's 64 bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "vidar_hip.h"
#include "vidar_common.h"

namespace {

constexpr int kZ = 16;
constexpr int kThreads = 256;
constexpr int kCellsPerBlock = kThreads / 64;

// Every ray starts at the BEV centre, so the first waypoints of all H*W rays scatter onto the same few cells and the
// atomics on those addresses serialise: the backward kernels add into kCopies private copies of the gradient maps
// (workgroup i -> copy i mod n; neighbouring workgroups = neighbouring cells = nearly the same ray), kept in the
// CALLER's workspace (vidar_latent_render_bwd_workspace_bytes; without one they add straight into the outputs), and a
// small kernel sums the copies -- n times fewer atomics per hot address for the same total number.  Measured on MI355X
// (profiles/r04_staged_variants_kernel_times.log): lr_prob_bwd 0.66 -> 0.48 ms, lr_gather_bwd 1.32 -> 1.08 ms with 8
// copies, memset and sum included: the kernels were serialised on the hot addresses, not bound by the atomic rate.
constexpr int kCopies = 8;
// which private copy this workgroup adds into, as an offset in maps of [bs, Q, 16] (0 when the variant is off)
__device__ __forceinline__ size_t copy_of_block(int ncopies) { return (size_t)(blockIdx.x % ncopies) * gridDim.y; }

struct Geo {
  int H, W, G;
  float step;   // grid_step / (min(H,W)//2), rounded to f32 (latent_rendering.py:102-104)
  int act;      // 0 = sigmoid, 1 = exp
  float eps;
};

struct Cell {
  float rnx, rny;   // unit direction (nan_to_num'ed)
  float ncx, ncy;   // the cell itself in [-1,1]
  float len_c;      // |n_cell|
  float bound;      // min(1/|rnx|, 1/|rny|)
};

__device__ __forceinline__ Cell make_cell(int q, const Geo& g) {
  const int i = q / g.W, j = q % g.W;
  const float gx = (j + 0.5f) / g.W, gy = (i + 0.5f) / g.H;
  const float rx = gx - 0.5f, ry = gy - 0.5f;
  const float nrm = sqrtf(rx * rx + ry * ry);
  Cell c;
  c.rnx = rx / nrm; c.rny = ry / nrm;
  if (c.rnx != c.rnx) c.rnx = 0.f;
  if (c.rny != c.rny) c.rny = 0.f;
  c.ncx = gx * 2.f - 1.f; c.ncy = gy * 2.f - 1.f;
  c.len_c = sqrtf(c.ncx * c.ncx + c.ncy * c.ncy);
  c.bound = fminf(1.f / fabsf(c.rnx), 1.f / fabsf(c.rny));
  return c;
}

struct Tap {        // one bilinear footprint
  int o[4];         // cell index of the 4 corners or -1
  float w[4];
};

__device__ __forceinline__ Tap make_tap(float nx, float ny, const Geo& g) {
  const float ix = ((nx + 1.f) * g.W - 1.f) / 2.f;
  const float iy = ((ny + 1.f) * g.H - 1.f) / 2.f;
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
  const float ax = ix - fx, ay = iy - fy;
  Tap t;
  t.w[0] = (1.f - ax) * (1.f - ay); t.w[1] = ax * (1.f - ay);
  t.w[2] = (1.f - ax) * ay;         t.w[3] = ax * ay;
  const bool l = x0 >= 0 && x0 < g.W, r = x1 >= 0 && x1 < g.W;
  const bool u = y0 >= 0 && y0 < g.H, d = y1 >= 0 && y1 < g.H;
  t.o[0] = (l && u) ? y0 * g.W + x0 : -1;
  t.o[1] = (r && u) ? y0 * g.W + x1 : -1;
  t.o[2] = (l && d) ? y1 * g.W + x0 : -1;
  t.o[3] = (r && d) ? y1 * g.W + x1 : -1;
  return t;
}

__device__ __forceinline__ float4 tap_load(const float* __restrict__ map, const Tap& t, int zq) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (t.o[c] >= 0) {
      const float4 v = *reinterpret_cast<const float4*>(map + (size_t)t.o[c] * kZ + zq * 4);
      acc.x += t.w[c] * v.x; acc.y += t.w[c] * v.y; acc.z += t.w[c] * v.z; acc.w += t.w[c] * v.w;
    }
  }
  return acc;
}

__device__ __forceinline__ float act_f(float x, int act) {
  if (act == 0) return 1.f / (1.f + expf(-x));
  return 1.f - expf(-fmaxf(x, 0.f));
}
// d act / dx expressed with p = act(x)
__device__ __forceinline__ float act_d(float x, float p, int act) {
  if (act == 0) return p * (1.f - p);
  return x > 0.f ? (1.f - p) : 0.f;
}
__device__ __forceinline__ float4 act4(const float4& v, int act) {
  return make_float4(act_f(v.x, act), act_f(v.y, act), act_f(v.z, act), act_f(v.w, act));
}

// reduce over the 16 k-lanes (lane bits 2..5); every lane ends with the full result
__device__ __forceinline__ float4 kprod(float4 v) {
#pragma unroll
  for (int m = 4; m < 64; m <<= 1) {
    v.x *= __shfl_xor(v.x, m, 64); v.y *= __shfl_xor(v.y, m, 64);
    v.z *= __shfl_xor(v.z, m, 64); v.w *= __shfl_xor(v.w, m, 64);
  }
  return v;
}
__device__ __forceinline__ float4 ksum(float4 v) {
#pragma unroll
  for (int m = 4; m < 64; m <<= 1) {
    v.x += __shfl_xor(v.x, m, 64); v.y += __shfl_xor(v.y, m, 64);
    v.z += __shfl_xor(v.z, m, 64); v.w += __shfl_xor(v.w, m, 64);
  }
  return v;
}


// ---- backward mapping: lane = (waypoint k mod 4, height bin z): one atomic instruction covers the
// ---- 64 contiguous bytes of a tap corner (atomics cost per instruction x line, see msda.hip) ------
__device__ __forceinline__ float tap_load1(const float* __restrict__ map, const Tap& t, int z) {
  float acc = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c)
    if (t.o[c] >= 0) acc += t.w[c] * map[(size_t)t.o[c] * kZ + z];
  return acc;
}
__device__ __forceinline__ void tap_scatter1(float* __restrict__ map, const Tap& t, int z, float g) {
#pragma unroll
  for (int c = 0; c < 4; ++c)
    if (t.o[c] >= 0) unsafeAtomicAdd(map + (size_t)t.o[c] * kZ + z, t.w[c] * g);
}
__device__ __forceinline__ float kprod1(float v) {
  v *= __shfl_xor(v, 16, 64); v *= __shfl_xor(v, 32, 64);
  return v;
}

__device__ __forceinline__ void waypoint(const Cell& c, const Geo& g, int k, float& nx, float& ny,
                                         float& len) {
  const float s = (k + 0.5f) * g.step;
  const float ux = 0.5f + c.rnx * s, uy = 0.5f + c.rny * s;
  nx = ux * 2.f - 1.f; ny = uy * 2.f - 1.f;
  len = sqrtf(nx * nx + ny * ny);
}

// ---------------------------------------------------------------------------------------------
// stage 1 forward: occ [bs,Q,16] -> path_prob [bs,Q,16]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void lr_prob_fwd_kernel(const float* __restrict__ occ,
                                                               float* __restrict__ prob, int Q,
                                                               Geo g) {
  const int b = blockIdx.y;
  const int q = blockIdx.x * kCellsPerBlock + threadIdx.x / 64;
  if (q >= Q) return;
  const int lane = threadIdx.x & 63, zq = lane & 3, ks = lane >> 2;
  const float* map = occ + (size_t)b * Q * kZ;
  const Cell c = make_cell(q, g);
  float4 pr = make_float4(1.f, 1.f, 1.f, 1.f);
  for (int k = ks; k < g.G; k += 16) {
    float nx, ny, len;
    waypoint(c, g, k, nx, ny, len);
    if (len < c.len_c) {
      const float4 p = act4(tap_load(map, make_tap(nx, ny, g), zq), g.act);
      pr.x *= 1.f - p.x; pr.y *= 1.f - p.y; pr.z *= 1.f - p.z; pr.w *= 1.f - p.w;
    }
  }
  pr = kprod(pr);
  if (ks == 0) {
    const float4 pc = act4(tap_load(map, make_tap(c.ncx, c.ncy, g), zq), g.act);
    *reinterpret_cast<float4*>(prob + ((size_t)b * Q + q) * kZ + zq * 4) =
        make_float4(pr.x * pc.x, pr.y * pc.y, pr.z * pc.z, pr.w * pc.w);
  }
}

// stage 1 backward: grad_prob [bs,Q,16] -> grad_occ [bs,Q,16] (pre-zeroed, atomics)
__global__ __launch_bounds__(kThreads) void lr_prob_bwd_kernel(const float* __restrict__ occ,
                                                               const float* __restrict__ grad_prob,
                                                               float* __restrict__ grad_occ, int Q,
                                                               Geo g, int ncopies) {
  const int b = blockIdx.y;
  const int q = blockIdx.x * kCellsPerBlock + threadIdx.x / 64;
  if (q >= Q) return;
  const int lane = threadIdx.x & 63, z = lane & 15, ks = lane >> 4;
  const float* map = occ + (size_t)b * Q * kZ;
  float* gmap = grad_occ + (copy_of_block(ncopies) + b) * Q * kZ;
  const Cell c = make_cell(q, g);
  // pass 1: the transmittance product
  float pr = 1.f;
  for (int k = ks; k < g.G; k += 4) {
    float nx, ny, len;
    waypoint(c, g, k, nx, ny, len);
    if (len < c.len_c) pr *= 1.f - act_f(tap_load1(map, make_tap(nx, ny, g), z), g.act);
  }
  pr = kprod1(pr);
  const Tap tc = make_tap(c.ncx, c.ncy, g);
  const float xc = tap_load1(map, tc, z);
  const float pc = act_f(xc, g.act);
  const float go = grad_prob[((size_t)b * Q + q) * kZ + z];
  // d/d(1-p_k) of prod * pc  = prod/(1-p_k) * pc ; guarded against (1-p_k) == 0
  const float gp = go * pr * pc;
  for (int k = ks; k < g.G; k += 4) {
    float nx, ny, len;
    waypoint(c, g, k, nx, ny, len);
    if (len < c.len_c) {
      const Tap t = make_tap(nx, ny, g);
      const float x = tap_load1(map, t, z);
      const float p = act_f(x, g.act);
      const float gs = (1.f - p) > 0.f ? -gp / (1.f - p) * act_d(x, p, g.act) : 0.f;
      tap_scatter1(gmap, t, z, gs);
    }
  }
  if (ks == 0) tap_scatter1(gmap, tc, z, go * pr * act_d(xc, pc, g.act));
}

// ---------------------------------------------------------------------------------------------
// stage 2 forward: prob [bs,Q,16], a [bs,Q,16] -> feat [bs,Q,16], msum [bs,Q,16] (= sum_k m_k)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void lr_gather_fwd_kernel(const float* __restrict__ prob,
                                                                 const float* __restrict__ a,
                                                                 float* __restrict__ feat,
                                                                 float* __restrict__ msum, int Q,
                                                                 Geo g) {
  const int b = blockIdx.y;
  const int q = blockIdx.x * kCellsPerBlock + threadIdx.x / 64;
  if (q >= Q) return;
  const int lane = threadIdx.x & 63, zq = lane & 3, ks = lane >> 2;
  const float* pm = prob + (size_t)b * Q * kZ;
  const float* am = a + (size_t)b * Q * kZ;
  const Cell c = make_cell(q, g);
  float4 M = make_float4(0.f, 0.f, 0.f, 0.f), Nn = M;
  for (int k = ks; k < g.G; k += 16) {
    float nx, ny, len;
    waypoint(c, g, k, nx, ny, len);
    if (len < c.bound) {
      const Tap t = make_tap(nx, ny, g);
      const float4 m = tap_load(pm, t, zq);
      const float4 av = tap_load(am, t, zq);
      M.x += m.x; M.y += m.y; M.z += m.z; M.w += m.w;
      Nn.x += av.x * m.x; Nn.y += av.y * m.y; Nn.z += av.z * m.z; Nn.w += av.w * m.w;
    }
  }
  M = ksum(M); Nn = ksum(Nn);
  if (ks == 0) {
    const size_t o = ((size_t)b * Q + q) * kZ + zq * 4;
    *reinterpret_cast<float4*>(feat + o) = make_float4(Nn.x / (M.x + g.eps), Nn.y / (M.y + g.eps),
                                                       Nn.z / (M.z + g.eps), Nn.w / (M.w + g.eps));
    *reinterpret_cast<float4*>(msum + o) = M;
  }
}

// stage 2 backward: grad_feat -> grad_prob, grad_a (pre-zeroed, atomics)
__global__ __launch_bounds__(kThreads) void lr_gather_bwd_kernel(
    const float* __restrict__ prob, const float* __restrict__ a, const float* __restrict__ feat,
    const float* __restrict__ msum, const float* __restrict__ grad_feat,
    float* __restrict__ grad_prob, float* __restrict__ grad_a, int Q, Geo g, int ncopies) {
  const int b = blockIdx.y;
  const int q = blockIdx.x * kCellsPerBlock + threadIdx.x / 64;
  if (q >= Q) return;
  const int lane = threadIdx.x & 63, z = lane & 15, ks = lane >> 4;
  const float* pm = prob + (size_t)b * Q * kZ;
  const float* am = a + (size_t)b * Q * kZ;
  float* gpm = grad_prob + (copy_of_block(ncopies) + b) * Q * kZ;
  float* gam = grad_a + (copy_of_block(ncopies) + b) * Q * kZ;
  const Cell c = make_cell(q, g);
  const size_t o = ((size_t)b * Q + q) * kZ + z;
  const float f = feat[o];
  const float s = grad_feat[o] / (msum[o] + g.eps);
  for (int k = ks; k < g.G; k += 4) {
    float nx, ny, len;
    waypoint(c, g, k, nx, ny, len);
    if (len < c.bound) {
      const Tap t = make_tap(nx, ny, g);
      const float m = tap_load1(pm, t, z);
      const float av = tap_load1(am, t, z);
      tap_scatter1(gam, t, z, s * m);
      tap_scatter1(gpm, t, z, s * (av - f));
    }
  }
}

__global__ __launch_bounds__(256) void lr_sum_copies_kernel(const float4* __restrict__ copies, float4* __restrict__ out,
                                                            size_t n4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 a = copies[i];
  for (int c = 1; c < kCopies; ++c) {
    const float4 v = copies[(size_t)c * n4 + i];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  out[i] = a;
}
inline bool lr_bad(int bs, int H, int W, int Z, int G, int act) {
  return bs < 0 || H <= 0 || W <= 0 || Z != kZ || G <= 0 || (act != 0 && act != 1);
}
inline dim3 lr_grid(int bs, int Q) { return dim3((Q + kCellsPerBlock - 1) / kCellsPerBlock, bs); }

}  // namespace

extern "C" {

size_t vidar_latent_render_bwd_workspace_bytes(int bs, int H, int W, int Z, int maps) {
  if (bs <= 0 || H <= 0 || W <= 0 || Z <= 0 || maps < 1 || maps > 2) return 0;
  return sizeof(float) * (size_t)maps * bs * H * W * Z * kCopies;   // maps: 1 = prob_bwd (grad_occ), 2 = gather_bwd
}

int vidar_latent_render_prob_fwd_f32(const float* occ, float* path_prob, int bs, int H, int W, int Z,
                                     int grid_num, float step, int act, void* stream) {
  VIDAR_ENTER();
  if (lr_bad(bs, H, W, Z, grid_num, act)) return VIDAR_ERR_BAD_ARG;
  if (bs == 0) return 0;
  Geo g{H, W, grid_num, step, act, 0.f};
  hipLaunchKernelGGL(lr_prob_fwd_kernel, lr_grid(bs, H * W), dim3(kThreads), 0, (hipStream_t)stream,
                     occ, path_prob, H * W, g);
  return vidar_last_error();
}

int vidar_latent_render_prob_bwd_f32(const float* occ, const float* grad_path_prob, float* grad_occ,
                                     int bs, int H, int W, int Z, int grid_num, float step, int act,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  VIDAR_ENTER();
  if (lr_bad(bs, H, W, Z, grid_num, act)) return VIDAR_ERR_BAD_ARG;
  if (bs == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  Geo g{H, W, grid_num, step, act, 0.f};
  const size_t n = (size_t)bs * H * W * Z;
  const bool copies = workspace != nullptr && workspace_bytes >= sizeof(float) * n * kCopies &&
                      (((uintptr_t)workspace | (uintptr_t)grad_occ) & 15u) == 0;
  float* acc = copies ? (float*)workspace : grad_occ;
  hipError_t e = hipMemsetAsync(acc, 0, sizeof(float) * n * (copies ? kCopies : 1), s);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(lr_prob_bwd_kernel, lr_grid(bs, H * W), dim3(kThreads), 0, s, occ, grad_path_prob, acc, H * W, g,
                     copies ? kCopies : 1);
  if (copies)
    hipLaunchKernelGGL(lr_sum_copies_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s,
                       (const float4*)acc, (float4*)grad_occ, n / 4);
  return vidar_last_error();
}

int vidar_latent_render_gather_fwd_f32(const float* path_prob, const float* lora_a, float* feat,
                                       float* msum, int bs, int H, int W, int Z, int grid_num,
                                       float step, float eps, void* stream) {
  VIDAR_ENTER();
  if (lr_bad(bs, H, W, Z, grid_num, 0)) return VIDAR_ERR_BAD_ARG;
  if (bs == 0) return 0;
  Geo g{H, W, grid_num, step, 0, eps};
  hipLaunchKernelGGL(lr_gather_fwd_kernel, lr_grid(bs, H * W), dim3(kThreads), 0,
                     (hipStream_t)stream, path_prob, lora_a, feat, msum, H * W, g);
  return vidar_last_error();
}

int vidar_latent_render_gather_bwd_f32(const float* path_prob, const float* lora_a, const float* feat,
                                       const float* msum, const float* grad_feat,
                                       float* grad_path_prob, float* grad_lora_a, int bs, int H,
                                       int W, int Z, int grid_num, float step, float eps,
                                       void* workspace, size_t workspace_bytes, void* stream) {
  VIDAR_ENTER();
  if (lr_bad(bs, H, W, Z, grid_num, 0)) return VIDAR_ERR_BAD_ARG;
  if (bs == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  Geo g{H, W, grid_num, step, 0, eps};
  const size_t n = (size_t)bs * H * W * Z;
  const bool copies = workspace != nullptr && workspace_bytes >= vidar_latent_render_bwd_workspace_bytes(bs, H, W, Z, 2) &&
                      (((uintptr_t)workspace | (uintptr_t)grad_path_prob | (uintptr_t)grad_lora_a) & 15u) == 0;
  float* sp = copies ? (float*)workspace : grad_path_prob;
  float* sa = copies ? (float*)workspace + n * kCopies : grad_lora_a;
  hipError_t e;
  if (copies) {
    e = hipMemsetAsync(sp, 0, sizeof(float) * 2 * n * kCopies, s);
  } else {
    e = hipMemsetAsync(sp, 0, sizeof(float) * n, s);
    if (e == hipSuccess) e = hipMemsetAsync(sa, 0, sizeof(float) * n, s);
  }
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(lr_gather_bwd_kernel, lr_grid(bs, H * W), dim3(kThreads), 0, s, path_prob, lora_a, feat, msum,
                     grad_feat, sp, sa, H * W, g, copies ? kCopies : 1);
  if (copies) {
    const dim3 rg((unsigned)((n / 4 + 255) / 256));
    hipLaunchKernelGGL(lr_sum_copies_kernel, rg, dim3(256), 0, s, (const float4*)sp, (float4*)grad_path_prob, n / 4);
    hipLaunchKernelGGL(lr_sum_copies_kernel, rg, dim3(256), 0, s, (const float4*)sa, (float4*)grad_lora_a, n / 4);
  }
  return vidar_last_error();
}

}  // extern "C"
