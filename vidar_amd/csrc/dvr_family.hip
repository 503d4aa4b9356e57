// Differentiable voxel rendering kernels (dvr / dvxlr / dvxlr_v2) for gfx950.
//
// Replaces (behaviour, not code) the reference CUDA extensions
//   third_lib/dvr/dvr.cu          : init :14-63, render_forward :65-383, render :385-694
//   third_lib/dvxlr/dvxlr.cu      : get_grad_sigma :63-156, render :160-517, init :528-561
//   third_lib/dvxlr/dvxlr_v2.cu   : get_grad_sigma_v2 :12-115, render_v2 :119-493
//
// Design (MI355X-first, not a translation):
//  * The reference keeps 5-6 per-thread arrays of MAX_D doubles (53-75 KB of
//    scratch per thread).  Here a ray keeps O(1) state: the march is an
//    *online* recurrence.  With T_k = exp(-csd_k) and W_k = T_k (d_{k+1}-d_k):
//        pred_dist      = d_0 + sum_k W_k                (summation by parts)
//        dd_dsigma[i]   = -dt_i * sum_{k>=i} W_k         (suffix sum)
//    dvr.render re-marches the ray once the total S is known and scatters
//    dl_dd * dt_i * (P_i - S); dvxlr.render marches ONCE, parks (W, dt) in its own
//    output rows and finishes them with a coalesced wave-scan pass.  No per-ray
//    arrays, no scratch, registers only.
//  * The "consecutive duplicate voxel" merge of dvxlr (dvxlr.cu:366-373) is a
//    one-slot pending sample that is either widened or committed.
//  * The call owns the padding of the API-mandated [N,M,MAX_D(,3)] rows (one
//    full-rate device fill ahead of the march kernel, which then writes only
//    the live prefixes), so callers pass torch.empty() buffers.
//  * All traversal decisions are IEEE fp64 in the reference's operation order;
//    this file must be compiled with -ffp-contract=off (see build.py).
//  * Small launches (at most one wave per SIMD) use one wave (64 rays) per workgroup so that M=30k
//    rays spread over all CUs.  Larger launches are issue bound (~200 instructions per step and
//    a wave runs as long as its longest ray), so 256-ray workgroups first rank their rays by an
//    estimate of the step count and hand each wave one quartile.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

#include "vidar_hip.h"
#include "vidar_common.h"
#include "dvr_march.h"
#include "dvr_par_kernels.h"

namespace {

using namespace vidar_march;

constexpr int kWave = 64;
constexpr int kSortBlock = 256;  // rays ranked together when a launch has more than one wave per SIMD
constexpr int kSimds = 1024;     // 256 CUs x 4

// Which ray does this thread walk?  kBlock == 64: its own.  kBlock > 64: the block's rays are
// ranked by estimate_steps() and wave w takes quartile (w + blockIdx.x) mod #waves (rotated so
// that no SIMD always receives the longest rays).  Returns a ray index that may be >= M.
template <int kBlock>
__device__ __forceinline__ int pick_ray(const float* __restrict__ origin,
                                        const float* __restrict__ points,
                                        const float* __restrict__ tindex, int n, int M, const Vol& g) {
  const int tid = threadIdx.x;
  const int c0 = blockIdx.x * kBlock;
  if (kBlock == kWave) return c0 + tid;
  __shared__ unsigned skey[kBlock];
  __shared__ unsigned short sperm[kBlock];
  int key = 0;
  if (c0 + tid < M) key = estimate_steps(load_ray(origin, points, tindex, n, c0 + tid, M, g), g);
  const unsigned mine = ((unsigned)key << 10) | (unsigned)tid;
  skey[tid] = mine;
  __syncthreads();
  int rank = 0;
#pragma unroll 8
  for (int j = 0; j < kBlock; ++j) rank += (skey[j] < mine) ? 1 : 0;
  sperm[rank] = (unsigned short)tid;
  __syncthreads();
  constexpr int kWaves = kBlock / kWave;
  const int q = (tid / kWave + (int)blockIdx.x) % kWaves;
  return c0 + sperm[q * kWave + tid % kWave];
}

int g_sort_min_waves = kSimds;
int g_dvxlr_pad_mode = 1;   // 0: the finish pass pads the rows; 1: device fill first, finish pass only fixes the
                            // live prefixes.  (A third variant forked the fills of the rows the march never touches
                            // onto a side stream: 0.749 vs 0.727 ms at 150 k rays, 0.253 vs 0.251 ms at 30 k --
                            // the fills already saturate HBM, overlap buys nothing; removed.)

inline bool sort_rays(int N, int M) {
  return (long)N * ((M + kWave - 1) / kWave) > (long)g_sort_min_waves;
}

// Which traversal a launch uses: -1 (default) = the step-parallel kernels (dvr_par_kernels.h) while the launch has at
// most kParAutoMaxRays rays -- there the lane-per-ray kernels leave most SIMDs empty and last as long as their longest
// ray's serial chain -- and the lane-per-ray kernels above (they are the throughput-efficient form once every SIMD
// holds several waves); 0 = always lane-per-ray, 1 = always step-parallel.
int g_traversal = -1;
// persistent padding workgroups of the one-launch dvxlr.render / render_v2: enough of them to finish the rows' padding
// when the compute workgroups finish the march (one streams ~40-50 GB/s; sweeps 16 ... 1024 in
// profiles/r05_kbench_dvr_traversal.log: fewer leave the call waiting for the padding, more take wave slots from the march)
constexpr int kParPadWgs = 80, kParPadWgsV2 = 128;
inline bool step_parallel(int N, int M, const Vol& g, long auto_max_rays) {
  if (g.X > 32767 || g.Y > 32767 || g.Z > 32767) return false;      // staged voxel coordinates are int16
  if (g_traversal >= 0) return g_traversal == 1;
  return (long)N * M <= auto_max_rays;
}
// measured crossovers (profiles/r05_kbench_dvr_traversal.log): render_forward stores nothing, so its step-parallel form
// only wins while the lane-per-ray launch is far from filling the chip; dvr.render (coalesced lane-per-step atomics)
// wins at every measured size; the one-launch dvxlr.render / render_v2 (padding streamed next to the march) win up to
// ~90 000 rays and tie with the three-launch lane-per-ray form above (box-to-box spread larger than the difference)
constexpr long kParAutoForward = 24576, kParAutoRender = 1L << 40, kParAutoDvxlr = 98304;
template <int KIND>
inline void launch_par(const float* sigma, const float* sigma_regul, const float* origin, const float* points,
                       const float* tindex, float* pred_dist, float* gt_dist, float* dd_dsigma, float* indices,
                       float* ray_pred, float* indicator, float* grad_sigma, int N, int M, const Vol& g, int aux,
                       hipStream_t s_) {
  // dvxlr: the first pad_wgs workgroups of every grid row are the persistent padding workgroups (dvr_par_kernels.h)
  const int nb = (M + kParRays - 1) / kParRays;
  const bool rows = (KIND == kParDvxlr || KIND == kParDvxlrV2);
  const int pad_wgs = rows ? std::min(KIND == kParDvxlrV2 ? kParPadWgsV2 : kParPadWgs, (M + 63) / 64) : 0;
  hipLaunchKernelGGL(dvr_par_kernel<KIND>, dim3(nb + pad_wgs, N), dim3(kParThreads), 0, s_, sigma,
                     sigma_regul, origin, points, tindex, pred_dist, gt_dist, dd_dsigma, indices, ray_pred, indicator,
                     grad_sigma, M, g, aux, pad_wgs);
}

// ----------------------------------------------------------------------------------------------
// dvr.render_forward
// ----------------------------------------------------------------------------------------------
template <int kBlock>
__global__ __launch_bounds__(kBlock) void dvr_render_forward_kernel(
    const float* __restrict__ sigma, const float* __restrict__ origin,
    const float* __restrict__ points, const float* __restrict__ tindex,
    float* __restrict__ pred_dist, float* __restrict__ gt_dist, int M, Vol g, int train_phase) {
  const int n = blockIdx.y;
  const int c = pick_ray<kBlock>(origin, points, tindex, n, M, g);
  if (c >= M) return;
  seq_forward_ray(sigma, origin, points, tindex, pred_dist, gt_dist, n, c, M, g, train_phase);
}

// ----------------------------------------------------------------------------------------------
// dvr.render (fused loss gradient, dvr.cu:594-623).  Reference accumulates with a racy "+=";
// we use hardware fp32 atomics, which is the race-free reading of the same sum.
// ----------------------------------------------------------------------------------------------
template <int kBlock>
__global__ __launch_bounds__(kBlock) void dvr_render_kernel(
    const float* __restrict__ sigma, const float* __restrict__ origin,
    const float* __restrict__ points, const float* __restrict__ tindex,
    float* __restrict__ pred_dist, float* __restrict__ gt_dist, float* __restrict__ grad_sigma,
    int M, Vol g, int loss_type) {
  const int n = blockIdx.y;
  const int c = pick_ray<kBlock>(origin, points, tindex, n, M, g);
  if (c >= M) return;
  seq_render_ray(sigma, origin, points, tindex, pred_dist, gt_dist, grad_sigma, n, c, M, g, loss_type);
}

// ----------------------------------------------------------------------------------------------
// dvxlr.render / dvxlr_v2.render_v2: two launches, no scratch, every output byte written once.
//
//  1. march (issue bound, lane per ray): while a lane walks its ray it parks (dt_k, voxel id,
//     W_{k-1}) as one 12-byte store per sample inside the ray's own `indices` row (RowStager,
//     dvr_march.h); the unused W slot of sample 0 receives count / k_surface / NaN flag.
//  2. finish (HBM bound, wave per ray, coalesced): a reverse wave scan turns W into the suffix sums
//     R_k, dd_dsigma[k] = -dt_k R_k, (z, y, x) are unpacked, the v2 extras are gathered and the
//     rest of the API-mandated [1026] rows is padded -- the padding is 95 % of the bytes, and a
//     launch with one wave per ray writes it at fill rate, which the few march waves cannot.
// One traversal, one exp per sample.
// ----------------------------------------------------------------------------------------------
template <int kBlock>
__global__ __launch_bounds__(kBlock) void dvxlr_march_kernel(
    const float* __restrict__ sigma, const float* __restrict__ origin,
    const float* __restrict__ points, const float* __restrict__ tindex,
    float* __restrict__ pred_dist, float* __restrict__ gt_dist, float* __restrict__ indices, int M,
    Vol g) {
  const int n = blockIdx.y;
  const int c = pick_ray<kBlock>(origin, points, tindex, n, M, g);
  if (c >= M) return;
  dvxlr_march_ray(sigma, origin, points, tindex, pred_dist, gt_dist, indices, n, c, M, g);
}

// grid: (ceil(M/4), N), 256 threads = 4 rays.  Rows are 8-byte aligned (1026 floats, aligned base).
template <bool V2, bool PAD>
__global__ __launch_bounds__(256) void dvxlr_finish_kernel(
    const float* __restrict__ sigma_regul, const float* __restrict__ tindex,
    float* __restrict__ dd_dsigma, float* __restrict__ indices, float* __restrict__ ray_pred,
    float* __restrict__ indicator, int M, Vol g) {
  constexpr int L = kDvxlrMaxD;
  const int n = blockIdx.y;
  const int lane = threadIdx.x % kWave;
  const int c = blockIdx.x * (256 / kWave) + threadIdx.x / kWave;
  if (c >= M) return;
  const size_t row = (size_t)n * M + c;
  float* ddr = dd_dsigma + row * L;
  float* idr = indices + row * L * 3;
  float* rpr = V2 ? ray_pred + row * L : nullptr;
  float* inr = V2 ? indicator + row * L : nullptr;

  const float* reg = nullptr;
  if (V2) {
    int cnt0, ks0;
    bool nt0;
    decode_stash(idr[2], cnt0, ks0, nt0);
    if (cnt0 > 0) {
      const long ti = (long)tindex[row];          // a ray with samples has a valid time index
      reg = sigma_regul + ((size_t)n * g.T + (g.T == 1 ? 0 : ti)) * ((size_t)g.Z * g.Y * g.X);
    }
  }
  const int cnt = dvxlr_finish_row<V2>(reg, ddr, idr, rpr, inr, g, lane);

  if (!PAD) return;
  // padding: one odd element if needed, then 8-byte stores
  const float2 zero2 = make_float2(0.f, 0.f), neg2 = make_float2(-1.f, -1.f);
  if ((cnt & 1) && lane == 0 && cnt < L) {
    ddr[cnt] = 0.f;
    idr[3 * cnt] = 0.f;
    if (V2) { rpr[cnt] = 0.f; inr[cnt] = -1.f; }
  }
  const int e1 = (cnt + 1) >> 1;             // first whole float2 of a [L] row
  const int e3 = (3 * cnt + 1) >> 1;         // ... of the [3L] row
  float2* dd2 = reinterpret_cast<float2*>(ddr);
  float2* id2 = reinterpret_cast<float2*>(idr);
  for (int i = e1 + lane; i < L / 2; i += kWave) dd2[i] = zero2;
  for (int i = e3 + lane; i < 3 * L / 2; i += kWave) id2[i] = zero2;
  if (V2) {
    float2* rp2 = reinterpret_cast<float2*>(rpr);
    float2* in2 = reinterpret_cast<float2*>(inr);
    for (int i = e1 + lane; i < L / 2; i += kWave) rp2[i] = zero2;
    for (int i = e1 + lane; i < L / 2; i += kWave) in2[i] = neg2;
  }
}

// ----------------------------------------------------------------------------------------------
// get_grad_sigma{,_v2}: one wave per ray, the padded rows are streamed with coalesced loads.
// Zero contributions are skipped (adding +0 is the identity), NaNs are propagated like the reference.
// ----------------------------------------------------------------------------------------------
template <bool V2>
__global__ __launch_bounds__(256) void dvxlr_scatter_kernel(
    const float* __restrict__ em, const float* __restrict__ indices,
    const float* __restrict__ tindex, const float* __restrict__ indicator,
    const float* __restrict__ grad_ray_pred, float* __restrict__ grad_sigma,
    float* __restrict__ grad_regul, int M, int L, Vol g, int ncopies, size_t copy_stride) {
  const int n = blockIdx.y;
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  const int c = blockIdx.x * (256 / kWave) + wave;
  if (c >= M) return;
  // rays of one frame share their first voxels (the sensor origin): workgroup i adds into private copy i mod n
  grad_sigma += (size_t)(blockIdx.x % ncopies) * copy_stride;
  if (V2) grad_regul += (size_t)(blockIdx.x % ncopies) * copy_stride;
  const float t = tindex[(size_t)n * M + c];
  if (t < 0.f || t != t) return;
  const long ti = (long)t;
  if (!(g.T == 1 || ti < g.T)) return;
  const int ts = (g.T == 1) ? 0 : (int)ti;
  const size_t vol = (size_t)g.Z * g.Y * g.X;
  float* gs = grad_sigma + ((size_t)n * g.T + ts) * vol;
  float* gr = V2 ? grad_regul + ((size_t)n * g.T + ts) * vol : nullptr;
  const size_t row = ((size_t)n * M + c) * L;
  for (int i = lane; i < L; i += kWave) {
    const float v = em[row + i];
    bool live = (v != 0.f);
    float rv = 0.f;
    if (V2) {
      if (indicator[row + i] >= 0.f) {
        rv = grad_ray_pred[row + i];
        live = live || (rv != 0.f);
      }
    }
    if (!live) continue;
    const int z = (int)indices[(row + i) * 3 + 0];
    const int y = (int)indices[(row + i) * 3 + 1];
    const int x = (int)indices[(row + i) * 3 + 2];
    const size_t o = ((size_t)z * g.Y + y) * g.X + x;
    if (v != 0.f) unsafeAtomicAdd(gs + o, v);
    if (V2 && rv != 0.f) unsafeAtomicAdd(gr + o, rv);
  }
}

constexpr int kScatterCopies = 8;
__global__ __launch_bounds__(256) void dvxlr_sum_copies_kernel(const float* __restrict__ copies, float* __restrict__ out,
                                                               size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float a = copies[i];
  for (int c = 1; c < kScatterCopies; ++c) a += copies[(size_t)c * n + i];
  out[i] = a;
}

// ----------------------------------------------------------------------------------------------
// init (occupancy rasterisation, dvr.cu:14-63 == dvxlr.cu:12-61)
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dvr_init_kernel(const float* __restrict__ points,
                                                       const float* __restrict__ tindex,
                                                       float* __restrict__ occupancy, int M, Vol g) {
  const int n = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= M) return;
  const float t = tindex[(size_t)n * M + c];
  if (t < 0.f || t != t) return;
  const long ti = (long)t;
  if (!(g.T == 1 || ti < g.T)) return;
  const int ts = (g.T == 1) ? 0 : (int)ti;
  const float* p = points + ((size_t)n * M + c) * 3;
  const int vx = (int)p[0], vy = (int)p[1], vz = (int)p[2];
  if (0 <= vx && vx < g.X && 0 <= vy && vy < g.Y && 0 <= vz && vz < g.Z)
    occupancy[((((size_t)n * g.T + ts) * g.Z + vz) * g.Y + vy) * g.X + vx] = 1.f;
}

inline int hip_ret() { return vidar_last_error(); }
inline bool bad_dims(int N, int M, int T, int Z, int Y, int X) {
  return N < 0 || M < 0 || T <= 0 || Z <= 0 || Y <= 0 || X <= 0;
}

}  // namespace

extern "C" {

int vidar_dvr_max_d(void) { return kDvrMaxD; }
int vidar_dvxlr_set_pad_mode(int mode) {
  const int prev = g_dvxlr_pad_mode;
  g_dvxlr_pad_mode = (mode < 0 || mode > 1) ? 1 : mode;
  return prev;
}
int vidar_dvr_set_sort_min_waves(int min_waves) {
  const int prev = g_sort_min_waves;
  g_sort_min_waves = min_waves < 0 ? 0 : min_waves;
  return prev;
}
int vidar_dvr_set_traversal(int mode) {
  const int prev = g_traversal;
  g_traversal = (mode < -1 || mode > 1) ? -1 : mode;
  return prev;
}
#ifdef VIDAR_PAR_TIMING
int vidar_dbg_par_cycles(unsigned long long* out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(vidar_march::g_par_cycles), sizeof(unsigned long long) * 8);
  if (reset) {
    unsigned long long z[8] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(vidar_march::g_par_cycles), z, sizeof(z));
  }
  return 0;
}
#endif
int vidar_dvxlr_max_d(void) { return kDvxlrMaxD; }

int vidar_dvr_render_forward_f32(const float* sigma, const float* origin, const float* points,
                                 const float* tindex, float* pred_dist, float* gt_dist, int N, int M,
                                 int T, int TO, int Z, int Y, int X, int train_phase, void* stream) {
  VIDAR_ENTER();
  if (bad_dims(N, M, T, Z, Y, X) || TO <= 0 || (train_phase != 0 && train_phase != 1))
    return VIDAR_ERR_BAD_ARG;
  if (N == 0 || M == 0) return 0;
  Vol g{T, TO, Z, Y, X};
  if (step_parallel(N, M, g, kParAutoForward))
    launch_par<kParForward>(sigma, nullptr, origin, points, tindex, pred_dist, gt_dist, nullptr, nullptr, nullptr,
                            nullptr, nullptr, N, M, g, train_phase, (hipStream_t)stream);
  else if (sort_rays(N, M))
    hipLaunchKernelGGL(dvr_render_forward_kernel<kSortBlock>, dim3((M + kSortBlock - 1) / kSortBlock, N),
                       dim3(kSortBlock), 0, (hipStream_t)stream, sigma, origin, points, tindex,
                       pred_dist, gt_dist, M, g, train_phase);
  else
    hipLaunchKernelGGL(dvr_render_forward_kernel<kWave>, dim3((M + kWave - 1) / kWave, N), dim3(kWave),
                       0, (hipStream_t)stream, sigma, origin, points, tindex, pred_dist, gt_dist, M, g,
                       train_phase);
  return vidar_last_error();
}

int vidar_dvr_render_f32(const float* sigma, const float* origin, const float* points,
                         const float* tindex, float* pred_dist, float* gt_dist, float* grad_sigma,
                         int N, int M, int T, int TO, int Z, int Y, int X, int loss_type,
                         void* stream) {
  VIDAR_ENTER();
  if (bad_dims(N, M, T, Z, Y, X) || TO <= 0 || loss_type < 0 || loss_type > 2)
    return VIDAR_ERR_BAD_ARG;
  hipError_t e = hipMemsetAsync(grad_sigma, 0, sizeof(float) * (size_t)N * T * Z * Y * X,
                                (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  if (N == 0 || M == 0) return 0;
  Vol g{T, TO, Z, Y, X};
  if (step_parallel(N, M, g, kParAutoRender))
    launch_par<kParRender>(sigma, nullptr, origin, points, tindex, pred_dist, gt_dist, nullptr, nullptr, nullptr,
                           nullptr, grad_sigma, N, M, g, loss_type, (hipStream_t)stream);
  else if (sort_rays(N, M))
    hipLaunchKernelGGL(dvr_render_kernel<kSortBlock>, dim3((M + kSortBlock - 1) / kSortBlock, N),
                       dim3(kSortBlock), 0, (hipStream_t)stream, sigma, origin, points, tindex,
                       pred_dist, gt_dist, grad_sigma, M, g, loss_type);
  else
    hipLaunchKernelGGL(dvr_render_kernel<kWave>, dim3((M + kWave - 1) / kWave, N), dim3(kWave), 0,
                       (hipStream_t)stream, sigma, origin, points, tindex, pred_dist, gt_dist,
                       grad_sigma, M, g, loss_type);
  return vidar_last_error();
}

int vidar_dvr_init_f32(const float* points, const float* tindex, float* occupancy, int N, int M,
                       int T, int Z, int Y, int X, void* stream) {
  VIDAR_ENTER();
  if (bad_dims(N, M, T, Z, Y, X)) return VIDAR_ERR_BAD_ARG;
  hipError_t e = hipMemsetAsync(occupancy, 0, sizeof(float) * (size_t)N * T * Z * Y * X,
                                (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  if (N == 0 || M == 0) return 0;
  Vol g{T, T, Z, Y, X};
  dim3 grid((M + 255) / 256, N);
  hipLaunchKernelGGL(dvr_init_kernel, grid, dim3(256), 0, (hipStream_t)stream, points, tindex,
                     occupancy, M, g);
  return vidar_last_error();
}

static int dvxlr_render_launch(bool v2, const float* sigma, const float* sigma_regul,
                               const float* origin, const float* points, const float* tindex,
                               float* pred_dist, float* gt_dist, float* dd_dsigma, float* indices,
                               float* ray_pred, float* indicator, int N, int M, int T, int TO, int Z,
                               int Y, int X, hipStream_t s_) {
  if (bad_dims(N, M, T, Z, Y, X) || TO <= 0 || (size_t)Z * Y * X >= (1u << 24)) return VIDAR_ERR_BAD_ARG;
  // rows are padded with 8-byte stores
  if ((((uintptr_t)dd_dsigma | (uintptr_t)indices | (uintptr_t)ray_pred | (uintptr_t)indicator) & 7u) != 0)
    return VIDAR_ERR_BAD_ARG;
  if (N == 0 || M == 0) return 0;
  Vol g{T, TO, Z, Y, X};
  // one launch: padding, march and row finish in the step-parallel kernel (float4 fills need 16-byte aligned rows)
  if (step_parallel(N, M, g, kParAutoDvxlr) &&
      (((uintptr_t)dd_dsigma | (uintptr_t)indices | (uintptr_t)ray_pred | (uintptr_t)indicator) & 15u) == 0) {
    if (v2)
      launch_par<kParDvxlrV2>(sigma, sigma_regul, origin, points, tindex, pred_dist, gt_dist, dd_dsigma, indices,
                              ray_pred, indicator, nullptr, N, M, g, 0, s_);
    else
      launch_par<kParDvxlr>(sigma, nullptr, origin, points, tindex, pred_dist, gt_dist, dd_dsigma, indices, nullptr,
                            nullptr, nullptr, N, M, g, 0, s_);
    return vidar_last_error();
  }
  if (g_dvxlr_pad_mode != 0) {
    const size_t rows = (size_t)N * M * kDvxlrMaxD;
    hipError_t e = hipMemsetAsync(dd_dsigma, 0, rows * sizeof(float), s_);
    if (v2 && e == hipSuccess) e = hipMemsetAsync(ray_pred, 0, rows * sizeof(float), s_);
    if (v2 && e == hipSuccess)
      e = hipMemsetD32Async((hipDeviceptr_t)indicator, 0xBF800000 /* -1.0f */, rows, s_);
    if (e == hipSuccess) e = hipMemsetAsync(indices, 0, rows * 3 * sizeof(float), s_);
    if (e != hipSuccess) return (int)e;
  }
  if (sort_rays(N, M))
    hipLaunchKernelGGL(dvxlr_march_kernel<kSortBlock>, dim3((M + kSortBlock - 1) / kSortBlock, N),
                       dim3(kSortBlock), 0, s_, sigma, origin, points, tindex, pred_dist, gt_dist,
                       indices, M, g);
  else
    hipLaunchKernelGGL(dvxlr_march_kernel<kWave>, dim3((M + kWave - 1) / kWave, N), dim3(kWave), 0, s_,
                       sigma, origin, points, tindex, pred_dist, gt_dist, indices, M, g);
  const dim3 fgrid((M + 3) / 4, N);
  const bool pad = (g_dvxlr_pad_mode == 0);
#define VIDAR_FINISH(V2_, PAD_)                                                                      \
  hipLaunchKernelGGL((dvxlr_finish_kernel<V2_, PAD_>), fgrid, dim3(256), 0, s_, sigma_regul, tindex, \
                     dd_dsigma, indices, ray_pred, indicator, M, g)
  if (v2) { if (pad) VIDAR_FINISH(true, true); else VIDAR_FINISH(true, false); }
  else    { if (pad) VIDAR_FINISH(false, true); else VIDAR_FINISH(false, false); }
#undef VIDAR_FINISH
  return vidar_last_error();
}

int vidar_dvxlr_render_f32(const float* sigma, const float* origin, const float* points,
                           const float* tindex, float* pred_dist, float* gt_dist, float* dd_dsigma,
                           float* indices, int N, int M, int T, int TO, int Z, int Y, int X,
                           void* stream) {
  VIDAR_ENTER();
  return dvxlr_render_launch(false, sigma, nullptr, origin, points, tindex, pred_dist, gt_dist,
                             dd_dsigma, indices, nullptr, nullptr, N, M, T, TO, Z, Y, X,
                             (hipStream_t)stream);
}

int vidar_dvxlr2_render_f32(const float* sigma, const float* origin, const float* points,
                            const float* tindex, const float* sigma_regul, float* pred_dist,
                            float* gt_dist, float* dd_dsigma, float* indices, float* ray_pred,
                            float* indicator, int N, int M, int T, int TO, int Z, int Y, int X,
                            void* stream) {
  VIDAR_ENTER();
  return dvxlr_render_launch(true, sigma, sigma_regul, origin, points, tindex, pred_dist, gt_dist,
                             dd_dsigma, indices, ray_pred, indicator, N, M, T, TO, Z, Y, X,
                             (hipStream_t)stream);
}

size_t vidar_dvxlr_get_grad_sigma_workspace_bytes(int N, int T, int Z, int Y, int X, int volumes) {
  if (bad_dims(N, 0, T, Z, Y, X) || volumes < 1 || volumes > 2) return 0;
  return sizeof(float) * (size_t)N * T * Z * Y * X * kScatterCopies * volumes;   // 1: get_grad_sigma, 2: _v2 (two volumes)
}

int vidar_dvxlr_get_grad_sigma_f32(const float* elementwise_mult, const float* indices,
                                   const float* tindex, float* grad_sigma, int N, int M, int L, int T,
                                   int Z, int Y, int X, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  VIDAR_ENTER();
  if (bad_dims(N, M, T, Z, Y, X) || L < 0) return VIDAR_ERR_BAD_ARG;
  const size_t n = (size_t)N * T * Z * Y * X;
  const bool copies = workspace != nullptr && workspace_bytes >= sizeof(float) * n * kScatterCopies;
  float* acc = copies ? (float*)workspace : grad_sigma;
  hipStream_t s_ = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(acc, 0, sizeof(float) * n * (copies ? kScatterCopies : 1), s_);
  if (e != hipSuccess) return (int)e;
  if (N == 0 || M == 0 || L == 0) return copies ? (int)hipMemsetAsync(grad_sigma, 0, sizeof(float) * n, s_) : 0;
  Vol g{T, T, Z, Y, X};
  dim3 grid((M + 3) / 4, N);
  hipLaunchKernelGGL(dvxlr_scatter_kernel<false>, grid, dim3(256), 0, s_, elementwise_mult, indices, tindex,
                     (const float*)nullptr, (const float*)nullptr, acc, (float*)nullptr, M, L, g,
                     copies ? kScatterCopies : 1, n);
  if (copies)
    hipLaunchKernelGGL(dvxlr_sum_copies_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s_, acc, grad_sigma, n);
  return vidar_last_error();
}

int vidar_dvxlr2_get_grad_sigma_f32(const float* elementwise_mult, const float* indices,
                                    const float* tindex, const float* indicator,
                                    const float* grad_ray_pred, float* grad_sigma,
                                    float* grad_sigma_regul, int N, int M, int L, int T, int Z, int Y,
                                    int X, void* workspace, size_t workspace_bytes, void* stream) {
  VIDAR_ENTER();
  if (bad_dims(N, M, T, Z, Y, X) || L < 0) return VIDAR_ERR_BAD_ARG;
  const size_t n = (size_t)N * T * Z * Y * X;
  const bool copies = workspace != nullptr && workspace_bytes >= vidar_dvxlr_get_grad_sigma_workspace_bytes(N, T, Z, Y, X, 2);
  float* acc = copies ? (float*)workspace : grad_sigma;
  float* acc2 = copies ? (float*)workspace + n * kScatterCopies : grad_sigma_regul;
  hipStream_t s_ = (hipStream_t)stream;
  hipError_t e;
  if (copies) {
    e = hipMemsetAsync(acc, 0, sizeof(float) * n * kScatterCopies * 2, s_);
  } else {
    e = hipMemsetAsync(acc, 0, sizeof(float) * n, s_);
    if (e == hipSuccess) e = hipMemsetAsync(acc2, 0, sizeof(float) * n, s_);
  }
  if (e != hipSuccess) return (int)e;
  if (N == 0 || M == 0 || L == 0) {
    if (!copies) return 0;
    e = hipMemsetAsync(grad_sigma, 0, sizeof(float) * n, s_);
    if (e == hipSuccess) e = hipMemsetAsync(grad_sigma_regul, 0, sizeof(float) * n, s_);
    return (int)e;
  }
  Vol g{T, T, Z, Y, X};
  dim3 grid((M + 3) / 4, N);
  hipLaunchKernelGGL(dvxlr_scatter_kernel<true>, grid, dim3(256), 0, s_, elementwise_mult, indices, tindex, indicator,
                     grad_ray_pred, acc, acc2, M, L, g, copies ? kScatterCopies : 1, n);
  if (copies) {
    const dim3 rg((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(dvxlr_sum_copies_kernel, rg, dim3(256), 0, s_, acc, grad_sigma, n);
    hipLaunchKernelGGL(dvxlr_sum_copies_kernel, rg, dim3(256), 0, s_, acc2, grad_sigma_regul, n);
  }
  return vidar_last_error();
}

}  // extern "C"
