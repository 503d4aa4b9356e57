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
//  * one wave (64 rays) per workgroup so that M=30k rays spread over all CUs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include <math.h>

#include "vidar_hip.h"
#include "vidar_common.h"

namespace {

constexpr int kDvrMaxD = 1446;    // dvr.cu:9
constexpr int kDvxlrMaxD = 1026;  // dvxlr.cu:10, dvxlr_v2.cu:10
constexpr int kWave = 64;
constexpr long kStepCap = 1L << 22;  // reference has no cap (it would hang); we bound every loop

enum MarchMode : int {
  kClassic = 0,        // dvr.render: first boundary v+(step<0?0:1), integer voxel path
  kRounded = 1,        // dvr.render_forward: boundary v+(step<0?-1:1), path = round(position)
  kRoundedMerged = 2,  // dvxlr / dvxlr_v2: kRounded + merge of consecutive duplicate voxels
};

struct Vol {
  int T, TO, Z, Y, X;
};

struct RayIn {
  double xo, yo, zo, xe, ye, ze;
  int ts;      // time slice of sigma
  bool valid;  // false: padded ray (tindex < 0) or tindex out of range
};

__device__ __forceinline__ RayIn load_ray(const float* __restrict__ origin,
                                          const float* __restrict__ points,
                                          const float* __restrict__ tindex, int n, int c, int M,
                                          const Vol& v) {
  RayIn r;
  const float t = tindex[(size_t)n * M + c];
  r.valid = !(t < 0.f) && (t == t);
  long ti = r.valid ? (long)t : 0;
  if (!(v.T == 1 || ti < v.T) || ti >= v.TO) r.valid = false;  // reference: device assert
  if (!r.valid) ti = 0;
  r.ts = (v.T == 1) ? 0 : (int)ti;
  const float* o = origin + ((size_t)n * v.TO + ti) * 3;
  const float* p = points + ((size_t)n * M + c) * 3;
  r.xo = o[0]; r.yo = o[1]; r.zo = o[2];
  r.xe = p[0]; r.ye = p[1]; r.ze = p[2];
  return r;
}

// Amanatides-Woo traversal with the reference's modifications.  Sink::sample is
// called once per step spent inside the volume, in order, with the voxel the
// reference would record, the exit distance _d of that step and the previous
// step's exit distance.  Returns the un-clamped ray length.
template <int MODE, class Sink>
__device__ __forceinline__ double march(const RayIn& r, const Vol& g, Sink& sink) {
  int vx = (int)r.xo, vy = (int)r.yo, vz = (int)r.zo;
  double px = (double)vx, py = (double)vy, pz = (double)vz;
  const double rx = r.xe - r.xo, ry = r.ye - r.yo, rz = r.ze - r.zo;
  const double len = sqrt(rx * rx + ry * ry + rz * rz);
  const double dx = rx / len, dy = ry / len, dz = rz / len;
  const int sx = (dx >= 0) ? 1 : -1, sy = (dy >= 0) ? 1 : -1, sz = (dz >= 0) ? 1 : -1;
  const int back = (MODE == kClassic) ? 0 : -1;
  const double bx = vx + (sx < 0 ? back : 1);
  const double by = vy + (sy < 0 ? back : 1);
  const double bz = vz + (sz < 0 ? back : 1);
  double tx = (dx != 0) ? (bx - r.xo) / dx : DBL_MAX;
  double ty = (dy != 0) ? (by - r.yo) / dy : DBL_MAX;
  double tz = (dz != 0) ? (bz - r.zo) / dz : DBL_MAX;
  const double ddx = (dx != 0) ? sx / dx : DBL_MAX;
  const double ddy = (dy != 0) ? sy / dy : DBL_MAX;
  const double ddz = (dz != 0) ? sz / dz : DBL_MAX;

  double last_d = 0.0;
  bool was_inside = false;
  for (long step = 0; step < kStepCap; ++step) {
    const bool inside = (0 <= vx && vx < g.X) && (0 <= vy && vy < g.Y) && (0 <= vz && vz < g.Z);
    int qx = vx, qy = vy, qz = vz;
    if (inside) {
      was_inside = true;
      if (MODE != kClassic) {
        qx = (int)round(px); qx = qx < g.X ? qx : g.X - 1; qx = qx >= 0 ? qx : 0;
        qy = (int)round(py); qy = qy < g.Y ? qy : g.Y - 1; qy = qy >= 0 ? qy : 0;
        qz = (int)round(pz); qz = qz < g.Z ? qz : g.Z - 1; qz = qz >= 0 ? qz : 0;
      }
    } else if (was_inside) {
      break;
    } else if (last_d > len) {
      break;
    }
    double d;
    if (tx < ty) {
      if (tx < tz) { d = tx; vx += sx; tx += ddx; }
      else         { d = tz; vz += sz; tz += ddz; }
    } else {
      if (ty < tz) { d = ty; vy += sy; ty += ddy; }
      else         { d = tz; vz += sz; tz += ddz; }
    }
    if (MODE != kClassic) {
      const double adv = fmax(0.0, d - last_d);
      px += adv * dx; py += adv * dy; pz += adv * dz;
    }
    if (inside) {
      if (!sink.sample(qx, qy, qz, d, last_d)) break;
    }
    last_d = d;
  }
  sink.finish();
  return len;
}

// Online integrator shared by every variant.  Emit::commit(k, x,y,z, d, dt, P_k, W_{k-1})
// is called once per *final* sample k in order (P_k = prefix of W before k).
template <int MODE, int MAXD, class Emit>
struct Integrator {
  const float* __restrict__ sig;  // sigma[n][ts] slice
  int Y, X;
  Emit& emit;
  // committed state
  int k = 0;
  double csd = 0.0, Tprev = 1.0, dprev = 0.0, d0 = 0.0, S = 0.0;
  // pending sample (merged mode only)
  bool pending = false;
  int ux = 0, uy = 0, uz = 0;
  double ud = 0.0, udt = 0.0;

  __device__ __forceinline__ Integrator(const float* s, int Y_, int X_, Emit& e)
      : sig(s), Y(Y_), X(X_), emit(e) {}

  __device__ __forceinline__ void commit(int x, int y, int z, double d, double dt) {
    const double sg = (double)sig[((size_t)z * Y + y) * X + x];
    double w_prev = 0.0;                       // W_{k-1} = T_{k-1} (d_k - d_{k-1})
    if (k == 0) {
      d0 = d;
    } else {
      w_prev = Tprev * (d - dprev);
      S += w_prev;
    }
    emit.commit(k, x, y, z, d, dt, S, w_prev);
    csd = (k == 0) ? sg * dt : csd + sg * dt;
    // the transmittance only scales value outputs (1e-7 relative is plenty for fp32 results);
    // csd itself and every traversal quantity stay fp64
    Tprev = (double)expf((float)(-csd));
    dprev = d;
    ++k;
  }

  __device__ __forceinline__ bool sample(int x, int y, int z, double d, double last_d) {
    if (MODE == kRoundedMerged) {
      if (pending && x == ux && y == uy && z == uz) {
        // dvxlr.cu:366-377: drop the previous sample, rewind last_d by its dt
        udt = fmax(0.0, d - (last_d - udt));
        ud = d;
        return true;
      }
      if (pending) commit(ux, uy, uz, ud, udt);
      if (k >= MAXD) { pending = false; return false; }
      ux = x; uy = y; uz = z; ud = d; udt = fmax(0.0, d - last_d);
      pending = true;
      return true;
    } else {
      if (k >= MAXD) return false;
      commit(x, y, z, d, fmax(0.0, d - last_d));
      return true;
    }
  }
  __device__ __forceinline__ void finish() {
    if (MODE == kRoundedMerged && pending) { commit(ux, uy, uz, ud, udt); pending = false; }
  }
  // after finish(): count = k, p_out = Tprev, max_d = dprev, pred = d0 + S
};

struct NoEmit {
  __device__ __forceinline__ void commit(int, int, int, int, double, double, double, double) {}
};

// ----------------------------------------------------------------------------------------------
// dvr.render_forward
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) void dvr_render_forward_kernel(
    const float* __restrict__ sigma, const float* __restrict__ origin,
    const float* __restrict__ points, const float* __restrict__ tindex,
    float* __restrict__ pred_dist, float* __restrict__ gt_dist, int M, Vol g, int train_phase) {
  const int n = blockIdx.y;
  const int c = blockIdx.x * kWave + threadIdx.x;
  if (c >= M) return;
  float pred = -1.f, gt = -1.f;
  const RayIn r = load_ray(origin, points, tindex, n, c, M, g);
  if (r.valid) {
    NoEmit ne;
    const size_t vol = (size_t)g.Z * g.Y * g.X;
    Integrator<kRounded, kDvrMaxD, NoEmit> integ(sigma + ((size_t)n * g.T + r.ts) * vol, g.Y, g.X, ne);
    const double len = march<kRounded>(r, g, integ);
    if (integ.k > 0) {
      pred = (float)(integ.d0 + integ.S);
      gt = (float)(train_phase ? fmin(len, integ.dprev) : len);
    }
  }
  pred_dist[(size_t)n * M + c] = pred;
  gt_dist[(size_t)n * M + c] = gt;
}

// ----------------------------------------------------------------------------------------------
// dvr.render (fused loss gradient, dvr.cu:594-623).  Reference accumulates with a racy "+=";
// we use hardware fp32 atomics, which is the race-free reading of the same sum.
// ----------------------------------------------------------------------------------------------
struct GradScatter {
  float* __restrict__ grad;  // grad_sigma[n][ts] slice
  int Y, X;
  double S_total, dl_dd;
  __device__ __forceinline__ void commit(int, int x, int y, int z, double, double dt, double P,
                                         double) {
    const double g = dl_dd * (dt * (P - S_total));
    if (g != 0.0) unsafeAtomicAdd(grad + ((size_t)z * Y + y) * X + x, (float)g);
  }
};

__global__ __launch_bounds__(kWave) void dvr_render_kernel(
    const float* __restrict__ sigma, const float* __restrict__ origin,
    const float* __restrict__ points, const float* __restrict__ tindex,
    float* __restrict__ pred_dist, float* __restrict__ gt_dist, float* __restrict__ grad_sigma,
    int M, Vol g, int loss_type) {
  const int n = blockIdx.y;
  const int c = blockIdx.x * kWave + threadIdx.x;
  if (c >= M) return;
  float pred = -1.f, gt = -1.f;
  const RayIn r = load_ray(origin, points, tindex, n, c, M, g);
  if (r.valid) {
    const size_t vol = (size_t)g.Z * g.Y * g.X;
    const size_t slice = ((size_t)n * g.T + r.ts) * vol;
    NoEmit ne;
    Integrator<kClassic, kDvrMaxD, NoEmit> a(sigma + slice, g.Y, g.X, ne);
    const double len = march<kClassic>(r, g, a);
    if (a.k > 0) {
      const double exp_d = a.d0 + a.S;
      const double gt_d = fmin(len, a.dprev);
      pred = (float)exp_d;
      gt = (float)gt_d;
      double dl = 1.0;
      if (loss_type == 0) dl = (exp_d >= gt_d) ? 1.0 : -1.0;
      else if (loss_type == 1) dl = exp_d - gt_d;
      else if (loss_type == 2) dl = (exp_d >= gt_d) ? (1.0 / gt_d) : -(1.0 / gt_d);
      GradScatter gs{grad_sigma + slice, g.Y, g.X, a.S, dl};
      Integrator<kClassic, kDvrMaxD, GradScatter> b(sigma + slice, g.Y, g.X, gs);
      march<kClassic>(r, g, b);
    }
  }
  pred_dist[(size_t)n * M + c] = pred;
  gt_dist[(size_t)n * M + c] = gt;
}

// ----------------------------------------------------------------------------------------------
// dvxlr.render / dvxlr_v2.render_v2
// ----------------------------------------------------------------------------------------------
// Single march.  While a lane walks its ray it parks, inside the ray's own output rows,
//   dd_row[k-1]   <- W_{k-1} (fp32)     idx_row[3k+0] <- dt_k (fp32)
//   idx_row[3k+1] <- linear voxel id (z*Y + y)*X + x, exact in fp32 below 2^24 voxels per slice
// and afterwards the whole wave revisits the 64 rows it owns with coalesced accesses: a reverse
// wave scan turns W into the suffix sums R_k, dd_dsigma[k] = -dt_k R_k, (z, y) are unpacked, the
// v2 extras are gathered and the tails are padded.  One traversal, one exp per sample, no scratch.
struct RowStager {
  float* __restrict__ dd;
  float* __restrict__ idx;
  int Y, X;
  double true_len;
  int k_surface = -1;
  __device__ __forceinline__ void commit(int k, int x, int y, int z, double d, double dt, double,
                                         double w_prev) {
    if (k > 0) dd[k - 1] = (float)w_prev;
    idx[3 * k + 0] = (float)dt;
    idx[3 * k + 1] = (float)((z * Y + y) * X + x);
    if (k_surface < 0 && d >= true_len) k_surface = k;    // dvxlr_v2.cu:408-424
  }
};

template <bool V2>
__global__ __launch_bounds__(kWave) void dvxlr_render_kernel(
    const float* __restrict__ sigma, const float* __restrict__ sigma_regul,
    const float* __restrict__ origin, const float* __restrict__ points,
    const float* __restrict__ tindex, float* __restrict__ pred_dist, float* __restrict__ gt_dist,
    float* __restrict__ dd_dsigma, float* __restrict__ indices, float* __restrict__ ray_pred,
    float* __restrict__ indicator, int M, Vol g) {
  constexpr int L = kDvxlrMaxD;
  const int n = blockIdx.y;
  const int c0 = blockIdx.x * kWave;
  const int lane = threadIdx.x;
  const int c = c0 + lane;
  const size_t rowbase = (size_t)n * M;
  const size_t vol = (size_t)g.Z * g.Y * g.X;
  int count = 0, ksurf = -1, ts = 0;
  if (c < M) {
    float pred = -1.f, gt = -1.f;
    const RayIn r = load_ray(origin, points, tindex, n, c, M, g);
    ts = r.ts;
    if (r.valid) {
      RowStager st;
      st.dd = dd_dsigma + (rowbase + c) * L;
      st.idx = indices + (rowbase + c) * L * 3;
      st.Y = g.Y; st.X = g.X;
      {
        const double rx = r.xe - r.xo, ry = r.ye - r.yo, rz = r.ze - r.zo;
        st.true_len = sqrt(rx * rx + ry * ry + rz * rz);
      }
      Integrator<kRoundedMerged, kDvxlrMaxD, RowStager> a(sigma + ((size_t)n * g.T + r.ts) * vol, g.Y,
                                                          g.X, st);
      const double len = march<kRoundedMerged>(r, g, a);
      count = a.k;
      ksurf = st.k_surface;
      if (count > 0) {
        pred = (float)(a.d0 + a.S);
        gt = (float)fmin(len, a.dprev);
      }
    }
    pred_dist[rowbase + c] = pred;
    gt_dist[rowbase + c] = gt;
  }
  __threadfence_block();      // the staged rows were written by single lanes, now every lane reads them

  // fix-up: four rays at a time, 16 lanes each (the per-ray chain "load W -> scan -> store" is
  // latency bound, so independent rays run side by side)
  constexpr int kSub = 16;
  const int grp = lane / kSub, gl = lane % kSub, lead = lane & ~(kSub - 1);
  for (int rq = 0; rq < kWave / (kWave / kSub); ++rq) {
    const int r = rq * (kWave / kSub) + grp;
    const int cr = c0 + r;
    const int cnt_r = __shfl(count, r, kWave);       // unconditional: every lane takes part
    const int cnt = (cr < M) ? cnt_r : 0;
    const int ks = __shfl(ksurf, r, kWave);
    const int tsr = __shfl(ts, r, kWave);
    const size_t row = rowbase + (cr < M ? cr : c0);
    float* ddr = dd_dsigma + row * L;
    float* idr = indices + row * L * 3;
    float* rpr = V2 ? ray_pred + row * L : nullptr;
    float* inr = V2 ? indicator + row * L : nullptr;
    const float* reg = V2 ? sigma_regul + ((size_t)n * g.T + tsr) * vol : nullptr;
    double carry = 0.0;
    for (int base = cnt > 0 ? ((cnt - 1) / kSub) * kSub : -1; base >= 0; base -= kSub) {
      const int k = base + gl;
      double sfx = (k < cnt - 1) ? (double)ddr[k] : 0.0;
#pragma unroll
      for (int off = 1; off < kSub; off <<= 1) {
        const double t = __shfl_down(sfx, off, kWave);
        if (gl + off < kSub) sfx += t;
      }
      const double R = sfx + carry;
      carry += __shfl(sfx, lead, kWave);
      if (k < cnt) {
        const float dtk = idr[3 * k + 0];
        const int vid = (int)idr[3 * k + 1];
        const int zy = vid / g.X, x = vid - zy * g.X;
        const int z = zy / g.Y, y = zy - z * g.Y;
        ddr[k] = (float)(-(double)dtk * R);
        idr[3 * k + 0] = (float)z;
        idr[3 * k + 1] = (float)y;
        idr[3 * k + 2] = (float)x;
        if (V2) {
          rpr[k] = reg[vid];
          inr[k] = (k == ks) ? 1.f : 0.f;
        }
      }
    }
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
    float* __restrict__ grad_regul, int M, int L, Vol g) {
  const int n = blockIdx.y;
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  const int c = blockIdx.x * (256 / kWave) + wave;
  if (c >= M) return;
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
int vidar_dvxlr_max_d(void) { return kDvxlrMaxD; }

int vidar_dvr_render_forward_f32(const float* sigma, const float* origin, const float* points,
                                 const float* tindex, float* pred_dist, float* gt_dist, int N, int M,
                                 int T, int TO, int Z, int Y, int X, int train_phase, void* stream) {
  VIDAR_ENTER();
  if (bad_dims(N, M, T, Z, Y, X) || TO <= 0 || (train_phase != 0 && train_phase != 1))
    return VIDAR_ERR_BAD_ARG;
  if (N == 0 || M == 0) return 0;
  Vol g{T, TO, Z, Y, X};
  dim3 grid((M + kWave - 1) / kWave, N);
  hipLaunchKernelGGL(dvr_render_forward_kernel, grid, dim3(kWave), 0, (hipStream_t)stream, sigma,
                     origin, points, tindex, pred_dist, gt_dist, M, g, train_phase);
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
  dim3 grid((M + kWave - 1) / kWave, N);
  hipLaunchKernelGGL(dvr_render_kernel, grid, dim3(kWave), 0, (hipStream_t)stream, sigma, origin,
                     points, tindex, pred_dist, gt_dist, grad_sigma, M, g, loss_type);
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

int vidar_dvxlr_render_f32(const float* sigma, const float* origin, const float* points,
                           const float* tindex, float* pred_dist, float* gt_dist, float* dd_dsigma,
                           float* indices, int N, int M, int T, int TO, int Z, int Y, int X,
                           void* stream) {
  VIDAR_ENTER();
  if (bad_dims(N, M, T, Z, Y, X) || TO <= 0 || (size_t)Z * Y * X >= (1u << 24)) return VIDAR_ERR_BAD_ARG;
  if (N == 0 || M == 0) return 0;
  Vol g{T, TO, Z, Y, X};
  dim3 grid((M + kWave - 1) / kWave, N);
  // padding first, as one full-rate device fill (measured 6.0 TB/s); the march kernel then only
  // touches the live prefixes
  const size_t rows = (size_t)N * M * kDvxlrMaxD;
  hipError_t e = hipMemsetAsync(dd_dsigma, 0, rows * sizeof(float), (hipStream_t)stream);
  if (e == hipSuccess) e = hipMemsetAsync(indices, 0, rows * 3 * sizeof(float), (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(dvxlr_render_kernel<false>, grid, dim3(kWave), 0, (hipStream_t)stream, sigma,
                     (const float*)nullptr, origin, points, tindex, pred_dist, gt_dist, dd_dsigma,
                     indices, (float*)nullptr, (float*)nullptr, M, g);
  return vidar_last_error();
}

int vidar_dvxlr2_render_f32(const float* sigma, const float* origin, const float* points,
                            const float* tindex, const float* sigma_regul, float* pred_dist,
                            float* gt_dist, float* dd_dsigma, float* indices, float* ray_pred,
                            float* indicator, int N, int M, int T, int TO, int Z, int Y, int X,
                            void* stream) {
  VIDAR_ENTER();
  if (bad_dims(N, M, T, Z, Y, X) || TO <= 0 || (size_t)Z * Y * X >= (1u << 24)) return VIDAR_ERR_BAD_ARG;
  if (N == 0 || M == 0) return 0;
  Vol g{T, TO, Z, Y, X};
  dim3 grid((M + kWave - 1) / kWave, N);
  const size_t rows = (size_t)N * M * kDvxlrMaxD;
  hipStream_t s_ = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(dd_dsigma, 0, rows * sizeof(float), s_);
  if (e == hipSuccess) e = hipMemsetAsync(indices, 0, rows * 3 * sizeof(float), s_);
  if (e == hipSuccess) e = hipMemsetAsync(ray_pred, 0, rows * sizeof(float), s_);
  if (e == hipSuccess) e = hipMemsetD32Async((hipDeviceptr_t)indicator, 0xBF800000 /* -1.0f */, rows, s_);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(dvxlr_render_kernel<true>, grid, dim3(kWave), 0, (hipStream_t)stream, sigma,
                     sigma_regul, origin, points, tindex, pred_dist, gt_dist, dd_dsigma, indices,
                     ray_pred, indicator, M, g);
  return vidar_last_error();
}

int vidar_dvxlr_get_grad_sigma_f32(const float* elementwise_mult, const float* indices,
                                   const float* tindex, float* grad_sigma, int N, int M, int L, int T,
                                   int Z, int Y, int X, void* stream) {
  VIDAR_ENTER();
  if (bad_dims(N, M, T, Z, Y, X) || L < 0) return VIDAR_ERR_BAD_ARG;
  hipError_t e = hipMemsetAsync(grad_sigma, 0, sizeof(float) * (size_t)N * T * Z * Y * X,
                                (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  if (N == 0 || M == 0 || L == 0) return 0;
  Vol g{T, T, Z, Y, X};
  dim3 grid((M + 3) / 4, N);
  hipLaunchKernelGGL(dvxlr_scatter_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream,
                     elementwise_mult, indices, tindex, (const float*)nullptr, (const float*)nullptr,
                     grad_sigma, (float*)nullptr, M, L, g);
  return vidar_last_error();
}

int vidar_dvxlr2_get_grad_sigma_f32(const float* elementwise_mult, const float* indices,
                                    const float* tindex, const float* indicator,
                                    const float* grad_ray_pred, float* grad_sigma,
                                    float* grad_sigma_regul, int N, int M, int L, int T, int Z, int Y,
                                    int X, void* stream) {
  VIDAR_ENTER();
  if (bad_dims(N, M, T, Z, Y, X) || L < 0) return VIDAR_ERR_BAD_ARG;
  const size_t bytes = sizeof(float) * (size_t)N * T * Z * Y * X;
  hipError_t e = hipMemsetAsync(grad_sigma, 0, bytes, (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  e = hipMemsetAsync(grad_sigma_regul, 0, bytes, (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  if (N == 0 || M == 0 || L == 0) return 0;
  Vol g{T, T, Z, Y, X};
  dim3 grid((M + 3) / 4, N);
  hipLaunchKernelGGL(dvxlr_scatter_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream,
                     elementwise_mult, indices, tindex, indicator, grad_ray_pred, grad_sigma,
                     grad_sigma_regul, M, L, g);
  return vidar_last_error();
}

}  // extern "C"
