// Step-parallel dvr / dvxlr kernels (round 5) -- device side of dvr_par.h; included by dvr_family.hip.
//
// One 256-thread workgroup owns kParRays rays.  Per pass over as many of its rays as fit the LDS staging area:
//   setup   lanes (ray, axis)   the reference's per-axis terms (par_setup), element counts
//   chain 1 lanes (ray, axis)   t_a[i+1] = t_a[i] + tDelta_a into LDS                    serial, 1 add / element
//   rank    lane per element    merged position by exact comparisons (par_rank) -> LDS   parallel
//   chain 2 lanes (ray, axis)   rounded path p_a += max(0, d_s - d_{s-1}) dir_a -> voxel coordinate per step
//   consume wave per ray        lane per step: duplicate merge, density gather, fp64 wave scans for the
//                               cumulative optical depth, exp, W_k, outputs (coalesced)
// Irregular rays (dvr_par.h) and rays that do not fit fall back to the sequential per-ray code at the end.
// LDS-staged ray segments + wavefront-shuffle reductions; nothing is spilled to HBM besides the API's outputs.
#pragma once
#include "dvr_par.h"

namespace vidar_march {

constexpr int kParRays = 16;        // rays per workgroup -> 48 chain lanes
constexpr int kParThreads = 256;
constexpr int kParBudget = 2048;    // staged elements per pass: 2 x 16 KB of LDS

enum ParEmit : int { kEmitNone = 0, kEmitPark = 1, kEmitScatter = 2 };

struct ParHdr {
  double dir[3], tmax[3];
  double len;
  int v0[3], s[3], n[3], m[3];
  int last_rank[3];
  int off, elems, S, state;         // state: 0 regular & pending, 1 sequential fallback, 2 done / absent
  int ts, valid;
};

template <bool CLASSIC>
struct ParStage {
  double seq[kParBudget];           // per-axis sequences; rounded modes re-use them as 4 x int16 voxel coordinates per step
  double md[kParBudget];            // merged exit distances d_s
  short4 vox[CLASSIC ? kParBudget : 1];   // classic mode: integer voxel of every step, written by the rank phase
  ParHdr hdr[kParRays];
  int first, end;
};

__device__ __forceinline__ double wave_shfl_up_f64(double v, int delta) { return __shfl_up(v, delta, 64); }

// inclusive prefix sum over the 64 lanes (fp64)
__device__ __forceinline__ double wave_scan_f64(double v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double t = __shfl_up(v, off, 64);
    if (lane >= off) v += t;
  }
  return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
  return v;
}

struct ParResult {
  int count;           // committed samples
  int k_surface;       // first sample whose exit distance reaches the ray length (-1: none)
  double d0, S, dprev; // pred = d0 + S, dprev = exit distance of the last sample
};

// Lane-per-step integration of one ray by one wave.  steps: S entries of (voxel coordinates, d).  Chunks of 63 new
// steps; lane 0 carries the still-open sample of the previous chunk (merged mode: the pending run; otherwise the
// last sample, whose W needs the next sample's distance).
//   EMIT == kEmitPark     parks (dt, voxel id, W_{k-1}) in the ray's `indices` row like RowStager
//   EMIT == kEmitScatter  adds dl_dd * dt_k * (P_k - S_total) to grad[voxel]  (second pass of dvr.render)
template <int MODE, int EMIT>
__device__ __forceinline__ ParResult par_consume(const double* __restrict__ md, const short* __restrict__ qb, int S,
                                                 const float* __restrict__ sig, const Vol& g, double true_len,
                                                 float* __restrict__ idr, float* __restrict__ grad, double S_total,
                                                 double dl_dd, int lane) {
  constexpr bool kMerged = (MODE == kRoundedMerged);
  int k_base = 0, ksurf = 1 << 30;
  double csd_c = 0.0, T_c = 1.0, dl_c = 0.0, d0 = 0.0, ssum = 0.0, P_c = 0.0;
  bool has_carry = false;
  int c_vid = 0;
  double c_d = 0.0, c_udt = 0.0;
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int sb = 0; sb < S; sb += 63) {
    const int s = sb + lane - 1;
    const bool done = (sb + 63 >= S);
    const int last_lane = min(63, S - sb);
    bool valid = (lane == 0) ? has_carry : (s < S);
    int vid = c_vid;
    double d = c_d, lastd = 0.0, udt = c_udt;
    if (lane > 0 && valid) {
      const short4 q = reinterpret_cast<const short4*>(qb)[s];
      vid = ((int)q.z * g.Y + (int)q.y) * g.X + (int)q.x;
      d = md[s];
      lastd = (s > 0) ? md[s - 1] : 0.0;
      udt = fmax(0.0, d - lastd);
    }
    bool same = false;
    if (kMerged) {
      const int vid_up = __shfl_up(vid, 1, 64);
      const int valid_up = __shfl_up((int)valid, 1, 64);
      same = (lane > 0) && valid && (valid_up != 0) && (vid == vid_up);
      // udt_s = max(0, d_s - (d_{s-1} - udt_{s-1})) inside a run of equal voxels: a short serial recurrence;
      // iterate to the fixed point (one sweep per run depth)
      bool any = __ballot(same) != 0ull;
      while (any) {
        const double up = wave_shfl_up_f64(udt, 1);
        const double nu = same ? par_dt(d, lastd, true, up) : udt;
        any = __ballot(__double_as_longlong(nu) != __double_as_longlong(udt)) != 0ull;
        udt = nu;
      }
    }
    const int same_dn = kMerged ? __shfl_down((int)same, 1, 64) : 0;
    const bool commit = valid && ((lane == last_lane) ? done : (same_dn == 0));
    const unsigned long long cm = __ballot(commit);
    const int kl = k_base + __popcll(cm & lt);
    float sg = 0.f;
    if (commit) sg = sig[vid];
    const double sd = commit ? (double)sg * udt : 0.0;
    const double csd = csd_c + wave_scan_f64(sd, lane);
    const double T = (double)expf((float)(-csd));
    const unsigned long long below = cm & lt;
    const int pl = below ? (63 - __clzll((long long)below)) : 0;
    double Tp = __shfl(T, pl, 64), dp = __shfl(d, pl, 64);
    if (!below) { Tp = T_c; dp = dl_c; }
    double w_prev = 0.0;
    if (commit && kl > 0) w_prev = Tp * (d - dp);
    if (cm != 0ull && k_base == 0) d0 = __shfl(d, __ffsll((long long)cm) - 1, 64);
    ssum += w_prev;
    if (EMIT == kEmitPark) {
      if (commit) {
        idr[3 * kl + 0] = (float)udt;
        idr[3 * kl + 1] = (float)vid;
        if (kl > 0) idr[3 * kl + 2] = (float)w_prev;       // slot 2 of sample 0 receives the stash
        if (d >= true_len) ksurf = min(ksurf, kl);
      }
    }
    if (EMIT == kEmitScatter) {
      const double P = P_c + wave_scan_f64(w_prev, lane);
      if (commit) {
        const double gr = dl_dd * (udt * (P - S_total));
        if (gr != 0.0) unsafeAtomicAdd(grad + vid, (float)gr);
      }
      P_c = __shfl(P, 63, 64);
    }
    csd_c = __shfl(csd, 63, 64);
    if (cm != 0ull) {
      const int hl = 63 - __clzll((long long)cm);
      T_c = __shfl(T, hl, 64);
      dl_c = __shfl(d, hl, 64);
      k_base += __popcll(cm);
    }
    has_carry = !done;
    if (!done) {
      c_vid = __shfl(vid, 63, 64);
      c_d = __shfl(d, 63, 64);
      c_udt = __shfl(udt, 63, 64);
    }
  }
  ParResult R;
  R.count = k_base;
  R.d0 = d0;
  R.S = wave_sum_f64(ssum);
  R.dprev = dl_c;
  R.k_surface = -1;
  if (EMIT == kEmitPark) {
    const int ks = wave_min_i32(ksurf);
    R.k_surface = (ks == (1 << 30)) ? -1 : ks;
  }
  return R;
}

enum ParKind : int { kParForward = 0, kParDvxlr = 1, kParRender = 2 };

template <int KIND> struct ParMode;
template <> struct ParMode<kParForward> { static constexpr int mode = kRounded; };
template <> struct ParMode<kParDvxlr> { static constexpr int mode = kRoundedMerged; };
template <> struct ParMode<kParRender> { static constexpr int mode = kClassic; };

// sequential per-ray code (the lane-per-ray kernels' bodies): irregular rays of a step-parallel launch
__device__ __forceinline__ void seq_forward_ray(const float* __restrict__ sigma, const float* __restrict__ origin,
                                                const float* __restrict__ points, const float* __restrict__ tindex,
                                                float* __restrict__ pred_dist, float* __restrict__ gt_dist, int n,
                                                int c, int M, const Vol& g, int train_phase) {
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

struct GradScatter {
  float* __restrict__ grad;  // grad_sigma[n][ts] slice
  double S_total, dl_dd;
  __device__ __forceinline__ void commit(int, int vid, double, double dt, double P, double) {
    const double g = dl_dd * (dt * (P - S_total));
    if (g != 0.0) unsafeAtomicAdd(grad + vid, (float)g);
  }
};

__device__ __forceinline__ double dvr_loss_slope(int loss_type, double exp_d, double gt_d) {
  if (loss_type == 0) return (exp_d >= gt_d) ? 1.0 : -1.0;
  if (loss_type == 1) return exp_d - gt_d;
  if (loss_type == 2) return (exp_d >= gt_d) ? (1.0 / gt_d) : -(1.0 / gt_d);
  return 1.0;
}

__device__ __forceinline__ void seq_render_ray(const float* __restrict__ sigma, const float* __restrict__ origin,
                                               const float* __restrict__ points, const float* __restrict__ tindex,
                                               float* __restrict__ pred_dist, float* __restrict__ gt_dist,
                                               float* __restrict__ grad_sigma, int n, int c, int M, const Vol& g,
                                               int loss_type) {
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
      GradScatter gs{grad_sigma + slice, a.S, dvr_loss_slope(loss_type, exp_d, gt_d)};
      Integrator<kClassic, kDvrMaxD, GradScatter> b(sigma + slice, g.Y, g.X, gs);
      march<kClassic>(r, g, b);
    }
  }
  pred_dist[(size_t)n * M + c] = pred;
  gt_dist[(size_t)n * M + c] = gt;
}

// grid (ceil(M / kParRays), N), block kParThreads.  `aux`: train_phase (forward) / loss_type (render).
template <int KIND>
__global__ __launch_bounds__(kParThreads) void dvr_par_kernel(
    const float* __restrict__ sigma, const float* __restrict__ origin, const float* __restrict__ points,
    const float* __restrict__ tindex, float* __restrict__ pred_dist, float* __restrict__ gt_dist,
    float* __restrict__ indices, float* __restrict__ grad_sigma, int M, Vol g, int aux) {
  constexpr int MODE = ParMode<KIND>::mode;
  __shared__ ParStage<MODE == kClassic> st;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.y;
  const int c0 = blockIdx.x * kParRays;
  const size_t vol = (size_t)g.Z * g.Y * g.X;

  // ---- setup: lane (ray, axis); every lane evaluates the whole ray (its own axis stays in registers) ----
  const int cr = tid / 3, ca = tid - 3 * cr;
  const bool chain_lane = tid < 3 * kParRays;
  ParRay P;
  P.regular = false;
  if (chain_lane) {
    ParHdr& h = st.hdr[cr];
    const int c = c0 + cr;
    if (c < M) {
      const RayIn r = load_ray(origin, points, tindex, n, c, M, g);
      P = par_setup<MODE>(r, g);
      if (ca == 0) {
        for (int a = 0; a < 3; ++a) {
          h.dir[a] = P.ax[a].dir; h.tmax[a] = P.ax[a].tmax;
          h.v0[a] = P.ax[a].v0; h.s[a] = P.ax[a].s; h.n[a] = P.ax[a].n; h.m[a] = P.ax[a].m;
          h.last_rank[a] = -1;
        }
        h.len = P.len; h.elems = P.elems; h.S = 0; h.off = 0;
        h.state = P.regular ? 0 : 1;
        h.ts = r.ts; h.valid = r.valid ? 1 : 0;
      }
    } else if (ca == 0) {
      h.state = 2; h.elems = 0;
    }
  }
  if (tid == 0) st.first = 0;
  __syncthreads();

  // this lane's own axis as scalars (a run-time index into P.ax[] would push the struct into scratch)
  const int size_a = (ca == 0) ? g.X : (ca == 1 ? g.Y : g.Z);
  const double my_tmax = (ca == 0) ? P.ax[0].tmax : (ca == 1 ? P.ax[1].tmax : P.ax[2].tmax);
  const double my_tdelta = (ca == 0) ? P.ax[0].tdelta : (ca == 1 ? P.ax[1].tdelta : P.ax[2].tdelta);
  const double my_dir = (ca == 0) ? P.ax[0].dir : (ca == 1 ? P.ax[1].dir : P.ax[2].dir);
  const int my_m = (ca == 0) ? P.ax[0].m : (ca == 1 ? P.ax[1].m : P.ax[2].m);
  const int my_v0 = (ca == 0) ? P.ax[0].v0 : (ca == 1 ? P.ax[1].v0 : P.ax[2].v0);
  while (true) {
    // ---- allocation of this pass (thread 0) ----
    if (tid == 0) {
      int used = 0, r = st.first;
      for (; r < kParRays; ++r) {
        ParHdr& h = st.hdr[r];
        if (h.state != 0) continue;
        if (used + h.elems > kParBudget) break;
        h.off = used;
        used += h.elems;
      }
      st.end = r;
    }
    __syncthreads();
    const int first = st.first, end = st.end;
    if (first >= kParRays) break;
    const bool mine = chain_lane && cr >= first && cr < end && st.hdr[cr].state == 0;
    int off_a = 0;
    if (mine) off_a = st.hdr[cr].off + (ca == 0 ? 0 : (ca == 1 ? P.ax[0].m : P.ax[0].m + P.ax[1].m));

    // ---- chain 1: the per-axis boundary distances, the reference's own adds ----
    if (mine) {
      double t = my_tmax;
      double* out = st.seq + off_a;
      for (int i = 0; i < my_m; ++i) { out[i] = t; t += my_tdelta; }
    }
    __syncthreads();

    // ---- rank: every element finds its merged position ----
    for (int r = first + wave; r < end; r += kParThreads / 64) {
      ParHdr& h = st.hdr[r];
      if (h.state != 0) continue;
      ParRay Q;
      for (int a = 0; a < 3; ++a) {
        Q.ax[a].dir = h.dir[a]; Q.ax[a].tmax = h.tmax[a]; Q.ax[a].m = h.m[a];
      }
      const double* base = st.seq + h.off;
      const double* tb[3] = {base, base + h.m[0], base + h.m[0] + h.m[1]};
      const int E = h.elems;
      for (int e = lane; e < E; e += 64) {
        const int a = (e < h.m[0]) ? 0 : (e < h.m[0] + h.m[1] ? 1 : 2);
        const int i = e - (a == 0 ? 0 : (a == 1 ? h.m[0] : h.m[0] + h.m[1]));
        const double t = base[e];
        int before[3];
        const int k = par_rank(Q, a, i, t, tb, before);
        st.md[h.off + k] = t;
        if (i == h.m[a] - 1) h.last_rank[a] = k;
        if (MODE == kClassic)   // integer voxel of the step (before stepping): origin voxel + steps taken so far
          st.vox[h.off + k] = make_short4((short)(h.v0[0] + h.s[0] * before[0]), (short)(h.v0[1] + h.s[1] * before[1]),
                                          (short)(h.v0[2] + h.s[2] * before[2]), 0);
      }
    }
    __syncthreads();

    // ---- step count, bound check, chain 2: rounded-path voxel coordinates per step ----
    if (mine) {
      ParHdr& h = st.hdr[cr];
      int lr[3] = {h.last_rank[0], h.last_rank[1], h.last_rank[2]};
      int S = 0;
      const bool okb = par_steps(P, lr, S);
      if (!okb) {
        if (ca == 0) h.state = 1;
      } else {
        if (ca == 0) h.S = S;
        if (MODE != kClassic) {
          short* qb = reinterpret_cast<short*>(st.seq + h.off) + ca;
          const double* dm = st.md + h.off;
          double p = (double)my_v0, last = 0.0;
          const double dir = my_dir;
          for (int s = 0; s < S; ++s) {
            const double d = dm[s];
            qb[4 * s] = (short)par_round_clamp(p, size_a);
            const double adv = fmax(0.0, d - last);
            p += adv * dir;
            last = d;
          }
        }
      }
    }
    __syncthreads();

    // ---- consume: wave per ray, lane per step ----
    for (int r = first + wave; r < end; r += kParThreads / 64) {
      ParHdr& h = st.hdr[r];
      if (h.state != 0) continue;
      const int c = c0 + r;
      const size_t row = (size_t)n * M + c;
      const float* sig = sigma + ((size_t)n * g.T + h.ts) * vol;
      const double* dm = st.md + h.off;
      const short* qb = (MODE == kClassic) ? reinterpret_cast<const short*>(st.vox + h.off)
                                           : reinterpret_cast<const short*>(st.seq + h.off);
      const double len = h.len;
      float pred = -1.f, gt = -1.f;
      if (KIND == kParForward) {
        const ParResult R = par_consume<MODE, kEmitNone>(dm, qb, h.S, sig, g, len, nullptr, nullptr, 0.0, 0.0, lane);
        if (R.count > 0) {
          pred = (float)(R.d0 + R.S);
          gt = (float)(aux ? fmin(len, R.dprev) : len);
        }
      } else if (KIND == kParDvxlr) {
        float* idr = indices + row * kDvxlrMaxD * 3;
        const ParResult R = par_consume<MODE, kEmitPark>(dm, qb, h.S, sig, g, len, idr, nullptr, 0.0, 0.0, lane);
        float stash = 0.f;
        if (R.count > 0) {
          pred = (float)(R.d0 + R.S);
          gt = (float)fmin(len, R.dprev);
          stash = encode_stash(R.count, R.k_surface, false);
        }
        if (lane == 0) idr[2] = stash;
      } else {
        const ParResult R = par_consume<MODE, kEmitNone>(dm, qb, h.S, sig, g, len, nullptr, nullptr, 0.0, 0.0, lane);
        if (R.count > 0) {
          const double exp_d = R.d0 + R.S;
          const double gt_d = fmin(len, R.dprev);
          pred = (float)exp_d;
          gt = (float)gt_d;
          float* grad = grad_sigma + ((size_t)n * g.T + h.ts) * vol;
          par_consume<MODE, kEmitScatter>(dm, qb, h.S, sig, g, len, nullptr, grad, R.S,
                                          dvr_loss_slope(aux, exp_d, gt_d), lane);
        }
      }
      if (lane == 0) {
        pred_dist[row] = pred;
        gt_dist[row] = gt;
        h.state = 2;
      }
    }
    __syncthreads();
    if (tid == 0) st.first = end;
    __syncthreads();
  }

  // ---- sequential fallback: one lane per irregular ray ----
  if (tid < kParRays && st.hdr[tid].state == 1) {
    const int c = c0 + tid;
    if (KIND == kParForward) seq_forward_ray(sigma, origin, points, tindex, pred_dist, gt_dist, n, c, M, g, aux);
    else if (KIND == kParDvxlr) dvxlr_march_ray(sigma, origin, points, tindex, pred_dist, gt_dist, indices, n, c, M, g);
    else seq_render_ray(sigma, origin, points, tindex, pred_dist, gt_dist, grad_sigma, n, c, M, g, aux);
  }
}

}  // namespace vidar_march
