// Step-parallel dvr / dvxlr kernels (round 5) -- device side of dvr_par.h; included by dvr_family.hip.
//
// One 256-thread workgroup owns kParRays rays.  Per pass over as many of its rays as fit the LDS staging area:
//   setup   lanes (ray, axis)   the reference's per-axis terms (par_setup), element counts
//   chain 1 lanes (ray, axis)   t_a[i+1] = t_a[i] + tDelta_a into LDS                    serial, 1 add / element
//   rank    lane per element    merged position by exact comparisons (par_rank) -> LDS   parallel
//   chain 2 lanes (ray, axis)   rounded path p_a += max(0, d_s - d_{s-1}) dir_a -> voxel coordinate per step
//   consume wave per ray        lane per step: duplicate merge, density gather, fp64 wave scans (DPP) for the
//                               cumulative optical depth, exp, W_k, outputs (coalesced)
// Irregular rays (dvr_par.h) fall back to the sequential per-ray code at the end.
// dvxlr.render / render_v2 are ONE launch in this form: while the chain lanes of wave 0 run their serial adds, the
// other three waves stream the zero / -1 padding of the workgroup's API-mandated [1026] rows (95 % of the call's
// bytes), and the consume phase writes the final dd_dsigma / indices / ray_pred / indicator prefixes itself
// (R_k = S_total - P_k instead of a reverse scan in a second kernel).
// LDS-staged ray segments + wavefront-shuffle reductions; nothing is spilled to HBM besides the API's outputs.
#pragma once
#include "dvr_par.h"

namespace vidar_march {

constexpr int kParRays = 16;        // rays per workgroup -> 48 chain lanes
constexpr int kParThreads = 256;
constexpr int kParBudget = 1280;    // staged elements per pass: 2 x 10 KB of LDS -> 7 workgroups per CU

enum ParEmit : int { kEmitNone = 0, kEmitScatter = 2, kEmitRows = 3 };

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

// ---- cross-lane helpers: DPP moves (no LDS crossbar round trip), fp64 as two dwords ----
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp0_f64(double v) {     // lanes without a source / in masked rows read 0.0
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROW_MASK, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xF, true);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <int CTRL>
__device__ __forceinline__ int dpp0_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
constexpr int kWaveShr1 = 0x138;    // lane i reads lane i-1 (lane 0: 0)
constexpr int kWaveShl1 = 0x130;    // lane i reads lane i+1 (lane 63: 0)

// inclusive prefix sum over the 64 lanes (fp64): 4 row_shr steps inside the rows of 16, then row_bcast 15 / 31
__device__ __forceinline__ double wave_scan_f64(double v) {
  v += dpp0_f64<0x111, 0xF>(v);
  v += dpp0_f64<0x112, 0xF>(v);
  v += dpp0_f64<0x114, 0xF>(v);
  v += dpp0_f64<0x118, 0xF>(v);
  v += dpp0_f64<0x142, 0xA>(v);
  v += dpp0_f64<0x143, 0xC>(v);
  return v;
}
__device__ __forceinline__ double readlane_f64(double v, int l) {     // l wave-uniform
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)b, l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
  return v;
}

struct ParResult {
  int count;           // committed samples
  double d0, S, dprev; // pred = d0 + S, dprev = exit distance of the last sample
};

struct RowsOut {       // dvxlr rows of one ray (kEmitRows)
  float* dd;           // [1026]
  float* idx;          // [1026][3]
  float* rp;           // v2: ray_pred [1026]
  float* ind;          // v2: indicator [1026]
  const float* reg;    // v2: sigma_regul slice
  int pad_to;          // zeros / -1 up to here (the early fill starts there)
};

// Lane-per-step integration of one ray by one wave.  steps: S entries of (voxel coordinates, d).  Chunks of 63 new
// steps; lane 0 carries the still-open sample of the previous chunk (merged mode: the pending run; otherwise the
// last sample, whose W needs the next sample's distance).
//   EMIT == kEmitScatter  adds dl_dd * dt_k * (P_k - S_total) to grad[voxel]          (second pass of dvr.render)
//   EMIT == kEmitRows     writes the final dvxlr rows: dd_dsigma[k] = -dt_k (S_total - P_k), (z, y, x), v2 extras.
//                         S_total: known (a first kEmitNone pass) when the ray has more than one chunk, else taken
//                         from this chunk's own scan.
template <int MODE, int EMIT, bool V2>
__device__ __forceinline__ ParResult par_consume(const double* __restrict__ md, const short* __restrict__ qb, int S,
                                                 const float* __restrict__ sig, const Vol& g, double true_len,
                                                 const RowsOut& rows, float* __restrict__ grad, double S_total,
                                                 double dl_dd, int lane, float pre0, float pre1) {
  constexpr bool kMerged = (MODE == kRoundedMerged);
  int k_base = 0;
  float c_sg = 0.f;
  double csd_c = 0.0, T_c = 1.0, dl_c = 0.0, d0 = 0.0, P_c = 0.0;
  bool has_carry = false;
  int c_vid = 0, c_qx = 0, c_qy = 0, c_qz = 0;
  double c_d = 0.0, c_udt = 0.0;
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const bool single = (S <= 63);
  for (int sb = 0; sb < S; sb += 63) {
    const int s = sb + lane - 1;
    const bool done = (sb + 63 >= S);
    const int last_lane = min(63, S - sb);
    const bool valid = (lane == 0) ? has_carry : (s < S);
    int vid = c_vid, qx = c_qx, qy = c_qy, qz = c_qz;
    double d = c_d, lastd = 0.0, udt = c_udt;
    // densities of the first two chunks were requested before the wave started on its rays (a gather round trip per
    // chunk in the middle of the scans was the largest single cost of this phase); later chunks load here
    float sg = (sb == 0) ? pre0 : (sb == 63 ? pre1 : 0.f);
    if (lane == 0) sg = c_sg;
    if (lane > 0 && valid) {
      const short4 q = reinterpret_cast<const short4*>(qb)[s];
      qx = q.x; qy = q.y; qz = q.z;
      vid = (qz * g.Y + qy) * g.X + qx;
      if (sb > 63) sg = sig[vid];
      d = md[s];
      lastd = (s > 0) ? md[s - 1] : 0.0;
      udt = fmax(0.0, d - lastd);
    }
    bool same = false;
    if (kMerged) {
      const int vid_up = dpp0_i32<kWaveShr1>(vid);
      const int valid_up = dpp0_i32<kWaveShr1>((int)valid);
      same = (lane > 0) && valid && (valid_up != 0) && (vid == vid_up);
      // udt_s = max(0, d_s - (d_{s-1} - udt_{s-1})) inside a run of equal voxels: a short serial recurrence;
      // iterate to the fixed point (one sweep per run depth)
      bool any = __ballot(same) != 0ull;
      while (any) {
        const double up = dpp0_f64<kWaveShr1, 0xF>(udt);
        const double nu = same ? par_dt(d, lastd, true, up) : udt;
        any = __ballot(__double_as_longlong(nu) != __double_as_longlong(udt)) != 0ull;
        udt = nu;
      }
    }
    const int same_dn = kMerged ? dpp0_i32<kWaveShl1>((int)same) : 0;
    const bool commit = valid && ((lane == last_lane) ? done : (same_dn == 0));
    const unsigned long long cm = __ballot(commit);
    const int kl = k_base + __popcll(cm & lt);
    const double sd = commit ? (double)sg * udt : 0.0;
    const double csd = csd_c + wave_scan_f64(sd);
    const double T = (double)expf((float)(-csd));
    const unsigned long long below = cm & lt;
    const int pl = below ? (63 - __clzll((long long)below)) : 0;
    double Tp = __shfl(T, pl, 64), dp = __shfl(d, pl, 64);
    if (!below) { Tp = T_c; dp = dl_c; }
    double w_prev = 0.0;
    if (commit && kl > 0) w_prev = Tp * (d - dp);
    const int ncommit = __popcll(cm);
    if (ncommit != 0 && k_base == 0) d0 = readlane_f64(d, __ffsll((long long)cm) - 1);
    const double P = P_c + wave_scan_f64(w_prev);           // P_k = sum_{i <= k} W_{i-1}
    if (EMIT == kEmitRows) {
      const double St = single ? readlane_f64(P, 63) : S_total;
      if (commit) {
        rows.dd[kl] = (float)(-udt * (St - P));
        float* id = rows.idx + 3 * kl;
        id[0] = (float)qz; id[1] = (float)qy; id[2] = (float)qx;
        if (V2) {
          rows.rp[kl] = rows.reg[vid];
          // first sample whose exit distance reaches the un-clamped ray length (d is non-decreasing in k)
          rows.ind[kl] = (d >= true_len && (kl == 0 || !(dp >= true_len))) ? 1.f : 0.f;
        }
      }
    }
    if (EMIT == kEmitScatter) {
      if (commit) {
        const double gr = dl_dd * (udt * (P - S_total));
        if (gr != 0.0) unsafeAtomicAdd(grad + vid, (float)gr);
      }
    }
    P_c = readlane_f64(P, 63);
    csd_c = readlane_f64(csd, 63);
    if (ncommit != 0) {
      const int hl = 63 - __clzll((long long)cm);
      T_c = readlane_f64(T, hl);
      dl_c = readlane_f64(d, hl);
      k_base += ncommit;
    }
    has_carry = !done;
    if (!done) {
      c_vid = __builtin_amdgcn_readlane(vid, 63);
      c_sg = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sg), 63));
      c_d = readlane_f64(d, 63);
      c_udt = readlane_f64(udt, 63);
      if (EMIT == kEmitRows) {   // the carried sample's coordinates travel with it
        c_qx = __builtin_amdgcn_readlane(qx, 63); c_qy = __builtin_amdgcn_readlane(qy, 63);
        c_qz = __builtin_amdgcn_readlane(qz, 63);
      }
    }
  }
  if (EMIT == kEmitRows) {        // the gap between the live prefix and the early fill
    for (int k = k_base + lane; k < rows.pad_to; k += 64) {
      rows.dd[k] = 0.f;
      rows.idx[3 * k] = 0.f; rows.idx[3 * k + 1] = 0.f; rows.idx[3 * k + 2] = 0.f;
      if (V2) { rows.rp[k] = 0.f; rows.ind[k] = -1.f; }
    }
  }
  ParResult R;
  R.count = k_base;
  R.d0 = d0;
  R.S = P_c;
  R.dprev = dl_c;
  return R;
}

// inclusive wave sum at every lane -> total at lane 63 (fp64, DPP)
__device__ __forceinline__ double wave_total_f64(double v) { return readlane_f64(wave_scan_f64(v), 63); }

// Lean lane-per-step integrator of the modes WITHOUT the duplicate merge (dvr.render_forward, dvr.render): every step
// is a sample, so lane = step, the neighbouring distances come straight from LDS and nothing is carried between
// chunks except the running optical depth:  W_s = exp(-csd_s) (d_{s+1} - d_s),  pred = d_0 + sum_s W_s.
//   SCATTER: second pass of dvr.render -- grad[voxel_s] += dl_dd * dt_s * (P_s - S_total), P_s = sum_{j < s} W_j
template <bool SCATTER>
__device__ __forceinline__ ParResult par_consume_plain(const double* __restrict__ md, const short* __restrict__ qb,
                                                       int S, const float* __restrict__ sig, const Vol& g,
                                                       float* __restrict__ grad, double S_total, double dl_dd,
                                                       int lane, float pre0, float pre1) {
  double csd_c = 0.0, acc = 0.0, P_c = 0.0;
  for (int sb = 0; sb < S; sb += 64) {
    const int s = sb + lane;
    const bool valid = s < S;
    double d = 0.0, dp = 0.0, dn = 0.0;
    float sg = 0.f;
    int vid = 0;
    if (valid) {
      const short4 q = reinterpret_cast<const short4*>(qb)[s];
      vid = ((int)q.z * g.Y + (int)q.y) * g.X + (int)q.x;
      sg = (sb == 0) ? pre0 : (sb == 64 ? pre1 : sig[vid]);     // first two chunks: requested ahead (see the kernel)
      d = md[s];
      dp = (s > 0) ? md[s - 1] : 0.0;
      dn = (s + 1 < S) ? md[s + 1] : d;
    }
    const double dt = fmax(0.0, d - dp);
    const double csd = csd_c + wave_scan_f64(valid ? (double)sg * dt : 0.0);
    const double T = (double)expf((float)(-csd));
    const double w = valid ? T * (dn - d) : 0.0;              // W_s; 0 for the last sample (dn == d)
    if (SCATTER) {
      const double incl = wave_scan_f64(w);
      const double P = P_c + (incl - w);                      // exclusive: sum_{j < s} W_j
      if (valid) {
        const double gr = dl_dd * (dt * (P - S_total));
        if (gr != 0.0) unsafeAtomicAdd(grad + vid, (float)gr);
      }
      P_c += readlane_f64(incl, 63);
    } else {
      acc += w;
    }
    csd_c = readlane_f64(csd, 63);
  }
  ParResult R;
  R.count = S;
  R.d0 = md[0];
  R.dprev = md[S - 1];
  R.S = SCATTER ? P_c : wave_total_f64(acc);
  return R;
}

enum ParKind : int { kParForward = 0, kParDvxlr = 1, kParRender = 2, kParDvxlrV2 = 3 };

template <int KIND> struct ParMode;
template <> struct ParMode<kParForward> { static constexpr int mode = kRounded; };
template <> struct ParMode<kParDvxlr> { static constexpr int mode = kRoundedMerged; };
template <> struct ParMode<kParRender> { static constexpr int mode = kClassic; };
template <> struct ParMode<kParDvxlrV2> { static constexpr int mode = kRoundedMerged; };

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

// Finish pass of one parked dvxlr row by one wave (the body of dvxlr_finish_kernel, dvr_family.hip): reverse wave
// scan of the parked W -> suffix sums, dd = -dt R, (z, y, x) unpacked, v2 extras.  Used by the two-launch form and,
// in the one-launch form, for the rows of irregular rays (parked by the sequential march).
template <bool V2>
__device__ __forceinline__ int dvxlr_finish_row(const float* __restrict__ reg, float* __restrict__ ddr,
                                                float* __restrict__ idr, float* __restrict__ rpr,
                                                float* __restrict__ inr, const Vol& g, int lane) {
  int cnt, ks;
  bool nan_tail;
  decode_stash(idr[2], cnt, ks, nan_tail);
  // chunks of 64 samples from the far end; a lane only touches the slots of its own sample.
  // W_k sits in sample k+1's slot: neighbour lane, or lane 0 of the chunk handled just before.
  double carry = nan_tail ? (double)NAN : 0.0;
  float w_above = 0.f;
  for (int base = cnt > 0 ? ((cnt - 1) / 64) * 64 : -1; base >= 0; base -= 64) {
    const int k = base + lane;
    Parked p{0.f, 0.f, 0.f};
    if (k < cnt) p = reinterpret_cast<const Parked*>(idr)[k];
    float w = __shfl_down(p.w_prev, 1, 64);
    if (lane == 63) w = w_above;
    w_above = __shfl(p.w_prev, 0, 64);
    double sfx = (k < cnt - 1) ? (double)w : 0.0;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const double t = __shfl_down(sfx, off, 64);
      if (lane + off < 64) sfx += t;
    }
    const double R = sfx + carry;
    carry += __shfl(sfx, 0, 64);
    if (k < cnt) {
      const int vid = (int)p.vid;
      const int zy = vid / g.X, x = vid - zy * g.X;
      const int z = zy / g.Y, y = zy - z * g.Y;
      ddr[k] = (float)(-(double)p.dt * R);
      idr[3 * k + 0] = (float)z;
      idr[3 * k + 1] = (float)y;
      idr[3 * k + 2] = (float)x;
      if (V2) {
        rpr[k] = reg[vid];
        inr[k] = (k == ks) ? 1.f : 0.f;
      }
    }
  }
  return cnt;
}

// A wave stores `val` over the floats [a, b) of `base` (16-byte aligned): float4 body, scalar head / tail.
__device__ __forceinline__ void wave_fill(float* __restrict__ base, size_t a, size_t b, float val, int lane) {
  if (a >= b) return;
  size_t a4 = (a + 3) & ~(size_t)3, b4 = b & ~(size_t)3;
  if (a4 > b4) { a4 = b; b4 = b; }
  if (a + lane < a4) base[a + lane] = val;
  const float4 v4 = make_float4(val, val, val, val);
  float4* p4 = reinterpret_cast<float4*>(base);
  for (size_t i = (a4 >> 2) + lane; i < (b4 >> 2); i += 64) p4[i] = v4;
  if (b4 + lane < b) base[b4 + lane] = val;
}

#ifdef VIDAR_PAR_TIMING   // experiment only: cycles per phase, summed over workgroups (wave 0)
__device__ unsigned long long g_par_cycles[8];
#define PAR_STAMP(i)                                                       \
  do {                                                                     \
    if (tid == 0) {                                                        \
      const unsigned long long now_ = __builtin_readcyclecounter();        \
      atomicAdd(&g_par_cycles[i], now_ - stamp_);                          \
      stamp_ = now_;                                                       \
    }                                                                      \
  } while (0)
#else
#define PAR_STAMP(i) do {} while (0)
#endif

// grid (ceil(M / kParRays), N), block kParThreads.  `aux`: train_phase (forward) / loss_type (render).
template <int KIND>
__global__ __launch_bounds__(kParThreads) void dvr_par_kernel(
    const float* __restrict__ sigma, const float* __restrict__ sigma_regul, const float* __restrict__ origin,
    const float* __restrict__ points, const float* __restrict__ tindex, float* __restrict__ pred_dist,
    float* __restrict__ gt_dist, float* __restrict__ dd_dsigma, float* __restrict__ indices,
    float* __restrict__ ray_pred, float* __restrict__ indicator, float* __restrict__ grad_sigma, int M, Vol g,
    int aux, int pad_wgs) {
  constexpr int MODE = ParMode<KIND>::mode;
  constexpr bool kRows = (KIND == kParDvxlr || KIND == kParDvxlrV2);
  constexpr bool V2 = (KIND == kParDvxlrV2);
  constexpr int L = kDvxlrMaxD;
  __shared__ ParStage<MODE == kClassic> st;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.y;
  const size_t vol = (size_t)g.Z * g.Y * g.X;

  // ---- one-launch dvxlr: the FIRST `pad_wgs` workgroups of the grid (about one per CU, dispatched first) only PAD,
  // persistently: each walks blocks of 64 rays and writes the zeros / -1 of their rows from each ray's element bound
  // (the same par_setup as the compute workgroups, so the two never touch the same bytes) to the row end.  95 % of
  // the call's bytes stream at fill rate for the whole launch while the compute workgroups, which share no barrier
  // with them, take the remaining wave slots.  (Measured alternatives: padding from the compute workgroups' idle
  // waves serialises behind their barriers; one padding workgroup per compute workgroup takes half the wave slots;
  // padding workgroups at the end of the grid run after the march -- profiles/r05_kbench_dvr_traversal.log) ----
  if (kRows && (int)blockIdx.x < pad_wgs) {
    int* bound = reinterpret_cast<int*>(st.seq);
    const int nblk = (M + 63) / 64;
    for (int blk = (int)blockIdx.x; blk < nblk; blk += pad_wgs) {
      const int c0f = blk * 64;
      const int nrows = min(64, M - c0f);
      if (tid < nrows) {
        const RayIn rf = load_ray(origin, points, tindex, n, c0f + tid, M, g);
        const ParRay F = par_setup<MODE>(rf, g);
        // regular ray: from its element bound.  Padded ray (no samples): the whole row (the compute workgroup only
        // stores the same 0.0 stash into it).  Valid but irregular ray: nothing -- its row belongs to the compute
        // workgroup's sequential fallback, which pads it itself.
        bound[tid] = F.regular ? F.elems : (rf.valid ? L : 0);
      }
      __syncthreads();
      const size_t row0 = (size_t)n * M + c0f;
      for (int r = wave; r < nrows; r += kParThreads / 64) {      // a wave pads whole rows
        const size_t e = (size_t)bound[r];
        const size_t row = row0 + r;
        wave_fill(dd_dsigma, row * L + e, (row + 1) * L, 0.f, lane);
        wave_fill(indices, (row * L + e) * 3, (row + 1) * L * 3, 0.f, lane);
        if (V2) {
          wave_fill(ray_pred, row * L + e, (row + 1) * L, 0.f, lane);
          wave_fill(indicator, row * L + e, (row + 1) * L, -1.f, lane);
        }
      }
      __syncthreads();
    }
    return;
  }
  const int c0 = ((int)blockIdx.x - (kRows ? pad_wgs : 0)) * kParRays;
#ifdef VIDAR_PAR_TIMING
  unsigned long long stamp_ = __builtin_readcyclecounter();
#endif

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
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          h.dir[a] = P.ax[a].dir; h.tmax[a] = P.ax[a].tmax;
          h.v0[a] = P.ax[a].v0; h.s[a] = P.ax[a].s; h.n[a] = P.ax[a].n; h.m[a] = P.ax[a].m;
          h.last_rank[a] = -1;
        }
        h.len = P.len; h.elems = P.regular ? P.elems : 0; h.S = 0; h.off = 0;
        h.state = P.regular ? 0 : 1;
        h.ts = r.ts; h.valid = r.valid ? 1 : 0;
      }
    } else if (ca == 0) {
      h.state = 2; h.elems = 0; h.valid = 0;
    }
  }
  if (tid == 0) st.first = 0;
  __syncthreads();
  PAR_STAMP(0);

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
    PAR_STAMP(1);

    // ---- rank: every element finds its merged position ----
    for (int r = first + wave; r < end; r += kParThreads / 64) {
      ParHdr& h = st.hdr[r];
      if (h.state != 0) continue;
      ParRay Q;
#pragma unroll
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
    PAR_STAMP(2);

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
          int s = 0;
          for (; s + 4 <= S; s += 4) {          // the distances of four steps are requested before the serial adds
            const double d0 = dm[s], d1 = dm[s + 1], d2 = dm[s + 2], d3 = dm[s + 3];
            qb[4 * s] = (short)par_round_clamp(p, size_a);
            p += fmax(0.0, d0 - last) * dir;
            qb[4 * s + 4] = (short)par_round_clamp(p, size_a);
            p += fmax(0.0, d1 - d0) * dir;
            qb[4 * s + 8] = (short)par_round_clamp(p, size_a);
            p += fmax(0.0, d2 - d1) * dir;
            qb[4 * s + 12] = (short)par_round_clamp(p, size_a);
            p += fmax(0.0, d3 - d2) * dir;
            last = d3;
          }
          for (; s < S; ++s) {
            const double d = dm[s];
            qb[4 * s] = (short)par_round_clamp(p, size_a);
            p += fmax(0.0, d - last) * dir;
            last = d;
          }
        }
      }
    }
    __syncthreads();
    PAR_STAMP(3);

    // ---- consume: wave per ray, lane per step.  First the density gathers of the first two chunks of all (up to 4)
    // rays of this wave are put in flight together ----
    constexpr int kChunk = (MODE == kRoundedMerged) ? 63 : 64;
    float pre[4][2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      pre[k][0] = 0.f; pre[k][1] = 0.f;
      const int r = first + wave + 4 * k;
      if (r < end && st.hdr[r].state == 0) {
        const ParHdr& h = st.hdr[r];
        const float* sig = sigma + ((size_t)n * g.T + h.ts) * vol;
        const short4* qv = (MODE == kClassic) ? reinterpret_cast<const short4*>(st.vox + h.off)
                                              : reinterpret_cast<const short4*>(st.seq + h.off);
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          const int s = cc * kChunk + lane - (kChunk == 63 ? 1 : 0);
          if (s >= 0 && s < h.S) {
            const short4 q = qv[s];
            pre[k][cc] = sig[((int)q.z * g.Y + (int)q.y) * g.X + (int)q.x];
          }
        }
      }
    }
    int kk = 0;
    for (int r = first + wave; r < end; r += kParThreads / 64, ++kk) {
      ParHdr& h = st.hdr[r];
      if (h.state != 0) continue;
      const float pre0 = kk == 0 ? pre[0][0] : (kk == 1 ? pre[1][0] : (kk == 2 ? pre[2][0] : pre[3][0]));
      const float pre1 = kk == 0 ? pre[0][1] : (kk == 1 ? pre[1][1] : (kk == 2 ? pre[2][1] : pre[3][1]));
      const int c = c0 + r;
      const size_t row = (size_t)n * M + c;
      const float* sig = sigma + ((size_t)n * g.T + h.ts) * vol;
      const double* dm = st.md + h.off;
      const short* qb = (MODE == kClassic) ? reinterpret_cast<const short*>(st.vox + h.off)
                                           : reinterpret_cast<const short*>(st.seq + h.off);
      const double len = h.len;
      const int S = h.S;
      float pred = -1.f, gt = -1.f;
      RowsOut none{};
      if (KIND == kParForward) {
        const ParResult R = par_consume_plain<false>(dm, qb, S, sig, g, nullptr, 0.0, 0.0, lane, pre0, pre1);
        if (R.count > 0) {
          pred = (float)(R.d0 + R.S);
          gt = (float)(aux ? fmin(len, R.dprev) : len);
        }
      } else if (kRows) {
        RowsOut ro;
        ro.dd = dd_dsigma + row * L; ro.idx = indices + row * L * 3;
        ro.rp = V2 ? ray_pred + row * L : nullptr; ro.ind = V2 ? indicator + row * L : nullptr;
        ro.reg = V2 ? sigma_regul + ((size_t)n * g.T + h.ts) * vol : nullptr;
        ro.pad_to = h.elems;
        double St = 0.0;
        if (S > 63)
          St = par_consume<MODE, kEmitNone, false>(dm, qb, S, sig, g, len, none, nullptr, 0.0, 0.0, lane, pre0, pre1).S;
        const ParResult R = par_consume<MODE, kEmitRows, V2>(dm, qb, S, sig, g, len, ro, nullptr, St, 0.0, lane, pre0,
                                                              pre1);
        if (R.count > 0) {
          pred = (float)(R.d0 + R.S);
          gt = (float)fmin(len, R.dprev);
        }
      } else {
        const ParResult R = par_consume_plain<false>(dm, qb, S, sig, g, nullptr, 0.0, 0.0, lane, pre0, pre1);
        if (R.count > 0) {
          const double exp_d = R.d0 + R.S;
          const double gt_d = fmin(len, R.dprev);
          pred = (float)exp_d;
          gt = (float)gt_d;
          float* grad = grad_sigma + ((size_t)n * g.T + h.ts) * vol;
          par_consume_plain<true>(dm, qb, S, sig, g, grad, R.S, dvr_loss_slope(aux, exp_d, gt_d), lane, pre0, pre1);
        }
      }
      if (lane == 0) {
        pred_dist[row] = pred;
        gt_dist[row] = gt;
        h.state = 2;
      }
    }
    __syncthreads();
    PAR_STAMP(4);
    if (tid == 0) st.first = end;
    __syncthreads();
  }
  // ---- sequential fallback: one lane per irregular ray ----
  if (tid < kParRays && st.hdr[tid].state == 1) {
    const int c = c0 + tid;
    if (KIND == kParForward) seq_forward_ray(sigma, origin, points, tindex, pred_dist, gt_dist, n, c, M, g, aux);
    else if (kRows) dvxlr_march_ray(sigma, origin, points, tindex, pred_dist, gt_dist, indices, n, c, M, g);
    else seq_render_ray(sigma, origin, points, tindex, pred_dist, gt_dist, grad_sigma, n, c, M, g, aux);
  }
  if (kRows) {                        // ... whose parked rows are finished by a wave each (their padding is in place)
    __syncthreads();
    for (int r = wave; r < kParRays; r += kParThreads / 64) {
      if (st.hdr[r].state != 1) continue;
      const size_t row = (size_t)n * M + c0 + r;
      const float* reg = nullptr;
      if (V2 && st.hdr[r].valid) reg = sigma_regul + ((size_t)n * g.T + st.hdr[r].ts) * vol;
      const int cnt = dvxlr_finish_row<V2>(reg, dd_dsigma + row * L, indices + row * L * 3,
                                           V2 ? ray_pred + row * L : nullptr, V2 ? indicator + row * L : nullptr, g,
                                           lane);
      // the rest of the row: a valid irregular ray's row is not touched by the padding workgroup; a ray whose bound
      // check failed (mathematically impossible, kept as a net) was padded from its element bound
      const int pad_to = !st.hdr[r].valid ? 0 : (st.hdr[r].elems > 0 ? st.hdr[r].elems : L);
      for (int k = cnt + lane; k < pad_to; k += 64) {
        dd_dsigma[row * L + k] = 0.f;
        float* id = indices + (row * L + k) * 3;
        id[0] = 0.f; id[1] = 0.f; id[2] = 0.f;
        if (V2) { ray_pred[row * L + k] = 0.f; indicator[row * L + k] = -1.f; }
      }
    }
  }
}

}  // namespace vidar_march
