// TEST INFRASTRUCTURE.  Compiles vidar_amd/csrc/dvr_march.h (the traversal + integrator every
// dvr / dvxlr kernel instantiates per lane) with g++ and drives it with scalar loops that mirror
// what the kernels of dvr_family.hip do around it, so that the traversal logic can be compared
// with the oracle on a machine without a GPU (tests/test_march_host_cpu.py).  Never used by the
// product.
#include <stdint.h>
#include <string.h>
#include <vector>

#include "dvr_march.h"
#include "dvr_par.h"

using namespace vidar_march;

// -DVIDAR_MARCH_PAR: every traversal below runs in its step-parallel form (dvr_par.h: per-axis chains, merge by
// rank, rounded-path chains) instead of the sequential loop; the integrator and everything after it are shared.
#ifdef VIDAR_MARCH_PAR
struct HostTraversal {
  template <int MODE, class Sink>
  double run(const RayIn& r, const Vol& g, Sink& sink) const { return march_par<MODE>(r, g, sink); }
};
#else
using HostTraversal = SequentialTraversal;
#endif
static int g_par_regular = 0, g_par_total = 0;

extern "C" {

int host_dvr_render_forward(const float* sigma, const float* origin, const float* points,
                            const float* tindex, float* pred_dist, float* gt_dist, int N, int M,
                            int T, int TO, int Z, int Y, int X, int train_phase) {
  Vol g{T, TO, Z, Y, X};
  const size_t vol = (size_t)Z * Y * X;
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < M; ++c) {
      float pred = -1.f, gt = -1.f;
      const RayIn r = load_ray(origin, points, tindex, n, c, M, g);
      if (r.valid) {
        NoEmit ne;
        Integrator<kRounded, kDvrMaxD, NoEmit> a(sigma + ((size_t)n * T + r.ts) * vol, Y, X, ne);
        const double len = HostTraversal().run<kRounded>(r, g, a);
        if (a.k > 0) {
          pred = (float)(a.d0 + a.S);
          gt = (float)(train_phase ? fmin(len, a.dprev) : len);
        }
      }
      pred_dist[(size_t)n * M + c] = pred;
      gt_dist[(size_t)n * M + c] = gt;
    }
  return 0;
}

struct HostGrad {
  double* grad;
  double S_total, dl_dd;
  void commit(int, int vid, double, double dt, double P, double) {
    grad[vid] += (double)(float)(dl_dd * (dt * (P - S_total)));
  }
};

int host_dvr_render(const float* sigma, const float* origin, const float* points,
                    const float* tindex, float* pred_dist, float* gt_dist, double* grad_sigma,
                    int N, int M, int T, int TO, int Z, int Y, int X, int loss_type) {
  Vol g{T, TO, Z, Y, X};
  const size_t vol = (size_t)Z * Y * X;
  memset(grad_sigma, 0, sizeof(double) * N * T * vol);
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < M; ++c) {
      float pred = -1.f, gt = -1.f;
      const RayIn r = load_ray(origin, points, tindex, n, c, M, g);
      if (r.valid) {
        const size_t slice = ((size_t)n * T + r.ts) * vol;
        NoEmit ne;
        Integrator<kClassic, kDvrMaxD, NoEmit> a(sigma + slice, Y, X, ne);
        const double len = HostTraversal().run<kClassic>(r, g, a);
        if (a.k > 0) {
          const double exp_d = a.d0 + a.S, gt_d = fmin(len, a.dprev);
          pred = (float)exp_d; gt = (float)gt_d;
          double dl = 1.0;
          if (loss_type == 0) dl = (exp_d >= gt_d) ? 1.0 : -1.0;
          else if (loss_type == 1) dl = exp_d - gt_d;
          else if (loss_type == 2) dl = (exp_d >= gt_d) ? (1.0 / gt_d) : -(1.0 / gt_d);
          HostGrad hg{grad_sigma + slice, a.S, dl};
          Integrator<kClassic, kDvrMaxD, HostGrad> b(sigma + slice, Y, X, hg);
          HostTraversal().run<kClassic>(r, g, b);
        }
      }
      pred_dist[(size_t)n * M + c] = pred;
      gt_dist[(size_t)n * M + c] = gt;
    }
  return 0;
}

// dvxlr.render / render_v2: launch 1 is the shared dvxlr_march_ray(); launch 2 (the finish pass of
// dvr_family.hip, a wave scan on the GPU) is mirrored here as a scalar loop that reads the same
// stash slots.  Output buffers arrive uninitialised (filled with a poison value by the test).
int host_dvxlr_render(const float* sigma, const float* sigma_regul, const float* origin,
                      const float* points, const float* tindex, float* pred_dist, float* gt_dist,
                      float* dd_dsigma, float* indices, float* ray_pred, float* indicator,
                      int32_t* est_steps, int N, int M, int T, int TO, int Z, int Y, int X) {
  constexpr int L = kDvxlrMaxD;
  Vol g{T, TO, Z, Y, X};
  const size_t vol = (size_t)Z * Y * X;
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < M; ++c) {
      est_steps[(size_t)n * M + c] = estimate_steps(load_ray(origin, points, tindex, n, c, M, g), g);
      dvxlr_march_ray(sigma, origin, points, tindex, pred_dist, gt_dist, indices, n, c, M, g, HostTraversal());
      const RayIn rr = load_ray(origin, points, tindex, n, c, M, g);
      if (rr.valid) { ++g_par_total; g_par_regular += par_setup<kRoundedMerged>(rr, g).regular ? 1 : 0; }
    }
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < M; ++c) {
      const size_t row = (size_t)n * M + c;
      float* ddr = dd_dsigma + row * L;
      float* idr = indices + row * L * 3;
      int cnt, ks;
      bool nan_tail;
      decode_stash(idr[2], cnt, ks, nan_tail);
      const float* reg = nullptr;
      if (sigma_regul && cnt > 0) {
        const long ti = (long)tindex[row];
        reg = sigma_regul + ((size_t)n * T + (T == 1 ? 0 : ti)) * vol;
      }
      double R = nan_tail ? (double)NAN : 0.0;
      float w_above = 0.f;                     // W_k lives in sample k+1's slot
      for (int k = cnt - 1; k >= 0; --k) {
        const Parked p = reinterpret_cast<const Parked*>(idr)[k];
        if (k < cnt - 1) R += (double)w_above;
        w_above = p.w_prev;
        const int vid = (int)p.vid;
        const int zy = vid / X, x = vid - zy * X;
        const int z = zy / Y, y = zy - z * Y;
        ddr[k] = (float)(-(double)p.dt * R);
        idr[3 * k + 0] = (float)z; idr[3 * k + 1] = (float)y; idr[3 * k + 2] = (float)x;
        if (sigma_regul) {
          ray_pred[row * L + k] = reg[vid];
          indicator[row * L + k] = (k == ks) ? 1.f : 0.f;
        }
      }
      for (int k = cnt; k < L; ++k) {
        ddr[k] = 0.f;
        idr[3 * k + 0] = idr[3 * k + 1] = idr[3 * k + 2] = 0.f;
        if (sigma_regul) { ray_pred[row * L + k] = 0.f; indicator[row * L + k] = -1.f; }
      }
    }
  return 0;
}

// par_round_clamp (trunc + exact remainder) against the library round() it replaces; returns the mismatch count
int host_round_clamp_mismatches(const double* p, int count, int size) {
  int bad = 0;
  for (int i = 0; i < count; ++i) {
    int q = (int)round(p[i]);
    q = q < size ? q : size - 1;
    q = q >= 0 ? q : 0;
    bad += (q != par_round_clamp(p[i], size)) ? 1 : 0;
  }
  return bad;
}

// how many of the valid rays seen by host_dvxlr_render so far were of the regular (step-parallel) class
int host_par_regular_count(int* total) { *total = g_par_total; return g_par_regular; }

}  // extern "C"
