// Step-parallel form of the dvr / dvxlr ray traversal (round 5).
//
// The reference's DDA (third_lib/dvr/dvr.cu:232-251 / :528-547, third_lib/dvxlr/dvxlr.cu:334-353) advances
// `tMaxX += tDeltaX` only on X steps: the three per-axis boundary-distance sequences
//        t_a[0] = tMax_a,   t_a[i+1] = t_a[i] + tDelta_a                      (one fp64 add per element)
// are INDEPENDENT chains, and the order in which the loop consumes them is a comparison-only three-way merge
// with a fixed tie priority (Z over Y over X, from the `<` nest).  So a ray's steps need no serial loop of
// ~150 dependent instructions per step:
//   1. chain lanes (ray, axis) generate the per-axis sequences with the reference's own adds        (serial: 1 add / element)
//   2. every element finds its position in the merged order by exact comparisons against the other
//      two sequences (a closed-form estimate + a correction loop on the stored values)             (parallel)
//   3. chain lanes (ray, axis) run the rounded-path recurrence p_a += max(0, d_s - d_{s-1}) * dir_a
//      (dvr.cu:256-258) over the merged distances, in order                                        (serial: 1 add / step)
//   4. rounding, duplicate merge, density loads, the exp integral run lane-per-step               (parallel)
// Every value that decides a voxel index is produced by the same fp64 operations in the same order as in the
// sequential march (dvr_march.h) -- tests/march_host.cpp compiles this header with g++ and
// tests/test_march_host_cpu.py proves the voxel lists / counts / gt_dist bit-equal before a GPU is involved.
//
// Rays outside the "regular" class (origin voxel outside the volume, zero / non-finite direction terms,
// more elements than the staging area holds, or a failed bound check) take the sequential march.
#pragma once
#include "dvr_march.h"

namespace vidar_march {

constexpr int kParMaxElems = 1024;   // <= kDvxlrMaxD: a regular ray can never reach the sample cap

struct ParAxis {
  double dir, tmax, tdelta;
  int v0, s, n, m;   // n: elements until the ray leaves the volume along this axis; m: elements generated
};

struct ParRay {
  ParAxis ax[3];     // x, y, z
  double len;
  bool regular;
  int elems;         // m_x + m_y + m_z
};

// Per-axis terms exactly as march() computes them (same expressions, same order).
template <int MODE>
VIDAR_DEV ParRay par_setup(const RayIn& r, const Vol& g) {
  ParRay P;
  const double o[3] = {r.xo, r.yo, r.zo};
  const double e[3] = {r.xe, r.ye, r.ze};
  const int size[3] = {g.X, g.Y, g.Z};
  const double rx = e[0] - o[0], ry = e[1] - o[1], rz = e[2] - o[2];
  const double len = sqrt(rx * rx + ry * ry + rz * rz);
  const double rr[3] = {rx, ry, rz};
  P.len = len;
  bool ok = r.valid && (len > 0.0) && (len < DBL_MAX);
  const int back = (MODE == kClassic) ? 0 : -1;
  double t_exit = DBL_MAX;
  for (int a = 0; a < 3; ++a) {
    ParAxis& A = P.ax[a];
    ok = ok && (o[a] > -1.0) && (o[a] < 2147483000.0);       // (int) conversion defined; NaN fails
    A.v0 = ok ? (int)o[a] : 0;
    A.dir = rr[a] / len;
    A.s = (A.dir >= 0) ? 1 : -1;
    const double b = A.v0 + (A.s < 0 ? back : 1);
    A.tmax = (A.dir != 0) ? (b - o[a]) / A.dir : DBL_MAX;
    A.tdelta = (A.dir != 0) ? A.s / A.dir : DBL_MAX;
    ok = ok && ((unsigned)A.v0 < (unsigned)size[a]);
    A.n = 0;
    if (A.dir != 0) {
      A.n = (A.s > 0) ? size[a] - A.v0 : A.v0 + 1;
      ok = ok && (A.tmax >= 0.0) && (A.tmax < DBL_MAX) && (A.tdelta < DBL_MAX);
      const double T = A.tmax + (double)(A.n - 1) * A.tdelta;   // estimate only
      t_exit = fmin(t_exit, T);
    }
  }
  // how many elements of each axis can precede the exit: a generous closed-form bound, CHECKED after the merge
  // (par_bounds_hold) -- correctness never rests on it
  const double bound = t_exit + fabs(t_exit) * 0x1p-20 + 0x1p-20;
  P.elems = 0;
  for (int a = 0; a < 3; ++a) {
    ParAxis& A = P.ax[a];
    A.m = 0;
    if (A.n > 0) {
      double q = (bound - A.tmax) / A.tdelta;
      if (!(q >= 0.0)) q = 0.0;
      q = fmin(q, 1.0e9);
      const int want = (int)q + 2;
      A.m = want < A.n ? want : A.n;
    }
    P.elems += A.m;
  }
  P.regular = ok && P.elems > 0 && P.elems <= kParMaxElems;
  return P;
}

// Position of element (axis a, index i, value t) in the merged order, and how many elements of each axis
// precede it.  tb[b] points at axis b's generated sequence (m_b values).  The reference picks X iff
// tX < tY && tX < tZ, otherwise Y iff tY < tZ (given !(tX < tY)), otherwise Z: on equal values Z precedes Y
// precedes X, i.e. an element of axis b precedes an equal element of axis a iff b > a.
VIDAR_DEV int par_rank(const ParRay& P, int a, int i, double t, const double* const tb[3], int before[3]) {
  int rank = i;
  for (int b = 0; b < 3; ++b) {
    if (b == a) { before[b] = i; continue; }
    const int m = P.ax[b].m;
    int j = 0;
    if (m > 0) {
      const double inv = fabs(P.ax[b].dir);                     // 1 / tdelta_b
      double x = (t - P.ax[b].tmax) * inv;                      // estimate of the boundary, corrected below
      if (!(x > -1.0)) x = -1.0;
      if (x > (double)m) x = (double)m;
      j = (int)floor(x) + 1;                                    // floor, not truncation: x in (-1, 0) means "before all"
      j = j < 0 ? 0 : (j > m ? m : j);
      const double* s = tb[b];
      // both neighbours of the estimated boundary are requested at once; the loops only run when it is off
      const double lo = (j > 0) ? s[j - 1] : -DBL_MAX, hi = (j < m) ? s[j] : DBL_MAX;
      if (b > a) {
        if (!((lo <= t || j == 0) && (!(hi <= t) || j == m))) {
          while (j < m && s[j] <= t) ++j;
          while (j > 0 && !(s[j - 1] <= t)) --j;
        }
      } else {
        if (!((lo < t || j == 0) && (!(hi < t) || j == m))) {
          while (j < m && s[j] < t) ++j;
          while (j > 0 && !(s[j - 1] < t)) --j;
        }
      }
    }
    before[b] = j;
    rank += j;
  }
  return rank;
}

// After the merge: S = 1 + position of the first element that takes the ray out of the volume; the bound of
// par_setup holds iff every truncated axis' last generated element lies at or beyond the exit.
// last_rank[a]: merged position of element (a, m_a - 1)  (unused when m_a == 0).
VIDAR_DEV bool par_steps(const ParRay& P, const int last_rank[3], int& S) {
  int exit_rank = 1 << 30;
  for (int a = 0; a < 3; ++a)
    if (P.ax[a].n > 0 && P.ax[a].m == P.ax[a].n && last_rank[a] < exit_rank) exit_rank = last_rank[a];
  if (exit_rank == (1 << 30)) return false;
  S = exit_rank + 1;
  for (int a = 0; a < 3; ++a)
    if (P.ax[a].m > 0 && P.ax[a].m < P.ax[a].n && last_rank[a] < S) return false;
  return true;
}

// clamp((int)round(p), 0, size - 1) with round() = half away from zero, written as trunc + an exact remainder test
// (p - trunc(p) is exact in fp64): a handful of instructions instead of the library call in the serial chain.
// |p| stays far below 2^31 on a regular ray.  tests/test_march_host_cpu.py checks it against round().
VIDAR_DEV int par_round_clamp(double p, int size) {
  const double t = trunc(p);
  const double f = p - t;
  int q = (int)t;
  if (fabs(f) >= 0.5) q += (p < 0.0) ? -1 : 1;
  q = q < size ? q : size - 1;
  return q >= 0 ? q : 0;
}

// Duplicate-merge recurrence of the merged mode (dvxlr.cu:366-377 as restated by Integrator::sample): the
// interval of a step that lands in the pending voxel again is measured from the rewound start.
VIDAR_DEV double par_dt(double d, double last_d, bool same, double prev_dt) {
  const double rewind = same ? prev_dt : 0.0;
  return fmax(0.0, d - (last_d - rewind));
}

#ifndef __HIPCC__
// Host emulation of the step-parallel traversal with the call signature of march(): the phases run one
// after the other over plain arrays, the sink sees the same (voxel, d, last_d) stream.  TEST HARNESS.
template <int MODE, class Sink>
inline double march_par(const RayIn& r, const Vol& g, Sink& sink) {
  const ParRay P = par_setup<MODE>(r, g);
  if (!P.regular) return march<MODE>(r, g, sink);
  static thread_local double seq[3][kParMaxElems];
  static thread_local double md[kParMaxElems];
  static thread_local int vox[kParMaxElems][3];
  for (int a = 0; a < 3; ++a) {                                  // phase 1: per-axis chains
    double t = P.ax[a].tmax;
    for (int i = 0; i < P.ax[a].m; ++i) { seq[a][i] = t; t += P.ax[a].tdelta; }
  }
  const double* tb[3] = {seq[0], seq[1], seq[2]};
  int last_rank[3] = {-1, -1, -1};
  for (int a = 0; a < 3; ++a)                                    // phase 2: merge by rank
    for (int i = 0; i < P.ax[a].m; ++i) {
      int before[3];
      const int k = par_rank(P, a, i, seq[a][i], tb, before);
      md[k] = seq[a][i];
      for (int b = 0; b < 3; ++b) vox[k][b] = P.ax[b].v0 + P.ax[b].s * before[b];
      if (i == P.ax[a].m - 1) last_rank[a] = k;
    }
  int S = 0;
  if (!par_steps(P, last_rank, S)) return march<MODE>(r, g, sink);
  const int size[3] = {g.X, g.Y, g.Z};
  if (MODE != kClassic) {                                        // phase 3: rounded-path chains
    for (int a = 0; a < 3; ++a) {
      double p = (double)P.ax[a].v0, last = 0.0;
      for (int s = 0; s < S; ++s) {
        vox[s][a] = par_round_clamp(p, size[a]);
        const double adv = fmax(0.0, md[s] - last);
        p += adv * P.ax[a].dir;
        last = md[s];
      }
    }
  }
  for (int s = 0; s < S; ++s)                                    // phase 4 (here: the sequential sink)
    if (!sink.sample(vox[s][0], vox[s][1], vox[s][2], md[s], s > 0 ? md[s - 1] : 0.0)) break;
  sink.finish();
  return P.len;
}
#endif

}  // namespace vidar_march
