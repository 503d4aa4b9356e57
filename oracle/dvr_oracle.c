/* dvr_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded (optionally OpenMP-over-rays) CPU restatement of the reference's
 * differentiable-voxel-rendering kernels.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library.
 *
 * Follows (behaviour, fp64 operation order):
 *   third_lib/dvr/dvr.cu       :31-62 init, :87-316 render_forward, :409-626 render,
 *                              host fills :354-355, :655-657
 *   third_lib/dvxlr/dvxlr.cu   :84-111 get_grad_sigma, :185-456 render, fills :490-493
 *   third_lib/dvxlr/dvxlr_v2.cu:37-66 get_grad_sigma_v2, :147-426 render_v2, fills :462-468
 *
 * Parity pin: this restatement is checked bit-for-bit (indices) against the reference .cu
 * kernels themselves compiled for the host by oracle/build_ref.py (oracle/_ref), see
 * tests/test_oracle_vs_ref.py, and frozen in tests/golden/dvr_family_*.npz.
 *
 * Unlike the reference the per-ray work arrays are heap blocks per call, and a ray that would
 * overflow MAX_D samples is truncated (the reference asserts).  scalar_t == float only: the
 * reference allocates its outputs with the default dtype (dvxlr.cu:490-493), so fp64 inputs fail
 * inside packed_accessor32 and never worked.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define DVR_MAX_D 1446
#define DVXLR_MAX_D 1026

enum { MODE_CLASSIC = 0, MODE_ROUNDED = 1, MODE_ROUNDED_MERGED = 2 };

typedef struct {
  int count;
  double len; /* un-clamped origin->end distance */
  int* vx; int* vy; int* vz;
  double* d; double* dt;
} path_t;

static int clampi(int v, int hi) { v = v < hi ? v : hi - 1; return v >= 0 ? v : 0; }

/* voxel traversal; fills p (capacity cap) */
static void trace(int mode, double xo, double yo, double zo, double xe, double ye, double ze,
                  int X, int Y, int Z, int cap, path_t* p) {
  int vx = (int)xo, vy = (int)yo, vz = (int)zo;
  double pvx = (double)vx, pvy = (double)vy, pvz = (double)vz;
  const double rx = xe - xo, ry = ye - yo, rz = ze - zo;
  double gt_d = sqrt(rx * rx + ry * ry + rz * rz);
  const double dx = rx / gt_d, dy = ry / gt_d, dz = rz / gt_d;
  const int stepX = (dx >= 0) ? 1 : -1, stepY = (dy >= 0) ? 1 : -1, stepZ = (dz >= 0) ? 1 : -1;
  const int neg = (mode == MODE_CLASSIC) ? 0 : -1;
  const double nbx = vx + (stepX < 0 ? neg : 1);
  const double nby = vy + (stepY < 0 ? neg : 1);
  const double nbz = vz + (stepZ < 0 ? neg : 1);
  double tMaxX = (dx != 0) ? (nbx - xo) / dx : DBL_MAX;
  double tMaxY = (dy != 0) ? (nby - yo) / dy : DBL_MAX;
  double tMaxZ = (dz != 0) ? (nbz - zo) / dz : DBL_MAX;
  const double tDeltaX = (dx != 0) ? stepX / dx : DBL_MAX;
  const double tDeltaY = (dy != 0) ? stepY / dy : DBL_MAX;
  const double tDeltaZ = (dz != 0) ? stepZ / dz : DBL_MAX;

  int count = 0;
  long step = 0;
  double last_d = 0.0;
  int was_inside = 0;
  p->len = gt_d;
  while (step < (1L << 22)) {
    const int inside = (0 <= vx && vx < X) && (0 <= vy && vy < Y) && (0 <= vz && vz < Z);
    if (inside) {
      was_inside = 1;
      if (count >= cap) break; /* reference: assert(count <= MAX_D) */
      if (mode == MODE_CLASSIC) {
        p->vx[count] = vx; p->vy[count] = vy; p->vz[count] = vz;
      } else {
        p->vx[count] = clampi((int)round(pvx), X);
        p->vy[count] = clampi((int)round(pvy), Y);
        p->vz[count] = clampi((int)round(pvz), Z);
      }
    } else if (was_inside) {
      break;
    } else if (last_d > gt_d) {
      break;
    }
    double _d;
    if (tMaxX < tMaxY) {
      if (tMaxX < tMaxZ) { _d = tMaxX; vx += stepX; tMaxX += tDeltaX; }
      else               { _d = tMaxZ; vz += stepZ; tMaxZ += tDeltaZ; }
    } else {
      if (tMaxY < tMaxZ) { _d = tMaxY; vy += stepY; tMaxY += tDeltaY; }
      else               { _d = tMaxZ; vz += stepZ; tMaxZ += tDeltaZ; }
    }
    if (mode != MODE_CLASSIC) {
      pvx += fmax(0.0, _d - last_d) * dx;
      pvy += fmax(0.0, _d - last_d) * dy;
      pvz += fmax(0.0, _d - last_d) * dz;
    }
    if (inside) {
      if (mode == MODE_ROUNDED_MERGED && count >= 1) {
        if (p->vx[count - 1] == p->vx[count] && p->vy[count - 1] == p->vy[count] &&
            p->vz[count - 1] == p->vz[count]) {
          count--;
          last_d -= p->dt[count];
        }
      }
      p->d[count] = _d;
      p->dt[count] = fmax(0.0, _d - last_d);
      count++;
    }
    last_d = _d;
    step++;
  }
  p->count = count;
}

typedef struct {
  double exp_d, p_out, max_d;
  double* csd; double* dd; /* dd = d(pred)/d(sigma_i), length count */
} integ_t;

static void integrate(const float* vol, int Y, int X, const path_t* p, integ_t* g, int want_dd) {
  const int n = p->count;
  double exp_d = 0.0;
  for (int i = 0; i < n; i++) {
    const double s = (double)vol[((size_t)p->vz[i] * Y + p->vy[i]) * X + p->vx[i]];
    const double sd = s * p->dt[i];
    double pi;
    if (i == 0) { g->csd[0] = sd; pi = 1 - exp(-sd); }
    else { g->csd[i] = g->csd[i - 1] + sd; pi = exp(-g->csd[i - 1]) - exp(-g->csd[i]); }
    exp_d += pi * p->d[i];
  }
  g->p_out = exp(-g->csd[n - 1]);
  g->max_d = p->d[n - 1];
  g->exp_d = exp_d + g->p_out * g->max_d;
  if (!want_dd) return;
  for (int i = n - 1; i >= 0; i--) {
    if (i == n - 1) g->dd[i] = g->p_out * g->max_d;
    else g->dd[i] = g->dd[i + 1] - exp(-g->csd[i]) * (p->d[i + 1] - p->d[i]);
  }
  for (int i = n - 1; i >= 0; i--) g->dd[i] *= p->dt[i];
  for (int i = n - 1; i >= 0; i--) g->dd[i] -= p->dt[i] * g->p_out * g->max_d;
}

typedef struct { path_t p; integ_t g; void* block; } work_t;

static int work_alloc(work_t* w, int cap) {
  const size_t ni = (size_t)cap * 3 * sizeof(int), nd = (size_t)cap * 4 * sizeof(double);
  w->block = malloc(ni + nd);
  if (!w->block) return -1;
  double* dbl = (double*)w->block;
  w->p.d = dbl; w->p.dt = dbl + cap; w->g.csd = dbl + 2 * cap; w->g.dd = dbl + 3 * cap;
  int* it = (int*)(dbl + 4 * cap);
  w->p.vx = it; w->p.vy = it + cap; w->p.vz = it + 2 * cap;
  return 0;
}

/* returns 0 if the ray is skipped (padded / bad tindex) */
static int ray_setup(const float* origin, const float* points, const float* tindex, int n, int c,
                     int M, int T, int TO, double* o, double* e, int* ts) {
  const float t = tindex[(size_t)n * M + c];
  if (t < 0 || t != t) return 0;
  const long ti = (long)t;
  if (!(T == 1 || ti < T) || ti >= TO) return 0;
  *ts = (T == 1) ? 0 : (int)ti;
  for (int k = 0; k < 3; k++) {
    o[k] = origin[((size_t)n * TO + ti) * 3 + k];
    e[k] = points[((size_t)n * M + c) * 3 + k];
  }
  return 1;
}

/* ------------------------------------------------------------------------------------------ */
int oracle_dvr_render_forward(const float* sigma, const float* origin, const float* points,
                              const float* tindex, float* pred, float* gt, int N, int M, int T,
                              int TO, int Z, int Y, int X, int train_phase) {
  const size_t vol = (size_t)Z * Y * X;
  int err = 0;
#pragma omp parallel
  {
    work_t w;
    if (work_alloc(&w, DVR_MAX_D)) { err = 1; }
    else {
#pragma omp for schedule(dynamic, 64)
      for (long r = 0; r < (long)N * M; r++) {
        const int n = (int)(r / M), c = (int)(r % M);
        pred[r] = -1.f; gt[r] = -1.f;
        double o[3], e[3]; int ts;
        if (!ray_setup(origin, points, tindex, n, c, M, T, TO, o, e, &ts)) continue;
        trace(MODE_ROUNDED, o[0], o[1], o[2], e[0], e[1], e[2], X, Y, Z, DVR_MAX_D, &w.p);
        if (w.p.count > 0) {
          integrate(sigma + ((size_t)n * T + ts) * vol, Y, X, &w.p, &w.g, 0);
          double gt_d = w.p.len;
          if (train_phase == 1) gt_d = fmin(gt_d, w.g.max_d);
          pred[r] = (float)w.g.exp_d; gt[r] = (float)gt_d;
        }
      }
      free(w.block);
    }
  }
  return err;
}

/* grad accumulation is sequential over rays in index order (the reference's is racy) */
int oracle_dvr_render(const float* sigma, const float* origin, const float* points,
                      const float* tindex, float* pred, float* gt, float* grad_sigma, int N, int M,
                      int T, int TO, int Z, int Y, int X, int loss_type) {
  const size_t vol = (size_t)Z * Y * X;
  work_t w;
  if (work_alloc(&w, DVR_MAX_D)) return 1;
  double* acc = (double*)calloc((size_t)N * T * vol, sizeof(double));
  if (!acc) { free(w.block); return 1; }
  for (long r = 0; r < (long)N * M; r++) {
    const int n = (int)(r / M), c = (int)(r % M);
    pred[r] = -1.f; gt[r] = -1.f;
    double o[3], e[3]; int ts;
    if (!ray_setup(origin, points, tindex, n, c, M, T, TO, o, e, &ts)) continue;
    trace(MODE_CLASSIC, o[0], o[1], o[2], e[0], e[1], e[2], X, Y, Z, DVR_MAX_D, &w.p);
    if (w.p.count <= 0) continue;
    const size_t slice = ((size_t)n * T + ts) * vol;
    integrate(sigma + slice, Y, X, &w.p, &w.g, 1);
    const double exp_d = w.g.exp_d, gt_d = fmin(w.p.len, w.g.max_d);
    pred[r] = (float)exp_d; gt[r] = (float)gt_d;
    double dl = 1.0;
    if (loss_type == 0) dl = (exp_d >= gt_d) ? 1 : -1;
    else if (loss_type == 1) dl = exp_d - gt_d;
    else if (loss_type == 2) dl = (exp_d >= gt_d) ? (1.0 / gt_d) : -(1.0 / gt_d);
    for (int i = 0; i < w.p.count; i++)
      acc[slice + ((size_t)w.p.vz[i] * Y + w.p.vy[i]) * X + w.p.vx[i]] += dl * w.g.dd[i];
  }
  for (size_t i = 0; i < (size_t)N * T * vol; i++) grad_sigma[i] = (float)acc[i];
  free(acc); free(w.block);
  return 0;
}

int oracle_dvr_init(const float* points, const float* tindex, float* occ, int N, int M, int T, int Z,
                    int Y, int X) {
  memset(occ, 0, sizeof(float) * (size_t)N * T * Z * Y * X);
  for (long r = 0; r < (long)N * M; r++) {
    const int n = (int)(r / M);
    const float t = tindex[r];
    if (t < 0 || t != t) continue;
    const long ti = (long)t;
    if (!(T == 1 || ti < T)) continue;
    const int ts = (T == 1) ? 0 : (int)ti;
    const int vx = (int)points[r * 3 + 0], vy = (int)points[r * 3 + 1], vz = (int)points[r * 3 + 2];
    if (0 <= vx && vx < X && 0 <= vy && vy < Y && 0 <= vz && vz < Z)
      occ[((((size_t)n * T + ts) * Z + vz) * Y + vy) * X + vx] = 1.f;
  }
  return 0;
}

/* dvxlr.render (sigma_regul == NULL) and dvxlr_v2.render_v2 */
int oracle_dvxlr_render(const float* sigma, const float* sigma_regul, const float* origin,
                        const float* points, const float* tindex, float* pred, float* gt,
                        float* dd_dsigma, float* indices, float* ray_pred, float* indicator, int N,
                        int M, int T, int TO, int Z, int Y, int X) {
  const size_t vol = (size_t)Z * Y * X;
  const int L = DVXLR_MAX_D;
  int err = 0;
#pragma omp parallel
  {
    work_t w;
    if (work_alloc(&w, L)) { err = 1; }
    else {
#pragma omp for schedule(dynamic, 64)
      for (long r = 0; r < (long)N * M; r++) {
        const int n = (int)(r / M), c = (int)(r % M);
        pred[r] = -1.f; gt[r] = -1.f;
        memset(dd_dsigma + r * L, 0, sizeof(float) * L);
        memset(indices + r * L * 3, 0, sizeof(float) * L * 3);
        if (sigma_regul) {
          memset(ray_pred + r * L, 0, sizeof(float) * L);
          for (int i = 0; i < L; i++) indicator[r * L + i] = -1.f;
        }
        double o[3], e[3]; int ts;
        if (!ray_setup(origin, points, tindex, n, c, M, T, TO, o, e, &ts)) continue;
        trace(MODE_ROUNDED_MERGED, o[0], o[1], o[2], e[0], e[1], e[2], X, Y, Z, L, &w.p);
        if (w.p.count <= 0) continue;
        const size_t slice = ((size_t)n * T + ts) * vol;
        integrate(sigma + slice, Y, X, &w.p, &w.g, 1);
        pred[r] = (float)w.g.exp_d;
        gt[r] = (float)fmin(w.p.len, w.g.max_d);
        int reached = 0;
        for (int i = 0; i < w.p.count; i++) {
          dd_dsigma[r * L + i] = (float)w.g.dd[i];
          indices[(r * L + i) * 3 + 0] = (float)w.p.vz[i];
          indices[(r * L + i) * 3 + 1] = (float)w.p.vy[i];
          indices[(r * L + i) * 3 + 2] = (float)w.p.vx[i];
          if (sigma_regul) {
            indicator[r * L + i] = 0.f;
            if (!reached && w.p.d[i] >= w.p.len) { indicator[r * L + i] = 1.f; reached = 1; }
            ray_pred[r * L + i] =
                sigma_regul[slice + ((size_t)w.p.vz[i] * Y + w.p.vy[i]) * X + w.p.vx[i]];
          }
        }
      }
      free(w.block);
    }
  }
  return err;
}

/* get_grad_sigma (indicator == NULL) / get_grad_sigma_v2; sequential double accumulation */
int oracle_dvxlr_get_grad_sigma(const float* em, const float* indices, const float* tindex,
                                const float* indicator, const float* grad_ray_pred,
                                float* grad_sigma, float* grad_regul, int N, int M, int L, int T,
                                int Z, int Y, int X) {
  const size_t vol = (size_t)Z * Y * X, tot = (size_t)N * T * vol;
  double* a = (double*)calloc(tot, sizeof(double));
  double* b = indicator ? (double*)calloc(tot, sizeof(double)) : NULL;
  if (!a || (indicator && !b)) { free(a); free(b); return 1; }
  for (long r = 0; r < (long)N * M; r++) {
    const int n = (int)(r / M);
    const float t = tindex[r];
    if (t < 0 || t != t) continue;
    const long ti = (long)t;
    if (!(T == 1 || ti < T)) continue;
    const int ts = (T == 1) ? 0 : (int)ti;
    const size_t slice = ((size_t)n * T + ts) * vol;
    for (int i = 0; i < L; i++) {
      const int z = (int)indices[(r * L + i) * 3 + 0];
      const int y = (int)indices[(r * L + i) * 3 + 1];
      const int x = (int)indices[(r * L + i) * 3 + 2];
      const size_t o = slice + ((size_t)z * Y + y) * X + x;
      a[o] += em[r * L + i];
      if (indicator && indicator[r * L + i] >= 0) b[o] += grad_ray_pred[r * L + i];
    }
  }
  for (size_t i = 0; i < tot; i++) grad_sigma[i] = (float)a[i];
  if (indicator) for (size_t i = 0; i < tot; i++) grad_regul[i] = (float)b[i];
  free(a); free(b);
  return 0;
}
