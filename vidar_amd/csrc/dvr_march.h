// Ray traversal + online integration shared by every dvr / dvxlr / dvxlr_v2 kernel.
//
// Plain C++ on purpose: the kernels of dvr_family.hip instantiate it per lane, and
// tests/march_host.cpp compiles the very same text with g++ so that the traversal logic can be
// checked by the CPU test-suite on a machine without a GPU.  Everything here is IEEE fp64 in the
// reference's operation order (compile with -ffp-contract=off).
//
// Follows (behaviour, not code): third_lib/dvr/dvr.cu:87-316 / :409-626,
// third_lib/dvxlr/dvxlr.cu:185-456, third_lib/dvxlr/dvxlr_v2.cu:147-426.
#pragma once
#include <float.h>
#include <math.h>
#include <stddef.h>

#ifdef __HIPCC__
#define VIDAR_DEV __device__ __forceinline__
#else
#define VIDAR_DEV inline
#endif

namespace vidar_march {

constexpr int kDvrMaxD = 1446;    // dvr.cu:9
constexpr int kDvxlrMaxD = 1026;  // dvxlr.cu:10, dvxlr_v2.cu:10
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

VIDAR_DEV RayIn load_ray(const float* __restrict__ origin, const float* __restrict__ points,
                         const float* __restrict__ tindex, int n, int c, int M, const Vol& v) {
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

// Cheap estimate of the number of traversal steps a ray spends inside the volume (fp32, used only
// to group rays of similar length into the same wave; never influences a result).
VIDAR_DEV int estimate_steps(const RayIn& r, const Vol& g) {
  if (!r.valid) return 0;
  const float ox = (float)r.xo, oy = (float)r.yo, oz = (float)r.zo;
  float dx = (float)r.xe - ox, dy = (float)r.ye - oy, dz = (float)r.ze - oz;
  const float len = sqrtf(dx * dx + dy * dy + dz * dz);
  if (!(len > 0.f)) return 0;
  dx /= len; dy /= len; dz /= len;
  const float big = 4.f * (float)(g.X + g.Y + g.Z);
  const float ex = dx > 0.f ? ((float)g.X - ox) / dx : (dx < 0.f ? -ox / dx : big);
  const float ey = dy > 0.f ? ((float)g.Y - oy) / dy : (dy < 0.f ? -oy / dy : big);
  const float ez = dz > 0.f ? ((float)g.Z - oz) / dz : (dz < 0.f ? -oz / dz : big);
  float t = fminf(ex, fminf(ey, ez));
  t = fminf(fmaxf(t, 0.f), big);
  const float steps = t * (fabsf(dx) + fabsf(dy) + fabsf(dz));
  return (int)fminf(steps, 65535.f);
}

// Amanatides-Woo traversal with the reference's modifications.  Sink::sample is called once per
// step spent inside the volume, in order, with the voxel the reference would record, the exit
// distance _d of that step and the previous step's exit distance.  Returns the un-clamped ray
// length.  The axis choice is written as selects (no divergent branches): neighbouring lanes step
// along different axes on almost every iteration.
template <int MODE, class Sink>
VIDAR_DEV double march(const RayIn& r, const Vol& g, Sink& sink) {
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
    const bool inside = ((unsigned)vx < (unsigned)g.X) & ((unsigned)vy < (unsigned)g.Y) &
                        ((unsigned)vz < (unsigned)g.Z);
    if (!inside && (was_inside || last_d > len)) break;   // left the volume / never reached it
    was_inside = was_inside || inside;
    int qx = vx, qy = vy, qz = vz;
    if (MODE != kClassic) {   // only consumed when inside; computed by every lane to stay branch-free
      qx = (int)round(px); qx = qx < g.X ? qx : g.X - 1; qx = qx >= 0 ? qx : 0;
      qy = (int)round(py); qy = qy < g.Y ? qy : g.Y - 1; qy = qy >= 0 ? qy : 0;
      qz = (int)round(pz); qz = qz < g.Z ? qz : g.Z - 1; qz = qz >= 0 ? qz : 0;
    }
    // if (tx < ty) { if (tx < tz) x else z } else { if (ty < tz) y else z }
    const bool xy = tx < ty, xz = tx < tz, yz = ty < tz;
    const bool ax = xy & xz;
    const bool ay = (!xy) & yz;
    const bool az = !(ax | ay);
    const double d = ax ? tx : (ay ? ty : tz);
    const double ntx = tx + ddx, nty = ty + ddy, ntz = tz + ddz;
    vx += ax ? sx : 0; tx = ax ? ntx : tx;
    vy += ay ? sy : 0; ty = ay ? nty : ty;
    vz += az ? sz : 0; tz = az ? ntz : tz;
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

// Online integrator shared by every variant.  Emit::commit(k, vid, d, dt, P_k, W_{k-1}) is called
// once per *final* sample k in order (vid = (z*Y + y)*X + x, P_k = prefix of W before k).
template <int MODE, int MAXD, class Emit>
struct Integrator {
  const float* __restrict__ sig;  // sigma[n][ts] slice, fewer than 2^31 voxels
  int Y, X;
  Emit& emit;
  // committed state
  int k = 0;
  double csd = 0.0, Tprev = 1.0, dprev = 0.0, d0 = 0.0, S = 0.0;
  // pending sample (merged mode only)
  bool pending = false;
  int uvid = -1;
  double ud = 0.0, udt = 0.0;

  VIDAR_DEV Integrator(const float* s, int Y_, int X_, Emit& e) : sig(s), Y(Y_), X(X_), emit(e) {}

  // Same arithmetic, same order per sample -- but the density of sample k is only CONSUMED by commit k+1 (the
  // transmittance T_k first matters for W_k = T_k (d_{k+1} - d_k)), so its load has a whole traversal step to
  // arrive instead of stalling the lane right after it is issued; the last sample's density is never needed.
  float sg_fly = 0.f;                          // density of sample k-1, in flight
  double dt_fly = 0.0;
  VIDAR_DEV void commit(int vid, double d, double dt) {
    double w_prev = 0.0;                       // W_{k-1} = T_{k-1} (d_k - d_{k-1})
    if (k == 0) {
      d0 = d;
    } else {
      const double sg = (double)sg_fly;
      csd = (k == 1) ? sg * dt_fly : csd + sg * dt_fly;
      Tprev = (double)expf((float)(-csd));
      w_prev = Tprev * (d - dprev);
      S += w_prev;
    }
    emit.commit(k, vid, d, dt, S, w_prev);
    // issued after the last read of the previous density, so that the load can land in the very register that
    // carries it to the next commit (a copy at the end of the block would wait for it)
    sg_fly = sig[vid];
    dt_fly = dt;
    dprev = d;
    ++k;
  }

  VIDAR_DEV bool sample(int x, int y, int z, double d, double last_d) {
    const int vid = (z * Y + y) * X + x;
    if (MODE == kRoundedMerged) {
      // dvxlr.cu:366-377: a step that lands in the pending voxel again replaces the pending sample
      // and rewinds last_d by its dt; anything else commits the pending sample and opens a new
      // one.  Written with one real branch (the commit): x - 0.0 == x bit for bit.
      const bool same = pending && vid == uvid;
      if (pending && !same) commit(uvid, ud, udt);
      if (!same && k >= MAXD) { pending = false; return false; }
      const double rewind = same ? udt : 0.0;
      udt = fmax(0.0, d - (last_d - rewind));
      ud = d;
      uvid = vid;
      pending = true;
      return true;
    } else {
      if (k >= MAXD) return false;
      commit(vid, d, fmax(0.0, d - last_d));
      return true;
    }
  }
  VIDAR_DEV void finish() {
    if (MODE == kRoundedMerged && pending) { commit(uvid, ud, udt); pending = false; }
  }
  // after finish(): count = k, p_out = Tprev, max_d = dprev, pred = d0 + S
};

struct NoEmit {
  VIDAR_DEV void commit(int, int, double, double, double, double) {}
};

// dvxlr staging: while a lane walks its ray it parks each sample in the ray's own `indices` row
// (three floats per sample, ONE 12-byte store):
//   idx_row[3k+0] <- dt_k    idx_row[3k+1] <- linear voxel id (exact in fp32 below 2^24 voxels)
//   idx_row[3k+2] <- W_{k-1} (k >= 1)
// The k = 0 slot idx_row[2] has no W and receives the per-ray scalars the finish pass needs
// (encode_stash).  The dd_dsigma row is not touched by the march.
struct Parked {
  float dt, vid, w_prev;
};

struct RowStager {
  Parked* __restrict__ slot;  // &idx_row[3k] for the next commit
  double true_len;
  int k_surface = -1;
  VIDAR_DEV void commit(int k, int vid, double d, double dt, double, double w_prev) {
    Parked p;
    p.dt = (float)dt;
    p.vid = (float)vid;
    p.w_prev = (float)w_prev;
    *slot = p;
    ++slot;
    if (k_surface < 0 && d >= true_len) k_surface = k;    // dvxlr_v2.cu:408-424
  }
};

// count in [0, 1026], k_surface in [-1, 1025], nan_tail: +-(count + 2048 (k_surface + 1)), exact in fp32
VIDAR_DEV float encode_stash(int count, int k_surface, bool nan_tail) {
  const float v = (float)(count + 2048 * (k_surface + 1));
  return nan_tail ? -v : v;
}
VIDAR_DEV void decode_stash(float stash, int& count, int& k_surface, bool& nan_tail) {
  nan_tail = stash < 0.f;
  const int v = (int)fabsf(stash);
  count = v & 2047;
  k_surface = (v >> 11) - 1;
}

struct SequentialTraversal {
  template <int MODE, class Sink>
  VIDAR_DEV double run(const RayIn& r, const Vol& g, Sink& sink) const { return march<MODE>(r, g, sink); }
};

// dvxlr / dvxlr_v2 march of ray c of sample n (launch 1 of dvxlr.render, see dvr_family.hip).  `trav` is the
// traversal that feeds the integrator: the sequential march, or (host harness only) its step-parallel form.
template <class Trav = SequentialTraversal>
VIDAR_DEV void dvxlr_march_ray(const float* __restrict__ sigma, const float* __restrict__ origin,
                               const float* __restrict__ points, const float* __restrict__ tindex,
                               float* __restrict__ pred_dist, float* __restrict__ gt_dist,
                               float* __restrict__ indices, int n, int c, int M, const Vol& g,
                               const Trav trav = Trav()) {
  constexpr int L = kDvxlrMaxD;
  const size_t row = (size_t)n * M + c;
  float* idr = indices + row * L * 3;
  float pred = -1.f, gt = -1.f, stash = 0.f;
  const RayIn r = load_ray(origin, points, tindex, n, c, M, g);
  if (r.valid) {
    RowStager st;
    st.slot = reinterpret_cast<Parked*>(idr);
    {
      const double rx = r.xe - r.xo, ry = r.ye - r.yo, rz = r.ze - r.zo;
      st.true_len = sqrt(rx * rx + ry * ry + rz * rz);
    }
    const size_t vol = (size_t)g.Z * g.Y * g.X;
    Integrator<kRoundedMerged, kDvxlrMaxD, RowStager> a(sigma + ((size_t)n * g.T + r.ts) * vol, g.Y, g.X,
                                                        st);
    const double len = trav.template run<kRoundedMerged>(r, g, a);
    if (a.k > 0) {
      pred = (float)(a.d0 + a.S);
      gt = (float)fmin(len, a.dprev);
      // the reference's (max_d - d_last) p_out term (dvxlr.cu:412-439) is 0, or NaN when the
      // distances are NaN (zero-length ray): such a ray poisons its whole row
      const double tail = a.dprev - a.dprev;
      stash = encode_stash(a.k, st.k_surface, !(tail == tail));
    }
  }
  idr[2] = stash;
  pred_dist[row] = pred;
  gt_dist[row] = gt;
}

}  // namespace vidar_march
