// Modulated deformable convolution v2 (DCNv2) sampling kernels for gfx950.
//
// Replaces mmcv.ops.ModulatedDeformConv2dPack's CUDA kernels (mmcv-full 1.4.0, third party, NOT
// vendored in the reference) used by the ResNet101 backbone of every ViDAR config
// (`dcn=dict(type='DCNv2', deform_groups=1, fallback_on_stride=False)`,
//  projects/configs/vidar_pretrain/nusc_1_8_subset/vidar_1_8_nusc_1future.py:96-97).
// The convolution itself stays a GEMM (weight [Cout, Cin*kh*kw] x columns, hipBLASLt / MFMA);
// these kernels build / differentiate the deformable column matrix:
//   cols[n, (c*K + t), p] = mask[n,t,p] * bilinear(x[n,c], p_y*s - pad + i*dil + off[n,2t,p],
//                                                          p_x*s - pad + j*dil + off[n,2t+1,p])
// with zero padding (a sample counts iff -1 < h < H and -1 < w < W; each corner zero outside).
// Layouts: x [N,C,H,W], offset [N,2K,Ho,Wo] ((dy,dx) interleaved per tap), mask [N,K,Ho,Wo],
// cols [N, C*K, Ho*Wo] -- lanes run over output pixels so column writes, offset/mask reads and
// (for small offsets) input reads are coalesced; the backward scatter therefore puts neighbouring
// lanes on the same 128-byte lines (atomics cost per instruction x line, see msda.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vidar_hip.h"
#include "vidar_common.h"

namespace {

// XCD-aware block order for kernels whose workgroups of one (channel group, image) PLANE re-read each other's cache lines
// (window halos, neighbouring taps): hardware deals consecutive workgroup ids round-robin to the 8 XCDs, each with its
// own L2, so the T workgroups of a plane would run on 8 different L2s and every shared line would come from HBM up to 8
// times.  Here XCD k walks planes k, k + 8, ...: linear id L -> xcd = L % 8, j = L / 8, plane = (j / T) * 8 + xcd,
// tile = j % T; the grid is padded to 8 * T * ceil(planes / 8) and the extra workgroups return.
struct PlaneTile { int plane, tile; bool ok; };
__device__ __forceinline__ PlaneTile xcd_plane_tile(int L, int T, int planes) {
  const int xcd = L & 7, j = L >> 3;
  const int pl = (j / T) * 8 + xcd;
  return PlaneTile{pl, j - (j / T) * T, pl < planes};
}
inline unsigned xcd_plane_grid(int T, int planes) { return 8u * (unsigned)T * (unsigned)((planes + 7) / 8); }

struct Conv {
  int C, H, W, Ho, Wo, kh, kw, stride, pad, dil;
};

struct Bil {
  int h0, w0;
  float lh, lw;
  bool in, t, b, l, r;
};

__device__ __forceinline__ Bil bil(float h, float w, int H, int W) {
  Bil q;
  q.in = h > -1.f && w > -1.f && h < H && w < W;
  q.h0 = (int)floorf(h); q.w0 = (int)floorf(w);
  q.lh = h - q.h0; q.lw = w - q.w0;
  q.t = q.h0 >= 0; q.b = q.h0 + 1 <= H - 1; q.l = q.w0 >= 0; q.r = q.w0 + 1 <= W - 1;
  return q;
}

__device__ __forceinline__ float sample(const float* __restrict__ im, const Bil& q, int W) {
  if (!q.in) return 0.f;
  const float v1 = (q.t && q.l) ? im[q.h0 * W + q.w0] : 0.f;
  const float v2 = (q.t && q.r) ? im[q.h0 * W + q.w0 + 1] : 0.f;
  const float v3 = (q.b && q.l) ? im[(q.h0 + 1) * W + q.w0] : 0.f;
  const float v4 = (q.b && q.r) ? im[(q.h0 + 1) * W + q.w0 + 1] : 0.f;
  const float hh = 1.f - q.lh, hw = 1.f - q.lw;
  return hh * hw * v1 + hh * q.lw * v2 + q.lh * hw * v3 + q.lh * q.lw * v4;
}

// A thread owns one output pixel and kCG consecutive channels: the 9 sampling footprints (index
// arithmetic + offset / mask loads) are computed once and reused for every channel of the group.
constexpr int kCG = 8;
constexpr int kMaxTaps = 9;

struct Foot {
  Bil q[kMaxTaps];
  float m[kMaxTaps];
};

__device__ __forceinline__ Foot footprints(const float* __restrict__ offset,
                                           const float* __restrict__ mask, int n, int p, const Conv& g) {
  const int P = g.Ho * g.Wo, K = g.kh * g.kw;
  const int py = p / g.Wo, px = p % g.Wo;
  Foot f;
#pragma unroll
  for (int t = 0; t < kMaxTaps; ++t) {
    if (t < K) {
      const int i = t / g.kw, j = t % g.kw;
      const float h = py * g.stride - g.pad + i * g.dil + offset[((size_t)n * 2 * K + 2 * t) * P + p];
      const float w = px * g.stride - g.pad + j * g.dil + offset[((size_t)n * 2 * K + 2 * t + 1) * P + p];
      f.q[t] = bil(h, w, g.H, g.W);
      f.m[t] = mask[((size_t)n * K + t) * P + p];
    } else {
      f.q[t] = bil(-2.f, -2.f, g.H, g.W);
      f.m[t] = 0.f;
    }
  }
  return f;
}

// grid: (ceil(P/256), ceil(C/kCG), N)
__global__ __launch_bounds__(256) void dcn_im2col_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ offset,
                                                         const float* __restrict__ mask,
                                                         float* __restrict__ cols, Conv g) {
  const int P = g.Ho * g.Wo, K = g.kh * g.kw;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int n = blockIdx.z, c0 = blockIdx.y * kCG;
  const Foot f = footprints(offset, mask, n, p, g);
  for (int c = c0; c < min(c0 + kCG, g.C); ++c) {
    const float* im = x + ((size_t)n * g.C + c) * g.H * g.W;
    float* out = cols + ((size_t)n * g.C + c) * K * P + p;
#pragma unroll
    for (int t = 0; t < kMaxTaps; ++t)
      if (t < K) out[(size_t)t * P] = sample(im, f.q[t], g.W) * f.m[t];
  }
}

// im2col, pair-load form (W >= 2).  The four corners of a tap are two horizontally adjacent PAIRS: one 8-byte load
// per row instead of two 4-byte loads (the kernel sits on the vector memory pipe: 4 gathers + 1 store per column
// element before, 2 + 1 now), and everything that does not depend on the channel -- clamped pair index, the four
// weights times the mask, with the zero padding folded into the weights -- is computed once per (pixel, tap) and
// reused for the kCP channels of the thread's group.  The pair starts at column c0 = clamp(w0, 0, W-2): for
// w0 == -1 the right corner is element 0 of the pair, for w0 == W-1 the left corner is element 1.
constexpr int kCP = 16;
typedef float pair_t __attribute__((ext_vector_type(2), aligned(4)));

struct PairFoot {
  int top[kMaxTaps], bot[kMaxTaps];          // element index of the pair in the top / bottom row
  float wt0[kMaxTaps], wt1[kMaxTaps], wb0[kMaxTaps], wb1[kMaxTaps];
};

__device__ __forceinline__ PairFoot pair_footprints(const float* __restrict__ offset, const float* __restrict__ mask,
                                                    int n, int p, const Conv& g) {
  const Foot f = footprints(offset, mask, n, p, g);
  PairFoot o;
#pragma unroll
  for (int t = 0; t < kMaxTaps; ++t) {
    const Bil& q = f.q[t];
    const float m = q.in ? f.m[t] : 0.f;
    const int c0 = min(max(q.w0, 0), g.W - 2);
    const int r0 = min(max(q.h0, 0), g.H - 1), r1 = min(max(q.h0 + 1, 0), g.H - 1);
    // weights of the left / right corner, then their place in the loaded pair
    const float wl = (q.l ? 1.f - q.lw : 0.f), wr = (q.r ? q.lw : 0.f);
    const float a0 = q.w0 == c0 ? wl : (q.w0 < c0 ? wr : 0.f);          // w0 == -1: the right corner is element 0
    const float a1 = q.w0 == c0 ? wr : (q.w0 > c0 ? wl : 0.f);          // w0 == W-1: the left corner is element 1
    const float ht = (q.t ? 1.f - q.lh : 0.f) * m, hb = (q.b ? q.lh : 0.f) * m;
    o.top[t] = r0 * g.W + c0; o.bot[t] = r1 * g.W + c0;
    // a sample outside the image (or NaN) contributes nothing: select, not a product with 0
    o.wt0[t] = q.in ? ht * a0 : 0.f; o.wt1[t] = q.in ? ht * a1 : 0.f;
    o.wb0[t] = q.in ? hb * a0 : 0.f; o.wb1[t] = q.in ? hb * a1 : 0.f;
  }
  return o;
}

// grid: (ceil(P/256), ceil(C/kCP), N)
// Counters (profiles/r03_pmc_dcn): with the channel loop outside (18 gathers, then 9 column stores per channel) the
// waves spent 91 % of their cycles in s_waitcnt at 7 % VALU -- loads and stores share gfx9's vmcnt counter and
// complete out of order with respect to each other, so the compiler has to drain the previous stores before it may
// consume the next gathers: the store acknowledgement latency is exposed once per (load batch, store batch) round.
// Hence the TAP loop is outside: per tap the 2 x kCP pair loads of all channels of the group are in flight together
// (plus the next tap's offsets and mask), then kCP stores -- 9 rounds per thread instead of 16, each twice as big.
struct TapFoot { unsigned top, bot; float wt0, wt1, wb0, wb1; };

__device__ __forceinline__ TapFoot tap_foot(float h, float w, float m, const Conv& g) {
  const Bil q = bil(h, w, g.H, g.W);
  const int c0 = min(max(q.w0, 0), g.W - 2);
  const int r0 = min(max(q.h0, 0), g.H - 1), r1 = min(max(q.h0 + 1, 0), g.H - 1);
  // weights of the left / right corner, then their place in the loaded pair
  const float wl = (q.l ? 1.f - q.lw : 0.f), wr = (q.r ? q.lw : 0.f);
  const float a0 = q.w0 == c0 ? wl : (q.w0 < c0 ? wr : 0.f);          // w0 == -1: the right corner is element 0
  const float a1 = q.w0 == c0 ? wr : (q.w0 > c0 ? wl : 0.f);          // w0 == W-1: the left corner is element 1
  const float ht = (q.t ? 1.f - q.lh : 0.f) * m, hb = (q.b ? q.lh : 0.f) * m;
  TapFoot f;
  f.top = (unsigned)(r0 * g.W + c0) * 4u; f.bot = (unsigned)(r1 * g.W + c0) * 4u;      // byte offsets inside a plane
  // a sample outside the image (or NaN) contributes nothing: select, not a product with 0
  f.wt0 = q.in ? ht * a0 : 0.f; f.wt1 = q.in ? ht * a1 : 0.f;
  f.wb0 = q.in ? hb * a0 : 0.f; f.wb1 = q.in ? hb * a1 : 0.f;
  return f;
}

__global__ __launch_bounds__(256) void dcn_im2col_pair_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ offset,
                                                              const float* __restrict__ mask,
                                                              float* __restrict__ cols, Conv g, int nimg) {
  const int P = g.Ho * g.Wo, K = g.kh * g.kw;
  // XCD-aware order: the (P / 256) workgroups of one (channel group, image) read the same kCP input planes through nine
  // shifted footprints -- on one XCD they share its L2 (FETCH_SIZE of the 24-image call: 4.6 GB for a 142 MB input before)
  const int cgroups = (g.C + kCP - 1) / kCP;
  const PlaneTile pt = xcd_plane_tile(blockIdx.x, (P + 255) / 256, cgroups * nimg);
  if (!pt.ok) return;
  const int p = pt.tile * 256 + threadIdx.x;
  if (p >= P) return;
  const int n = pt.plane / cgroups, c0 = (pt.plane - n * cgroups) * kCP, nc = min(kCP, g.C - c0);
  const int py = p / g.Wo, px = p % g.Wo;
  const size_t plane = (size_t)g.H * g.W, KP = (size_t)K * P;
  const char* im = reinterpret_cast<const char*>(x + ((size_t)n * g.C + c0) * plane);       // wave-uniform bases
  float* out = cols + ((size_t)n * g.C + c0) * KP + p;
  const float* off_n = offset + (size_t)n * 2 * K * P + p;
  const float* msk_n = mask + (size_t)n * K * P + p;
  float oh = off_n[0], ow = off_n[P], m = msk_n[0];
  for (int t = 0; t < K; ++t) {
    const int i = t / g.kw, j = t - i * g.kw;
    const TapFoot f = tap_foot(py * g.stride - g.pad + i * g.dil + oh, px * g.stride - g.pad + j * g.dil + ow, m, g);
    pair_t a[kCP], b[kCP];
#pragma unroll
    for (int c = 0; c < kCP; ++c) {
      const char* pl = im + (size_t)(c < nc ? c : 0) * plane * 4;
      a[c] = *reinterpret_cast<const pair_t*>(pl + f.top);
      b[c] = *reinterpret_cast<const pair_t*>(pl + f.bot);
    }
    if (t + 1 < K) {                                     // the next tap's offsets and mask ride in the same round
      oh = off_n[(size_t)(2 * t + 2) * P]; ow = off_n[(size_t)(2 * t + 3) * P]; m = msk_n[(size_t)(t + 1) * P];
    }
#pragma unroll
    for (int c = 0; c < kCP; ++c)
      if (c < nc) {
        const float v = f.wt0 * a[c].x + f.wt1 * a[c].y + f.wb0 * b[c].x + f.wb1 * b[c].y;
        // the column matrix is written once and read by the GEMM much later (1.28 GB per call at stage 3); the counters
        // showed the 71 MB input fetched 4 x from HBM (profiles/r03_pmc_dcn) -- the write stream evicted the planes the
        // next taps re-read.  Streaming (non-temporal) stores leave them in L2: 0.399 -> 0.330 ms per call at stage 3
        // (profiles/r04_staged_variants_kernel_times.log).
        __builtin_nontemporal_store(v, out + (size_t)c * KP + (size_t)t * P);
      }
  }
}

// grad wrt input: scatter grad_cols * mask * corner weights (atomics); grid as im2col.  Lanes are
// neighbouring pixels, so a wave instruction touches only 2-3 lines (atomics cost per instruction x line).
__global__ __launch_bounds__(256) void dcn_col2im_kernel(const float* __restrict__ grad_cols,
                                                         const float* __restrict__ offset,
                                                         const float* __restrict__ mask,
                                                         float* __restrict__ grad_x, Conv g) {
  const int P = g.Ho * g.Wo, K = g.kh * g.kw;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int n = blockIdx.z, c0 = blockIdx.y * kCG;
  const Foot f = footprints(offset, mask, n, p, g);
  for (int c = c0; c < min(c0 + kCG, g.C); ++c) {
    float* gim = grad_x + ((size_t)n * g.C + c) * g.H * g.W;
    const float* gc = grad_cols + ((size_t)n * g.C + c) * K * P + p;
#pragma unroll
    for (int t = 0; t < kMaxTaps; ++t) {
      if (t >= K) continue;
      const Bil& q = f.q[t];
      if (!q.in) continue;
      const float gv = gc[(size_t)t * P] * f.m[t];
      if (gv == 0.f) continue;
      const float hh = 1.f - q.lh, hw = 1.f - q.lw;
      if (q.t && q.l) unsafeAtomicAdd(gim + q.h0 * g.W + q.w0, hh * hw * gv);
      if (q.t && q.r) unsafeAtomicAdd(gim + q.h0 * g.W + q.w0 + 1, hh * q.lw * gv);
      if (q.b && q.l) unsafeAtomicAdd(gim + (q.h0 + 1) * g.W + q.w0, q.lh * hw * gv);
      if (q.b && q.r) unsafeAtomicAdd(gim + (q.h0 + 1) * g.W + q.w0 + 1, q.lh * q.lw * gv);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// grad wrt input without atomics.  deform_groups == 1: the nine sampling positions of an output pixel
// are shared by ALL channels, so the scatter  grad_x[c] += S^T (mask * grad_cols[c])  uses one sparse
// matrix S (4 non-zeros per (tap, pixel) row) for every channel of an image.  The call builds S^T once
// -- a counting sort of the <= 4*K*P corner entries by destination pixel: count / scan / fill, tiny --
// and then every destination pixel GATHERS its ~36 contributions for a group of channels: lanes are
// neighbouring destination pixels, whose entries point at neighbouring columns, so the loads coalesce.
// The scatter it replaces was the top kernel of the step (1.4 ms x 26 calls, atomic-request bound).
// ---------------------------------------------------------------------------------------------
struct Entry { int src; float w; };            // src = tap * P + output pixel

template <bool FILL>
__global__ __launch_bounds__(256) void dcn_revmap_kernel(const float* __restrict__ offset,
                                                         const float* __restrict__ mask, int* __restrict__ cursor,
                                                         Entry* __restrict__ rec, Conv g) {
  const int P = g.Ho * g.Wo, K = g.kh * g.kw, HW = g.H * g.W;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int t = blockIdx.y, n = blockIdx.z;
  const int py = p / g.Wo, px = p % g.Wo;
  const int i = t / g.kw, j = t % g.kw;
  const size_t o = ((size_t)n * 2 * K + 2 * t) * P + p;
  const Bil q = bil(py * g.stride - g.pad + i * g.dil + offset[o], px * g.stride - g.pad + j * g.dil + offset[o + P],
                    g.H, g.W);
  if (!q.in) return;
  const float m = FILL ? mask[((size_t)n * K + t) * P + p] : 0.f;
  const float hh = 1.f - q.lh, hw = 1.f - q.lw;
  int* cur = cursor + ((size_t)n * K + t) * HW;      // bins are (image, tap, destination pixel): tap-major lists keep
  const int src = t * P + p;                         // neighbouring destination pixels in step (coalesced gathers)
  auto put = [&](bool ok, int pix, float w) {
    if (!ok) return;
    if (FILL) rec[atomicAdd(cur + pix, 1)] = Entry{src, w * m};
    else atomicAdd(cur + pix, 1);
  };
  put(q.t && q.l, q.h0 * g.W + q.w0, hh * hw);
  put(q.t && q.r, q.h0 * g.W + q.w0 + 1, hh * q.lw);
  put(q.b && q.l, (q.h0 + 1) * g.W + q.w0, q.lh * hw);
  put(q.b && q.r, (q.h0 + 1) * g.W + q.w0 + 1, q.lh * q.lw);
}

// block per image: counts -> first[] (exclusive prefix, offset by the image's slice of `rec`), cursor = first
__global__ __launch_bounds__(1024) void dcn_revmap_scan_kernel(int* __restrict__ cursor, int* __restrict__ first,
                                                               int HW, int per_image) {
  __shared__ int s_part[1024];
  const int n = blockIdx.x;
  int* cnt = cursor + (size_t)n * HW;
  int* fst = first + (size_t)n * HW;
  const int per = (HW + 1023) / 1024;
  const int r0 = min(HW, (int)threadIdx.x * per), r1 = min(HW, r0 + per);
  int sum = 0;
  for (int i = r0; i < r1; ++i) sum += cnt[i];
  s_part[threadIdx.x] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int a = (int)threadIdx.x >= d ? s_part[threadIdx.x - d] : 0;
    __syncthreads();
    s_part[threadIdx.x] += a;
    __syncthreads();
  }
  int run = n * per_image + s_part[threadIdx.x] - sum;
  for (int i = r0; i < r1; ++i) { const int c = cnt[i]; fst[i] = run; cnt[i] = run; run += c; }
}

constexpr int kGC = 16;                        // channels per thread of the gather
// grid: (ceil(HW/256), ceil(C/kGC), N); after the fill pass cursor[i] = end of bin i's entries
__global__ __launch_bounds__(256) void dcn_col2im_gather_kernel(const float* __restrict__ grad_cols,
                                                                const int* __restrict__ first,
                                                                const int* __restrict__ last,
                                                                const Entry* __restrict__ rec,
                                                                float* __restrict__ grad_x, Conv g) {
  const int P = g.Ho * g.Wo, K = g.kh * g.kw, HW = g.H * g.W;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= HW) return;
  const int n = blockIdx.z, c0 = blockIdx.y * kGC;
  float acc[kGC];
#pragma unroll
  for (int c = 0; c < kGC; ++c) acc[c] = 0.f;
  const size_t KP = (size_t)K * P;
  const float* gc = grad_cols + ((size_t)n * g.C + c0) * KP;
  const int nc = min(kGC, g.C - c0);
  for (int t = 0; t < K; ++t) {
    const size_t bin = ((size_t)n * K + t) * HW + pix;
    const int e0 = first[bin], e1 = last[bin];
    if (nc == kGC) {
      int e = e0;
      for (; e + 1 < e1; e += 2) {               // two entries in flight: 32 independent loads per iteration
        const Entry r0 = rec[e], r1 = rec[e + 1];
        float v0[kGC], v1[kGC];
#pragma unroll
        for (int c = 0; c < kGC; ++c) { v0[c] = gc[(size_t)c * KP + r0.src]; v1[c] = gc[(size_t)c * KP + r1.src]; }
#pragma unroll
        for (int c = 0; c < kGC; ++c) acc[c] += r0.w * v0[c] + r1.w * v1[c];
      }
      if (e < e1) {
        const Entry r = rec[e];
#pragma unroll
        for (int c = 0; c < kGC; ++c) acc[c] += r.w * gc[(size_t)c * KP + r.src];
      }
    } else {
      for (int e = e0; e < e1; ++e) {
        const Entry r = rec[e];
        for (int c = 0; c < nc; ++c) acc[c] += r.w * gc[(size_t)c * KP + r.src];
      }
    }
  }
  for (int c = 0; c < nc; ++c) grad_x[((size_t)n * g.C + c0 + c) * HW + pix] = acc[c];
}

// ---------------------------------------------------------------------------------------------
// grad wrt input, gather through an LDS window (round 6; 3x3 / stride 1 / dilation 1 layers -- all 26 of the backbone's).
// The gather above reads `grad_cols` with one 4-byte load per (entry, channel) and lane; with learned offsets the sources
// of neighbouring destination pixels are no longer neighbours, a wave instruction touches 10-20 cache lines for 256
// useful bytes and the kernel runs at the L2's line rate: 150 us per call on integer offsets, 445 us in the trained-like
// step, 1.2 ms on i.i.d. 1.5-pixel offsets (N = 6, C = 256, 58 x 100).  Here a workgroup owns a 16 x 32 tile of destination
// pixels (2 rows per thread) and 8 channels; per tap it copies the window of `grad_cols` that can reach the tile -- tile + 1
// (bilinear footprint) + kGHalo pixels of learned offset on every side -- into LDS with coalesced row segments and gathers
// from LDS (one ds_read_b32 per entry and channel, 8 channels behind one address); an entry whose source lies outside the
// window (|offset| > kGHalo) takes the global load.  Measured (kernel alone, same shape): 245 us on integer offsets, 360 us on
// smooth 1.5-pixel offsets, 270 us in the step (-4.5 ms per step).  What the sweep said (profiles/
// r06_kbench_dcn_col2im_lds_gather.log): the copy into LDS is 2/3 of the kernel and is bound by its instruction count, not by
// HBM (8 x 32, 16 x 32 and 32 x 32 tiles -- 3.1 x / 2.3 x / 1.8 x the column matrix -- take 274 / 245 / 311 us; float4 pieces
// of rows, dword aligned on the global side, 1.5 x slower than 4-byte loads; a 6-pixel halo 13 % slower than 4 or 5;
// global_load_lds_dword straight into the lane-linear LDS image instead of load + ds_write: 245 -> 237 / 360 -> 328 us, kept).
// ---------------------------------------------------------------------------------------------
#ifndef VIDAR_DCN_GLDS
#define VIDAR_DCN_GLDS 1
#endif
#ifndef VIDAR_DCN_GRY
#define VIDAR_DCN_GRY 2
#endif
#ifndef VIDAR_DCN_GLC
#define VIDAR_DCN_GLC 8
#endif
constexpr int kGRY = VIDAR_DCN_GRY;           // destination rows per thread (rows ry, ry + 8, ...)
constexpr int kGTH = 8 * kGRY, kGTW = 32;     // destination tile: 256 threads x kGRY pixels
#ifndef VIDAR_DCN_GHALO
#define VIDAR_DCN_GHALO 5
#endif
constexpr int kGHalo = VIDAR_DCN_GHALO;       // learned offset served from LDS, pixels
constexpr int kGWH = kGTH + 2 * kGHalo + 2;   // source rows of the window
constexpr int kGWW = kGTW + 2 * kGHalo + 2;   // 42 source columns ...
constexpr int kGWP = (kGWW + 3) / 4 * 4;      // ... at a pitch of 44
constexpr int kGWS = kGWH * kGWP;             // floats per channel (32 x 32 tile: 1 848 for 1 024 pixels -- 1.8 x, 8 x 32: 3.1 x)
constexpr int kGLC = VIDAR_DCN_GLC;           // channels per workgroup (LDS: kGLC * kGWS * 4 bytes = 59 KB)
constexpr int kGPos = (kGWS + 255) / 256;     // window positions a thread stages per tap

__global__ __launch_bounds__(256) void dcn_col2im_gather_lds_kernel(
    const float* __restrict__ grad_cols, const int* __restrict__ first, const int* __restrict__ last,
    const Entry* __restrict__ rec, float* __restrict__ grad_x, Conv g, int tiles_x, int tiles_y, int nimg, float inv_wo) {
  __shared__ __attribute__((aligned(16))) float s_gc[kGLC * kGWS];
  constexpr int K = 9;
  const int P = g.Ho * g.Wo, HW = g.H * g.W;
  // XCD-aware order: the tiles of one (channel group, image) share their window halos (and the 128-byte lines the
  // 176-byte row pieces straddle) through ONE L2
  const int cgroups = (g.C + kGLC - 1) / kGLC;
  const PlaneTile pt = xcd_plane_tile(blockIdx.x, tiles_x * tiles_y, cgroups * nimg);
  if (!pt.ok) return;
  const int tile = pt.tile, tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
  const int n = pt.plane / cgroups, c0 = (pt.plane - n * cgroups) * kGLC, nc = min(kGLC, g.C - c0);
  const int ry = threadIdx.x / kGTW, rx = threadIdx.x % kGTW;
  const int dx = txi * kGTW + rx;
  const size_t KP = (size_t)K * P;
  const float* gc = grad_cols + ((size_t)n * g.C + c0) * KP;
  float acc[kGRY][kGLC];
#pragma unroll
  for (int k = 0; k < kGRY; ++k)
#pragma unroll
    for (int c = 0; c < kGLC; ++c) acc[k][c] = 0.f;
  for (int t = 0; t < K; ++t) {
    const int i = t / 3, j = t - i * 3;
    // source (= output-pixel) coordinates of window position (0, 0): a source row py reaches destination rows
    // floor(py - pad + i + off) + {0, 1}, so rows dy0 .. dy0 + kGTH - 1 are reached from py in [dy0 + pad - i - 1 - halo, ..]
    const int wy0 = tyi * kGTH + g.pad - i - kGHalo - 1, wx0 = txi * kGTW + g.pad - j - kGHalo - 1;
    const float* gct = gc + (size_t)t * P;
    __syncthreads();                          // the previous tap's gathers are done
#if VIDAR_DCN_GLDS
    // staging straight into LDS (global_load_lds_dword: destination = wave-uniform base + lane x 4 -- window position
    // e = thread + k * 256 is lane-linear by construction); positions outside the image are never read and stay as they are
    {
      const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
#pragma unroll
      for (int k = 0; k < kGPos; ++k) {
        const int e = threadIdx.x + k * 256;
        const int r = e / kGWP, cc = e - r * kGWP;
        const int sy = wy0 + r, sx = wx0 + cc;
        const bool ok = e < kGWS && sy >= 0 && sy < g.Ho && sx >= 0 && sx < g.Wo && cc < kGWW;
        const int off = ok ? sy * g.Wo + sx : 0;
        if (ok) {
#pragma unroll
          for (int c = 0; c < kGLC; ++c)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gct + (size_t)(c < nc ? c : 0) * KP + off),
                                             (__attribute__((address_space(3))) void*)(s_gc + c * kGWS + k * 256 + wave * 64), 4, 0, 0);
        }
      }
    }
#else
    // staging in rounds of 4 window positions per thread: their 4 x kGLC loads are requested before the first LDS store.
    // (float4 pieces of rows -- the global side is only dword aligned -- measured 1.5 x SLOWER than these 4-byte loads,
    //  profiles/r06_kbench_dcn_col2im_lds_gather.log)
#pragma unroll
    for (int k0 = 0; k0 < kGPos; k0 += 4) {
      float v[4][kGLC];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = threadIdx.x + (k0 + u) * 256;
        const int r = e / kGWP, cc = e - r * kGWP;
        const int sy = wy0 + r, sx = wx0 + cc;
        ok[u] = e < kGWS && sy >= 0 && sy < g.Ho && sx >= 0 && sx < g.Wo && cc < kGWW;
        const int off = ok[u] ? sy * g.Wo + sx : 0;
#pragma unroll
        for (int c = 0; c < kGLC; ++c) v[u][c] = gct[(size_t)(c < nc ? c : 0) * KP + off];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = threadIdx.x + (k0 + u) * 256;
        if (e < kGWS) {
#pragma unroll
          for (int c = 0; c < kGLC; ++c) s_gc[c * kGWS + e] = ok[u] ? v[u][c] : 0.f;
        }
      }
    }
#endif
    __syncthreads();
    if (dx >= g.W) continue;
#pragma unroll
    for (int k = 0; k < kGRY; ++k) {
      const int dy = tyi * kGTH + ry + 8 * k;
      if (dy >= g.H) continue;
      const size_t bin = ((size_t)n * K + t) * HW + dy * g.W + dx;
      const int e1 = last[bin];
      // entries of this (tap, destination pixel): the next entry is requested before the current one is used
      int e = first[bin];
      Entry nxt = e < e1 ? rec[e] : Entry{0, 0.f};
      while (e < e1) {
        const Entry r = nxt;
        ++e;
        if (e < e1) nxt = rec[e];
        const int p = r.src - t * P;
        const int py = (int)((p + 0.5f) * inv_wo), px = p - py * g.Wo;   // exact: (p + 0.5) / Wo is >= 0.5 / Wo off an integer
        const int ly = py - wy0, lx = px - wx0;
        if (ly >= 0 && ly < kGWH && lx >= 0 && lx < kGWW) {
          const float* q = s_gc + ly * kGWP + lx;
#pragma unroll
          for (int c = 0; c < kGLC; ++c) acc[k][c] += r.w * q[c * kGWS];
        } else {                              // the source sits beyond the halo: rare, from global memory
          for (int c = 0; c < nc; ++c) acc[k][c] += r.w * gct[(size_t)c * KP + p];
        }
      }
    }
  }
  if (dx < g.W) {
#pragma unroll
    for (int k = 0; k < kGRY; ++k) {
      const int dy = tyi * kGTH + ry + 8 * k;
      if (dy < g.H)
        for (int c = 0; c < nc; ++c) grad_x[((size_t)n * g.C + c0 + c) * HW + dy * g.W + dx] = acc[k][c];
    }
  }
}

int g_dcn_variant = 1;         // bit 0: col2im gathers through the LDS window kernel where it applies (vidar_dcn_set_variant)

// grad wrt offset and mask: thread per (n, tap, pixel), loop over channels (no atomics)
// grid: (ceil(P/256), K, N)
__global__ __launch_bounds__(256) void dcn_col2im_coord_kernel(
    const float* __restrict__ grad_cols, const float* __restrict__ x,
    const float* __restrict__ offset, const float* __restrict__ mask,
    float* __restrict__ grad_offset, float* __restrict__ grad_mask, Conv g) {
  const int P = g.Ho * g.Wo, K = g.kh * g.kw;
  // (the XCD-aware order of the other kernels -- the K workgroups of a pixel block on one XCD -- measured 8 % slower here)
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int t = blockIdx.y, n = blockIdx.z;
  const int py = p / g.Wo, px = p % g.Wo;
  const int i = t / g.kw, j = t % g.kw;
  const size_t o = ((size_t)n * 2 * K + 2 * t) * P + p;
  const float h = py * g.stride - g.pad + i * g.dil + offset[o];
  const float w = px * g.stride - g.pad + j * g.dil + offset[o + P];
  const float m = mask[((size_t)n * K + t) * P + p];
  const Bil q = bil(h, w, g.H, g.W);
  float gh = 0.f, gw = 0.f, gm = 0.f;
  if (q.in && g.W >= 2) {
    // pair loads like dcn_im2col_pair_kernel: per row one 8-byte load; the corner -> pair-element map and the zero
    // padding are folded into 12 coefficients computed once per (pixel, tap)
    const float hh = 1.f - q.lh, hw = 1.f - q.lw;
    const int c0 = min(max(q.w0, 0), g.W - 2);
    const int top = min(max(q.h0, 0), g.H - 1) * g.W + c0, bot = min(max(q.h0 + 1, 0), g.H - 1) * g.W + c0;
    const float tl = (q.t && q.l) ? 1.f : 0.f, tr = (q.t && q.r) ? 1.f : 0.f;
    const float bl = (q.b && q.l) ? 1.f : 0.f, br = (q.b && q.r) ? 1.f : 0.f;
    const bool same = q.w0 == c0, left = q.w0 < c0;          // left: w0 == -1 (right corner is element 0)
    auto e0 = [&](float cl, float cr) { return same ? cl : (left ? cr : 0.f); };
    auto e1 = [&](float cl, float cr) { return same ? cr : (left ? 0.f : cl); };
    const float m_t0 = e0(hh * hw * tl, hh * q.lw * tr), m_t1 = e1(hh * hw * tl, hh * q.lw * tr);
    const float m_b0 = e0(q.lh * hw * bl, q.lh * q.lw * br), m_b1 = e1(q.lh * hw * bl, q.lh * q.lw * br);
    const float h_t0 = e0(-hw * tl, -q.lw * tr), h_t1 = e1(-hw * tl, -q.lw * tr);
    const float h_b0 = e0(hw * bl, q.lw * br), h_b1 = e1(hw * bl, q.lw * br);
    const float w_t0 = e0(-hh * tl, hh * tr), w_t1 = e1(-hh * tl, hh * tr);
    const float w_b0 = e0(-q.lh * bl, q.lh * br), w_b1 = e1(-q.lh * bl, q.lh * br);
    int c = 0;
    // one channel per iteration keeps 3 loads in flight per thread and waits for them 256 times in a row (the compiler
    // does not cluster the loads of an unrolled body by itself): the loads of kCB channels are issued together, the
    // sums are still taken in channel order (0.654 -> 0.577 ms at stage 4, profiles/r04_staged_variants_kernel_times.log)
    constexpr int kCB = 8;
    const unsigned top_b = (unsigned)top * 4u, bot_b = (unsigned)bot * 4u, p_b = (unsigned)p * 4u;   // plane < 2^30 B
    const size_t plane_b = (size_t)g.H * g.W * 4, col_b = (size_t)K * P * 4;
    for (; c + kCB <= g.C; c += kCB) {
      // wave-uniform bases + 32-bit lane offsets: scalar-base addressing, no 64-bit address pair per load
      const char* im0 = reinterpret_cast<const char*>(x) + ((size_t)n * g.C + c) * plane_b;
      const char* gc0 = reinterpret_cast<const char*>(grad_cols) + (((size_t)n * g.C + c) * K + t) * (size_t)P * 4;
      pair_t a[kCB], b[kCB];
      float gc[kCB];
#pragma unroll
      for (int u = 0; u < kCB; ++u) {
        gc[u] = *reinterpret_cast<const float*>(gc0 + u * col_b + p_b);
        a[u] = *reinterpret_cast<const pair_t*>(im0 + u * plane_b + top_b);
        b[u] = *reinterpret_cast<const pair_t*>(im0 + u * plane_b + bot_b);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < kCB; ++u) {
        gm += gc[u] * (m_t0 * a[u].x + m_t1 * a[u].y + m_b0 * b[u].x + m_b1 * b[u].y);
        gh += gc[u] * (h_t0 * a[u].x + h_t1 * a[u].y + h_b0 * b[u].x + h_b1 * b[u].y);
        gw += gc[u] * (w_t0 * a[u].x + w_t1 * a[u].y + w_b0 * b[u].x + w_b1 * b[u].y);
      }
    }
    for (; c < g.C; ++c) {
      const float* im = x + ((size_t)n * g.C + c) * g.H * g.W;
      const float gc = grad_cols[(((size_t)n * g.C + c) * K + t) * P + p];
      const pair_t a = *reinterpret_cast<const pair_t*>(im + top);
      const pair_t b = *reinterpret_cast<const pair_t*>(im + bot);
      gm += gc * (m_t0 * a.x + m_t1 * a.y + m_b0 * b.x + m_b1 * b.y);
      gh += gc * (h_t0 * a.x + h_t1 * a.y + h_b0 * b.x + h_b1 * b.y);
      gw += gc * (w_t0 * a.x + w_t1 * a.y + w_b0 * b.x + w_b1 * b.y);
    }
  } else if (q.in) {
    const float hh = 1.f - q.lh, hw = 1.f - q.lw;
    for (int c = 0; c < g.C; ++c) {
      const float* im = x + ((size_t)n * g.C + c) * g.H * g.W;
      const float gc = grad_cols[(((size_t)n * g.C + c) * K + t) * P + p];
      const float v1 = (q.t && q.l) ? im[q.h0 * g.W + q.w0] : 0.f;
      const float v2 = (q.t && q.r) ? im[q.h0 * g.W + q.w0 + 1] : 0.f;
      const float v3 = (q.b && q.l) ? im[(q.h0 + 1) * g.W + q.w0] : 0.f;
      const float v4 = (q.b && q.r) ? im[(q.h0 + 1) * g.W + q.w0 + 1] : 0.f;
      gm += gc * (hh * hw * v1 + hh * q.lw * v2 + q.lh * hw * v3 + q.lh * q.lw * v4);
      gh += gc * (-hw * v1 - q.lw * v2 + hw * v3 + q.lw * v4);
      gw += gc * (-hh * v1 + hh * v2 - q.lh * v3 + q.lh * v4);
    }
  }
  grad_offset[o] = gh * m;
  grad_offset[o + P] = gw * m;
  grad_mask[((size_t)n * K + t) * P + p] = gm;
}

inline bool dcn_bad(int N, const Conv& g) {
  return N < 0 || g.C <= 0 || g.H <= 0 || g.W <= 0 || g.Ho <= 0 || g.Wo <= 0 || g.kh <= 0 ||
         g.kw <= 0 || g.stride <= 0 || g.dil <= 0 || g.pad < 0;
}

}  // namespace

extern "C" {

int vidar_dcn_set_variant(int variant) {
  const int prev = g_dcn_variant;
  if (variant >= 0 && variant <= 1) g_dcn_variant = variant;
  return prev;
}

int vidar_dcn_im2col_f32(const float* x, const float* offset, const float* mask, float* cols, int N,
                         int C, int H, int W, int Ho, int Wo, int kh, int kw, int stride, int pad,
                         int dil, void* stream) {
  VIDAR_ENTER();
  Conv g{C, H, W, Ho, Wo, kh, kw, stride, pad, dil};
  if (dcn_bad(N, g)) return VIDAR_ERR_BAD_ARG;
  if (N == 0) return 0;
  if (kh * kw > kMaxTaps) return VIDAR_ERR_BAD_ARG;
  if (W >= 2)
    hipLaunchKernelGGL(dcn_im2col_pair_kernel, dim3(xcd_plane_grid((Ho * Wo + 255) / 256, ((C + kCP - 1) / kCP) * N)), dim3(256),
                       0, (hipStream_t)stream, x, offset, mask, cols, g, N);
  else
    hipLaunchKernelGGL(dcn_im2col_kernel, dim3((Ho * Wo + 255) / 256, (C + kCG - 1) / kCG, N), dim3(256),
                       0, (hipStream_t)stream, x, offset, mask, cols, g);
  return vidar_last_error();
}

size_t vidar_dcn_col2im_workspace_bytes(int N, int H, int W, int Ho, int Wo, int kh, int kw) {
  if (N <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || kh <= 0 || kw <= 0) return 0;
  const size_t hw = (size_t)N * kh * kw * H * W, ent = (size_t)N * kh * kw * Ho * Wo * 4;
  if (ent >= (1ull << 31)) return 0;
  return sizeof(int) * 2 * hw + sizeof(Entry) * ent;
}

int vidar_dcn_col2im_f32(const float* grad_cols, const float* x, const float* offset,
                         const float* mask, float* grad_x, float* grad_offset, float* grad_mask,
                         int N, int C, int H, int W, int Ho, int Wo, int kh, int kw, int stride,
                         int pad, int dil, void* workspace, size_t workspace_bytes, void* stream) {
  VIDAR_ENTER();
  Conv g{C, H, W, Ho, Wo, kh, kw, stride, pad, dil};
  if (dcn_bad(N, g)) return VIDAR_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (N == 0) return 0;
  if (kh * kw > kMaxTaps) return VIDAR_ERR_BAD_ARG;
  const int P = Ho * Wo, K = kh * kw, HW = H * W;
  if (workspace) {
    const size_t need = vidar_dcn_col2im_workspace_bytes(N, H, W, Ho, Wo, kh, kw);
    if (need == 0 || workspace_bytes < need) return VIDAR_ERR_BAD_ARG;
    int* cursor = (int*)workspace;
    int* first = cursor + (size_t)N * K * HW;
    Entry* rec = (Entry*)(first + (size_t)N * K * HW);
    hipError_t e = hipMemsetAsync(cursor, 0, sizeof(int) * (size_t)N * K * HW, s);
    if (e != hipSuccess) return (int)e;
    const dim3 rgrid((P + 255) / 256, K, N);
    hipLaunchKernelGGL(dcn_revmap_kernel<false>, rgrid, dim3(256), 0, s, offset, mask, cursor, rec, g);
    // a (image, tap) list never holds more than 4 P entries, so every list owns a fixed slice of `rec` and is scanned
    // by its own workgroup -- N K workgroups instead of N (a scan per image kept 6 CUs busy for 0.1 ms per call)
    hipLaunchKernelGGL(dcn_revmap_scan_kernel, dim3(N * K), dim3(1024), 0, s, cursor, first, HW, P * 4);
    hipLaunchKernelGGL(dcn_revmap_kernel<true>, rgrid, dim3(256), 0, s, offset, mask, cursor, rec, g);
    if ((g_dcn_variant & 1) && kh == 3 && kw == 3 && stride == 1 && dil == 1 && Ho == H && Wo == W) {
      const int tiles_x = (W + kGTW - 1) / kGTW, tiles_y = (H + kGTH - 1) / kGTH;
      hipLaunchKernelGGL(dcn_col2im_gather_lds_kernel, dim3(xcd_plane_grid(tiles_x * tiles_y, ((C + kGLC - 1) / kGLC) * N)), dim3(256), 0,
                         s, grad_cols, first, cursor, rec, grad_x, g, tiles_x, tiles_y, N, 1.0f / (float)Wo);
    } else {
      hipLaunchKernelGGL(dcn_col2im_gather_kernel, dim3((HW + 255) / 256, (C + kGC - 1) / kGC, N), dim3(256), 0, s,
                         grad_cols, first, cursor, rec, grad_x, g);
    }
  } else {
    hipError_t e = hipMemsetAsync(grad_x, 0, sizeof(float) * (size_t)N * C * H * W, s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(dcn_col2im_kernel, dim3((P + 255) / 256, (C + kCG - 1) / kCG, N), dim3(256), 0, s, grad_cols,
                       offset, mask, grad_x, g);
  }
  hipLaunchKernelGGL(dcn_col2im_coord_kernel, dim3((P + 255) / 256, K, N), dim3(256), 0,
                     s, grad_cols, x, offset, mask, grad_offset, grad_mask, g);
  return vidar_last_error();
}

}  // extern "C"
