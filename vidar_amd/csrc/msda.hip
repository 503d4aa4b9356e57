// Multi-scale deformable attention (bilinear gather / scatter) for gfx950.
//
// Replaces the op the reference reaches through mmcv-full 1.4.0 (NOT vendored in the reference):
//   mmcv._ext.ms_deform_attn_forward / ms_deform_attn_backward, called from
//   projects/mmdet3d_plugin/bevformer/modules/multi_scale_deformable_attn_function.py:118-124, :150-160
// used by TemporalSelfAttention (temporal_self_attention.py:246-248), MSDeformableAttention3D
// (spatial_cross_attention.py:384-391) and PredictionMSDeformableAttention (vidar_decoder.py:500-505).
// Semantics (== grid_sample(bilinear, zeros, align_corners=False) per level, weighted sum):
//   pixel = loc * (W, H) - 0.5 ; sample counts iff -1 < pixel < size ; each corner zero outside.
//
// Layout in HBM (as the reference): value [B, Nv, H, 32] f32 with Nv = sum_l H_l*W_l,
// loc [B, Nq, H, L, P, 2] (x, y in [0,1]), w [B, Nq, H, L, P], out [B, Nq, H*32].
//
// Mapping (wave64): one (b, q, head) "item" is owned by 8 adjacent lanes, each lane carrying 4 of
// the 32 head channels as a float4, so every corner fetch of an item is one 128-byte line and a
// wave covers 8 items (= all 8 heads of one query) per instruction.  Sampling locations and
// weights of the 32 items of a workgroup are staged through LDS with fully coalesced loads
// (they are contiguous in HBM); one thread per sample turns them into a 32-byte record (corner weights
// and byte offsets) that the item's 8 lanes re-read as broadcasts.  Workgroups are remapped so that each
// XCD walks a contiguous band of queries: neighbouring BEV queries sample neighbouring pixels, so
// a band keeps its slice of `value` resident in that XCD's private 4 MiB L2.
// Backward: two grad_value strategies, chosen per call through the `workspace` argument -- the destination-binned
// LDS accumulation below (default for real sizes) and the one-launch atomic scatter `msda_bwd_kernel` (small launches);
// grad_loc / grad_w come from a gather shaped like the forward.  The fused entry points read the raw Linear outputs
// (softmax / offset normalisation / reference add happen in the staging phase, see `Prep`).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vidar_hip.h"
#include "vidar_common.h"

namespace {

constexpr int kCh = 32;            // channels per head (ViDAR: embed 256 / 8 heads)
constexpr int kLanes = kCh / 4;    // lanes per item
constexpr int kThreads = 256;
constexpr int kItems = kThreads / kLanes;  // 32 items per workgroup
constexpr int kMaxLP = 64;         // max levels*points staged per item

__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  // blocks are dealt round-robin to the 8 XCDs; give XCD k the k-th contiguous band (measured within
  // 1 % of plain order on random reference points, profiles/r01_kbench_msda_xcd_remap_ab.log; kept for
  // the spatially coherent queries the model produces)
  const int per = (nblocks + 7) >> 3;
  return (bid & 7) * per + (bid >> 3);
}

// Which 32 (b, q, head) items a workgroup of the gather kernels owns.
//   banded (round 1-3): 32 consecutive items = 4 queries x 8 heads, XCD k walks the k-th contiguous band of queries.
//     Every wave then touches all 8 head planes of `value`: the working set of an XCD's 4 MiB L2 is 8 planes wide and
//     the counters show it -- SpatialCrossAttention fetches the 189 MB `value` tensor 16 x from HBM, L2 hit rate 46-53 %.
//   head-major: 32 consecutive (b, q) pairs of ONE head, head = blockIdx % H.  Workgroups are dealt round-robin to the
//     8 XCDs, so with the 8 heads of ViDAR XCD k only ever reads head k's plane (3.9 MB per camera at the FPN sizes:
//     it fits the XCD's L2) and all XCDs walk the queries at the same pace.
// item(it) = base + it * stride covers both; `vidar_msda_set_item_order` switches (A/B, results are identical).
struct ItemMap {
  int64_t base;
  int stride, nvalid;
  __device__ __forceinline__ int64_t at(int it) const { return base + (int64_t)it * stride; }
};

__device__ __forceinline__ bool item_map(ItemMap& m, int bid, int nblocks, int64_t n_items, int H, int head_major) {
  if (head_major) {
    const int64_t BQ = n_items / H;
    const int h = bid % H;
    const int64_t bq0 = (int64_t)(bid / H) * kItems;
    if (bq0 >= BQ) return false;
    m.base = bq0 * H + h; m.stride = H; m.nvalid = (int)min((int64_t)kItems, BQ - bq0);
    return true;
  }
  const int blk = xcd_remap(bid, nblocks);
  if (blk >= nblocks) return false;
  m.base = (int64_t)blk * kItems; m.stride = 1; m.nvalid = (int)min((int64_t)kItems, n_items - m.base);
  return true;
}

// pixel coordinate of a normalised location: ONE rounding (fma), the same in every kernel -- the
// binned backward computes the tile of a sample in one kernel and its window offset in another
__device__ __forceinline__ float pix(float v, int n) { return __fmaf_rn(v, (float)n, -0.5f); }

struct Corner {
  int64_t o00, o01, o10, o11;  // element offsets (in floats) or -1
  float w00, w01, w10, w11;
  float lh, lw;
};

__device__ __forceinline__ Corner corners(float x, float y, int Hl, int Wl, int64_t base,
                                          int row_stride) {
  // x,y already in pixel units (loc*size - 0.5) and known to be inside (-1, size)
  Corner c;
  const int h0 = (int)floorf(y), w0 = (int)floorf(x);
  const int h1 = h0 + 1, w1 = w0 + 1;
  c.lh = y - h0; c.lw = x - w0;
  const float hh = 1.f - c.lh, hw = 1.f - c.lw;
  c.w00 = hh * hw; c.w01 = hh * c.lw; c.w10 = c.lh * hw; c.w11 = c.lh * c.lw;
  const bool t = h0 >= 0, b = h1 <= Hl - 1, l = w0 >= 0, r = w1 <= Wl - 1;
  c.o00 = (t && l) ? base + ((int64_t)h0 * Wl + w0) * row_stride : -1;
  c.o01 = (t && r) ? base + ((int64_t)h0 * Wl + w1) * row_stride : -1;
  c.o10 = (b && l) ? base + ((int64_t)h1 * Wl + w0) * row_stride : -1;
  c.o11 = (b && r) ? base + ((int64_t)h1 * Wl + w1) * row_stride : -1;
  return c;
}

// ---------------------------------------------------------------------------------------------
// Fused operand preparation.  The modules feed the op with the raw outputs of two Linear layers:
//   off_raw   [bs, Nq, H, Qn, L, P, 2]   sampling offsets in pixels          (sampling_offsets GEMM)
//   logit_raw [bs, Nq, H, Qn, L*P]       attention logits                    (attention_weights GEMM)
// and reference points ref [bs*Qn, Nq, R, 2]; the reference turns them into the op's operands with a
// softmax, a division by the level size, an add and (TemporalSelfAttention, Qn = 2 BEV queue entries)
// two permute copies (temporal_self_attention.py:218-245, spatial_cross_attention.py:359-383,
// vidar_decoder.py:463-490):
//   loc[b', q, h, l, p] = ref[b', q, r(l, p)] + off / (W_l, H_l),  r = l (mode 0) or p % R (mode 1, SCA:
//   point p belongs to pillar anchor p % R), b' = b * Qn + qn;   w = softmax over the L*P logits.
// Here the staging phase of the kernels does that arithmetic in LDS.  The forward also writes loc / w
// (the tensors the unfused path would have saved for the backward); the backward turns grad_loc / grad_w
// into grad_off_raw / grad_logit_raw in its store phase.
// ---------------------------------------------------------------------------------------------
struct Prep {
  const float* off_raw;      // nullptr: plain operands (loc / w given)
  const float* logit_raw;
  const float* ref;
  float* loc_out;            // forward: saved operands;   backward: unused
  float* w_out;
  float* g_off_raw;          // backward outputs
  float* g_logit_raw;
  int Qn, R, mode;
  // merge != 0 (fused forward only): the Qn queue entries of a (batch, query, head) are ONE item whose Qn * L * P samples
  // read Qn different `value` planes, and the kernel stores their MEAN -- TemporalSelfAttention's
  // `output.view(bs, Qn, Nq, C).mean(1)` (temporal_self_attention.py:256-264) without the [bs*Qn, Nq, C] round trip.
  // Items then count (b, q, h) of the bs batch rows; the saved loc / w keep their [bs*Qn, ...] layout for the backward.
  int merge;
};

// merged queue entries in the backward: grad_out is [bs, Nq, H*C], item' = ((b*Qn + qn)*Nq + q)*H + h reads the line of
// (b, q, h), scaled by 1/Qn
struct GoMap {
  int Qn, NqH;
  float scale;
  __device__ __forceinline__ int64_t at(int64_t item) const {
    if (Qn <= 1) return item;
    const int64_t bp = item / NqH;
    return item - (bp - bp / Qn) * NqH;
  }
};

// index of (item, lp) inside the raw [bs, Nq, H, Qn, LP] layout
__device__ __forceinline__ int64_t raw_index(int64_t item, int lp, int H, int Nq, int LP, int Qn) {
  const int h = (int)(item % H);
  const int64_t bq = item / H;
  const int64_t q = bq % Nq, bp = bq / Nq;
  const int64_t b = bp / Qn, qn = bp % Qn;
  return ((((b * Nq + q) * H + h) * Qn + qn) * LP) + lp;
}

// ---------------------------------------------------------------------------------------------
// Gather kernels (forward, and grad_loc / grad_w of the backward).  A workgroup owns 32 consecutive
// (b, q, head) items; item = 8 adjacent lanes x float4 = one 128-byte line per corner fetch.  Round 2 let
// each of the 8 lanes of an item redo the whole corner arithmetic of every sample (~75 VALU instructions
// per sample, 8-fold redundant, 4 loads in flight per wave).  Now the corner arithmetic is done ONCE per
// sample, by one thread, into a 32-byte LDS record; the gather loop of an item then is two (forward) or one
// (backward) broadcast ds_read_b128 + four `global_load_dwordx4 v, voffset, s[value]` per sample, unrolled
// by 4 (16 lines in flight per wave).  Records of item k are skewed by k records so that the 8 items of a
// wave read from 8 different bank groups.
//   forward record : {w00, w01, w10, w11 (bilinear x attention weight, 0 outside), o00, o01, o10, o11}
//   backward record: {lh, lw, w, meta (bit 0-3 corner validity, 4.. level, -1: sample outside),
//                     o00, o01, o10, o11 -> overwritten by the four corner dot products}
//   o = BYTE offset of the corner's 128-byte line of this item's head from `value` (32-bit: the host
//   checks B*Nv*H*32*4 < 2^32); an invalid corner points at a valid line and carries weight / mask 0.
// ---------------------------------------------------------------------------------------------
constexpr int kRecF = 8;                    // floats per sample record
constexpr int kGLv = 16;                    // levels supported by the gather kernels' level table

__device__ __forceinline__ int rec_at(int it, int lp, int LP) { return (it * (LP + 1) + lp) * kRecF; }
inline size_t rec_lds_bytes(int LP) { return sizeof(float) * (size_t)kItems * (LP + 1) * kRecF; }

struct GLevels { int Hl[kGLv], Wl[kGLv], start[kGLv]; };

__device__ __forceinline__ void load_levels(GLevels& t, const int64_t* __restrict__ shapes,
                                            const int64_t* __restrict__ lsi, int L) {
  if ((int)threadIdx.x < L) {
    t.Hl[threadIdx.x] = (int)shapes[2 * threadIdx.x];
    t.Wl[threadIdx.x] = (int)shapes[2 * threadIdx.x + 1];
    t.start[threadIdx.x] = (int)lsi[threadIdx.x];
  }
}

// stage (x, y, w) of the workgroup's items into fields 0..2 of their records
template <int kThr>
__device__ __forceinline__ void stage_records(const Prep& pr, const float* __restrict__ loc,
                                              const float* __restrict__ attw, const GLevels& lv, float* rec,
                                              const ItemMap& im, int H, int Nq, int L, int P) {
  const int LP = L * P;
  const int nvalid = im.nvalid;
  if (pr.off_raw == nullptr) {
    for (int i = threadIdx.x; i < nvalid * LP; i += kThr) {
      const int it = i / LP, lp = i - it * LP;
      const int64_t e = im.at(it) * LP + lp;
      const float2 xy = reinterpret_cast<const float2*>(loc)[e];
      float* r = rec + rec_at(it, lp, LP);
      r[0] = xy.x; r[1] = xy.y; r[2] = attw[e];
    }
    __syncthreads();
    return;
  }
  // merged queue entries: an item carries Qm = Qn groups of LPg = L*P samples (LP == Qn * L * P there); group g of item
  // (b, q, h) is queue entry g: batch row b*Qn + g of ref / value / the saved operands, and the raw layout
  // [bs, Nq, H, Qn, L*P] makes the item's Qn * L*P raw values contiguous
  const int Qm = pr.merge ? pr.Qn : 1, LPg = LP / Qm;
  for (int i = threadIdx.x; i < nvalid * LP; i += kThr) {
    const int it = i / LP, lp = i - it * LP;
    const int64_t item = im.at(it);
    const int g = lp / LPg, lpg = lp - g * LPg;
    const int64_t raw = pr.merge ? item * LP + lp : raw_index(item, lp, H, Nq, LP, pr.Qn);
    const int l = lpg / P, p = lpg - l * P;
    const int rr = pr.mode == 0 ? l : p % pr.R;
    const float2 o = reinterpret_cast<const float2*>(pr.off_raw)[raw];
    const int64_t bq = item / H;                     // merge: (b, q) of the bs rows -> row b*Qn + g of ref
    const int64_t refrow = pr.merge ? ((bq / Nq) * Qm + g) * Nq + bq % Nq : bq;
    const float2 rf = reinterpret_cast<const float2*>(pr.ref)[refrow * pr.R + rr];
    float* r = rec + rec_at(it, lp, LP);
    r[0] = rf.x + o.x / (float)lv.Wl[l];
    r[1] = rf.y + o.y / (float)lv.Hl[l];
    r[2] = pr.logit_raw[raw];
  }
  __syncthreads();
  // softmax over the LPg logits of every (item, group): 8 lanes per unit, values in registers (LPg <= 64)
  const int nlan = 8;
  for (int u = threadIdx.x / nlan; u < nvalid * Qm; u += kThr / nlan) {
    const int it = u / Qm, lp0 = (u - it * Qm) * LPg;
    const int sub = threadIdx.x % nlan;
    float e[kMaxLP / 8];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < kMaxLP / 8; ++k) {
      const int lp = sub + k * nlan;
      e[k] = lp < LPg ? rec[rec_at(it, lp0 + lp, LP) + 2] : -INFINITY;
      m = fmaxf(m, e[k]);
    }
#pragma unroll
    for (int d = 1; d < nlan; d <<= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < kMaxLP / 8; ++k) {
      const int lp = sub + k * nlan;
      e[k] = lp < LPg ? expf(e[k] - m) : 0.f;
      sum += e[k];
    }
#pragma unroll
    for (int d = 1; d < nlan; d <<= 1) sum += __shfl_xor(sum, d, 64);
#pragma unroll
    for (int k = 0; k < kMaxLP / 8; ++k) {
      const int lp = sub + k * nlan;
      if (lp < LPg) rec[rec_at(it, lp0 + lp, LP) + 2] = e[k] / sum;
    }
  }
  __syncthreads();
  if (pr.loc_out != nullptr) {
    for (int i = threadIdx.x; i < nvalid * LP; i += kThr) {
      const int it = i / LP, lp = i - it * LP;
      const float* r = rec + rec_at(it, lp, LP);
      int64_t e = im.at(it) * LP + lp;
      if (pr.merge) {                                  // saved operands stay [bs*Qn, Nq, H, L*P]
        const int64_t item = im.at(it), bq = item / H;
        const int g = lp / LPg;
        e = ((((bq / Nq) * Qm + g) * Nq + bq % Nq) * H + item % H) * LPg + (lp - g * LPg);
      }
      reinterpret_cast<float2*>(pr.loc_out)[e] = make_float2(r[0], r[1]);
      pr.w_out[e] = r[2];
    }
  }
}

// corner arithmetic of one sample, once: (x, y, w) in the record -> the forward or backward record
template <int kThr, bool BWD>
__device__ __forceinline__ void corner_records(const GLevels& lv, float* rec, const ItemMap& im, int Nv,
                                               int H, int Nq, int L, int P, int Qm = 1) {
  // Qm > 1 (merged queue entries, forward): LP = Qm * L * P samples per item, group g reads value plane b*Qm + g
  const int LP = Qm * L * P, LPg = L * P;
  const int nvalid = im.nvalid;
  for (int i = threadIdx.x; i < nvalid * LP; i += kThr) {
    const int it = i / LP, lp = i - it * LP;
    const int g = lp / LPg;
    const int l = (lp - g * LPg) / P;
    const int64_t item = im.at(it);
    const int h = (int)(item % H);
    const int b = (int)(item / H / Nq) * Qm + g;
    const int Hl = lv.Hl[l], Wl = lv.Wl[l];
    float* r = rec + rec_at(it, lp, LP);
    const float x = pix(r[0], Wl), y = pix(r[1], Hl), w = r[2];
    const bool in = y > -1.f && x > -1.f && y < Hl && x < Wl;
    const int h0 = (int)floorf(y), w0 = (int)floorf(x);
    const float lh = y - h0, lw = x - w0;
    const bool t = h0 >= 0, bo = h0 + 1 <= Hl - 1, le = w0 >= 0, ri = w0 + 1 <= Wl - 1;
    const int r0 = min(max(h0, 0), Hl - 1) * Wl, r1 = min(max(h0 + 1, 0), Hl - 1) * Wl;
    const int c0 = min(max(w0, 0), Wl - 1), c1 = min(max(w0 + 1, 0), Wl - 1);
    const unsigned line0 = ((unsigned)b * Nv + lv.start[l]) * H + h;          // line index of pixel 0 (128 B per line)
    const unsigned o00 = (line0 + (unsigned)(r0 + c0) * H) * (kCh * 4), o01 = (line0 + (unsigned)(r0 + c1) * H) * (kCh * 4);
    const unsigned o10 = (line0 + (unsigned)(r1 + c0) * H) * (kCh * 4), o11 = (line0 + (unsigned)(r1 + c1) * H) * (kCh * 4);
    if (!BWD) {
      // a sample outside the level (or with a NaN location) contributes nothing: selects, not products with 0
      const float hh = 1.f - lh, hw = 1.f - lw;
      reinterpret_cast<float4*>(r)[0] = make_float4((in && t && le) ? hh * hw * w : 0.f, (in && t && ri) ? hh * lw * w : 0.f,
                                                    (in && bo && le) ? lh * hw * w : 0.f, (in && bo && ri) ? lh * lw * w : 0.f);
    } else {
      const int meta = in ? ((int)(t && le) | ((int)(t && ri) << 1) | ((int)(bo && le) << 2) | ((int)(bo && ri) << 3) | (l << 4)) : -1;
      reinterpret_cast<float4*>(r)[0] = make_float4(lh, lw, w, __int_as_float(meta));
    }
    reinterpret_cast<uint4*>(r)[1] = make_uint4(o00, o01, o10, o11);
  }
  __syncthreads();
}

__device__ __forceinline__ float4 ldv(const float* __restrict__ value, unsigned byte_off) {
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(value) + byte_off);
}

__global__ __launch_bounds__(kThreads) void msda_fwd_kernel(
    const float* __restrict__ value, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lsi, const float* __restrict__ loc, const float* __restrict__ attw,
    float* __restrict__ out, int Nv, int H, int Nq, int L, int P, int64_t n_items, int nblocks, int head_major,
    Prep pr) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ GLevels lv;
  const int Qm = pr.merge ? pr.Qn : 1;
  const int LP = Qm * L * P;                  // samples per item (merged queue entries: Qn groups of L * P)
  ItemMap im;
  if (!item_map(im, blockIdx.x, nblocks, n_items, H, head_major)) return;
  load_levels(lv, shapes, lsi, L);
  __syncthreads();
  const int nvalid = im.nvalid;
  stage_records<kThreads>(pr, loc, attw, lv, smem, im, H, Nq, Qm * L, P);
  corner_records<kThreads, false>(lv, smem, im, Nv, H, Nq, L, P, Qm);
  const int it = threadIdx.x / kLanes, sub = threadIdx.x % kLanes;
  if (it >= nvalid) return;
  const unsigned sub16 = sub * 16;
  const float* r = smem + rec_at(it, 0, LP);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 2        // 8 corner lines in flight per wave: 4 measured 12 % slower on random points (the L1 thrashes), 8 slower still
  for (int lp = 0; lp < LP; ++lp, r += kRecF) {
    const float4 w = reinterpret_cast<const float4*>(r)[0];
    const uint4 o = reinterpret_cast<const uint4*>(r)[1];
    const float4 v00 = ldv(value, o.x + sub16), v01 = ldv(value, o.y + sub16), v10 = ldv(value, o.z + sub16),
                 v11 = ldv(value, o.w + sub16);
    acc.x += w.x * v00.x + w.y * v01.x + w.z * v10.x + w.w * v11.x;
    acc.y += w.x * v00.y + w.y * v01.y + w.z * v10.y + w.w * v11.y;
    acc.z += w.x * v00.z + w.y * v01.z + w.z * v10.z + w.w * v11.z;
    acc.w += w.x * v00.w + w.y * v01.w + w.z * v10.w + w.w * v11.w;
  }
  if (Qm > 1) { const float s = 1.f / Qm; acc.x *= s; acc.y *= s; acc.z *= s; acc.w *= s; }
  *reinterpret_cast<float4*>(out + im.at(it) * kCh + sub * 4) = acc;
}

// store phase of the backward kernels: s_loc / s_w hold grad_loc / grad_w of `nvalid` items
template <int kThr>
__device__ __forceinline__ void store_grads(const Prep& pr, const float* __restrict__ attw,
                                            const int64_t* __restrict__ shapes, float* __restrict__ grad_loc,
                                            float* __restrict__ grad_w, float* s_loc, float* s_w, float* s_dot,
                                            int64_t item0, int nvalid, int H, int Nq, int L, int P) {
  const int LP = L * P;
  if (pr.g_off_raw == nullptr) {
    for (int i = threadIdx.x; i < nvalid * LP * 2; i += kThr) grad_loc[item0 * LP * 2 + i] = s_loc[i];
    for (int i = threadIdx.x; i < nvalid * LP; i += kThr) grad_w[item0 * LP + i] = s_w[i];
    return;
  }
  // softmax backward: g_logit = w * (g_w - sum_j w_j g_w_j);  d loc / d off = 1 / (W_l, H_l)
  for (int it = threadIdx.x; it < nvalid; it += kThr) {
    float dot = 0.f;
    for (int lp = 0; lp < LP; ++lp) dot += attw[(item0 + it) * LP + lp] * s_w[it * LP + lp];
    s_dot[it] = dot;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nvalid * LP; i += kThr) {
    const int it = i / LP, lp = i - it * LP;
    const int64_t raw = raw_index(item0 + it, lp, H, Nq, LP, pr.Qn);
    const int l = lp / P;
    pr.g_logit_raw[raw] = attw[item0 * LP + i] * (s_w[i] - s_dot[it]);
    reinterpret_cast<float2*>(pr.g_off_raw)[raw] =
        make_float2(s_loc[2 * i] / (float)shapes[2 * l + 1], s_loc[2 * i + 1] / (float)shapes[2 * l]);
  }
}

// ---------------------------------------------------------------------------------------------
// backward.  Measured on MI355X (tools/micro/atomic_bench.hip): fp32 global atomics retire at
// ~10 G (instruction x 128-byte line) requests per second no matter how many dwords of the line an
// instruction touches.  So the scatter uses ONE instruction per corner line: 32 adjacent lanes own
// the 32 channels of a head (two items per wave).  grad_loc / grad_w reduce over those 32 lanes
// with xor shuffles and leave through LDS as coalesced stores.
// ---------------------------------------------------------------------------------------------
constexpr int kBLanes = 32;
constexpr int kBItems = kThreads / kBLanes;   // 8 items per workgroup

__device__ __forceinline__ float half_wave_sum(float v) {
#pragma unroll
  for (int m = 1; m < kBLanes; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ float ld1(const float* __restrict__ p, int64_t o) {
  return o >= 0 ? p[o] : 0.f;
}

__global__ __launch_bounds__(kThreads) void msda_bwd_kernel(
    const float* __restrict__ value, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lsi, const float* __restrict__ loc, const float* __restrict__ attw,
    const float* __restrict__ grad_out, float* __restrict__ grad_value,
    float* __restrict__ grad_loc, float* __restrict__ grad_w, int Nv, int H, int Nq, int L, int P,
    int64_t n_items, int nblocks, Prep pr, GoMap gm) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int LP = L * P;
  float* s_loc = smem;                         // [kBItems][LP*2]  in: loc, out: grad_loc
  float* s_w = smem + kBItems * LP * 2;        // [kBItems][LP]    in: w,   out: grad_w
  float* s_dot = s_w + kBItems * LP;           // [kBItems]
  const int blk = xcd_remap(blockIdx.x, nblocks);
  if (blk >= nblocks) return;
  const int64_t item0 = (int64_t)blk * kBItems;
  const int nvalid = (int)min((int64_t)kBItems, n_items - item0);
  for (int i = threadIdx.x; i < nvalid * LP * 2; i += kThreads) s_loc[i] = loc[item0 * LP * 2 + i];
  for (int i = threadIdx.x; i < nvalid * LP; i += kThreads) s_w[i] = attw[item0 * LP + i];
  __syncthreads();
  const int it = threadIdx.x / kBLanes, ch = threadIdx.x % kBLanes;
  if (it < nvalid) {
    const int64_t item = item0 + it;
    const int h = (int)(item % H);
    const int64_t bq = item / H;
    const int b = (int)(bq / Nq);
    const int row_stride = H * kCh;
    const int64_t voff = (int64_t)b * Nv * row_stride + h * kCh + ch;
    const float* vb = value + voff;
    float* gvb = grad_value + voff;
    const float go = grad_out[gm.at(item) * kCh + ch] * gm.scale;
    float* ml = s_loc + it * LP * 2;
    float* mw = s_w + it * LP;
    for (int l = 0; l < L; ++l) {
      const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
      const int64_t base = lsi[l] * row_stride;
      for (int p = 0; p < P; ++p) {
        const float x = pix(ml[(l * P + p) * 2], Wl);
        const float y = pix(ml[(l * P + p) * 2 + 1], Hl);
        const float w = mw[l * P + p];
        float gx = 0.f, gy = 0.f, gw = 0.f;
        if (y > -1.f && x > -1.f && y < Hl && x < Wl) {     // uniform over the item's 32 lanes
          const Corner c = corners(x, y, Hl, Wl, base, row_stride);
          const float d00 = ld1(vb, c.o00) * go, d01 = ld1(vb, c.o01) * go,
                      d10 = ld1(vb, c.o10) * go, d11 = ld1(vb, c.o11) * go;
          const float hh = 1.f - c.lh, hw = 1.f - c.lw;
          gw = c.w00 * d00 + c.w01 * d01 + c.w10 * d10 + c.w11 * d11;
          gx = w * Wl * (-hh * d00 + hh * d01 - c.lh * d10 + c.lh * d11);
          gy = w * Hl * (-hw * d00 - c.lw * d01 + hw * d10 + c.lw * d11);
          const float wg = w * go;
          if (c.o00 >= 0) unsafeAtomicAdd(gvb + c.o00, c.w00 * wg);
          if (c.o01 >= 0) unsafeAtomicAdd(gvb + c.o01, c.w01 * wg);
          if (c.o10 >= 0) unsafeAtomicAdd(gvb + c.o10, c.w10 * wg);
          if (c.o11 >= 0) unsafeAtomicAdd(gvb + c.o11, c.w11 * wg);
        }
        gx = half_wave_sum(gx); gy = half_wave_sum(gy); gw = half_wave_sum(gw);
        // all 32 lanes consumed (x, y, w) of this point before the shuffles finished
        if (ch == 0) {
          ml[(l * P + p) * 2] = gx;
          ml[(l * P + p) * 2 + 1] = gy;
          mw[l * P + p] = gw;
        }
      }
    }
  }
  __syncthreads();
  store_grads<kThreads>(pr, attw, shapes, grad_loc, grad_w, s_loc, s_w, s_dot, item0, nvalid, H, Nq, L, P);
}

// ---------------------------------------------------------------------------------------------
// backward, destination-binned variant (the default for launches big enough to amortise 5 launches).
// Measured on MI355X (tools/micro/{atomic_scope_bench,scatter_bench}.hip, profiles/r02_micro.log):
//   * global fp32 atomics retire at ~9-10 G (instruction x 128-byte line) requests/s no matter the
//     scope or whether the target is an XCD-private copy  -> the scatter above is on that wall;
//   * LDS ds_add_f32 is no faster (6 G line-adds/s), scattered 4-byte stores run at 85 G/s and random
//     128-byte line gathers at 56 G lines/s (7 TB/s out of L2/MALL).
// So the scatter becomes: counting-sort the (b,q,head,level,point) samples by DESTINATION tile
// (batch, level, 8x8 block of top-left corner pixels, head), then one wave per <=1024-sample chunk of
// a tile accumulates the 9x9-pixel x 32-channel window of that tile in its PRIVATE 10 KB LDS window
// with plain (non-atomic) read-modify-writes -- all 64 lanes work on ONE sample: lanes 0-31 own the 32
// channels of the left corner column, lanes 32-63 the right column, top row then bottom row, so the
// 64 addresses of a step are always distinct and a wave's LDS operations execute in order -- and
// flushes the non-zero window lines with one atomic per line: ~2-3 M requests instead of 61 M.
// grad_loc / grad_w come from a gather kernel shaped like the forward (no atomics at all).
// ---------------------------------------------------------------------------------------------
#ifndef VIDAR_MSDA_TILE_SHIFT
#define VIDAR_MSDA_TILE_SHIFT 3
#endif
constexpr int kTileShift = VIDAR_MSDA_TILE_SHIFT;
constexpr int kTile = 1 << kTileShift;         // tile edge (top-left corner pixels)
constexpr int kWin = kTile + 1;                // window edge (corner pixels)
constexpr int kWinLines = kWin * kWin;         // 81 lines of 32 floats
static_assert(kWinLines < 128, "the window line index shares a word with a 128-byte aligned offset");
#ifndef VIDAR_MSDA_CHUNK
#define VIDAR_MSDA_CHUNK 1024
#endif
constexpr int kChunk = VIDAR_MSDA_CHUNK;       // samples per chunk descriptor (tuning sweep: tools/tune_msda_tile.sh)
constexpr int kMaxL = 16;                      // levels supported by the binned path
constexpr int kTWaves = 4;                     // waves (= chunks, private 10 KB windows) per workgroup of the accumulate kernel
#ifndef VIDAR_MSDA_TILE_GROUP
#define VIDAR_MSDA_TILE_GROUP 8
#endif
constexpr int kGrp = VIDAR_MSDA_TILE_GROUP;    // samples per software-pipeline group of the accumulate kernel (grad_out lines in flight: 2 kGrp)
static_assert(64 % kGrp == 0, "groups tile a 64-sample batch");
// (16-byte sort records {x, y, attention weight, sample} that the accumulate kernel would read coalesced instead of
//  gathering loc[s] / attw[s]: 1.24 vs 1.19 ms for the SCA backward -- the fill pass's scattered stores grow 4x; removed)

struct LevelTab {
  int Hl[kMaxL], Wl[kMaxL], ntx[kMaxL], toff[kMaxL];
  int T;                                       // tiles per batch element
};

__device__ __forceinline__ void build_tab(LevelTab& t, const int64_t* __restrict__ shapes, int L) {
  if (threadIdx.x == 0) {
    int off = 0;
    for (int l = 0; l < L; ++l) {
      const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
      t.Hl[l] = Hl; t.Wl[l] = Wl; t.ntx[l] = (Wl >> kTileShift) + 1; t.toff[l] = off;
      off += t.ntx[l] * ((Hl >> kTileShift) + 1);       // px = floor(x)+1 in [0, Wl], py likewise
    }
    t.T = off;
  }
  __syncthreads();
}

// tile (within its level) of a sample, or -1 when the sample falls outside the level -- exactly the
// test of the forward
__device__ __forceinline__ int sample_tile(const LevelTab& t, const float* __restrict__ loc, int64_t s, int l) {
  const float2 xy = reinterpret_cast<const float2*>(loc)[s];
  const int Hl = t.Hl[l], Wl = t.Wl[l];
  const float x = pix(xy.x, Wl), y = pix(xy.y, Hl);
  if (!(y > -1.f && x > -1.f && y < Hl && x < Wl)) return -1;
  const int px = (int)floorf(x) + 1, py = (int)floorf(y) + 1;
  return (py >> kTileShift) * t.ntx[l] + (px >> kTileShift);
}

// Counting sort, passes 1 and 3.  Global int atomics on the tile counters would put the sort on the
// same atomic wall as the scatter it replaces (measured: 3.9 + 6.8 ms for the 15 M samples of SCA, hot
// tiles serialise).  So a workgroup owns kBinQ consecutive queries of ONE (batch, head, level) plane --
// its samples can only fall into that level's tiles -- histograms them in LDS and touches the global
// counters once per (workgroup, touched tile).
// queries per workgroup of the binning passes: ~4 096 samples (one register-resident batch of kThreads x kBinSamples) --
// 128 queries for SpatialCrossAttention (L * P = 32), 512 for the one-level shapes (measured: 128 / 256 / 512 queries give
// 1.278 / 1.287 / 1.304 ms on the SCA backward and 0.382 / 0.349 / 0.342 ms on TemporalSelfAttention's)
inline int bin_queries(int L, int P) {
  const int q = 4096 / (L * P);
  return q < 64 ? 64 : (q > 512 ? 512 : q);
}
constexpr int kBinSamples = 16;                // samples per thread held in registers (fill pass)
constexpr int kMaxTilesLds = 16000;            // LDS histogram capacity: 2 x 4 B x 16000 + the level table < 160 KB

template <bool FILL>
__global__ __launch_bounds__(kThreads) void msda_bin_kernel(
    const int64_t* __restrict__ shapes, const float* __restrict__ loc, int* __restrict__ counts,
    int* __restrict__ rec, int H, int Nq, int L, int P, int kBinQ) {
  extern __shared__ int s_hist[];              // [ntl] counts; (FILL) then the per-tile record cursors
  __shared__ LevelTab t;
  build_tab(t, shapes, L);
  // a workgroup owns kBinQ queries of one (batch element, head) with ALL their levels: the L * P locations of a
  // (query, head) are 256 contiguous bytes, read once as whole lines (one workgroup per level fetched every line twice:
  // FETCH_SIZE 189 MB per pass for the 94 MB of locations)
  const int h = blockIdx.y % H, b = blockIdx.y / H;
  const int ntl = t.T;                         // tiles of all levels
  const int q0 = blockIdx.x * kBinQ, nq = min(kBinQ, Nq - q0);
  const int LP = L * P;
  const int n = nq * LP;                       // samples of this workgroup
  for (int i = threadIdx.x; i < ntl; i += kThreads) s_hist[i] = 0;
  __syncthreads();
  // bins are (batch element, head, level, tile): all tiles of one (batch element, head) GROUP are consecutive, so the
  // chunk table lists a group's chunks together and the accumulate kernel can hand whole groups to one XCD
  const int64_t gbin0 = ((int64_t)b * H + h) * t.T;                // bin = gbin0 + toff[level] + tile
  int tile_of[kBinSamples];
  for (int i0 = 0; i0 < n; i0 += kThreads * kBinSamples) {
#pragma unroll
    for (int u = 0; u < kBinSamples; ++u) {
      const int i = i0 + u * kThreads + threadIdx.x;
      int tl = -1;
      if (i < n) {
        const int ql = i / LP, lp = i - ql * LP, l = lp / P;
        tl = sample_tile(t, loc, (((int64_t)b * Nq + q0 + ql) * H + h) * LP + lp, l);
        if (tl >= 0) { tl += t.toff[l]; atomicAdd(s_hist + tl, 1); }
      }
      tile_of[u] = tl;
    }
    if (!FILL) continue;
    __syncthreads();
    // reserve a run of record slots per touched tile (the cursor was initialised with the bin starts)
    for (int i = threadIdx.x; i < ntl; i += kThreads) {
      const int c = s_hist[i];
      if (c) s_hist[i] = atomicAdd(counts + gbin0 + i, c);    // the LDS counter becomes the cursor
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kBinSamples; ++u) {
      const int i = i0 + u * kThreads + threadIdx.x;
      const int tl = tile_of[u];
      if (tl >= 0) {
        const int ql = i / LP, lp = i - ql * LP;
        rec[atomicAdd(s_hist + tl, 1)] = (int)((((int64_t)b * Nq + q0 + ql) * H + h) * LP + lp);
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ntl; i += kThreads) s_hist[i] = 0;   // next batch of this workgroup
    __syncthreads();
  }
  if (FILL) return;
  __syncthreads();
  for (int i = threadIdx.x; i < ntl; i += kThreads) {
    const int c = s_hist[i];
    if (c) atomicAdd(counts + gbin0 + i, c);
  }
}

// counts -> exclusive prefix (`cursor`: the fill pass's per-bin record cursor) and the chunk table: two int4 per chunk,
// {first record, number of records, level, batch element} and {head, tile row, tile column, 0}, so the accumulate
// kernel needs no level table of its own.  One workgroup per slab of 8192 bins (rounds 2-3: ONE workgroup walked all
// slabs, 0.05 ms on a single CU per backward); a workgroup first sums the counts of every earlier slab itself -- a few
// hundred KB out of L2, redundant but without a second launch or a hand-off between workgroups.
constexpr int kScanThreads = 1024;
constexpr int kScanPer = 8;                            // consecutive bins per thread
constexpr int kScanSlab = kScanThreads * kScanPer;
__global__ __launch_bounds__(kScanThreads) void msda_bin_scan_kernel(
    const int64_t* __restrict__ shapes, const int* __restrict__ counts, int* __restrict__ cursor,
    int4* __restrict__ desc, int* __restrict__ n_chunks, int* __restrict__ group_start, int B, int H, int L) {
  __shared__ LevelTab t;
  __shared__ int s_wsum[kScanThreads / 64], s_wchk[kScanThreads / 64];
  build_tab(t, shapes, L);
  const int nbins = B * t.T * H;
  const int base = (int)blockIdx.x * kScanSlab;
  if (base >= nbins) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int kPer = kScanPer;
  // records / chunks before this slab
  int ps = 0, pk = 0;
  for (int i = threadIdx.x; i < base; i += kScanThreads) {
    const int c0 = counts[i];
    ps += c0; pk += (c0 + kChunk - 1) / kChunk;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { ps += __shfl_xor(ps, d, 64); pk += __shfl_xor(pk, d, 64); }
  if (lane == 0) { s_wsum[wave] = ps; s_wchk[wave] = pk; }
  __syncthreads();
  int carry_s = 0, carry_k = 0;
#pragma unroll
  for (int w = 0; w < kScanThreads / 64; ++w) { carry_s += s_wsum[w]; carry_k += s_wchk[w]; }
  __syncthreads();
  int c[kPer];
#pragma unroll
  for (int j = 0; j < kPer; ++j) { const int bin = base + threadIdx.x * kPer + j; c[j] = bin < nbins ? counts[bin] : 0; }
  int ts = 0, tk = 0;
#pragma unroll
  for (int j = 0; j < kPer; ++j) { ts += c[j]; tk += (c[j] + kChunk - 1) / kChunk; }
  int xs = ts, xk = tk;                                // inclusive wave scans of the per-thread totals
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int us = __shfl_up(xs, d, 64), uk = __shfl_up(xk, d, 64);
    if (lane >= d) { xs += us; xk += uk; }
  }
  if (lane == 63) { s_wsum[wave] = xs; s_wchk[wave] = xk; }
  __syncthreads();
  int ws = 0, wk = 0, tot_k = 0;
#pragma unroll
  for (int w = 0; w < kScanThreads / 64; ++w) {
    const int a = s_wsum[w], q = s_wchk[w];
    tot_k += q;
    if (w < wave) { ws += a; wk += q; }
  }
  int s = carry_s + ws + xs - ts, kk = carry_k + wk + xk - tk;
  // (tile, level, head, batch element) of this thread's first bin by division ONCE, then counted up
  int bin = base + threadIdx.x * kPer;
  int tl = bin % t.T, bh = bin / t.T, h = bh % H, b = bh / H;
  int l = 0;
  while (l + 1 < L && t.toff[l + 1] <= tl) ++l;
  int ty = (tl - t.toff[l]) / t.ntx[l], tx = (tl - t.toff[l]) - ty * t.ntx[l];
#pragma unroll
  for (int j = 0; j < kPer; ++j, ++bin) {
    if (bin < nbins) {
      if (tl == 0) group_start[b * H + h] = kk;        // first chunk of the (batch element, head) group
      cursor[bin] = s;
      for (int i = 0; i < c[j]; i += kChunk) {
        desc[2 * kk] = make_int4(s + i, min(kChunk, c[j] - i), l, b);
        desc[2 * kk + 1] = make_int4(h, ty, tx, 0);
        ++kk;
      }
      s += c[j];
    }
    ++tl;                                                // next tile
    if (++tx == t.ntx[l]) { tx = 0; ++ty; }
    if (tl == t.T) {                                     // next (batch element, head) group
      tl = 0; l = 0; tx = ty = 0;
      if (++h == H) { h = 0; ++b; }
    } else if (l + 1 < L && tl == t.toff[l + 1]) { ++l; tx = ty = 0; }
  }
  if (threadIdx.x == 0 && base + kScanSlab >= nbins) {   // the last slab
    *n_chunks = carry_k + tot_k;
    group_start[B * H] = carry_k + tot_k;
  }
}

// Accumulate kernel.  One chunk (<= kChunk records of one destination tile) per wave: the wave accumulates the tile's
// 9x9-pixel x 32-channel window in its PRIVATE 10 KB LDS window with plain read-modify-writes.  All 64 lanes work on
// ONE sample -- lanes 0-31 own the 32 channels of the left corner column, lanes 32-63 the right column, top row
// and bottom row are two LDS updates -- so the 64 addresses of a step are always distinct (no atomics needed) and a
// wave's LDS operations execute in order.  What round 3 measured on the way here (profiles/r03_msda_tile_sweep_*,
// profiles/r03_pmc_msda_sca_*; SCA shape, whole backward):
//   * round 2 handed the per-sample scalars to the lanes with ~18 VALU instructions (readlane + select chains);
//     cutting that to ~6 (below) bought 1.28 -> 1.18 ms together with the cheaper scan: the kernel is not VALU bound
//     but LDS bound -- two 4-byte reads + two 4-byte writes per sample = 12 LDS cycles, SQ_WAIT_INST_LDS 101 M of
//     800 M wave cycles -- and bound by the read -> fma -> write chain of the 12 windows that fit a CU;
//   * ds_add_f32 on a window shared by the workgroup (no chain): 10 ms -- LDS float atomics retire at ~175 clocks per
//     wave instruction;
//   * the window in REGISTERS (4x4 tile, 25 accumulators per lane addressed through the VGPR index mode): the
//     compiler's extract / fma / insert sequence costs 9 VALU + 10 SALU per sample: 1.36 ms;
//   * one 8-byte read-modify-write covering the four corners (lane = corner x channel pair): 1.18 ms vs 1.15 ms;
//   * a resident grid striding over the chunk table (1.33 ms) or claiming chunks from a counter (1.30 ms) instead of
//     one wave per descriptor slot of the host's bound (1.15 ms): the hardware dispatcher balances the very unequal
//     chunks better than either, and waves that find no chunk cost almost nothing.
// Lane k of a 64-sample batch prepares sample k and parks its four corner weights in LDS as two 8-byte records (left
// column, right column); in the sample loop every lane picks its column's (top, bottom) pair with ONE ds_read_b64
// whose address does not depend on the sample, the window line and the grad_out line travel in one v_readlane, and
// the grad_out line is a buffer load (channel offset in a VGPR, line offset in an SGPR: no vector address arithmetic).
__device__ __forceinline__ float go_one(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0));
}

__global__ __launch_bounds__(64 * kTWaves) void msda_bwd_tile_kernel(
    const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi, const float* __restrict__ loc,
    const float* __restrict__ attw, const float* __restrict__ grad_out, float* __restrict__ grad_value,
    const int* __restrict__ rec, const int4* __restrict__ desc, const int* __restrict__ group_start, int n_groups, int Nv,
    int H, int L, int P, int go_bytes, GoMap gm) {
  __shared__ __attribute__((aligned(16))) float s_win[kTWaves][kWinLines * kCh];
  __shared__ float4 s_par[kTWaves][64];                // per wave and sample: {top-left, bottom-left, top-right, bottom-right}
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // XCD-affine chunk assignment (round 6).  Every sample of a (batch element, head) group reads that group's grad_out
  // lines ([b, q, h*32 .. h*32+31]: exactly one 128-byte line per query) -- ~32 times per line, from different tiles.
  // Workgroups are dealt round-robin to the 8 XCDs; with chunks assigned in table order every XCD's L2 ended up fetching
  // nearly ALL of grad_out (FETCH_SIZE 600 MB against 235 MB compulsory, profiles/r05_pmc_msda_sca_coherent_nq7680).
  // Group g belongs to XCD g % 8: the workgroups of XCD k = blockIdx % 8 walk the chunks of groups k, k + 8, ... in order.
  int chunk = -1;
  {
    int c = (int)(blockIdx.x >> 3) * kTWaves + wave;   // index among this XCD's chunks (wave-uniform)
    for (int g = (int)(blockIdx.x & 7); g < n_groups; g += 8) {
      const int g0 = group_start[g], cnt = group_start[g + 1] - g0;
      if (c < cnt) { chunk = g0 + c; break; }
      c -= cnt;
    }
  }
  if (chunk < 0) return;                               // wave-uniform
  const int4 d0 = desc[2 * chunk], d1 = desc[2 * chunk + 1];
  const int s0 = d0.x, n = d0.y, l = d0.z, b = d0.w, h = d1.x, ty = d1.y, tx = d1.z;
  const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
  float* win = s_win[wave];
  const int half = lane >> 5, ch = lane & 31;
  float* wq = win + half * kCh + ch;                   // + (window line of the top-left corner) * kCh
  const float2* par = reinterpret_cast<const float2*>(&s_par[wave][0]) + half;
  const int LP = L * P;
  const __amdgpu_buffer_rsrc_t go_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(grad_out), 0, go_bytes, 0x00020000);
  for (int i = lane; i < kWinLines * kCh / 4; i += 64) reinterpret_cast<float4*>(win)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  // The per-sample operands come through two DEPENDENT gathers (record -> sample index -> location / weight): fetched at
  // the top of a batch they leave the wave with nothing in flight for two memory latencies per 64 samples.  They are
  // software-pipelined ACROSS batches instead: the sample indices are requested two batches ahead, locations / weights
  // one batch ahead, under the current batch's window updates (round 6: SCA backward 1.206 -> 1.177 ms, coherent queries
  // 1.336 -> 1.298 ms; twice / four times as many grad_out lines in flight per wave -- VIDAR_MSDA_TILE_GROUP 16 / 32 --
  // measured 1.182 / 1.309 ms: latency is not what is left, profiles/r06_kbench_msda_tile_prefetch_ab.log).
  auto rec_at_batch = [&](int base) { return base < n ? rec[s0 + min(base + lane, n - 1)] : 0; };
  int s_cur = rec_at_batch(0);
  int s_nxt = rec_at_batch(64);
  float2 xy_cur = reinterpret_cast<const float2*>(loc)[s_cur];
  float aw_cur = attw[s_cur];
  for (int base = 0; base < n; base += 64) {
    // lane k prepares sample base+k: window line of its top-left corner, the four corner weights
    // (times the attention weight) and the offset of its grad_out line
    const bool valid = base + lane < n;
    const int s = s_cur;
    const float2 xy = xy_cur;
    const float aw = valid ? aw_cur * gm.scale : 0.f;    // (merged queue entries: grad_out / Qn rides on the weight)
    // next batch's operands (their sample indices arrived during the previous batch), and the indices after them
    s_cur = s_nxt;
    if (base + 64 < n) {
      xy_cur = reinterpret_cast<const float2*>(loc)[s_cur];
      aw_cur = attw[s_cur];
    }
    s_nxt = rec_at_batch(base + 128);
    const float x = pix(xy.x, Wl), y = pix(xy.y, Hl);
    const int h0 = (int)floorf(y), w0 = (int)floorf(x);
    const float lh = y - h0, lw = x - w0;
    const float hh = (1.f - lh) * aw, lha = lh * aw;
    __builtin_amdgcn_wave_barrier();                   // the previous batch's reads of s_par are done (in-order LDS)
    s_par[wave][lane] = make_float4(hh * (1.f - lw), lha * (1.f - lw), hh * lw, lha * lw);
    const int line = min(max(h0 + 1 - ty * kTile, 0), kTile - 1) * kWin + min(max(w0 + 1 - tx * kTile, 0), kTile - 1);
    // one word per sample for the v_readlane hand-off: byte offset of its grad_out line (a multiple of 128) | window line
    const int pack = ((int)gm.at(s / LP) * (kCh * 4)) | line;
    __builtin_amdgcn_wave_barrier();
    // groups of kGrp samples, software-pipelined by hand: the weight pairs and grad_out lines of group k+1 are
    // requested before the window updates of group k (samples past the end of the chunk carry zero weights)
    float g[kGrp], gn[kGrp];
    float2 a[kGrp], an[kGrp];
#pragma unroll
    for (int u = 0; u < kGrp; ++u) {
      g[u] = go_one(go_rsrc, ch * 4, __builtin_amdgcn_readlane(pack, u) & ~127);
      a[u] = par[2 * u];                               // (top, bottom) weight of this lane's column
    }
#pragma unroll
    for (int j0 = 0; j0 < 64; j0 += kGrp) {
      if (j0 + kGrp < 64) {
#pragma unroll
        for (int u = 0; u < kGrp; ++u) {
          const int j = (j0 + kGrp + u) & 63;
          gn[u] = go_one(go_rsrc, ch * 4, __builtin_amdgcn_readlane(pack, j) & ~127);
          an[u] = par[2 * j];
        }
      }
      __builtin_amdgcn_sched_barrier(0);               // keep the requests above ahead of the updates below
#pragma unroll
      for (int u = 0; u < kGrp; ++u) {
        float* p = wq + (__builtin_amdgcn_readlane(pack, j0 + u) & 127) * kCh;
        const float t0 = p[0], t1 = p[kWin * kCh];
        p[0] = t0 + a[u].x * g[u];
        p[kWin * kCh] = t1 + a[u].y * g[u];
      }
#pragma unroll
      for (int u = 0; u < kGrp; ++u) { g[u] = gn[u]; a[u] = an[u]; }
    }
  }
  // flush: window line i = (row r, column c) is pixel (ty*kTile + r - 1, tx*kTile + c - 1)
  float* gv = grad_value + (((int64_t)b * Nv + lsi[l]) * H + h) * kCh + ch;
  for (int i = half; i < kWinLines; i += 2) {
    const int r = i / kWin, c = i - r * kWin;
    const int py = ty * kTile + r - 1, px = tx * kTile + c - 1;
    if (py < 0 || py >= Hl || px < 0 || px >= Wl) continue;
    const float v = win[i * kCh + ch];
    if (v != 0.f) unsafeAtomicAdd(gv + ((int64_t)py * Wl + px) * H * kCh, v);
  }
}

// grad_sampling_loc / grad_attn_weight: a gather shaped like the forward.  Per sample the 8 lanes of an item
// fetch the four corner lines, dot them with the item's grad_out line and reduce the four dot products over the
// 8 lanes with DPP adds (quad_perm x2 + row_half_mirror: 3 VALU operations per value; __shfl_xor would be a
// ds_bpermute round trip per step); lane 0 parks them in the sample's record and ONE thread per sample turns
// them into (grad_x, grad_y, grad_w) afterwards.  No atomics.
__device__ __forceinline__ float reduce8(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  return v;
}

__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
  return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}

// N consecutive samples of one item: all 4N corner lines are requested before the first reduction
template <int N>
__device__ __forceinline__ void locw_samples(const float* __restrict__ value, float* r, const float4& go, int sub,
                                             unsigned sub16) {
  float4 v[N][4];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const uint4 o = reinterpret_cast<const uint4*>(r + k * kRecF)[1];
    v[k][0] = ldv(value, o.x + sub16); v[k][1] = ldv(value, o.y + sub16);
    v[k][2] = ldv(value, o.z + sub16); v[k][3] = ldv(value, o.w + sub16);
  }
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const float d00 = reduce8(dot4(v[k][0], go)), d01 = reduce8(dot4(v[k][1], go));
    const float d10 = reduce8(dot4(v[k][2], go)), d11 = reduce8(dot4(v[k][3], go));
    if (sub == 0) reinterpret_cast<float4*>(r + k * kRecF)[1] = make_float4(d00, d01, d10, d11);
  }
}

__global__ __launch_bounds__(kThreads) void msda_bwd_locw_kernel(
    const float* __restrict__ value, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lsi, const float* __restrict__ loc, const float* __restrict__ attw,
    const float* __restrict__ grad_out, float* __restrict__ grad_loc, float* __restrict__ grad_w, int Nv,
    int H, int Nq, int L, int P, int64_t n_items, int nblocks, int head_major, Prep pr, GoMap gm) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ GLevels lv;
  __shared__ float s_dot[kItems];
  const int LP = L * P;
  ItemMap im;
  if (!item_map(im, blockIdx.x, nblocks, n_items, H, head_major)) return;
  load_levels(lv, shapes, lsi, L);
  __syncthreads();
  const int nvalid = im.nvalid;
  stage_records<kThreads>(Prep{}, loc, attw, lv, smem, im, H, Nq, L, P);     // saved (prepared) operands
  corner_records<kThreads, true>(lv, smem, im, Nv, H, Nq, L, P);
  const int it = threadIdx.x / kLanes, sub = threadIdx.x % kLanes;
  if (it < nvalid) {
    const unsigned sub16 = sub * 16;
    float4 go = *reinterpret_cast<const float4*>(grad_out + gm.at(im.at(it)) * kCh + sub * 4);
    go.x *= gm.scale; go.y *= gm.scale; go.z *= gm.scale; go.w *= gm.scale;
    float* r = smem + rec_at(it, 0, LP);
    // (the DPP reductions are convergent operations, which keeps the compiler from unrolling a loop with a
    //  run-time trip count: unrolled by hand, 16 lines in flight per wave)
    int lp = 0;
    for (; lp + 4 <= LP; lp += 4, r += 4 * kRecF) locw_samples<4>(value, r, go, sub, sub16);
    for (; lp < LP; ++lp, r += kRecF) locw_samples<1>(value, r, go, sub, sub16);
  }
  __syncthreads();
  // one thread per sample: corner dot products -> gradients (fields 0..2 of the record; field 3 keeps w)
  const bool fused = pr.g_off_raw != nullptr;
  for (int i = threadIdx.x; i < nvalid * LP; i += kThreads) {
    const int itx = i / LP, lp = i - itx * LP;
    float* r = smem + rec_at(itx, lp, LP);
    const float4 a = reinterpret_cast<const float4*>(r)[0];
    const float4 d = reinterpret_cast<const float4*>(r)[1];
    const int meta = __float_as_int(a.w);
    float gx = 0.f, gy = 0.f, gw = 0.f;
    if (meta >= 0) {
      const int l = meta >> 4;
      const float lh = a.x, lw = a.y, w = a.z, hh = 1.f - lh, hw = 1.f - lw;
      const float d00 = (meta & 1) ? d.x : 0.f, d01 = (meta & 2) ? d.y : 0.f, d10 = (meta & 4) ? d.z : 0.f,
                  d11 = (meta & 8) ? d.w : 0.f;
      gw = hh * hw * d00 + hh * lw * d01 + lh * hw * d10 + lh * lw * d11;
      gx = w * lv.Wl[l] * (-hh * d00 + hh * d01 - lh * d10 + lh * d11);
      gy = w * lv.Hl[l] * (-hw * d00 - lw * d01 + hw * d10 + lw * d11);
    }
    if (!fused) {
      const int64_t e = im.at(itx) * LP + lp;
      reinterpret_cast<float2*>(grad_loc)[e] = make_float2(gx, gy);
      grad_w[e] = gw;
    } else {
      reinterpret_cast<float4*>(r)[0] = make_float4(gx, gy, gw, a.z);
    }
  }
  if (!fused) return;
  __syncthreads();
  // softmax backward: g_logit = w * (g_w - sum_j w_j g_w_j);  d loc / d off = 1 / (W_l, H_l)
  for (int itx = threadIdx.x; itx < nvalid; itx += kThreads) {
    float dot = 0.f;
    for (int lp = 0; lp < LP; ++lp) {
      const float* r = smem + rec_at(itx, lp, LP);
      dot += r[3] * r[2];
    }
    s_dot[itx] = dot;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nvalid * LP; i += kThreads) {
    const int itx = i / LP, lp = i - itx * LP;
    const float4 a = reinterpret_cast<const float4*>(smem + rec_at(itx, lp, LP))[0];
    const int64_t raw = raw_index(im.at(itx), lp, H, Nq, LP, pr.Qn);
    const int l = lp / P;
    pr.g_logit_raw[raw] = a.w * (a.z - s_dot[itx]);
    reinterpret_cast<float2*>(pr.g_off_raw)[raw] = make_float2(a.x / (float)lv.Wl[l], a.y / (float)lv.Hl[l]);
  }
}

// workspace layout of the binned backward (all int32): [counts: nbins_bound][cursor: nbins_bound][n_chunks: 4][group starts: B*H+1]
// [chunk table: 4 * max_chunks][records: n_samples].  The level shapes live on the device, so the host
// sizes the tables from a bound (bin_plan).
struct BinPlan {
  int64_t n_samples, nbins_bound, max_chunks, tiles_bound;
  size_t off_cursor, off_chunks, off_groups, off_desc, off_rec, bytes;
  int n_groups;
  int64_t xcd_chunks;          // bound on the chunks of the groups one XCD walks
  bool ok;
};
inline BinPlan bin_plan(int B, int Nv, int H, int Nq, int L, int P) {
  BinPlan p{};
  p.n_samples = (int64_t)B * Nq * H * L * P;
  // a level of h x w pixels has (h/k+1)(w/k+1) <= hw/k^2 + (h+w)/k + 1 <= hw (1+k)/k^2 + 2 tiles (h + w <= hw + 1)
  p.tiles_bound = ((int64_t)Nv * (1 + kTile)) / (kTile * kTile) + 2 * L + 1;
  p.nbins_bound = (int64_t)B * H * p.tiles_bound;
  p.max_chunks = p.n_samples / kChunk + p.nbins_bound;
  p.off_cursor = (sizeof(int) * (size_t)p.nbins_bound + 15) & ~(size_t)15;      // [counts][cursor]: 16-byte aligned tables
  p.off_chunks = 2 * p.off_cursor;
  p.off_groups = p.off_chunks + 16;                    // first chunk of every (batch element, head) group, + the total
  p.n_groups = B * H;
  p.off_desc = (p.off_groups + sizeof(int) * ((size_t)p.n_groups + 1) + 15) & ~(size_t)15;
  // a group holds Nq * L * P samples in at most tiles_bound bins: XCD k walks groups k, k + 8, ...
  p.xcd_chunks = (int64_t)((p.n_groups + 7) / 8) * ((int64_t)Nq * L * P / kChunk + p.tiles_bound);
  p.off_rec = p.off_desc + 32 * (size_t)p.max_chunks;
  p.bytes = p.off_rec + sizeof(int) * (size_t)p.n_samples;
  p.ok = L <= kMaxL && p.n_samples > 0 && p.n_samples < (1ll << 31) &&
         (int64_t)B * Nq * H * kCh * 4 < (1ll << 31) && p.nbins_bound < (1ll << 28) &&
         p.tiles_bound <= kMaxTilesLds && (int64_t)B * H * L < 65536;
  return p;
}

inline bool msda_bad(int B, int Nv, int H, int C, int Nq, int L, int P) {
  if ((int64_t)B * Nv * H * kCh * 4 >= (1ll << 32)) return true;   // 32-bit corner byte offsets into `value`
  if (L > kGLv) return true;
  return B < 0 || Nv < 0 || H <= 0 || C != kCh || Nq < 0 || L <= 0 || P <= 0 || L * P > kMaxLP;
}

int g_head_major = 1;          // item order of the gather kernels (see ItemMap); vidar_msda_set_item_order

// The gather kernels' records take 1 KiB x (L*P + 1) of dynamic LDS: past 64 KiB (L*P >= 63) a kernel must opt in.
// Returns false when the device cannot provide it (the caller answers VIDAR_ERR_BAD_ARG instead of a failed launch).
template <typename K>
inline bool allow_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return true;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
}

// grid of a gather kernel: banded -- blocks of 32 items, padded to the 8 XCDs; head-major -- blocks of 32 (b, q) pairs x H
inline void gather_grid(int64_t n_items, int H, int& nblocks, int& grid) {
  if (g_head_major) {
    nblocks = (int)((n_items / H + kItems - 1) / kItems) * H;
    grid = nblocks;
  } else {
    nblocks = (int)((n_items + kItems - 1) / kItems);
    grid = ((nblocks + 7) / 8) * 8;
  }
}

}  // namespace

extern "C" {

int vidar_msda_set_item_order(int head_major) {
  const int prev = g_head_major;
  g_head_major = head_major ? 1 : 0;
  return prev;
}

static int msda_fwd_launch(const float* value, const int64_t* spatial_shapes,
                           const int64_t* level_start_index, const float* sampling_loc,
                           const float* attn_weight, float* out, int B, int Nv, int H, int C, int Nq, int L,
                           int P, const Prep& pr, void* stream) {
  if (msda_bad(B, Nv, H, C, Nq, L, P)) return VIDAR_ERR_BAD_ARG;
  const int Qm = pr.merge ? pr.Qn : 1;         // merged queue entries: B = bs * Qn value planes, bs * Nq * H items
  if (Qm * L * P > kMaxLP || B % Qm != 0) return VIDAR_ERR_BAD_ARG;
  const int64_t n_items = (int64_t)(B / Qm) * Nq * H;
  if (n_items == 0) return 0;
  int nblocks, grid;
  gather_grid(n_items, H, nblocks, grid);
  const size_t lds = rec_lds_bytes(Qm * L * P);
  if (!allow_lds(msda_fwd_kernel, lds)) { (void)hipGetLastError(); return VIDAR_ERR_BAD_ARG; }
  hipLaunchKernelGGL(msda_fwd_kernel, dim3(grid), dim3(kThreads), lds, (hipStream_t)stream, value,
                     spatial_shapes, level_start_index, sampling_loc, attn_weight, out, Nv, H, Nq, L,
                     P, n_items, nblocks, g_head_major, pr);
  return vidar_last_error();
}

static int msda_bwd_launch(const float* value, const int64_t* spatial_shapes,
                           const int64_t* level_start_index, const float* sampling_loc,
                           const float* attn_weight, const float* grad_out, float* grad_value,
                           float* grad_sampling_loc, float* grad_attn_weight, int B, int Nv, int H, int C,
                           int Nq, int L, int P, void* workspace, size_t workspace_bytes, const Prep& pr,
                           void* stream) {
  if (msda_bad(B, Nv, H, C, Nq, L, P)) return VIDAR_ERR_BAD_ARG;
  // merged queue entries (pr.merge): grad_out is [B / Qn, Nq, H*C] and every queue entry reads its (b, q, h) line / Qn
  const GoMap gm{pr.merge ? pr.Qn : 1, Nq * H, pr.merge ? 1.f / pr.Qn : 1.f};
  hipStream_t s = (hipStream_t)stream;
  const size_t vbytes = sizeof(float) * (size_t)B * Nv * H * C;
  if (vbytes) {
    hipError_t e = hipMemsetAsync(grad_value, 0, vbytes, s);
    if (e != hipSuccess) return (int)e;
  }
  const int64_t n_items = (int64_t)B * Nq * H;
  if (n_items == 0) return 0;
  if (workspace) {
    // destination-binned scatter + gather for grad_loc / grad_w
    if (Nv == 0) return VIDAR_ERR_BAD_ARG;
    const BinPlan p = bin_plan(B, Nv, H, Nq, L, P);
    if (!p.ok || workspace_bytes < p.bytes) return VIDAR_ERR_BAD_ARG;
    char* ws = (char*)workspace;
    int* counts = (int*)ws;
    int* cursor = (int*)(ws + p.off_cursor);
    int* n_chunks = (int*)(ws + p.off_chunks);
    int* group_start = (int*)(ws + p.off_groups);
    int4* desc = (int4*)(ws + p.off_desc);
    int* rec = (int*)(ws + p.off_rec);
    hipError_t e = hipMemsetAsync(counts, 0, p.off_desc, s);
    if (e != hipSuccess) return (int)e;
    const int binq = bin_queries(L, P);
    const dim3 bgrid((Nq + binq - 1) / binq, B * H);
    const size_t blds = sizeof(int) * (size_t)p.tiles_bound;
    hipLaunchKernelGGL(msda_bin_kernel<false>, bgrid, dim3(kThreads), blds, s, spatial_shapes, sampling_loc,
                       counts, rec, H, Nq, L, P, binq);
    const int nslabs = (int)((p.nbins_bound + kScanSlab - 1) / kScanSlab);
    hipLaunchKernelGGL(msda_bin_scan_kernel, dim3(nslabs), dim3(kScanThreads), 0, s, spatial_shapes, counts, cursor, desc,
                       n_chunks, group_start, B, H, L);
    hipLaunchKernelGGL(msda_bin_kernel<true>, bgrid, dim3(kThreads), blds, s, spatial_shapes, sampling_loc,
                       cursor, rec, H, Nq, L, P, binq);
    // 8 interleaved sequences of workgroups, one per XCD, each long enough for the chunks its groups can have
    const unsigned tgrid = 8u * (unsigned)((p.xcd_chunks + kTWaves - 1) / kTWaves);
    hipLaunchKernelGGL(msda_bwd_tile_kernel, dim3(tgrid), dim3(64 * kTWaves), 0, s, spatial_shapes,
                       level_start_index, sampling_loc, attn_weight, grad_out, grad_value, rec, desc, group_start,
                       p.n_groups, Nv, H, L, P, (int)(n_items / gm.Qn * kCh * 4), gm);
    int nblocks, grid;
    gather_grid(n_items, H, nblocks, grid);
    const size_t lds = rec_lds_bytes(L * P);
    if (!allow_lds(msda_bwd_locw_kernel, lds)) { (void)hipGetLastError(); return VIDAR_ERR_BAD_ARG; }
    hipLaunchKernelGGL(msda_bwd_locw_kernel, dim3(grid), dim3(kThreads), lds, s, value, spatial_shapes,
                       level_start_index, sampling_loc, attn_weight, grad_out, grad_sampling_loc,
                       grad_attn_weight, Nv, H, Nq, L, P, n_items, nblocks, g_head_major, pr, gm);
    return vidar_last_error();
  }
  const int nblocks = (int)((n_items + kBItems - 1) / kBItems);
  const int grid = ((nblocks + 7) / 8) * 8;
  const size_t lds = sizeof(float) * (kBItems * L * P * 3 + kBItems);
  hipLaunchKernelGGL(msda_bwd_kernel, dim3(grid), dim3(kThreads), lds, s, value, spatial_shapes,
                     level_start_index, sampling_loc, attn_weight, grad_out, grad_value,
                     grad_sampling_loc, grad_attn_weight, Nv, H, Nq, L, P, n_items, nblocks, pr, gm);
  return vidar_last_error();
}

static bool prep_bad(int bs, int Qn, int R, int mode, int L, int P) {
  return bs < 0 || Qn <= 0 || R <= 0 || (mode != 0 && mode != 1) || (mode == 0 && R != L) ||
         (mode == 1 && P % R != 0);
}

// The kernels address `value` with 32-bit byte offsets, so ONE launch covers at most 4 GiB of it.  A call with a larger
// `value` tensor (many frames x batch x cameras at once) is split over batch elements: each launch gets its own base
// pointers, the results are the same.  -> batch elements per launch, 0 when a single element does not fit.
static int batch_per_launch(int B, int Nv, int H) {
  const int64_t per = (int64_t)Nv * H * kCh * 4;
  if (per <= 0) return B > 0 ? B : 1;
  const int64_t fit = ((1ll << 32) - 1) / per;
  return (int)(fit < B ? fit : (B > 0 ? B : 1));
}

int vidar_msda_fwd_f32(const float* value, const int64_t* spatial_shapes,
                       const int64_t* level_start_index, const float* sampling_loc,
                       const float* attn_weight, float* out, int B, int Nv, int H, int C, int Nq,
                       int L, int P, void* stream) {
  VIDAR_ENTER();
  const int cb = batch_per_launch(B, Nv, H);
  if (cb <= 0) return VIDAR_ERR_BAD_ARG;
  const int64_t vs = (int64_t)Nv * H * kCh, is = (int64_t)Nq * H, LP = (int64_t)L * P;
  int b0 = 0;
  do {
    const int nb = B - b0 < cb ? B - b0 : cb;
    const int rc = msda_fwd_launch(value + b0 * vs, spatial_shapes, level_start_index, sampling_loc + b0 * is * LP * 2,
                                   attn_weight + b0 * is * LP, out + b0 * is * kCh, nb, Nv, H, C, Nq, L, P, Prep{}, stream);
    if (rc != 0) return rc;
    b0 += nb;
  } while (b0 < B);
  return 0;
}

size_t vidar_msda_bwd_workspace_bytes(int B, int Nv, int H, int Nq, int L, int P) {
  if (B <= 0 || Nv <= 0 || H <= 0 || Nq <= 0 || L <= 0 || P <= 0) return 0;
  const BinPlan p = bin_plan(B, Nv, H, Nq, L, P);
  return p.ok ? p.bytes : 0;
}

int vidar_msda_bwd_f32(const float* value, const int64_t* spatial_shapes,
                       const int64_t* level_start_index, const float* sampling_loc,
                       const float* attn_weight, const float* grad_out, float* grad_value,
                       float* grad_sampling_loc, float* grad_attn_weight, int B, int Nv, int H, int C,
                       int Nq, int L, int P, void* workspace, size_t workspace_bytes, void* stream) {
  VIDAR_ENTER();
  const int cb = batch_per_launch(B, Nv, H);
  if (cb <= 0) return VIDAR_ERR_BAD_ARG;
  const int64_t vs = (int64_t)Nv * H * kCh, is = (int64_t)Nq * H, LP = (int64_t)L * P;
  int b0 = 0;
  do {                                         // (the workspace is reused: the launches of one stream run in order)
    const int nb = B - b0 < cb ? B - b0 : cb;
    const int rc = msda_bwd_launch(value + b0 * vs, spatial_shapes, level_start_index, sampling_loc + b0 * is * LP * 2,
                                   attn_weight + b0 * is * LP, grad_out + b0 * is * kCh, grad_value + b0 * vs,
                                   grad_sampling_loc + b0 * is * LP * 2, grad_attn_weight + b0 * is * LP, nb, Nv, H, C, Nq,
                                   L, P, workspace, workspace_bytes, Prep{}, stream);
    if (rc != 0) return rc;
    b0 += nb;
  } while (b0 < B);
  return 0;
}

int vidar_msda_fused_fwd_f32(const float* value, const int64_t* spatial_shapes,
                             const int64_t* level_start_index, const float* off_raw, const float* logit_raw,
                             const float* ref, float* loc_out, float* w_out, float* out, int bs, int Qn, int Nv,
                             int H, int C, int Nq, int L, int P, int R, int mode, int merge_queue, void* stream) {
  VIDAR_ENTER();
  if (prep_bad(bs, Qn, R, mode, L, P) || (merge_queue != 0 && merge_queue != 1)) return VIDAR_ERR_BAD_ARG;
  if ((int64_t)bs * Qn * Nq * H == 0) return msda_bad(bs * Qn, Nv, H, C, Nq, L, P) ? VIDAR_ERR_BAD_ARG : 0;
  if (!off_raw || !logit_raw || !ref || !loc_out || !w_out) return VIDAR_ERR_BAD_ARG;   // (empty tensors are NULL)
  const int cb = batch_per_launch(bs * Qn, Nv, H) / Qn;            // whole batch items (Qn queue entries each) per launch
  if (cb <= 0) return VIDAR_ERR_BAD_ARG;
  const int64_t vs = (int64_t)Nv * H * kCh, is = (int64_t)Nq * H, LP = (int64_t)L * P;
  int b0 = 0;
  do {
    const int nb = bs - b0 < cb ? bs - b0 : cb;
    const int64_t q0 = (int64_t)b0 * Qn;                             // first (batch, queue) row of this launch
    Prep pr{};
    pr.off_raw = off_raw + q0 * is * LP * 2; pr.logit_raw = logit_raw + q0 * is * LP; pr.ref = ref + q0 * Nq * R * 2;
    pr.loc_out = loc_out + q0 * is * LP * 2; pr.w_out = w_out + q0 * is * LP;
    pr.Qn = Qn; pr.R = R; pr.mode = mode; pr.merge = merge_queue;
    float* out_b = out + (merge_queue ? (int64_t)b0 : q0) * is * kCh;          // merged: one output row block per batch row
    const int rc = msda_fwd_launch(value + q0 * vs, spatial_shapes, level_start_index, nullptr, nullptr, out_b,
                                   nb * Qn, Nv, H, C, Nq, L, P, pr, stream);
    if (rc != 0) return rc;
    b0 += nb;
  } while (b0 < bs);
  return 0;
}

int vidar_msda_fused_bwd_f32(const float* value, const int64_t* spatial_shapes,
                             const int64_t* level_start_index, const float* sampling_loc,
                             const float* attn_weight, const float* grad_out, float* grad_value,
                             float* grad_off_raw, float* grad_logit_raw, int bs, int Qn, int Nv, int H, int C,
                             int Nq, int L, int P, int merge_queue, void* workspace, size_t workspace_bytes, void* stream) {
  VIDAR_ENTER();
  if (bs < 0 || Qn <= 0 || (merge_queue != 0 && merge_queue != 1)) return VIDAR_ERR_BAD_ARG;
  if ((int64_t)bs * Qn * Nq * H != 0 && (!grad_off_raw || !grad_logit_raw)) return VIDAR_ERR_BAD_ARG;
  if (bs == 0) return msda_bwd_launch(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_out,
                                      grad_value, nullptr, nullptr, 0, Nv, H, C, Nq, L, P, workspace, workspace_bytes,
                                      Prep{}, stream);
  const int cb = batch_per_launch(bs * Qn, Nv, H) / Qn;
  if (cb <= 0) return VIDAR_ERR_BAD_ARG;
  const int64_t vs = (int64_t)Nv * H * kCh, is = (int64_t)Nq * H, LP = (int64_t)L * P;
  int b0 = 0;
  do {
    const int nb = bs - b0 < cb ? bs - b0 : cb;
    const int64_t q0 = (int64_t)b0 * Qn;
    Prep pr{};
    pr.g_off_raw = grad_off_raw + q0 * is * LP * 2; pr.g_logit_raw = grad_logit_raw + q0 * is * LP; pr.Qn = Qn;
    pr.merge = merge_queue;
    const int rc = msda_bwd_launch(value + q0 * vs, spatial_shapes, level_start_index, sampling_loc + q0 * is * LP * 2,
                                   attn_weight + q0 * is * LP, grad_out + (merge_queue ? (int64_t)b0 : q0) * is * kCh,
                                   grad_value + q0 * vs, nullptr,
                                   nullptr, nb * Qn, Nv, H, C, Nq, L, P, workspace, workspace_bytes, pr, stream);
    if (rc != 0) return rc;
    b0 += nb;
  } while (b0 < bs);
  return 0;
}

}  // extern "C"
