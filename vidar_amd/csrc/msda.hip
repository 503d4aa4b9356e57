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
// (they are contiguous in HBM) and re-read as broadcasts.  Workgroups are remapped so that each
// XCD walks a contiguous band of queries: neighbouring BEV queries sample neighbouring pixels, so
// a band keeps its slice of `value` resident in that XCD's private 4 MiB L2.
// Backward: see the comment above msda_bwd_kernel (one atomic instruction per 128-byte corner line).
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

__device__ int g_no_remap = 0;   // A/B switch for tools/kbench.py (VIDAR_NO_XCD_REMAP=1)
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  // blocks are dealt round-robin to the 8 XCDs; give XCD k the k-th contiguous band
  if (g_no_remap) return bid;
  const int per = (nblocks + 7) >> 3;
  return (bid & 7) * per + (bid >> 3);
}

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

__device__ __forceinline__ float4 ld4(const float* __restrict__ p, int64_t o) {
  return o >= 0 ? *reinterpret_cast<const float4*>(p + o) : make_float4(0.f, 0.f, 0.f, 0.f);
}

__global__ __launch_bounds__(kThreads) void msda_fwd_kernel(
    const float* __restrict__ value, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lsi, const float* __restrict__ loc, const float* __restrict__ attw,
    float* __restrict__ out, int Nv, int H, int Nq, int L, int P, int64_t n_items, int nblocks) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int LP = L * P;
  float* s_loc = smem;                       // [kItems][LP*2]
  float* s_w = smem + kItems * LP * 2;       // [kItems][LP]
  const int blk = xcd_remap(blockIdx.x, nblocks);
  if (blk >= nblocks) return;
  const int64_t item0 = (int64_t)blk * kItems;
  const int nvalid = (int)min((int64_t)kItems, n_items - item0);
  // coalesced staging
  for (int i = threadIdx.x; i < nvalid * LP * 2; i += kThreads) s_loc[i] = loc[item0 * LP * 2 + i];
  for (int i = threadIdx.x; i < nvalid * LP; i += kThreads) s_w[i] = attw[item0 * LP + i];
  __syncthreads();
  const int it = threadIdx.x / kLanes, sub = threadIdx.x % kLanes;
  if (it >= nvalid) return;
  const int64_t item = item0 + it;
  const int h = (int)(item % H);
  const int64_t bq = item / H;
  const int b = (int)(bq / Nq);
  const int row_stride = H * kCh;
  const float* vb = value + (int64_t)b * Nv * row_stride + h * kCh + sub * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* ml = s_loc + it * LP * 2;
  const float* mw = s_w + it * LP;
  for (int l = 0; l < L; ++l) {
    const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
    const int64_t base = lsi[l] * row_stride;
    for (int p = 0; p < P; ++p) {
      const float x = ml[(l * P + p) * 2] * Wl - 0.5f;
      const float y = ml[(l * P + p) * 2 + 1] * Hl - 0.5f;
      const float w = mw[l * P + p];
      if (y > -1.f && x > -1.f && y < Hl && x < Wl) {
        const Corner c = corners(x, y, Hl, Wl, base, row_stride);
        const float4 v00 = ld4(vb, c.o00), v01 = ld4(vb, c.o01), v10 = ld4(vb, c.o10),
                     v11 = ld4(vb, c.o11);
        acc.x += w * (c.w00 * v00.x + c.w01 * v01.x + c.w10 * v10.x + c.w11 * v11.x);
        acc.y += w * (c.w00 * v00.y + c.w01 * v01.y + c.w10 * v10.y + c.w11 * v11.y);
        acc.z += w * (c.w00 * v00.z + c.w01 * v01.z + c.w10 * v10.z + c.w11 * v11.z);
        acc.w += w * (c.w00 * v00.w + c.w01 * v01.w + c.w10 * v10.w + c.w11 * v11.w);
      }
    }
  }
  *reinterpret_cast<float4*>(out + item * kCh + sub * 4) = acc;
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
    int64_t n_items, int nblocks) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int LP = L * P;
  float* s_loc = smem;                         // [kBItems][LP*2]  in: loc, out: grad_loc
  float* s_w = smem + kBItems * LP * 2;        // [kBItems][LP]    in: w,   out: grad_w
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
    const float go = grad_out[item * kCh + ch];
    float* ml = s_loc + it * LP * 2;
    float* mw = s_w + it * LP;
    for (int l = 0; l < L; ++l) {
      const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
      const int64_t base = lsi[l] * row_stride;
      for (int p = 0; p < P; ++p) {
        const float x = ml[(l * P + p) * 2] * Wl - 0.5f;
        const float y = ml[(l * P + p) * 2 + 1] * Hl - 0.5f;
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
  for (int i = threadIdx.x; i < nvalid * LP * 2; i += kThreads) grad_loc[item0 * LP * 2 + i] = s_loc[i];
  for (int i = threadIdx.x; i < nvalid * LP; i += kThreads) grad_w[item0 * LP + i] = s_w[i];
}

// ---------------------------------------------------------------------------------------------
// backward, pair-merging variant (A/B switch vidar_msda_set_bwd_pair_merge, default off until it has
// been measured).  The scatter sits on the atomic-unit wall (one request per instruction x 128-byte
// line), so the only lever is fewer requests.  Here the two half-waves of a wave own the SAME head of
// two ADJACENT queries (bq, bq+1): neighbouring BEV queries project next to each other, and on the
// coarse pyramid levels / for far queries the same (level, point) sample of both lands in the same
// pixel cell or in cells that share an edge.  The half-waves exchange their four corner line ids and
// values by shuffle; a line both of them touch is written once, by the lower half, with the sum.
// Results differ from the plain kernel only in fp32 summation order.
// Workgroup = 4 waves = 4 heads x 2 queries; grid = ceil(B*Nq/2) query pairs x ceil(H/4) head groups.
// ---------------------------------------------------------------------------------------------
constexpr int kPHeads = kThreads / 64;         // heads per workgroup (one wave each)
constexpr int kPItems = 2 * kPHeads;           // LDS slots: [half][head]

__global__ __launch_bounds__(kThreads) void msda_bwd_pair_kernel(
    const float* __restrict__ value, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lsi, const float* __restrict__ loc, const float* __restrict__ attw,
    const float* __restrict__ grad_out, float* __restrict__ grad_value,
    float* __restrict__ grad_loc, float* __restrict__ grad_w, int Nv, int H, int Nq, int L, int P,
    int64_t n_bq, int n_hg, int nblocks) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int LP = L * P;
  float* s_loc = smem;                         // [kPItems][LP*2]  in: loc, out: grad_loc
  float* s_w = smem + kPItems * LP * 2;        // [kPItems][LP]    in: w,   out: grad_w
  const int blk = xcd_remap(blockIdx.x, nblocks);
  if (blk >= nblocks) return;
  const int64_t pair = blk / n_hg;
  const int h0 = (blk % n_hg) * kPHeads;       // first head of this workgroup
  const int nh = min(kPHeads, H - h0);         // valid heads here
  const int64_t bq0 = pair * 2;
  const int nhalf = (bq0 + 1 < n_bq) ? 2 : 1;  // valid queries here
  // staging: per half one contiguous run of nh items (heads h0 .. h0+nh-1 of query bq0+half)
  for (int half = 0; half < nhalf; ++half) {
    const int64_t item_first = (bq0 + half) * H + h0;
    float* dl = s_loc + half * kPHeads * LP * 2;
    float* dw = s_w + half * kPHeads * LP;
    for (int i = threadIdx.x; i < nh * LP * 2; i += kThreads) dl[i] = loc[item_first * LP * 2 + i];
    for (int i = threadIdx.x; i < nh * LP; i += kThreads) dw[i] = attw[item_first * LP + i];
  }
  __syncthreads();
  const int wv = threadIdx.x / 64, lane = threadIdx.x % 64;
  const int half = lane / kBLanes, ch = lane % kBLanes;
  if (wv < nh) {                                // wave-uniform: both halves take part in the shuffles
    const bool live = half < nhalf;
    const int64_t bq = bq0 + (live ? half : 0);
    const int h = h0 + wv;
    const int b = (int)(bq / Nq);
    const int row_stride = H * kCh;
    const int64_t item = bq * H + h;
    const int64_t gbase = (int64_t)b * Nv * row_stride + h * kCh;   // element offset of channel 0
    const float* vb = value + gbase + ch;
    float* gvb = grad_value + ch;               // + global line offset below
    const float go = live ? grad_out[item * kCh + ch] : 0.f;
    const int slot = half * kPHeads + wv;
    float* ml = s_loc + slot * LP * 2;
    float* mw = s_w + slot * LP;
    for (int l = 0; l < L; ++l) {
      const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
      const int64_t base = lsi[l] * row_stride;
      for (int p = 0; p < P; ++p) {
        float x = 0.f, y = 0.f, w = 0.f;
        if (live) {
          x = ml[(l * P + p) * 2] * Wl - 0.5f;
          y = ml[(l * P + p) * 2 + 1] * Hl - 0.5f;
          w = mw[l * P + p];
        }
        float gx = 0.f, gy = 0.f, gw = 0.f;
        int64_t o[4] = {-1, -1, -1, -1};        // global line offsets (channel 0) of my corners
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (live && y > -1.f && x > -1.f && y < Hl && x < Wl) {   // uniform over a half-wave
          const Corner c = corners(x, y, Hl, Wl, base, row_stride);
          const float d00 = ld1(vb, c.o00) * go, d01 = ld1(vb, c.o01) * go,
                      d10 = ld1(vb, c.o10) * go, d11 = ld1(vb, c.o11) * go;
          const float hh = 1.f - c.lh, hw = 1.f - c.lw;
          gw = c.w00 * d00 + c.w01 * d01 + c.w10 * d10 + c.w11 * d11;
          gx = w * Wl * (-hh * d00 + hh * d01 - c.lh * d10 + c.lh * d11);
          gy = w * Hl * (-hw * d00 - c.lw * d01 + hw * d10 + c.lw * d11);
          const float wg = w * go;
          if (c.o00 >= 0) { o[0] = gbase + c.o00; v[0] = c.w00 * wg; }
          if (c.o01 >= 0) { o[1] = gbase + c.o01; v[1] = c.w01 * wg; }
          if (c.o10 >= 0) { o[2] = gbase + c.o10; v[2] = c.w10 * wg; }
          if (c.o11 >= 0) { o[3] = gbase + c.o11; v[3] = c.w11 * wg; }
        }
        // exchange with the partner half (same channel, other query)
        int64_t po[4];
        float pv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          po[j] = __shfl_xor((long long)o[j], kBLanes, 64);
          pv[j] = __shfl_xor(v[j], kBLanes, 64);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float total = v[i];
          bool issue = o[i] >= 0;
          if (half == 0) {                      // owner of shared lines: add the partner's share
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (issue && po[j] == o[i]) total += pv[j];
          } else {                              // a line the lower half also writes is left to it
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (po[j] >= 0 && po[j] == o[i]) issue = false;
          }
          if (issue) unsafeAtomicAdd(gvb + o[i], total);
        }
        gx = half_wave_sum(gx); gy = half_wave_sum(gy); gw = half_wave_sum(gw);
        if (ch == 0 && live) {
          ml[(l * P + p) * 2] = gx;
          ml[(l * P + p) * 2 + 1] = gy;
          mw[l * P + p] = gw;
        }
      }
    }
  }
  __syncthreads();
  for (int half = 0; half < nhalf; ++half) {
    const int64_t item_first = (bq0 + half) * H + h0;
    const float* sl = s_loc + half * kPHeads * LP * 2;
    const float* sw = s_w + half * kPHeads * LP;
    for (int i = threadIdx.x; i < nh * LP * 2; i += kThreads) grad_loc[item_first * LP * 2 + i] = sl[i];
    for (int i = threadIdx.x; i < nh * LP; i += kThreads) grad_w[item_first * LP + i] = sw[i];
  }
}

int g_bwd_pair_merge = 0;

inline bool msda_bad(int B, int Nv, int H, int C, int Nq, int L, int P) {
  return B < 0 || Nv < 0 || H <= 0 || C != kCh || Nq < 0 || L <= 0 || P <= 0 || L * P > kMaxLP;
}

}  // namespace

extern "C" {

int vidar_msda_set_xcd_remap(int enabled) {
  const int v = enabled ? 0 : 1;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_no_remap), &v, sizeof(int));
}

int vidar_msda_set_bwd_pair_merge(int enabled) {
  const int prev = g_bwd_pair_merge;
  g_bwd_pair_merge = enabled ? 1 : 0;
  return prev;
}

int vidar_msda_fwd_f32(const float* value, const int64_t* spatial_shapes,
                       const int64_t* level_start_index, const float* sampling_loc,
                       const float* attn_weight, float* out, int B, int Nv, int H, int C, int Nq,
                       int L, int P, void* stream) {
  VIDAR_ENTER();
  if (msda_bad(B, Nv, H, C, Nq, L, P)) return VIDAR_ERR_BAD_ARG;
  const int64_t n_items = (int64_t)B * Nq * H;
  if (n_items == 0) return 0;
  const int nblocks = (int)((n_items + kItems - 1) / kItems);
  const int grid = ((nblocks + 7) / 8) * 8;
  const size_t lds = sizeof(float) * kItems * L * P * 3;
  hipLaunchKernelGGL(msda_fwd_kernel, dim3(grid), dim3(kThreads), lds, (hipStream_t)stream, value,
                     spatial_shapes, level_start_index, sampling_loc, attn_weight, out, Nv, H, Nq, L,
                     P, n_items, nblocks);
  return vidar_last_error();
}

int vidar_msda_bwd_f32(const float* value, const int64_t* spatial_shapes,
                       const int64_t* level_start_index, const float* sampling_loc,
                       const float* attn_weight, const float* grad_out, float* grad_value,
                       float* grad_sampling_loc, float* grad_attn_weight, int B, int Nv, int H, int C,
                       int Nq, int L, int P, void* stream) {
  VIDAR_ENTER();
  if (msda_bad(B, Nv, H, C, Nq, L, P)) return VIDAR_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  const size_t vbytes = sizeof(float) * (size_t)B * Nv * H * C;
  if (vbytes) {
    hipError_t e = hipMemsetAsync(grad_value, 0, vbytes, s);
    if (e != hipSuccess) return (int)e;
  }
  const int64_t n_items = (int64_t)B * Nq * H;
  if (n_items == 0) return 0;
  if (g_bwd_pair_merge) {
    const int64_t n_bq = (int64_t)B * Nq;
    const int n_hg = (H + kPHeads - 1) / kPHeads;
    const int nb = (int)(((n_bq + 1) / 2) * n_hg);
    const int grid_p = ((nb + 7) / 8) * 8;
    const size_t lds_p = sizeof(float) * kPItems * L * P * 3;
    hipLaunchKernelGGL(msda_bwd_pair_kernel, dim3(grid_p), dim3(kThreads), lds_p, s, value, spatial_shapes,
                       level_start_index, sampling_loc, attn_weight, grad_out, grad_value,
                       grad_sampling_loc, grad_attn_weight, Nv, H, Nq, L, P, n_bq, n_hg, nb);
    return vidar_last_error();
  }
  const int nblocks = (int)((n_items + kBItems - 1) / kBItems);
  const int grid = ((nblocks + 7) / 8) * 8;
  const size_t lds = sizeof(float) * kBItems * L * P * 3;
  hipLaunchKernelGGL(msda_bwd_kernel, dim3(grid), dim3(kThreads), lds, s, value, spatial_shapes,
                     level_start_index, sampling_loc, attn_weight, grad_out, grad_value,
                     grad_sampling_loc, grad_attn_weight, Nv, H, Nq, L, P, n_items, nblocks);
  return vidar_last_error();
}

}  // extern "C"
