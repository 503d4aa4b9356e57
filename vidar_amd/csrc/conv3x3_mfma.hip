// 3x3 convolution with FEW output channels (<= 32) as an implicit GEMM on the fp32 matrix cores (round 5).
//
// The path's user: the `conv_offset` of every DCNv2 bottleneck (mmcv ModulatedDeformConv2dPack: a plain 3x3, stride 1,
// pad 1 convolution C -> 27 channels that predicts the 18 offsets + 9 masks; vidar_1_8_nusc_1future.py:93-95 switches
// DCNv2 on for ResNet101's stages 3 and 4 = 26 of the backbone's 37 3x3 convolutions, and with queue_length 4 each runs on
// 24 no-grad history images + the 6 images of the current frame).  The library's Winograd kernel spends 0.8 ms on the
// [24, 256, 58, 100] case (21 TFLOP/s): with 27 output channels it cannot fill its tiles.  Here:
//
//   out[n, o, p] = bias[o] + sum_{c, ky, kx} w[o, c, ky, kx] * x[n, c, p + (ky-1) W + (kx-1)]      (zero outside the image)
//
// is the GEMM  D[o, p] = sum_k A[o, k] B[k, p]  with k = (c, tap): A = the weights, padded to 32 rows, B = the image plane
// SHIFTED by the tap -- no column matrix exists anywhere.  A workgroup owns 128 consecutive pixels of one image (they may
// span rows) and walks the channels in chunks of 8; per chunk it stages
//   slab[8][SW]   : the 8 planes' pixels [p0 - W - 1, p0 + 128 + W + 1)  (zero outside [0, HW): the top / bottom padding)
//   wl[8][9][32]  : the chunk's weights, from the packed copy a tiny pre-pass writes (k-major, 32 outputs contiguous)
// in LDS (double buffered: the next chunk's global loads fly under the MFMAs, one barrier per chunk).  A wave owns 32
// pixels x 32 outputs: per channel PAIR and tap one v_mfma_f32_32x32x2_f32 whose A operand is wl[2 pair + h][tap][lane & 31]
// and whose B operand is slab[2 pair + h][32 wave + (lane & 31) + ky W + kx]  (h = lane >> 5 picks the channel of the pair:
// the k index of the 32x32x2 instruction), both plain conflict-free ds_read_b32.  Left / right padding = the pixel's own
// column test, constant over the whole K loop (two flags per lane; a wave without a border pixel skips the selects).  Two accumulators take turns so that consecutive MFMAs
// are independent.  fp32 products are exact in the matrix core; the sum runs channel-major, taps inside.
//
// Bounds: 2 * 32 * C * 9 flops per pixel on the fp32 matrix peak (157.3 TFLOP/s; 27 of the 32 rows are useful) -- the HBM
// side is one read of x and one write of out (0.02 ms for the case above).
#include "vidar_common.h"
#include "vidar_hip.h"
#include <type_traits>

namespace {

constexpr int TP = 128;        // pixels per workgroup
constexpr int THREADS = 256;   // 4 waves x 32 pixels
constexpr int CK = 8;          // channels per chunk (chunks of 4 -- half the LDS, twice the workgroups per CU -- measured the same)
constexpr int OB = 32;         // output rows of the MFMA tile
constexpr int MAXI = 8;        // slab columns per lane: TP + 2 W + 2 <= 64 * MAXI  ->  W <= 191

typedef float f32x16 __attribute__((ext_vector_type(16)));

// timing ablations of tools/conv_ablate.py (never set in the library build; the results are then wrong, only the time is of
// interest): 1 no border masks, 2 operands not read from LDS, 4 no global fetch / LDS staging, 8 no per-chunk barrier,
// 16 no MFMA.  What they showed on [24, 256, 58, 100] (profiles/r05_conv3x3_ablation.log): MFMAs alone 0.163 ms (80 % of
// the padded fp32 matrix peak: the 1104 workgroups balance to 86 %), the full kernel 0.251 -- and the whole difference
// is the LDS operand reads (without them 0.162 with fetch, staging and barriers still in place).  A 32-row tile consumes two
// fresh operand dwords per lane per MFMA, four to eight times what a 128 x 128 GEMM tile with its 2 x 2 fragment reuse
// needs, and neither twice the occupancy (chunks of 4 channels), nor the weights as three float4 per lane, nor reading a
// pair's / a whole chunk's operands ahead of its MFMAs behind a sched_barrier moved it by more than 6 % (0.229 - 0.237 ms).
#ifndef VIDAR_CONV_ABL
#define VIDAR_CONV_ABL 0
#endif
constexpr int ABL = VIDAR_CONV_ABL;

// packed weights: [c][tap][32 outputs], zero rows beyond Cout (one conflict-free ds_read_b32 per operand)
__global__ void conv3x3_pack_kernel(const float* __restrict__ w, float* __restrict__ wt, int C, int Cout) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= C * 9 * OB) return;
  const int o = e & (OB - 1), tap = (e >> 5) % 9, c = e / (9 * OB);
  wt[e] = o < Cout ? w[((int64_t)o * C + c) * 9 + tap] : 0.0f;
}

struct ConvArgs {
  const float* x; const float* wt; const float* bias; float* out;
  int C, H, W, HW, Cout, tiles, total, per_xcd;
};

template <int ITERS>
__global__ __launch_bounds__(THREADS, 3) void conv3x3_few_kernel(ConvArgs g) {
  constexpr int SW = 64 * ITERS + 32;              // == 32 (mod 64): the pair's two planes sit on different bank halves
  constexpr int WL = CK * 9 * OB;
  constexpr int CPW = CK / 4;                      // channels a wave stages per chunk
  constexpr int WLD = (WL + THREADS - 1) / THREADS; // weight dwords a thread stages per chunk (the last one may be partial)
  __shared__ float slab[2][CK * SW];
  __shared__ float wl[2][WL];
  // XCD-aware order: workgroup b runs on XCD b % 8; give each XCD a contiguous range of tiles so that the workgroups
  // sharing an L2 are neighbours in the image (their halos overlap)
  const int t = (blockIdx.x & 7) * g.per_xcd + (blockIdx.x >> 3);
  if (t >= g.total) return;
  const int n = t / g.tiles, p0 = (t - n * g.tiles) * TP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int W = g.W, HW = g.HW;
  const float* xn = g.x + (int64_t)n * g.C * HW;
  const int q0 = p0 - W - 1;                       // global pixel of slab column 0

  // the chunk's operands travel global -> registers -> LDS; every load is unconditional (clamped address, value
  // selected afterwards) so that the whole batch is in flight at once
  float sx[CPW][ITERS], sw[WLD];
  int qc[ITERS]; bool qv[ITERS];
#pragma unroll
  for (int i = 0; i < ITERS; ++i) {
    const int q = q0 + lane + 64 * i;
    qv[i] = q >= 0 && q < HW;
    qc[i] = q < 0 ? 0 : (q < HW ? q : HW - 1);
  }
  auto fetch = [&](int c0) {
#pragma unroll
    for (int u = 0; u < CPW; ++u) {                // wave w stages channels CPW w ... of the chunk
      const float* plane = xn + (int64_t)(c0 + CPW * wave + u) * HW;
#pragma unroll
      for (int i = 0; i < ITERS; ++i) sx[u][i] = plane[qc[i]];
    }
    const float* wsrc = g.wt + (int64_t)c0 * 9 * OB;
#pragma unroll
    for (int i = 0; i < WLD; ++i) sw[i] = (WL % THREADS == 0 || tid + THREADS * i < WL) ? wsrc[tid + THREADS * i] : 0.0f;
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int u = 0; u < CPW; ++u)
#pragma unroll
      for (int i = 0; i < ITERS; ++i) slab[buf][(CPW * wave + u) * SW + lane + 64 * i] = qv[i] ? sx[u][i] : 0.0f;
#pragma unroll
    for (int i = 0; i < WLD; ++i)
      if (WL % THREADS == 0 || tid + THREADS * i < WL) wl[buf][tid + THREADS * i] = sw[i];
  };

  const int p = p0 + 32 * wave + l31;              // this lane's pixel (B column / D column)
  const int col = p % W;
  const bool left = col == 0, right = col == W - 1;
  const bool any_border = __ballot(left || right) != 0;   // wave-uniform: two thirds of the 32-pixel blocks have none
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }

  fetch(0);
  stage(0);
  __syncthreads();
  int buf = 0;
  for (int c0 = 0; c0 < g.C; c0 += CK) {
    const bool more = c0 + CK < g.C;
    if (more && !(ABL & 4)) fetch(c0 + CK);
    const float* sl = slab[buf] + h * SW + 32 * wave + l31;
    const float* wp = wl[buf] + h * 9 * OB + l31;
    // one channel pair = 9 taps = 9 MFMAs; the NEXT pair's 18 operands are read from LDS before this pair's MFMAs issue.
    // `masked`: only a wave whose 32 pixels include a first / last column runs the variant with the padding selects
    auto compute = [&](auto masked) {
      auto frags = [&](int cp, float (&a)[9], float (&b)[9]) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int tap = ky * 3 + kx;
            if (ABL & 2) { a[tap] = (float)(lane + tap); b[tap] = (float)(cp + tap); continue; }
            a[tap] = wp[(2 * cp * 9 + tap) * OB];
            float v = sl[2 * cp * SW + ky * W + kx];
            if (decltype(masked)::value && !(ABL & 1)) {
              if (kx == 0 && left) v = 0.0f;       // left / right zero padding: the pixel's own column
              if (kx == 2 && right) v = 0.0f;
            }
            b[tap] = v;
          }
      };
      auto mfmas = [&](int cp, const float (&a)[9], const float (&b)[9]) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          if (ABL & 16) { acc0[tap] += a[tap] * b[tap]; continue; }
          if ((cp * 9 + tap) & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tap], b[tap], acc1, 0, 0, 0);
          else acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tap], b[tap], acc0, 0, 0, 0);
        }
      };
      float fa[2][9], fb[2][9];
      frags(0, fa[0], fb[0]);
#pragma unroll
      for (int cp = 0; cp < CK / 2; ++cp) {
        if (cp + 1 < CK / 2) frags(cp + 1, fa[(cp + 1) & 1], fb[(cp + 1) & 1]);
        mfmas(cp, fa[cp & 1], fb[cp & 1]);
      }
    };
    if (any_border) compute(std::true_type{});
    else compute(std::false_type{});
    if (more && !(ABL & 4)) stage(buf ^ 1);
    if (!(ABL & 8)) __syncthreads();
    buf ^= 1;
  }

  if (p >= HW) return;
  float* on = g.out + (int64_t)n * g.Cout * HW + p;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int o = (r & 3) + 8 * (r >> 2) + 4 * h;  // accumulator register r of the 32x32 tile: row o, column lane & 31
    if (o < g.Cout) on[(int64_t)o * HW] = acc0[r] + acc1[r] + (g.bias ? g.bias[o] : 0.0f);
  }
}

template <int ITERS>
void launch_conv(const ConvArgs& g, hipStream_t stream) {
  hipLaunchKernelGGL(conv3x3_few_kernel<ITERS>, dim3(g.per_xcd * 8), dim3(THREADS), 0, stream, g);
}

}  // namespace

extern "C" size_t vidar_conv3x3_few_workspace_bytes(int C) {
  return C > 0 ? (size_t)C * 9 * OB * sizeof(float) : 0;
}

extern "C" int vidar_conv3x3_few_f32(const float* x, const float* weight, const float* bias, float* out, int N, int C,
                                     int H, int W, int Cout, void* workspace, size_t workspace_bytes, void* stream) {
  VIDAR_ENTER();
  if (N < 0 || C <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cout > OB || C % CK != 0) return VIDAR_ERR_BAD_ARG;
  if (TP + 2 * W + 2 > 64 * MAXI) return VIDAR_ERR_BAD_ARG;
  if ((int64_t)H * W >= (1 << 30) || (int64_t)N * (((int64_t)H * W + TP - 1) / TP) >= (1 << 28)) return VIDAR_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < vidar_conv3x3_few_workspace_bytes(C)) return VIDAR_ERR_BAD_ARG;
  if (N == 0) return 0;
  float* wt = static_cast<float*>(workspace);
  const int elems = C * 9 * OB;
  hipLaunchKernelGGL(conv3x3_pack_kernel, dim3((elems + 255) / 256), dim3(256), 0, (hipStream_t)stream, weight, wt, C, Cout);
  ConvArgs g;
  g.x = x; g.wt = wt; g.bias = bias; g.out = out;
  g.C = C; g.H = H; g.W = W; g.HW = H * W; g.Cout = Cout;
  g.tiles = (g.HW + TP - 1) / TP;
  g.total = g.tiles * N;
  g.per_xcd = (g.total + 7) / 8;
  const int iters = (TP + 2 * W + 2 + 63) / 64;    // slab columns per lane (3 .. MAXI)
  hipStream_t st = (hipStream_t)stream;
  switch (iters) {
    case 3: launch_conv<3>(g, st); break;
    case 4: launch_conv<4>(g, st); break;
    case 5: launch_conv<5>(g, st); break;
    case 6: launch_conv<6>(g, st); break;
    case 7: launch_conv<7>(g, st); break;
    default: launch_conv<8>(g, st); break;
  }
  return vidar_last_error();
}
