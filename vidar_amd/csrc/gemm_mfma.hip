// gemm_mfma.hip -- the hot path's matrix products on the gfx950 matrix cores, hand written.
//
//   C[z] = epilogue( A[z] (M x K)  *  B[z] (K x N) )        fp32 in HBM on both sides, fp32 accumulate
//
// Two arithmetic modes share every line of the tile machinery:
//   VIDAR_GEMM_F32    v_mfma_f32_32x32x2_f32 : exact fp32 products, one rounding per product (the arithmetic of
//                     the rocBLAS / hipBLASLt kernels it replaces), 1/16 of the bf16 matrix rate
//   VIDAR_GEMM_BF16X3 every operand is split while it is staged, x = hi + lo with hi = bf16(x), lo = bf16(x - hi),
//                     and a product is three v_mfma_f32_32x32x16_bf16 into ONE fp32 accumulator:
//                     lo*hi + hi*lo + hi*hi.  hi + lo carries 16 significand bits (relative error <= 2^-16;
//                     TF32 -- what the reference's pinned torch 1.10 runs these products in on A100,
//                     README.md:96, tools/train.py:141-144 -- carries 11), the dropped lo*lo term is <= 2^-16 of
//                     a product, at 3/16 of the fp32 matrix-core cost.
//
// Operand layouts (the four combinations cover forward, grad-input and grad-weight of nn.Linear on [rows, C]
// activations and of 1x1 / deformable convolutions on NCHW activations without a transposed copy):
//   K-major : the contraction index is contiguous   (x [M,K] of F.linear; weight [N,K] as B)
//   MN-major: the row / column index is contiguous  (NCHW activations [C, H*W] as B; grad_out^T as A)
//
// Tile: 128 x 128 x 32 per 256-thread workgroup, 4 waves as 2 x 2, each wave 64 x 64 = 2 x 2 MFMA tiles of
// 32 x 32 (64 accumulator registers).  Staging is global -> registers -> (split) -> LDS with the next k-tile's global
// loads in flight under the MFMAs.  LDS images (a "unit" is one dword = one fp32 k or one bf16 k-pair; U = 32 units per
// k-step in fp32 mode, 16 in bf16 mode):
//   K-major  operand: fp32 [128 rows][32 units + 4 pad] (row stride 144 B, slot stride 9 odd: ds_read_b128 fragments
//                     conflict free); bf16 [128 rows][16 units], unpadded, 16-byte chunks XOR-swizzled (Geo::KM_STRIDE)
//   MN-major operand: [U units][128 rows + 8 pad]    four ds_read_b32 per fragment, 32 consecutive dwords per group
// A lane's fragment is always the four units 8*s + 4*(lane>>5) + {0..3} of its row -- in bf16 mode they ARE the
// eight consecutive k of one 32x32x16 operand, in fp32 mode they feed four 32x32x2 MFMAs whose two k slots
// (lane halves) take units t and 4 + t; both operands use the same assignment, which is all a contraction needs.
// An accumulator register of a wave is two rows of 32 consecutive n: the epilogue's stores are whole 128-byte lines.
//
// Memory side: every global access is a buffer instruction whose descriptor starts at the workgroup's tile -- one 32-bit
// per-thread offset, the k / row advance in a scalar register, and the descriptor's END doing the bounds checking
// (rows past the matrix read 0 / are not stored); the lanes of a load run along the contiguous index so that a wave
// instruction moves whole 128-byte lines.  Workgroups are persistent (one chip residency) and request the next tile's
// first operands before they store the current tile.  profiles/r04_kbench_gemm*.log, r04_pmc_gemm/: DESIGN.md section 4a.
//
// Epilogue: y = relu?( acc * scale[m|n] + shift[m|n] + residual[m,n] ) -- the bias of F.linear, the frozen
// BatchNorm + residual + ReLU that follows every 1x1 convolution of the ResNet bottlenecks
// (config vidar_1_8_nusc_1future.py:88-106: norm_eval, requires_grad False), the FFN's ReLU.
// Split-K / batch-reduced products (weight gradients) write fp32 slabs and a second kernel sums them in a
// fixed order (deterministic; no atomics).
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "vidar_hip.h"
#include "vidar_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));     // 4-byte aligned: rows of H*W = 1450 floats
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int BM = 128, BN = 128, THREADS = 256;
constexpr int BK = 32;                    // k per step in both modes
constexpr int MN_STRIDE = 128 + 8;        // dwords per unit row of an MN-major image
// geometry of an LDS image per mode: a "unit" is one dword of k = one fp32 k (32 per k-step) or one bf16 k pair (16)
template <int PREC>
struct Geo {
  static constexpr int UNITS = (PREC == 1) ? 16 : 32;
  // dwords per row of a K-major image.  fp32: 32 + 4 pad (slot stride 9, odd: ds_read_b128 fragments conflict free).
  // bf16: 16, UNPADDED, with the row's four 16-byte chunks XOR-swizzled by (row >> 2) & 3 (km_chunk): the 8-byte
  // staging stores of a 16-lane group then cover 32 consecutive dwords (the padded image made them 2-way conflicted:
  // a third of the LDS cycles, profiles/r04_pmc_gemm), and the 16 rows of a ds_read_b128 group still hit 16 distinct
  // 16-byte bank groups (rows with equal r & 3 differ in (r >> 2) & 3).
  static constexpr int KM_STRIDE = (PREC == 1) ? UNITS : UNITS + 4;
  static constexpr int IMG_DWORDS = (128 * KM_STRIDE > UNITS * MN_STRIDE) ? 128 * KM_STRIDE : UNITS * MN_STRIDE;
  static constexpr int NS = UNITS / 8;                 // fragment sub-steps per k-step
};
constexpr int PREC_F32 = 0, PREC_BF16X3 = 1;
constexpr int LAY_K = 0, LAY_MN = 1;

struct GemmArgs {
  const float* A; const float* B; float* C;
  int64_t lda, ldb, ldc, sA, sB, sC;
  int M, N, K;
  int splits, k_chunk;          // z = batch * splits + split ; the split covers k in [split*k_chunk, +k_chunk)
  int slabs;                    // 1: every z writes the slab ws[z] (no epilogue); 0: z = batch item, epilogue to C
  const float* scale; const float* shift; int vec_axis;      // 0: indexed by n, 1: indexed by m
  const float* residual; int64_t ldr, sR;
  int relu;
  int tiles_m, tiles_n, m_fastest;
  int total;       // tiles_m * tiles_n * (batch * splits)
  // a_rowsum != nullptr (reduced products with an MN-major A only): out[m] = sum_z sum_k A[z][m, k] -- the bias gradient of
  // an nn.Linear rides along with its weight gradient (A = grad_out^T).  The workgroups of tile column 0 add the A values
  // they stage anyway and write one partial row per slab behind the slabs; the slab reduction sums them in z order.
  float* a_rowsum;
  float* rowsum_ws;     // [Z][M] partial rows (inside the caller's workspace)
};

// ---- staging: 16 floats per operand per thread ------------------------------------------------------------------------
template <int PREC>
struct Staged {               // the registers a thread holds between its global loads and its LDS writes
  static constexpr int CH = 2;                                // chunks of 8 floats: 16 floats per operand per thread
  f32x4 v[CH][2];
};

// K-major operand: element (row, k) at P[row * ld + k].  The lanes of a wave instruction run along k first: LPR = BK/4 = 8
// lanes x 16 bytes cover the whole k-step of one row (one 128-byte line), 8 rows per instruction, so every request is a
// full contiguous piece (one lane per row, or a pair of
// lanes 64 bytes apart, made each dwordx4 touch 32 lines in 16-byte pieces: the kernel was bound by the texture
// addresser, not by MFMA or HBM).  float4 number i of a thread: row 32*wave + i*(64/LPR) + lane/LPR, k (lane%LPR)*4.
template <int PREC>
struct KMap {
  static constexpr int LPR = 8;                                 // lanes per row: 8 x 16 B = the 128-byte k-step of a row
  static constexpr int RPI = 64 / LPR;                          // rows per instruction
  static constexpr int NI = 32 / RPI;                           // instructions (float4 per thread)
};

// Addressing of both loaders: buffer loads through a descriptor whose base is the workgroup's (uniform) tile origin at
// the first k of its range, ONE 32-bit per-thread byte offset computed before the loop (voffset) and the uniform advance
// of the k loop in a scalar register (soffset): no vector address arithmetic inside the loop.  The descriptor is built
// from readfirstlane'd halves of the pointer so that the compiler can see it is uniform (no waterfall loops).
// The host checks that the offsets fit 31 bits.
typedef uint32_t raw4 __attribute__((__vector_size__(4 * sizeof(uint32_t))));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, int64_t bytes = 0x7fffffff) {
  const uint64_t a = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  const int32_t n = __builtin_amdgcn_readfirstlane((int32_t)(bytes > 0x7fffffff ? 0x7fffffff : bytes));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, n, 0x00020000);
}
__device__ __forceinline__ f32x4 ld16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ float ld4(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

template <int PREC>
__device__ __forceinline__ uint32_t voff_kmajor(int64_t ld, int tid) {
  using M = KMap<PREC>;
  const int lane = tid & 63, wave = tid >> 6;
  return (uint32_t)(((32 * wave + lane / M::LPR) * (uint32_t)ld + (lane % M::LPR) * 4) * 4);
}

// descriptor base = P + row0 * ld + kbeg, ending at the matrix' last element ; krel = k0 - kbeg.
// No per-lane bounds checks: rows past the matrix only feed output rows that are never stored (and read as 0 past the
// descriptor's end); a k-step that reaches past K reads the start of the next row, so its lanes are masked -- `tail` is
// workgroup-uniform and true for at most the last k-step of a product whose K is not a multiple of the k-step.
template <int PREC>
__device__ __forceinline__ void load_kmajor(Staged<PREC>& s, __amdgpu_buffer_rsrc_t rs, int64_t ld, int krel, int k0, int kend,
                                            bool tail, int tid, uint32_t voff) {
  using M = KMap<PREC>;
#pragma unroll
  for (int i = 0; i < M::NI; ++i)
    s.v[i >> 1][i & 1] = ld16(rs, voff, (uint32_t)((i * M::RPI * (uint32_t)ld + krel) * 4));
  if (tail) {
    const int k = k0 + ((tid & 63) % M::LPR) * 4;
#pragma unroll
    for (int i = 0; i < M::NI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) s.v[i >> 1][i & 1][j] = (k + j < kend) ? s.v[i >> 1][i & 1][j] : 0.0f;
  }
}

// MN-major operand: element (k, col) at P[k * ld + col].  bf16 mode: item = tid + 256c, column quad item&31, k pair
// item>>5 (two rows of four columns).  fp32 mode: items tid + 256 i, i = 0..3: column quad item&31, k = item>>5.
template <int PREC>
__device__ __forceinline__ uint32_t voff_mnmajor(int64_t ld, int tid) {
  const int kr = (PREC == PREC_BF16X3) ? (tid >> 5) * 2 : (tid >> 5);
  return (uint32_t)((kr * (uint32_t)ld + (tid & 31) * 4) * 4);
}

// descriptor base = P + kbeg * ld + col0, ending at the matrix' last element ; krel = k0 - kbeg.  Rows k >= K lie past the
// descriptor's end and read as 0 (the k tail needs no mask); columns past the matrix only feed outputs that are never stored.
template <int PREC>
__device__ __forceinline__ void load_mnmajor(Staged<PREC>& s, __amdgpu_buffer_rsrc_t rs, int64_t ld, int krel, uint32_t voff) {
#pragma unroll
  for (int i = 0; i < Staged<PREC>::CH * 2; ++i) {
    // uniform k of this load relative to the thread's own row: bf16 mode 16*(i>>1) + (i&1), fp32 mode 8*i
    const int ku = (PREC == PREC_BF16X3) ? 16 * (i >> 1) + (i & 1) : 8 * i;       // i = 0..3 in both modes
    s.v[i >> 1][i & 1] = ld16(rs, voff, (uint32_t)(krel + ku) * (uint32_t)ld * 4u);
  }
}

// x = hi + lo: hi = bf16_rne(x), lo = bf16_rne(x - float(hi)) ; two values at a time (v_cvt_pk_bf16_f32)
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
  f32x2 x = {a, b};
  bf16x2 h = __builtin_convertvector(x, bf16x2);
  f32x2 r = x - __builtin_convertvector(h, f32x2);
  bf16x2 l = __builtin_convertvector(r, bf16x2);
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, l);
}

// LDS writes.  `img` = the operand's image (bf16 mode: hi at img, lo at img + IMG_DWORDS).
template <int PREC>
__device__ __forceinline__ void store_kmajor(const Staged<PREC>& s, uint32_t* img, int tid) {
  using M = KMap<PREC>;
  const int lane = tid & 63, wave = tid >> 6;
  const int r0 = 32 * wave + lane / M::LPR, kq = lane % M::LPR;
#pragma unroll
  for (int i = 0; i < M::NI; ++i) {
    const int r = r0 + i * M::RPI;
    const f32x4 v = s.v[i >> 1][i & 1];
    if (PREC == PREC_BF16X3) {             // four k -> two units (k pairs) of hi and of lo
      uint32_t h0, l0, h1, l1;
      split2(v[0], v[1], h0, l0);
      split2(v[2], v[3], h1, l1);
      const int pos = r * Geo<PREC>::KM_STRIDE + ((((kq >> 1) ^ ((r >> 2) & 3)) << 2) | ((kq & 1) << 1));
      *(u32x2*)(img + pos) = u32x2{h0, h1};
      *(u32x2*)(img + Geo<PREC>::IMG_DWORDS + pos) = u32x2{l0, l1};
    } else {                               // (stored as dwords, the type every fragment read uses)
      *(u32x4*)(img + r * Geo<PREC>::KM_STRIDE + 4 * kq) = __builtin_bit_cast(u32x4, v);
    }
  }
}

template <int PREC>
__device__ __forceinline__ void store_mnmajor(const Staged<PREC>& s, uint32_t* img, int tid) {
  if (PREC == PREC_BF16X3) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int item = tid + 256 * c;
      const int cq = (item & 31) * 4, u = item >> 5;
      u32x4 hi, lo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint32_t h, l;
        split2(s.v[c][0][j], s.v[c][1][j], h, l);      // (k even, k odd) of column j
        hi[j] = h; lo[j] = l;
      }
      *(u32x4*)(img + u * MN_STRIDE + cq) = hi;
      *(u32x4*)(img + Geo<PREC>::IMG_DWORDS + u * MN_STRIDE + cq) = lo;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int item = tid + 256 * i;
      *(u32x4*)(img + (item >> 5) * MN_STRIDE + (item & 31) * 4) = __builtin_bit_cast(u32x4, s.v[i >> 1][i & 1]);
    }
  }
}

// a lane's fragment of row `row` (0..127 inside the tile) for sub-step s: units 8s + 4h + {0..3}
template <int PREC, int LAY>
__device__ __forceinline__ u32x4 frag(const uint32_t* img, int row, int s, int h) {
  if (LAY == LAY_K) {
    if (PREC == PREC_BF16X3)      // chunk 2s + h of the row, at its swizzled position
      return *(const u32x4*)(img + row * Geo<PREC>::KM_STRIDE + (((2 * s + h) ^ ((row >> 2) & 3)) << 2));
    return *(const u32x4*)(img + row * Geo<PREC>::KM_STRIDE + 8 * s + 4 * h);
  }
  u32x4 f;
  const uint32_t* p = img + (8 * s + 4 * h) * MN_STRIDE + row;
#pragma unroll
  for (int j = 0; j < 4; ++j) f[j] = p[j * MN_STRIDE];
  return f;
}

// one output tile of one workgroup: everything here is workgroup-uniform (scalar registers)
struct Tile {
  int z, batch, m0, n0, kbeg, kend;
  __amdgpu_buffer_rsrc_t rsA, rsB;
};

template <int PREC, int ALAY, int BLAY>
__device__ __forceinline__ Tile decode_tile(const GemmArgs& g, int t) {
  // tile order.  Consecutive ids of one XCD (t & 7 = blockIdx & 7: the persistent grid is a multiple of 8) walk the
  // shorter tile axis first, so the big operand's strip is fetched from HBM once and re-read from that XCD's L2.
  const int per_z = g.tiles_m * g.tiles_n;
  int id;
  {
    const int q = g.total >> 3, r = g.total & 7, x = t & 7;
    id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (t >> 3);
  }
  Tile T;
  // (integer division runs on the vector ALU: pin the workgroup-uniform results back into scalar registers)
  T.z = __builtin_amdgcn_readfirstlane(id / per_z);
  const int tt = id - T.z * per_z;
  const int tm = __builtin_amdgcn_readfirstlane(g.m_fastest ? tt % g.tiles_m : tt / g.tiles_n);
  const int tn = __builtin_amdgcn_readfirstlane(g.m_fastest ? tt / g.tiles_m : tt % g.tiles_n);
  T.batch = __builtin_amdgcn_readfirstlane(T.z / g.splits);
  const int split = T.z - T.batch * g.splits;
  T.m0 = tm * BM; T.n0 = tn * BN;
  T.kbeg = split * g.k_chunk;
  T.kend = min(g.K, T.kbeg + g.k_chunk);
  const float* A = g.A + (int64_t)T.batch * g.sA;
  const float* B = g.B + (int64_t)T.batch * g.sB;
  // descriptors: from the tile's first element to the operand's last one (bytes)
  const int64_t kleft = g.K - T.kbeg;
  T.rsA = ALAY == LAY_K ? make_rsrc(A + (int64_t)T.m0 * g.lda + T.kbeg, ((int64_t)(g.M - 1 - T.m0) * g.lda + kleft) * 4)
                        : make_rsrc(A + (int64_t)T.kbeg * g.lda + T.m0, ((kleft - 1) * g.lda + (g.M - T.m0)) * 4);
  T.rsB = BLAY == LAY_K ? make_rsrc(B + (int64_t)T.n0 * g.ldb + T.kbeg, ((int64_t)(g.N - 1 - T.n0) * g.ldb + kleft) * 4)
                        : make_rsrc(B + (int64_t)T.kbeg * g.ldb + T.n0, ((kleft - 1) * g.ldb + (g.N - T.n0)) * 4);
  return T;
}

// global loads of the k-step starting at k0 of tile T into the staging registers
template <int PREC, int ALAY, int BLAY>
__device__ __forceinline__ void fetch(Staged<PREC>& sa, Staged<PREC>& sb, const GemmArgs& g, const Tile& T, int k0, int tid,
                                      uint32_t voa, uint32_t vob) {
  const int krel = k0 - T.kbeg;
  const bool tail = k0 + BK > T.kend;
  if (ALAY == LAY_K) load_kmajor<PREC>(sa, T.rsA, g.lda, krel, k0, T.kend, tail, tid, voa);
  else load_mnmajor<PREC>(sa, T.rsA, g.lda, krel, voa);
  if (BLAY == LAY_K) load_kmajor<PREC>(sb, T.rsB, g.ldb, krel, k0, T.kend, tail, tid, vob);
  else load_mnmajor<PREC>(sb, T.rsB, g.ldb, krel, vob);
}

// the k loop of one tile.  On entry the staging registers hold (or are about to receive) the tile's first k-step.
// sum over the staged k of a thread's A values, per row of its row quad (MN-major staging: every float4 of a thread is
// the same four consecutive rows at another k).  k past the operand's END reads 0 (the descriptor's range check), but a
// float4 that straddles m >= M inside the operand does NOT: with lda == M it reads the first elements of the NEXT k-row,
// so the components of rows >= M hold garbage (possibly Inf / NaN).  Every row is its own vector component and the
// store's `cur.m0 + tid < g.M` guard drops those components: that guard is load-bearing
// (tests/test_gemm_gpu.py::test_rowsum_with_ragged_rows).  Buffer loads need dword alignment only (no 16-byte rule).
template <int PREC>
__device__ __forceinline__ void add_rowsum(f32x4& rs, const Staged<PREC>& sa) {
#pragma unroll
  for (int i = 0; i < Staged<PREC>::CH * 2; ++i) rs += sa.v[i >> 1][i & 1];
}

template <int PREC, int ALAY, int BLAY>
__device__ __forceinline__ void mainloop(f32x16 (&acc)[2][2], Staged<PREC>& sa, Staged<PREC>& sb, const GemmArgs& g,
                                         const Tile& T, uint32_t* imgA, uint32_t* imgB, int tid, uint32_t voa, uint32_t vob,
                                         f32x4& rowsum, bool want_rowsum) {
  constexpr int IMGS = (PREC == PREC_BF16X3) ? 2 : 1;
  constexpr int IMG_DWORDS = Geo<PREC>::IMG_DWORDS;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int l31 = lane & 31, h = lane >> 5;
  for (int k0 = T.kbeg; k0 < T.kend; k0 += BK) {
    if (ALAY == LAY_MN && want_rowsum) add_rowsum<PREC>(rowsum, sa);
    if (ALAY == LAY_K) store_kmajor<PREC>(sa, imgA, tid); else store_mnmajor<PREC>(sa, imgA, tid);
    if (BLAY == LAY_K) store_kmajor<PREC>(sb, imgB, tid); else store_mnmajor<PREC>(sb, imgB, tid);
    __syncthreads();
    if (k0 + BK < T.kend) fetch<PREC, ALAY, BLAY>(sa, sb, g, T, k0 + BK, tid, voa, vob);   // in flight under the MFMAs below
#pragma unroll
    for (int s = 0; s < Geo<PREC>::NS; ++s) {
      u32x4 fa[2][IMGS], fb[2][IMGS];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < IMGS; ++p) {
          fa[i][p] = frag<PREC, ALAY>(imgA + p * IMG_DWORDS, wm + 32 * i + l31, s, h);
          fb[i][p] = frag<PREC, BLAY>(imgB + p * IMG_DWORDS, wn + 32 * i + l31, s, h);
        }
      if (PREC == PREC_BF16X3) {
        // term-major order: the four accumulators of the wave take turns, so the three dependent MFMAs of one
        // accumulator (lo*hi, hi*lo, hi*hi) are four issues apart instead of back to back
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const bf16x8 a = __builtin_bit_cast(bf16x8, fa[i][term == 0 ? IMGS - 1 : 0]);
              const bf16x8 b = __builtin_bit_cast(bf16x8, fb[j][term == 1 ? IMGS - 1 : 0]);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i][j], 0, 0, 0);
            }
      } else {
        // (`__builtin_bit_cast(float, vec[u])` on a vector ELEMENT folds every u to element 0 on ROCm 7.2: cast the vector)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const f32x4 bf = __builtin_bit_cast(f32x4, fb[j][0]), af = __builtin_bit_cast(f32x4, fa[i][0]);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[u], bf[u], acc[i][j], 0, 0, 0);
            }
      }
    }
    __syncthreads();
  }
}

// epilogue of one tile: accumulator register r of MFMA tile (i, j) is row m = m0 + wm + 32i + (r&3) + 8(r>>2) + 4h, column
// n = n0 + wn + 32j + (lane&31): one store instruction writes two rows of 32 consecutive floats (two full 128-byte
// lines).  Buffer addressing again: the lane's part of the offset is ONE register, the register's part is scalar;
// the descriptors end at the matrix' last element, so rows past M read as 0 (a garbage load could only feed a row that
// is never stored anyway) and their stores are dropped by the hardware's range check -- and, independently of it,
// by an explicit row predicate; the column test n < N is made once per column tile.
template <bool RAGGED_M>
__device__ __forceinline__ void epilogue_impl(const f32x16 (&acc)[2][2], const GemmArgs& g, const Tile& T, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int l31 = lane & 31, h = lane >> 5;
  const int m0 = T.m0, n0 = T.n0;
  float* C; int64_t ldc;
  if (g.slabs) { C = g.C + (int64_t)T.z * g.M * g.N; ldc = g.N; }
  else { C = g.C + (int64_t)T.batch * g.sC; ldc = g.ldc; }
  const bool epi = !g.slabs;
  const bool by_n = epi && g.vec_axis == 0, by_m = epi && g.vec_axis == 1;
  const int rows_left = g.M - m0, cols_left = g.N - n0;                      // >= 1
  const int mleft = rows_left - wm - 4 * h;                                  // rows of this lane's part of the tile inside M
  const __amdgpu_buffer_rsrc_t rsC = make_rsrc(C + (int64_t)m0 * ldc + n0, ((int64_t)(rows_left - 1) * ldc + cols_left) * 4);
  const uint32_t voC = (uint32_t)(((wm + 4 * h) * (uint32_t)ldc + wn + l31) * 4);
  const bool has_res = epi && g.residual != nullptr;
  const float* Rp = has_res ? g.residual + (int64_t)T.batch * g.sR + (int64_t)m0 * g.ldr + n0 : C;
  const __amdgpu_buffer_rsrc_t rsR = make_rsrc(Rp, has_res ? ((int64_t)(rows_left - 1) * g.ldr + cols_left) * 4 : 0);
  const uint32_t voR = (uint32_t)(((wm + 4 * h) * (uint32_t)g.ldr + wn + l31) * 4);
  const bool m_scale = by_m && g.scale != nullptr, m_shift = by_m && g.shift != nullptr;
  const __amdgpu_buffer_rsrc_t rsS = make_rsrc(m_scale ? g.scale + m0 : C, m_scale ? (int64_t)rows_left * 4 : 0);
  const __amdgpu_buffer_rsrc_t rsH = make_rsrc(m_shift ? g.shift + m0 : C, m_shift ? (int64_t)rows_left * 4 : 0);
  const uint32_t voM = (uint32_t)((wm + 4 * h) * 4);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn + 32 * j + l31;
    if (n >= g.N) continue;
    float sn = 1.0f, bn = 0.0f;
    if (by_n && g.scale) sn = g.scale[n];
    if (by_n && g.shift) bn = g.shift[n];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r0 = 0; r0 < 16; r0 += 8) {               // groups of 8 registers: their loads are issued together
        float sc[8], sh[8], rs[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int row = 32 * i + (r & 3) + 8 * ((r0 + r) >> 2);         // uniform part of the row inside the wave's tile
          sc[r] = m_scale ? ld4(rsS, voM, (uint32_t)(row * 4)) : sn;
          sh[r] = m_shift ? ld4(rsH, voM, (uint32_t)(row * 4)) : bn;
          rs[r] = has_res ? ld4(rsR, voR, (uint32_t)((row * (uint32_t)g.ldr + 32 * j) * 4)) : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int row = 32 * i + (r & 3) + 8 * ((r0 + r) >> 2);
          float y = acc[i][j][r0 + r];
          if (epi) {
            y = y * sc[r] + sh[r] + rs[r];
            if (g.relu && !(y > 0.0f)) y = 0.0f;
          }
          // rows >= M (only a tile of the last tile row has any: RAGGED_M): an explicit predicate on top of the
          // descriptor's range check (the row part of the offset is a scalar offset, which the hardware only checks
          // through the gfx9 rule num_records - soffset).  Full tiles keep the bare store burst -- the predicate's
          // exec-mask juggling around each of the 64 stores cost the fused conv3 epilogue 19 % (528 -> 629 us).
          if (!RAGGED_M || row < mleft)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, y), rsC, voC,
                                                    (uint32_t)((row * (uint32_t)ldc + 32 * j) * 4), 0);
        }
      }
    }
  }
}

__device__ __forceinline__ void epilogue(const f32x16 (&acc)[2][2], const GemmArgs& g, const Tile& T, int tid) {
  if (g.M - T.m0 >= BM) epilogue_impl<false>(acc, g, T, tid);       // workgroup-uniform
  else epilogue_impl<true>(acc, g, T, tid);
}

// Persistent workgroups: the grid is (at most) one residency of the chip, and a workgroup walks tiles t = blockIdx,
// blockIdx + grid, ...  Between the k loop of a tile and its epilogue it decodes the NEXT tile and issues that tile's
// first global loads, so the store burst of the epilogue (and the launch / address set-up a fresh workgroup would pay)
// overlaps the latency of the next tile's first operands.
template <int PREC, int ALAY, int BLAY>
__global__ __launch_bounds__(THREADS, 3) void gemm_mfma_kernel(GemmArgs g) {
  constexpr int IMGS = (PREC == PREC_BF16X3) ? 2 : 1;
  __shared__ __attribute__((aligned(16))) uint32_t lds[2 * IMGS * Geo<PREC>::IMG_DWORDS];
  uint32_t* imgA = lds;
  uint32_t* imgB = lds + IMGS * Geo<PREC>::IMG_DWORDS;
  const int tid = threadIdx.x;
  const uint32_t voa = ALAY == LAY_K ? voff_kmajor<PREC>(g.lda, tid) : voff_mnmajor<PREC>(g.lda, tid);
  const uint32_t vob = BLAY == LAY_K ? voff_kmajor<PREC>(g.ldb, tid) : voff_mnmajor<PREC>(g.ldb, tid);
  Staged<PREC> sa, sb;
  int t = blockIdx.x;
  if (t >= g.total) return;
  Tile cur = decode_tile<PREC, ALAY, BLAY>(g, t);
  if (cur.kbeg < cur.kend) fetch<PREC, ALAY, BLAY>(sa, sb, g, cur, cur.kbeg, tid, voa, vob);
  for (;;) {
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    f32x4 rowsum = {0.f, 0.f, 0.f, 0.f};
    const bool want_rowsum = ALAY == LAY_MN && g.a_rowsum != nullptr && cur.n0 == 0;     // workgroup-uniform
    mainloop<PREC, ALAY, BLAY>(acc, sa, sb, g, cur, imgA, imgB, tid, voa, vob, rowsum, want_rowsum);
    if (want_rowsum) {
      // the 8 threads that staged the same row quad (tid & 31, both modes) combine through LDS -- free after the k loop's
      // last barrier -- in a fixed order; rows m0 .. m0 + 127 of slab z
      float* red = reinterpret_cast<float*>(lds);
      const int quad = tid & 31, part = tid >> 5;
      *(f32x4*)(red + (part * 32 + quad) * 4) = rowsum;
      __syncthreads();
      if (tid < BM) {
        float sum = 0.f;
#pragma unroll
        for (int p8 = 0; p8 < 8; ++p8) sum += red[(p8 * 32 + (tid >> 2)) * 4 + (tid & 3)];
        if (cur.m0 + tid < g.M) g.rowsum_ws[(int64_t)cur.z * g.M + cur.m0 + tid] = sum;
      }
      __syncthreads();
    }
    const int tn = t + (int)gridDim.x;
    const bool more = tn < g.total;
    if (more) {        // only the tile NUMBER survives the epilogue (registers): the next tile is decoded twice
      const Tile nxt = decode_tile<PREC, ALAY, BLAY>(g, tn);
      if (nxt.kbeg < nxt.kend) fetch<PREC, ALAY, BLAY>(sa, sb, g, nxt, nxt.kbeg, tid, voa, vob);
    }
    epilogue(acc, g, cur, tid);
    if (!more) break;
    cur = decode_tile<PREC, ALAY, BLAY>(g, tn);
    t = tn;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Wave-specialised form (round 5).  The kernel above gives every wave every job -- global loads, the bf16 split, LDS
// stores, fragment reads, MFMAs, the epilogue -- with two barriers per k-step, and its phases ADD UP instead of
// overlapping (profiles/r04_kbench_gemm_ablate*.log): the workgroups of a CU run in lockstep, and because loads and
// stores share the in-order vmcnt counter the 64 epilogue stores of a tile stand between its successor's second k-step
// and the matrix cores.  Here a 512-thread workgroup (one per CU: a matrix wave and a staging wave on every SIMD) splits
// the roles:
//   waves 4-7  PRODUCERS  global -> registers (two k-steps ahead, two register sets) -> split -> LDS stage (g & 1)
//   waves 0-3  CONSUMERS  fragment reads + MFMAs of stage (g & 1), then the tile's epilogue
// over ONE flat stream of k-steps g = 0, 1, ... that runs across the tiles a workgroup walks, with ONE barrier per
// k-step (LDS-only: `s_waitcnt lgkmcnt(0); s_barrier` -- the prefetch loads and the epilogue stores stay in flight
// across it, which __syncthreads()' vmcnt(0) would drain):
//   barrier g:  producers have filled stage g & 1 with step g  |  consumers have finished step g - 1 (stage (g-1) & 1)
// so after it the consumers compute step g while the producers overwrite stage (g+1) & 1 with step g + 1.  While the
// consumers store a tile, the producers are already staging the next tile's first two k-steps; the consumers' vmcnt
// only ever counts their own stores and is never waited on.
constexpr int WS_THREADS = 512;

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// a position in the flat k-step stream of one workgroup
template <int PREC, int ALAY, int BLAY>
struct Cursor {
  int t;          // tile number (blockIdx + n * grid); >= total: exhausted
  int k0;
  Tile T;
  __device__ __forceinline__ void start(const GemmArgs& g, int t0) {
    t = t0;
    while (t < g.total) {
      T = decode_tile<PREC, ALAY, BLAY>(g, t);
      if (T.kbeg < T.kend) break;
      t += (int)gridDim.x;
    }
    k0 = (t < g.total) ? T.kbeg : 0;
  }
  __device__ __forceinline__ bool valid(const GemmArgs& g) const { return t < g.total; }
  __device__ __forceinline__ void advance(const GemmArgs& g) {
    k0 += BK;
    if (k0 >= T.kend) start(g, t + (int)gridDim.x);
  }
};

template <int PREC, int ALAY, int BLAY>
__global__ __launch_bounds__(WS_THREADS, 2) void gemm_mfma_ws_kernel(GemmArgs g) {
  constexpr int IMGS = (PREC == PREC_BF16X3) ? 2 : 1;
  constexpr int IMG_DWORDS = Geo<PREC>::IMG_DWORDS;
  constexpr int STAGE_DWORDS = 2 * IMGS * IMG_DWORDS;
  __shared__ __attribute__((aligned(16))) uint32_t lds[2 * STAGE_DWORDS];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if ((int)blockIdx.x >= g.total) return;

  if (wave >= 4) {
    // ------------------------------------------------ producers ------------------------------------------------
    const int ptid = tid - 256;
    const uint32_t voa = ALAY == LAY_K ? voff_kmajor<PREC>(g.lda, ptid) : voff_mnmajor<PREC>(g.lda, ptid);
    const uint32_t vob = BLAY == LAY_K ? voff_kmajor<PREC>(g.ldb, ptid) : voff_mnmajor<PREC>(g.ldb, ptid);
    Staged<PREC> sa0, sb0, sa1, sb1;
    Cursor<PREC, ALAY, BLAY> rd;                       // next k-step to REQUEST
    rd.start(g, (int)blockIdx.x);
    // how many k-steps this workgroup owns in total (= barriers both roles execute)
    int steps = 0;
    for (int t = (int)blockIdx.x; t < g.total; t += (int)gridDim.x) {
      const Tile T = decode_tile<PREC, ALAY, BLAY>(g, t);
      steps += (T.kend > T.kbeg) ? (T.kend - T.kbeg + BK - 1) / BK : 0;
    }
    if (rd.valid(g)) { fetch<PREC, ALAY, BLAY>(sa0, sb0, g, rd.T, rd.k0, ptid, voa, vob); rd.advance(g); }
    if (rd.valid(g)) { fetch<PREC, ALAY, BLAY>(sa1, sb1, g, rd.T, rd.k0, ptid, voa, vob); rd.advance(g); }
    for (int gs = 0; gs < steps; gs += 2) {
      {
        uint32_t* imgA = lds;
        uint32_t* imgB = lds + IMGS * IMG_DWORDS;
        if (ALAY == LAY_K) store_kmajor<PREC>(sa0, imgA, ptid); else store_mnmajor<PREC>(sa0, imgA, ptid);
        if (BLAY == LAY_K) store_kmajor<PREC>(sb0, imgB, ptid); else store_mnmajor<PREC>(sb0, imgB, ptid);
        if (rd.valid(g)) { fetch<PREC, ALAY, BLAY>(sa0, sb0, g, rd.T, rd.k0, ptid, voa, vob); rd.advance(g); }
        lds_barrier();
      }
      if (gs + 1 < steps) {
        uint32_t* imgA = lds + STAGE_DWORDS;
        uint32_t* imgB = lds + STAGE_DWORDS + IMGS * IMG_DWORDS;
        if (ALAY == LAY_K) store_kmajor<PREC>(sa1, imgA, ptid); else store_mnmajor<PREC>(sa1, imgA, ptid);
        if (BLAY == LAY_K) store_kmajor<PREC>(sb1, imgB, ptid); else store_mnmajor<PREC>(sb1, imgB, ptid);
        if (rd.valid(g)) { fetch<PREC, ALAY, BLAY>(sa1, sb1, g, rd.T, rd.k0, ptid, voa, vob); rd.advance(g); }
        lds_barrier();
      }
    }
    return;
  }

  // ---------------------------------------------------- consumers ----------------------------------------------------
  const int lane = tid & 63;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int l31 = lane & 31, h = lane >> 5;
  int gs = 0;                                            // flat k-step counter: stage = gs & 1
  for (int t = (int)blockIdx.x; t < g.total; t += (int)gridDim.x) {
    const Tile T = decode_tile<PREC, ALAY, BLAY>(g, t);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    for (int k0 = T.kbeg; k0 < T.kend; k0 += BK, ++gs) {
      lds_barrier();
      const uint32_t* imgA = lds + (gs & 1) * STAGE_DWORDS;
      const uint32_t* imgB = imgA + IMGS * IMG_DWORDS;
#pragma unroll
      for (int s = 0; s < Geo<PREC>::NS; ++s) {
        u32x4 fa[2][IMGS], fb[2][IMGS];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int p = 0; p < IMGS; ++p) {
            fa[i][p] = frag<PREC, ALAY>(imgA + p * IMG_DWORDS, wm + 32 * i + l31, s, h);
            fb[i][p] = frag<PREC, BLAY>(imgB + p * IMG_DWORDS, wn + 32 * i + l31, s, h);
          }
        if (PREC == PREC_BF16X3) {
#pragma unroll
          for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                const bf16x8 a = __builtin_bit_cast(bf16x8, fa[i][term == 0 ? IMGS - 1 : 0]);
                const bf16x8 b = __builtin_bit_cast(bf16x8, fb[j][term == 1 ? IMGS - 1 : 0]);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i][j], 0, 0, 0);
              }
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                const f32x4 bf = __builtin_bit_cast(f32x4, fb[j][0]), af = __builtin_bit_cast(f32x4, fa[i][0]);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[u], bf[u], acc[i][j], 0, 0, 0);
              }
        }
      }
    }
    epilogue(acc, g, T, tid);
  }
}

// C[m, n] = epilogue( sum_z slab[z][m, n] ).  64 float4 columns x 4 z-groups per workgroup: a thread sums every fourth
// slab of its four elements (16-byte loads, Z/4 deep instead of Z), the four partial sums are combined through LDS in
// z-group order -- the order of the additions is fixed, the result is deterministic.
__global__ __launch_bounds__(256) void gemm_slab_reduce_kernel(const float* __restrict__ ws, int Z, GemmArgs g) {
  __shared__ f32x4 part[4][64];
  if (g.a_rowsum != nullptr) {
    // the bias-gradient rows: a wave per row, rows dealt over the whole grid; lane l adds the partial rows z = l, l + 64,
    // ... in order and the 64 lane sums are combined by a fixed butterfly -- the order of the additions never changes
    const int lane = threadIdx.x & 63;
    for (int m = (int)blockIdx.x * 4 + (threadIdx.x >> 6); m < g.M; m += (int)gridDim.x * 4) {
      float a = 0.f;
      for (int z = lane; z < Z; z += 64) a += g.rowsum_ws[(int64_t)z * g.M + m];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
      if (lane == 0) g.a_rowsum[m] = a;
    }
  }
  const int64_t total = (int64_t)g.M * g.N;                  // a multiple of 4 is NOT required: the tail is scalar
  const int64_t quads = total >> 2;
  const int col = threadIdx.x & 63, zg = threadIdx.x >> 6;
  for (int64_t q0 = (int64_t)blockIdx.x * 64; q0 < quads + 1; q0 += (int64_t)gridDim.x * 64) {
    const int64_t q = q0 + col;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (q < quads) {
      for (int z = zg; z < Z; z += 4) a += *(const f32x4u*)(ws + (int64_t)z * total + 4 * q);
    } else if (q == quads) {                                 // the last 0..3 elements
      for (int z = zg; z < Z; z += 4)
        for (int e = 0; e < (int)(total & 3); ++e) a[e] += ws[(int64_t)z * total + 4 * q + e];
    }
    part[zg][col] = a;
    __syncthreads();
    if (zg == 0 && q <= quads) {
      const f32x4 sum = ((part[0][col] + part[1][col]) + part[2][col]) + part[3][col];
      const int n_el = q < quads ? 4 : (int)(total & 3);
      for (int e = 0; e < n_el; ++e) {
        const int64_t idx = 4 * q + e;
        const int m = (int)(idx / g.N), n = (int)(idx - (int64_t)m * g.N);
        const int vi = g.vec_axis == 1 ? m : n;
        float y = sum[e] * (g.scale ? g.scale[vi] : 1.0f) + (g.shift ? g.shift[vi] : 0.0f);
        if (g.residual) y += g.residual[(int64_t)m * g.ldr + n];
        g.C[(int64_t)m * g.ldc + n] = (g.relu && !(y > 0.0f)) ? 0.0f : y;
      }
    }
    __syncthreads();
  }
}

template <int PREC>
void launch_ws(const GemmArgs& g, int a_layout, int b_layout, dim3 grid, hipStream_t st) {
  if (a_layout == LAY_K && b_layout == LAY_K) hipLaunchKernelGGL((gemm_mfma_ws_kernel<PREC, LAY_K, LAY_K>), grid, dim3(WS_THREADS), 0, st, g);
  else if (a_layout == LAY_K) hipLaunchKernelGGL((gemm_mfma_ws_kernel<PREC, LAY_K, LAY_MN>), grid, dim3(WS_THREADS), 0, st, g);
  else if (b_layout == LAY_K) hipLaunchKernelGGL((gemm_mfma_ws_kernel<PREC, LAY_MN, LAY_K>), grid, dim3(WS_THREADS), 0, st, g);
  else hipLaunchKernelGGL((gemm_mfma_ws_kernel<PREC, LAY_MN, LAY_MN>), grid, dim3(WS_THREADS), 0, st, g);
}

// 0 (default): every wave does every job (round 4); 1 / 2: wave-specialised producer / consumer workgroups, one /
// two of them per CU.  Measured (profiles/r05_kbench_gemm_variants.log): two wave-specialised workgroups per CU tie
// with the default within +-10 % on every shape, one per CU is 15 - 60 % slower -- the kernel is bound by the
// per-k-step LDS + split work that both forms share, not by the overlap structure.  Kept as the A/B it is.
int g_gemm_variant = 0;

template <int PREC>
void launch(const GemmArgs& g, int a_layout, int b_layout, dim3 grid, hipStream_t st) {
  if (a_layout == LAY_K && b_layout == LAY_K) hipLaunchKernelGGL((gemm_mfma_kernel<PREC, LAY_K, LAY_K>), grid, dim3(THREADS), 0, st, g);
  else if (a_layout == LAY_K) hipLaunchKernelGGL((gemm_mfma_kernel<PREC, LAY_K, LAY_MN>), grid, dim3(THREADS), 0, st, g);
  else if (b_layout == LAY_K) hipLaunchKernelGGL((gemm_mfma_kernel<PREC, LAY_MN, LAY_K>), grid, dim3(THREADS), 0, st, g);
  else hipLaunchKernelGGL((gemm_mfma_kernel<PREC, LAY_MN, LAY_MN>), grid, dim3(THREADS), 0, st, g);
}

int num_cus() {                       // of the CURRENT device (cached per device id)
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    hipDeviceProp_t prop;
    cached[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                      ? prop.multiProcessorCount : 256;
  }
  return cached[dev];
}

// how many k-splits a reduced product gets: enough workgroups for ~2 per CU, k chunks of at least 4 k-steps
int pick_splits(int M, int N, int K, int batch, int bk) {
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN) * batch;
  int s = (512 + tiles - 1) / tiles;
  const int max_s = (K + 4 * bk - 1) / (4 * bk);
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  if (s > 256) s = 256;
  return s;
}

}  // namespace

extern "C" {

int vidar_gemm_set_variant(int variant) {
  const int prev = g_gemm_variant;
  g_gemm_variant = (variant < 0 || variant > 2) ? 0 : variant;
  return prev;
}

int vidar_gemm_splits(int M, int N, int K, int batch, int precision, int reduce) {
  if (!reduce) return 1;
  return pick_splits(M, N, K, batch, BK);
}

size_t vidar_gemm_workspace_bytes(int M, int N, int K, int batch, int precision, int reduce) {
  if (!reduce) return 0;
  const int s = vidar_gemm_splits(M, N, K, batch, precision, reduce);
  // reduce == 1: the product alone -- Z = batch * splits slabs of [M, N], none when there is a single slab (it is
  // written straight to C);  reduce == 2: the product WITH the row sums of A (a_rowsum) -- the slabs plus Z partial
  // rows of [M], also for a single slab
  const size_t Z = (size_t)batch * s;
  if (reduce == 1) return Z > 1 ? Z * (size_t)M * N * sizeof(float) : 0;
  return Z * ((size_t)M * N + M) * sizeof(float);
}

int vidar_gemm_f32(const float* A, int64_t lda, int a_layout, const float* B, int64_t ldb, int b_layout, float* C,
                   int64_t ldc, int M, int N, int K, int batch, int64_t strideA, int64_t strideB, int64_t strideC,
                   const float* scale, const float* shift, int vec_axis, const float* residual, int64_t ldr,
                   int64_t strideR, int relu, int precision, int reduce, float* a_rowsum, void* workspace,
                   size_t workspace_bytes, void* stream) {
  VIDAR_ENTER();
  if (a_rowsum != nullptr && !(reduce && a_layout == LAY_MN)) return VIDAR_ERR_BAD_ARG;
  if (A == nullptr || B == nullptr || C == nullptr || M <= 0 || N <= 0 || K <= 0 || batch <= 0) return VIDAR_ERR_BAD_ARG;
  if ((a_layout != LAY_K && a_layout != LAY_MN) || (b_layout != LAY_K && b_layout != LAY_MN)) return VIDAR_ERR_BAD_ARG;
  if (precision != PREC_F32 && precision != PREC_BF16X3) return VIDAR_ERR_BAD_ARG;
  if (vec_axis != 0 && vec_axis != 1) return VIDAR_ERR_BAD_ARG;
  if (lda < (a_layout == LAY_K ? K : M) || ldb < (b_layout == LAY_K ? K : N) || ldc < N) return VIDAR_ERR_BAD_ARG;
  if (residual != nullptr && (ldr < N || ldr >= (1 << 22))) return VIDAR_ERR_BAD_ARG;
  if (ldc >= (1 << 22)) return VIDAR_ERR_BAD_ARG;
  // 31-bit byte offsets inside a workgroup's view of an operand: 128 rows x ld (K-major), k range x ld (MN-major)
  if (lda >= (1 << 22) || ldb >= (1 << 22)) return VIDAR_ERR_BAD_ARG;
  if ((a_layout == LAY_MN && (int64_t)K * lda >= (1LL << 29)) || (b_layout == LAY_MN && (int64_t)K * ldb >= (1LL << 29)))
    return VIDAR_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int bk = BK;
  GemmArgs g;
  g.A = A; g.B = B; g.C = C;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.sA = strideA; g.sB = strideB; g.sC = strideC;
  g.M = M; g.N = N; g.K = K;
  g.scale = scale; g.shift = shift; g.vec_axis = vec_axis; g.residual = residual; g.ldr = ldr; g.sR = strideR;
  g.relu = relu;
  g.tiles_m = (M + BM - 1) / BM; g.tiles_n = (N + BN - 1) / BN;
  g.m_fastest = M < N;
  g.splits = reduce ? pick_splits(M, N, K, batch, bk) : 1;
  g.k_chunk = ((K + g.splits - 1) / g.splits + bk - 1) / bk * bk;
  const int Z = batch * g.splits;
  g.slabs = (reduce && (Z > 1 || a_rowsum != nullptr)) ? 1 : 0;
  g.a_rowsum = a_rowsum;
  g.rowsum_ws = nullptr;
  if ((int64_t)g.tiles_m * g.tiles_n * Z > 0x3fffffff) return VIDAR_ERR_BAD_ARG;
  g.total = g.tiles_m * g.tiles_n * Z;
  // a reduced product writes slabs and the reduction kernel applies the epilogue on [M, N] of ONE output: batch and
  // residual strides have no meaning there
  if (reduce && (strideC != 0 || strideR != 0) && batch > 1) return VIDAR_ERR_BAD_ARG;
  GemmArgs k = g;
  if (g.slabs) {
    const size_t need = (size_t)Z * ((size_t)M * N + (a_rowsum ? M : 0)) * sizeof(float);
    if (workspace == nullptr || workspace_bytes < need) return VIDAR_ERR_BAD_ARG;
    k.C = (float*)workspace;
    g.rowsum_ws = a_rowsum ? (float*)workspace + (size_t)Z * M * N : nullptr;
    k.rowsum_ws = g.rowsum_ws;
  }
  k.total = g.tiles_m * g.tiles_n * Z;
  // one residency of the chip: 3 workgroups per CU (launch bounds: 3 waves per SIMD), a multiple of 8 so that t & 7
  // stays the workgroup's XCD for every tile it walks.  (Measured against one workgroup per tile and against fetching
  // the next tile after the epilogue: within noise of each other on MI355X, profiles/r04_kbench_gemm_*.)
  if (g_gemm_variant >= 1 && a_rowsum == nullptr) {
    // one (two) 512-thread workgroup(s) per CU (a multiple of 8 keeps t & 7 = the XCD for every tile a workgroup walks)
    const int resident = num_cus() * g_gemm_variant / 8 * 8;
    dim3 grid(k.total > resident ? resident : k.total);
    if (precision == PREC_BF16X3) launch_ws<PREC_BF16X3>(k, a_layout, b_layout, grid, st);
    else launch_ws<PREC_F32>(k, a_layout, b_layout, grid, st);
  } else {
    const int resident = num_cus() * 3 / 8 * 8;
    dim3 grid(k.total > resident ? resident : k.total);
    if (precision == PREC_BF16X3) launch<PREC_BF16X3>(k, a_layout, b_layout, grid, st);
    else launch<PREC_F32>(k, a_layout, b_layout, grid, st);
  }
  if (g.slabs) {
    const int64_t total = (int64_t)M * N;
    int blocks = (int)((total / 4 + 1 + 63) / 64);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gemm_slab_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)workspace, Z, g);
  }
  return vidar_last_error();
}

}  // extern "C"
