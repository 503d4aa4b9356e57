// micro-benchmark for the next design step of every atomic-bound kernel (msda_bwd, dcn_col2im,
// lr_*_bwd, ray_*_bwd): fp32 atomics to ONE buffer at agent scope run at ~10-14 G (instruction x line)
// requests/s on MI355X -- the rate of the memory-side atomic path that keeps the 8 XCDs coherent.
// Question: do narrower-scope atomics to XCD-PRIVATE copies of the accumulator (copy = XCC_ID of the
// issuing wave, so only one XCD's L2 ever owns a given line) retire in the L2 at a higher rate, and
// are the sums right after the kernel ends?  (8 copies of grad_value are 1.5 GB -- nothing on a 288 GB
// part -- and the final 8-way reduction is one 0.3 ms streaming pass.)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/atomic_scope_bench.hip -o /tmp/asb && /tmp/asb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ int xcc_id() {
  // HW_REG_XCC_ID (id 20), field XCC_ID = bits [3:0]:  imm = (size-1) << 11 | offset << 6 | id
  return (int)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);
}

// SCOPE: 0 agent (unsafeAtomicAdd), 1 workgroup, 2 wavefront.  PRIVATE: accumulate into copy xcc_id().
template <int SCOPE, bool PRIVATE>
__global__ __launch_bounds__(256) void k(float* buf, long copy_stride, const int* lines, long n_adds, int per_group,
                                         int* xcc_hist) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  const int group = tid / 32, sub = tid % 32;                 // 32 lanes = one 128-byte line per request
  const int xcc = xcc_id();
  if (threadIdx.x == 0) atomicAdd(xcc_hist + (xcc & 15), 1);
  float* base = buf + (PRIVATE ? (long)(xcc & 7) * copy_stride : 0);
  for (int i = 0; i < per_group; ++i) {
    const long idx = (long)group * per_group + i;
    if (idx >= n_adds) return;
    float* p = base + (long)lines[idx] * 32 + sub;
    if (SCOPE == 0) unsafeAtomicAdd(p, 1.0f);
    else if (SCOPE == 1) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  }
}

__global__ void reduce8(const float* buf, long copy_stride, float* out, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int c = 0; c < 8; ++c) s += buf[c * copy_stride + i];
  out[i] = s;
}

template <int SCOPE, bool PRIVATE>
static void run(const char* name, float* buf, long copy_stride, float* red, const int* lines, long n_adds,
                const std::vector<int>& expect_per_line, int* xcc_hist) {
  const int per_group = 32;
  const long groups = (n_adds + per_group - 1) / per_group;
  const int blocks = (int)((groups * 32 + 255) / 256);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipMemset(buf, 0, copy_stride * 4 * 8));
    CK(hipMemset(xcc_hist, 0, 64));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<SCOPE, PRIVATE>), dim3(blocks), dim3(256), 0, 0, buf, copy_stride, lines, n_adds, per_group, xcc_hist);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  float rms = 0;
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(reduce8, dim3((int)((copy_stride + 255) / 256)), dim3(256), 0, 0, buf, copy_stride, red, copy_stride);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&rms, e0, e1));
  std::vector<float> h(copy_stride);
  CK(hipMemcpy(h.data(), red, copy_stride * 4, hipMemcpyDeviceToHost));
  long bad = 0;
  for (long l = 0; l < copy_stride / 32; ++l)
    for (int c = 0; c < 32; ++c) bad += (h[l * 32 + c] != (float)expect_per_line[l]);
  int hist[16];
  CK(hipMemcpy(hist, xcc_hist, 64, hipMemcpyDeviceToHost));
  printf("%-44s %8.3f ms  %6.2f G requests/s   8-copy reduce %.3f ms   wrong elements %ld   blocks per XCC:", name, ms,
         n_adds / ms / 1e6, rms, bad);
  for (int i = 0; i < 8; ++i) printf(" %d", hist[i]);
  printf("\n");
}

int main() {
  const long n_lines = 1480000;                // SCA: 6 cams x 30825 px x 8 heads
  const long n_adds = 61440000;                // SCA: corner requests per backward
  const long copy_stride = n_lines * 32;       // floats per copy
  float *buf, *red; int *lines, *xcc_hist;
  CK(hipMalloc(&buf, copy_stride * 4 * 8));
  CK(hipMalloc(&red, copy_stride * 4));
  CK(hipMalloc(&xcc_hist, 64));
  std::vector<int> h(n_adds), expect(n_lines, 0);
  srand(1);
  for (long i = 0; i < n_adds; ++i) { h[i] = (int)(((i / 64) * 37 + rand() % 64) % n_lines); expect[h[i]]++; }
  CK(hipMalloc(&lines, n_adds * 4));
  CK(hipMemcpy(lines, h.data(), n_adds * 4, hipMemcpyHostToDevice));
  run<0, false>("agent scope, one buffer (today)", buf, copy_stride, red, lines, n_adds, expect, xcc_hist);
  run<0, true>("agent scope, XCD-private copies", buf, copy_stride, red, lines, n_adds, expect, xcc_hist);
  run<1, true>("workgroup scope, XCD-private copies", buf, copy_stride, red, lines, n_adds, expect, xcc_hist);
  run<2, true>("wavefront scope, XCD-private copies", buf, copy_stride, red, lines, n_adds, expect, xcc_hist);
  run<1, false>("workgroup scope, one buffer (expected WRONG)", buf, copy_stride, red, lines, n_adds, expect, xcc_hist);
  return 0;
}
