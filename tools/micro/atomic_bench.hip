// micro-benchmark: fp32 global atomic-add throughput vs lane->address mapping (MSDA backward design).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// lines: random line ids; each "corner add" = 32 floats (128 B line)
// MODE 0: 8 lanes x float4 (4 atomic instrs, stride 16 B)  -- 8 lines per wave-instr
// MODE 1: 32 lanes x 1 float  (1 atomic instr per line)    -- 2 lines per wave-instr
// MODE 2: 16 lanes x 2 floats (2 instrs)                   -- 4 lines per wave-instr
template <int MODE>
__global__ __launch_bounds__(256) void k(float* buf, const int* lines, int n_lines_total, int per_group) {
  constexpr int LPL = MODE == 0 ? 8 : (MODE == 1 ? 32 : 16);   // lanes per line
  constexpr int CH = 32 / LPL;
  const int tid = blockIdx.x * 256 + threadIdx.x;
  const int group = tid / LPL, sub = tid % LPL;
  for (int i = 0; i < per_group; ++i) {
    const long idx = (long)group * per_group + i;
    if (idx >= n_lines_total) return;
    float* p = buf + (long)lines[idx] * 32 + sub * CH;
#pragma unroll
    for (int c = 0; c < CH; ++c) unsafeAtomicAdd(p + c, 1.0f);
  }
}

int main() {
  const long n_line_slots = 1480000;           // SCA: 6 cams x 30825 px x 8 heads
  const long n_adds = 61440000;                // SCA: corner adds per backward
  float* buf; int* lines;
  CK(hipMalloc(&buf, n_line_slots * 128));
  CK(hipMemset(buf, 0, n_line_slots * 128));
  std::vector<int> h(n_adds);
  srand(1);
  for (int pass = 0; pass < 2; ++pass) {
    // pass 0: uniformly random lines; pass 1: locally clustered (consecutive adds hit nearby lines)
    for (long i = 0; i < n_adds; ++i) {
      if (pass == 0) h[i] = (int)(((long)rand() * 32768 + rand()) % n_line_slots);
      else h[i] = (int)(((i / 64) * 37 + rand() % 64) % n_line_slots);
    }
    CK(hipMalloc(&lines, n_adds * 4));
    CK(hipMemcpy(lines, h.data(), n_adds * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
      const int lpl = mode == 0 ? 8 : (mode == 1 ? 32 : 16);
      const int per_group = 32;
      const long groups = (n_adds + per_group - 1) / per_group;
      const long threads = groups * lpl;
      const int blocks = (int)((threads + 255) / 256);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, buf, lines, (int)n_adds, per_group);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, buf, lines, (int)n_adds, per_group);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, buf, lines, (int)n_adds, per_group);
        hipEventRecord(e1); CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep == 1) printf("pass %d mode %d (lanes/line %2d): %.3f ms  %.1f G dword-atomics/s  %.2f G lines/s\n",
                             pass, mode, lpl, ms, n_adds * 32 / ms / 1e6, n_adds / ms / 1e6);
      }
    }
    CK(hipFree(lines));
  }
  return 0;
}
