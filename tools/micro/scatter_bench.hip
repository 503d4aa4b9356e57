// micro-benchmarks that size the "bin by destination tile, accumulate in LDS" design for msda_bwd:
//   (a) ds_add_f32 rate: half-waves add 32-float lines into a random line of an LDS window
//   (b) scattered 4-byte / 16-byte plain stores (the record scatter of a counting sort)
//   (c) random 128-byte line gathers (grad_out lines fetched per binned sample)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/scatter_bench.hip -o /tmp/sb && /tmp/sb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int WLINES>
__global__ __launch_bounds__(256) void lds_add(const int* lines, float* out, int per_group) {
  __shared__ float win[WLINES * 32];
  for (int i = threadIdx.x; i < WLINES * 32; i += 256) win[i] = 0.f;
  __syncthreads();
  const int group = (blockIdx.x * 256 + threadIdx.x) / 32, sub = threadIdx.x % 32;
  const int* my = lines + (long)group * per_group;
  for (int i = 0; i < per_group; ++i) {
    const int l = my[i] % WLINES;
    atomicAdd(&win[l * 32 + sub], 1.0f);
  }
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < WLINES * 32; i += 256) s += win[i];
  if (s == -1.f) out[0] = s;
}

__global__ __launch_bounds__(256) void scatter4(const int* dst, int* out, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[dst[i]] = (int)i;
}
__global__ __launch_bounds__(256) void scatter16(const int* dst, int4* out, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[dst[i]] = make_int4((int)i, 1, 2, 3);
}
// 8 lanes x float4 fetch one random 128-B line each, sum
__global__ __launch_bounds__(256) void gather128(const int* src, const float4* buf, float4* out, long n, int per) {
  const long g = ((long)blockIdx.x * 256 + threadIdx.x) / 8; const int sub = threadIdx.x % 8;
  float4 a = make_float4(0, 0, 0, 0);
  for (int i = 0; i < per; ++i) {
    const long k = g * per + i;
    if (k >= n) break;
    const float4 v = buf[(long)src[k] * 8 + sub];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  out[g * 8 + sub] = a;
}

template <class F> static float timeit(F f) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); f(); f(); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 3;
}

int main() {
  const long n = 15360000;   // SCA samples per backward
  std::vector<int> h(n);
  srand(2);
  for (long i = 0; i < n; ++i) h[i] = rand();
  int* d_rand; CK(hipMalloc(&d_rand, n * 4)); CK(hipMemcpy(d_rand, h.data(), n * 4, hipMemcpyHostToDevice));
  float* d_out; CK(hipMalloc(&d_out, 1 << 20));
  {
    const int per = 64; const long groups = n * 4 / per;   // 61.4 M line adds
    const int blocks = (int)(groups * 32 / 256);
    std::vector<int> h4(n * 4);
    for (long i = 0; i < n * 4; ++i) h4[i] = rand();
    int* d4; CK(hipMalloc(&d4, n * 16)); CK(hipMemcpy(d4, h4.data(), n * 16, hipMemcpyHostToDevice));
    float ms = timeit([&] { hipLaunchKernelGGL(lds_add<289>, dim3(blocks), dim3(256), 0, 0, d4, d_out, per); });
    printf("LDS ds_add_f32, 289-line window (37 KB): %.3f ms  %.1f G line-adds/s\n", ms, n * 4 / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(lds_add<81>, dim3(blocks), dim3(256), 0, 0, d4, d_out, per); });
    printf("LDS ds_add_f32, 81-line window (10 KB):  %.3f ms  %.1f G line-adds/s\n", ms, n * 4 / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(lds_add<8>, dim3(blocks), dim3(256), 0, 0, d4, d_out, per); });
    printf("LDS ds_add_f32, 8-line window (hot):     %.3f ms  %.1f G line-adds/s\n", ms, n * 4 / ms / 1e6);
    CK(hipFree(d4));
  }
  {
    // destinations: a permutation-like random map (every slot written ~once), and a "binned" map where
    // consecutive threads mostly write consecutive slots of a few hundred bins
    std::vector<int> perm(n);
    for (long i = 0; i < n; ++i) perm[i] = (int)i;
    for (long i = n - 1; i > 0; --i) { long j = ((long)rand() * 32768 + rand()) % (i + 1); std::swap(perm[i], perm[j]); }
    int* d_dst; CK(hipMalloc(&d_dst, n * 4)); CK(hipMemcpy(d_dst, perm.data(), n * 4, hipMemcpyHostToDevice));
    int4* d_o; CK(hipMalloc(&d_o, n * 16));
    const int blocks = (int)((n + 255) / 256);
    float ms = timeit([&] { hipLaunchKernelGGL(scatter4, dim3(blocks), dim3(256), 0, 0, d_dst, (int*)d_o, n); });
    printf("scattered 4-B stores, random permutation:  %.3f ms  %.1f G stores/s\n", ms, n / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(scatter16, dim3(blocks), dim3(256), 0, 0, d_dst, d_o, n); });
    printf("scattered 16-B stores, random permutation: %.3f ms  %.1f G stores/s\n", ms, n / ms / 1e6);
    // binned: 3000 bins, thread i goes to bin (hash of i/8) with a running slot -> runs of 8 consecutive slots
    const int nb = 3000; std::vector<long> cur(nb);
    for (int b = 0; b < nb; ++b) cur[b] = (long)b * (n / nb);
    for (long i = 0; i < n; i += 8) { int b = rand() % nb; for (int k = 0; k < 8 && i + k < n; ++k) { long s = cur[b]++; perm[i + k] = (int)(s % n); } }
    CK(hipMemcpy(d_dst, perm.data(), n * 4, hipMemcpyHostToDevice));
    ms = timeit([&] { hipLaunchKernelGGL(scatter4, dim3(blocks), dim3(256), 0, 0, d_dst, (int*)d_o, n); });
    printf("scattered 4-B stores, runs of 8 per bin:   %.3f ms  %.1f G stores/s\n", ms, n / ms / 1e6);
    CK(hipFree(d_dst)); CK(hipFree(d_o));
  }
  {
    const long nlines = 480000;   // grad_out lines: 6 x 10^4 queries x 8 heads
    float4* buf; CK(hipMalloc(&buf, nlines * 128)); CK(hipMemset(buf, 0, nlines * 128));
    std::vector<int> src(n);
    for (long i = 0; i < n; ++i) src[i] = (int)((((long)rand() << 15) ^ rand()) % nlines);
    CK(hipMemcpy(d_rand, src.data(), n * 4, hipMemcpyHostToDevice));
    const int per = 32; const long groups = n / per; const int blocks = (int)(groups * 8 / 256);
    float4* o; CK(hipMalloc(&o, groups * 128));
    float ms = timeit([&] { hipLaunchKernelGGL(gather128, dim3(blocks), dim3(256), 0, 0, d_rand, buf, o, n, per); });
    printf("random 128-B line gathers from 61 MB:      %.3f ms  %.1f G lines/s  %.0f GB/s\n", ms, n / ms / 1e6, n * 128 / ms / 1e6);
  }
  return 0;
}
