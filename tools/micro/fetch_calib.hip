// FETCH_SIZE / WRITE_SIZE calibration on known byte counts (MI355X_MICROARCH.md, HBM section: gfx950 reports
// half the bytes of a wide coalesced streaming read; other patterns must be calibrated).  Three kernels over a
// 1 GiB buffer (4x the 256 MiB Infinity Cache), each touching every 128-byte line exactly once:
//   stream16  : coalesced 16 B / lane streaming read                      (the guide's reference pattern)
//   stream4   : coalesced 4 B / lane streaming read
//   gather128 : the MSDA access pattern -- 8 lanes x float4 fetch one RANDOM 128-byte line (a permutation)
//   write128  : the same permutation, written (WRITE_SIZE calibration)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/fetch_calib.hip -o tools/micro/_bin/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- tools/micro/_bin/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void calib_stream16(const float4* buf, float* out, long n4) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  float a = 0.f;
  for (; i < n4; i += (long)gridDim.x * 256) { const float4 v = buf[i]; a += v.x + v.y + v.z + v.w; }
  if (a == -1.f) out[0] = a;
}
__global__ __launch_bounds__(256) void calib_stream4(const float* buf, float* out, long n) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  float a = 0.f;
  for (; i < n; i += (long)gridDim.x * 256) a += buf[i];
  if (a == -1.f) out[0] = a;
}
__global__ __launch_bounds__(256) void calib_gather128(const int* perm, const float4* buf, float* out, long nlines) {
  const long g = ((long)blockIdx.x * 256 + threadIdx.x) / 8; const int sub = threadIdx.x % 8;
  float a = 0.f;
  for (long k = g; k < nlines; k += (long)gridDim.x * 32) { const float4 v = buf[(long)perm[k] * 8 + sub]; a += v.x + v.y + v.z + v.w; }
  if (a == -1.f) out[0] = a;
}
__global__ __launch_bounds__(256) void calib_write128(const int* perm, float4* buf, long nlines) {
  const long g = ((long)blockIdx.x * 256 + threadIdx.x) / 8; const int sub = threadIdx.x % 8;
  for (long k = g; k < nlines; k += (long)gridDim.x * 32) buf[(long)perm[k] * 8 + sub] = make_float4(1.f, 2.f, 3.f, (float)k);
}

int main() {
  const long bytes = 1l << 30, nlines = bytes / 128;
  float4* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
  float* out; CK(hipMalloc(&out, 64));
  std::vector<int> perm(nlines);
  for (long i = 0; i < nlines; ++i) perm[i] = (int)i;
  srand(3);
  for (long i = nlines - 1; i > 0; --i) { long j = (((long)rand() << 16) ^ rand()) % (i + 1); std::swap(perm[i], perm[j]); }
  int* dperm; CK(hipMalloc(&dperm, nlines * 4)); CK(hipMemcpy(dperm, perm.data(), nlines * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto f, double rd, double wr) {
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("{\"kernel\": \"%s\", \"ms\": %.4f, \"known_read_bytes\": %.0f, \"known_write_bytes\": %.0f, \"GBps\": %.1f}\n", name, ms, rd, wr, (rd + wr) / ms / 1e6);
  };
  run("calib_stream16", [&] { hipLaunchKernelGGL(calib_stream16, dim3(8192), dim3(256), 0, 0, buf, out, bytes / 16); }, (double)bytes, 0);
  run("calib_stream4", [&] { hipLaunchKernelGGL(calib_stream4, dim3(8192), dim3(256), 0, 0, (const float*)buf, out, bytes / 4); }, (double)bytes, 0);
  run("calib_gather128", [&] { hipLaunchKernelGGL(calib_gather128, dim3(8192), dim3(256), 0, 0, dperm, buf, out, nlines); }, (double)bytes + nlines * 4.0, 0);
  run("calib_write128", [&] { hipLaunchKernelGGL(calib_write128, dim3(8192), dim3(256), 0, 0, dperm, buf, nlines); }, nlines * 4.0, (double)bytes);
  return 0;
}
