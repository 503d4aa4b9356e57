/* Host stand-in for <cuda_runtime.h>, used ONLY by oracle/build_ref.py to compile the reference's
 * .cu kernels for the CPU (test infrastructure; never part of the product).  Our own code. */
#pragma once
#include <cassert>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_ { unsigned x, y, z; };
struct int3 { int x, y, z; };
static inline int3 make_int3(int x, int y, int z) { return int3{x, y, z}; }

extern thread_local uint3_ blockIdx, threadIdx;
extern thread_local dim3 blockDim, gridDim;
#ifdef VIDAR_REF_DEFINE_GLOBALS
thread_local uint3_ blockIdx, threadIdx;
thread_local dim3 blockDim, gridDim;
#endif

using std::max;
using std::min;

template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
static inline int cudaDeviceSynchronize() { return 0; }

/* one host call per CUDA thread, in (blockIdx.y, blockIdx.x, threadIdx.x) order */
template <class F, class... A>
static inline void vidar_ref_launch(F kernel, dim3 grid, int threads, A... args) {
  gridDim = grid;
  blockDim = dim3(threads, 1, 1);
  for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx)
      for (unsigned tx = 0; tx < (unsigned)threads; ++tx) {
        blockIdx = uint3_{bx, by, 0};
        threadIdx = uint3_{tx, 0, 0};
        kernel(args...);
      }
}
