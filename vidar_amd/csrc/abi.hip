// ABI bookkeeping for libvidar_hip.so
#include <hip/hip_runtime.h>

#include "vidar_hip.h"
#include "vidar_common.h"

namespace {
__global__ void vidar_marker_kernel(int* sink, int id) {
  if (sink != nullptr) *sink = id;
}
}  // namespace

extern "C" {

int vidar_abi_version(void) { return 2; }

// Launches a one-thread kernel named `vidar_marker_kernel`: profiling tools use a pair of them to
// delimit the timed region of bench.py inside a rocprofv3 kernel trace (warm-up excluded).
int vidar_marker(int id, void* stream) {
  VIDAR_ENTER();
  hipLaunchKernelGGL(vidar_marker_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (int*)nullptr, id);
  return vidar_last_error();
}

}  // extern "C"
