// Shared host-side helpers for the C-ABI entry points.
#pragma once
#include <hip/hip_runtime.h>

// hipGetLastError() is sticky per thread: an unrelated earlier failure (e.g. a benign probe made
// by the host framework) must not be reported as ours.  Every entry point starts with VIDAR_ENTER
// and ends with `return vidar_last_error();`.
#define VIDAR_ENTER() (void)hipGetLastError()
static inline int vidar_last_error() { return (int)hipGetLastError(); }
