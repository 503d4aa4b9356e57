/* Force-included (-include) ahead of a reference .cu translation unit compiled for the host.
 * torch 2.10 no longer lets AT_DISPATCH_FLOATING_TYPES take `tensor.type()`
 * (third_lib/dvxlr/dvxlr.cu:141,:496,:550 ...), so the macro is re-stated on scalarType(). */
#pragma once
#define VIDAR_REF_DEFINE_GLOBALS
#include <torch/extension.h>
#include "cuda_runtime.h"
/* torch only defines RestrictPtrTraits under a GPU compiler (headeronly/core/TensorAccessor.h:22) */
namespace at {
template <typename T>
struct RestrictPtrTraits {
  typedef T* __restrict__ PtrType;
};
}  // namespace at
#undef AT_DISPATCH_FLOATING_TYPES
#define AT_DISPATCH_FLOATING_TYPES(TYPE, NAME, ...)                         \
  do {                                                                      \
    const auto vidar_st_ = (TYPE).scalarType();                             \
    if (vidar_st_ == at::kFloat) { using scalar_t = float; __VA_ARGS__(); } \
    else if (vidar_st_ == at::kDouble) { using scalar_t = double; __VA_ARGS__(); } \
    else { TORCH_CHECK(false, NAME, ": unsupported dtype"); }               \
  } while (0)
