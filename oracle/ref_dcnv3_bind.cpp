// pybind11 glue appended (same translation unit, through g++'s stdin) to the reference's
// ops_dcnv3/src/cuda/dcnv3_im2col_cuda.cuh compiled for the host by oracle/build_ref.py.
// TEST INFRASTRUCTURE: it pins oracle/msda.py, the reference's in-tree copy of the deformable-attention
// sampling arithmetic being `dcnv3_im2col_bilinear` (:33-81), `dcnv3_col2im_bilinear_gm` (:149-213) and the
// kernels that call them, `dcnv3_im2col_gpu_kernel` (:217-276) and `dcnv3_col2im_gpu_kernel_gm` (:776-839).
// One host call per CUDA thread, one thread per index like the reference's launchers size their grids
// (`vidar_ref_launch` of the shim; the backward kernel advances its output pointers inside CUDA_KERNEL_LOOP,
// :809-810, so a thread must not walk more than one index); the atomicAdd of the shim is a plain add.
// Only the two kernels without __shared__ are kept by the build filter; the host launchers
// (`<<<...>>>`, :842-1044) are cut.
#include <torch/extension.h>
#include <vector>

namespace {
// input [N,H,W,G*C], offset [N,Ho,Wo,G*P*2], mask [N,Ho,Wo,G*P] (double, contiguous) -> [N,Ho,Wo,G*C]
torch::Tensor im2col(torch::Tensor input, torch::Tensor offset, torch::Tensor mask, int kh, int kw,
                     int stride, int pad, int dil, int group, int gc, double offset_scale) {
  TORCH_CHECK(input.scalar_type() == at::kDouble && input.is_contiguous() && offset.is_contiguous() &&
              mask.is_contiguous());
  const int N = input.size(0), H = input.size(1), W = input.size(2);
  const int Ho = offset.size(1), Wo = offset.size(2);
  auto out = torch::zeros({N, Ho, Wo, group * gc}, input.options());
  const int n = N * Ho * Wo * group * gc;
  vidar_ref_launch(dcnv3_im2col_gpu_kernel<double>, dim3(n), 1, n, input.data_ptr<double>(),
                   offset.data_ptr<double>(), mask.data_ptr<double>(), out.data_ptr<double>(), kh, kw, stride,
                   stride, pad, pad, dil, dil, group, gc, H, W, Ho, Wo, offset_scale);
  return out;
}

std::vector<torch::Tensor> col2im(torch::Tensor grad_out, torch::Tensor input, torch::Tensor offset,
                                  torch::Tensor mask, int kh, int kw, int stride, int pad, int dil, int group,
                                  int gc, double offset_scale) {
  TORCH_CHECK(input.scalar_type() == at::kDouble && input.is_contiguous() && offset.is_contiguous() &&
              mask.is_contiguous() && grad_out.is_contiguous());
  const int N = input.size(0), H = input.size(1), W = input.size(2);
  const int Ho = offset.size(1), Wo = offset.size(2);
  auto gi = torch::zeros_like(input), go = torch::zeros_like(offset), gm = torch::zeros_like(mask);
  const int n = N * Ho * Wo * group * gc;
  vidar_ref_launch(dcnv3_col2im_gpu_kernel_gm<double>, dim3(n), 1, n, grad_out.data_ptr<double>(),
                   input.data_ptr<double>(), offset.data_ptr<double>(), mask.data_ptr<double>(), kh, kw, stride,
                   stride, pad, pad, dil, dil, group, gc, H, W, Ho, Wo, offset_scale, gi.data_ptr<double>(),
                   go.data_ptr<double>(), gm.data_ptr<double>());
  return {gi, go, gm};
}
}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("im2col", &im2col);
  m.def("col2im", &col2im);
}
