// pybind11 glue for the host-compiled reference kernels (oracle/_ref).  Test infrastructure.
// Declares the reference's own host entry points (third_lib/dvr/dvr.cpp:9-21,
// third_lib/dvxlr/dvxlr.cpp:9-25, third_lib/dvxlr/dvxlr_v2.cpp:9-27) and exposes them WITHOUT the
// CHECK_CUDA guard of the reference's .cpp files, because here they run on CPU tensors.
#include <torch/extension.h>
#include <string>
#include <vector>

#if defined(REF_DVR)
std::vector<torch::Tensor> render_forward_cuda(torch::Tensor, torch::Tensor, torch::Tensor,
                                               torch::Tensor, const std::vector<int>, std::string);
std::vector<torch::Tensor> render_cuda(torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
                                       std::string);
torch::Tensor init_cuda(torch::Tensor, torch::Tensor, const std::vector<int>);
PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("render_forward", &render_forward_cuda);
  m.def("render", &render_cuda);
  m.def("init", &init_cuda);
}
#elif defined(REF_DVXLR)
std::vector<torch::Tensor> render_cuda(torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor);
std::vector<torch::Tensor> get_grad_sigma_cuda(torch::Tensor, torch::Tensor, torch::Tensor,
                                               torch::Tensor);
torch::Tensor init_cuda(torch::Tensor, torch::Tensor, const std::vector<int>);
PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("render", &render_cuda);
  m.def("get_grad_sigma", &get_grad_sigma_cuda);
  m.def("init", &init_cuda);
}
#elif defined(REF_DVXLR_V2)
std::vector<torch::Tensor> render_cuda_v2(torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
                                          torch::Tensor);
std::vector<torch::Tensor> get_grad_sigma_cuda_v2(torch::Tensor, torch::Tensor, torch::Tensor,
                                                  torch::Tensor, torch::Tensor, torch::Tensor);
PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("render_v2", &render_cuda_v2);
  m.def("get_grad_sigma_v2", &get_grad_sigma_cuda_v2);
}
#else
#error "pick a reference module"
#endif
