// Host glue between ATen's autograd engine and the C ABI of libhcmoco_hip.so.
//
// The C ABI (include/hcmoco_hip.h) stays the boundary; this file only removes the Python
// interpreter from the ~620 BatchNorm autograd nodes an HRNet pair executes per step (forward and
// backward of a Python torch.autograd.Function cost ~30 us of host time each on the MI355X host,
// tools/host_op_cost.py -- as much as the stock ops they replace; the same node in C++ is ~3x
// cheaper, and the step is host-paced once the kernels are fused).
// Registers torch.ops.hcmoco.bn_act; no pybind, no Python headers.
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/csrc/autograd/custom_function.h>
#include <torch/library.h>

#include "hcmoco_hip.h"

namespace {

using at::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

inline void* current_stream(const Tensor& t) {
  return static_cast<void*>(c10::hip::getCurrentHIPStream(t.get_device()).stream());
}

inline void check_rc(int rc, const char* what) {
  TORCH_CHECK(rc == 0, what, " failed: hip error ", rc, " (", hcm_error_string(rc), ")");
}

inline float* fptr(const Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }

// relu?(batch_norm(x; batch statistics) + residual?)   -- official_hrnet.py:40-105 block tails
struct BnAct : public torch::autograd::Function<BnAct> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x, const c10::optional<Tensor>& residual,
                        const Tensor& weight, const Tensor& bias, const c10::optional<Tensor>& running_mean,
                        const c10::optional<Tensor>& running_var, double momentum, double eps, bool relu) {
    TORCH_CHECK(x.is_cuda() && x.scalar_type() == at::kFloat && x.dim() == 4 && x.is_contiguous(),
                "hcmoco::bn_act needs fp32 NCHW-contiguous ROCm maps (no CPU fallback exists)");
    TORCH_CHECK(weight.defined() && bias.defined() && weight.is_contiguous() && bias.is_contiguous() &&
                    weight.scalar_type() == at::kFloat && bias.scalar_type() == at::kFloat &&
                    weight.numel() == x.size(1) && bias.numel() == x.size(1),
                "hcmoco::bn_act: affine parameters must be fp32 [C]");
    const int N = (int)x.size(0), C = (int)x.size(1), HW = (int)(x.size(2) * x.size(3));
    const size_t nf = hcm_bn_act_stats_floats(N, C, HW);
    TORCH_CHECK(nf > 0, "hcmoco::bn_act: unsupported shape (H*W must be a multiple of 4)");
    Tensor res;
    if (residual.has_value() && residual->defined()) {
      TORCH_CHECK(residual->sizes() == x.sizes() && residual->scalar_type() == at::kFloat,
                  "hcmoco::bn_act: residual must match x");
      res = residual->contiguous();
    }
    Tensor rm = running_mean.has_value() ? *running_mean : Tensor();
    Tensor rv = running_var.has_value() ? *running_var : Tensor();
    Tensor y = at::empty_like(x);
    Tensor stats = at::empty({(int64_t)nf}, x.options());
    check_rc(hcm_bn_act_forward(x.data_ptr<float>(), fptr(res), weight.data_ptr<float>(), bias.data_ptr<float>(),
                                fptr(rm), fptr(rv), (float)momentum, (float)eps, relu ? 1 : 0, N, C, HW,
                                y.data_ptr<float>(), stats.data_ptr<float>(), current_stream(x)),
             "hcm_bn_act_forward");
    ctx->saved_data["relu"] = relu;
    ctx->saved_data["has_res"] = res.defined();
    ctx->save_for_backward({x, weight, stats, relu ? y : Tensor()});
    return y;
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const auto saved = ctx->get_saved_variables();
    const Tensor& x = saved[0];
    const Tensor& weight = saved[1];
    const Tensor& stats = saved[2];
    const Tensor& y = saved[3];
    const bool relu = ctx->saved_data["relu"].toBool();
    const bool has_res = ctx->saved_data["has_res"].toBool();
    const int N = (int)x.size(0), C = (int)x.size(1), HW = (int)(x.size(2) * x.size(3));
    Tensor g = grads[0].contiguous();
    Tensor dx = ctx->needs_input_grad(0) ? at::empty_like(x) : Tensor();
    Tensor dz = relu ? at::empty_like(x) : Tensor();
    Tensor gstats = at::empty_like(stats);
    check_rc(hcm_bn_act_backward(g.data_ptr<float>(), x.data_ptr<float>(), fptr(y), weight.data_ptr<float>(),
                                 stats.data_ptr<float>(), relu ? 1 : 0, N, C, HW, fptr(dz), fptr(dx),
                                 gstats.data_ptr<float>(), current_stream(x)),
             "hcm_bn_act_backward");
    Tensor dres = has_res ? (relu ? dz : g) : Tensor();
    return {dx, dres, gstats.narrow(0, 0, C), gstats.narrow(0, C, C), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

Tensor bn_act(const Tensor& x, const c10::optional<Tensor>& residual, const Tensor& weight, const Tensor& bias,
              const c10::optional<Tensor>& running_mean, const c10::optional<Tensor>& running_var,
              double momentum, double eps, bool relu) {
  return BnAct::apply(x, residual, weight, bias, running_mean, running_var, momentum, eps, relu);
}

}  // namespace

TORCH_LIBRARY(hcmoco, m) {
  m.def("bn_act(Tensor x, Tensor? residual, Tensor weight, Tensor bias, Tensor? running_mean, "
        "Tensor? running_var, float momentum, float eps, bool relu) -> Tensor", &bn_act);
}
