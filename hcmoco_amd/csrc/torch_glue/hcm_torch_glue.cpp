// Host glue between ATen's autograd engine and the C ABI of libhcmoco_hip.so.
//
// The C ABI (include/hcmoco_hip.h) stays the boundary; this file only removes the Python
// interpreter from the ~620 BatchNorm autograd nodes an HRNet pair executes per step (forward and
// backward of a Python torch.autograd.Function cost ~30 us of host time each on the MI355X host,
// tools/host_op_cost.py -- as much as the stock ops they replace; the same node in C++ is ~3x
// cheaper, and the step is host-paced once the kernels are fused).
// Registers torch.ops.hcmoco.bn_act; no pybind, no Python headers.
#include <cstdlib>
#include <ATen/ATen.h>
#include <c10/hip/HIPCachingAllocator.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/csrc/autograd/custom_function.h>
#include <torch/custom_class.h>
#include <torch/library.h>

#include <miopen/miopen.h>

#include <stdlib.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

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

// ------------------------------------------------------------------------------------------------
// Normalisation: relu?(batch_norm(x; batch statistics) + residual?)  -- official_hrnet.py:40-105
// ------------------------------------------------------------------------------------------------
struct BnOut { Tensor y, stats; };

BnOut bn_forward_raw(const Tensor& x, const Tensor& res, const Tensor& weight, const Tensor& bias, const Tensor& rm,
                     const Tensor& rv, double momentum, double eps, bool relu) {
  TORCH_CHECK(x.is_cuda() && x.scalar_type() == at::kFloat && x.dim() == 4 && x.is_contiguous(),
              "hcmoco::bn_act needs fp32 NCHW-contiguous ROCm maps (no CPU fallback exists)");
  TORCH_CHECK(weight.defined() && bias.defined() && weight.is_contiguous() && bias.is_contiguous() &&
                  weight.scalar_type() == at::kFloat && bias.scalar_type() == at::kFloat &&
                  weight.numel() == x.size(1) && bias.numel() == x.size(1),
              "hcmoco::bn_act: affine parameters must be fp32 [C]");
  const int N = (int)x.size(0), C = (int)x.size(1), HW = (int)(x.size(2) * x.size(3));
  const size_t nf = hcm_bn_act_stats_floats(N, C, HW);
  TORCH_CHECK(nf > 0, "hcmoco::bn_act: unsupported shape (H*W must be a multiple of 4)");
  if (res.defined())
    TORCH_CHECK(res.sizes() == x.sizes() && res.scalar_type() == at::kFloat && res.is_contiguous(),
                "hcmoco::bn_act: residual must be a contiguous fp32 tensor shaped like x");
  BnOut o;
  o.y = at::empty_like(x);
  o.stats = at::empty({(int64_t)nf}, x.options());
  check_rc(hcm_bn_act_forward(x.data_ptr<float>(), fptr(res), weight.data_ptr<float>(), bias.data_ptr<float>(),
                              fptr(rm), fptr(rv), (float)momentum, (float)eps, relu ? 1 : 0, N, C, HW,
                              o.y.data_ptr<float>(), o.stats.data_ptr<float>(), current_stream(x)),
           "hcm_bn_act_forward");
  return o;
}

struct BnGrads { Tensor dx, dres, dgamma, dbeta; };

// g must be contiguous.  dres aliases dz (relu) or g (no relu).
BnGrads bn_backward_raw(const Tensor& g, const Tensor& x, const Tensor& y, const Tensor& weight, const Tensor& stats,
                        bool relu, bool has_res, bool need_dx) {
  const int N = (int)x.size(0), C = (int)x.size(1), HW = (int)(x.size(2) * x.size(3));
  BnGrads o;
  if (need_dx) o.dx = at::empty_like(x);
  Tensor dz = relu ? at::empty_like(x) : Tensor();
  Tensor gstats = at::empty_like(stats);
  check_rc(hcm_bn_act_backward(g.data_ptr<float>(), nullptr, x.data_ptr<float>(), fptr(y), weight.data_ptr<float>(),
                               stats.data_ptr<float>(), relu ? 1 : 0, N, C, HW, fptr(dz), fptr(o.dx),
                               gstats.data_ptr<float>(), current_stream(x)),
           "hcm_bn_act_backward");
  if (has_res) o.dres = relu ? dz : g;
  o.dgamma = gstats.narrow(0, 0, C);
  o.dbeta = gstats.narrow(0, C, C);
  return o;
}

inline Tensor opt_tensor(const c10::optional<Tensor>& t) { return t.has_value() ? *t : Tensor(); }
inline Tensor opt_contig(const c10::optional<Tensor>& t) {
  return (t.has_value() && t->defined()) ? t->contiguous() : Tensor();
}

struct BnAct : public torch::autograd::Function<BnAct> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x, const c10::optional<Tensor>& residual,
                        const Tensor& weight, const Tensor& bias, const c10::optional<Tensor>& running_mean,
                        const c10::optional<Tensor>& running_var, double momentum, double eps, bool relu) {
    Tensor res = opt_contig(residual);
    BnOut o = bn_forward_raw(x, res, weight, bias, opt_tensor(running_mean), opt_tensor(running_var), momentum, eps, relu);
    ctx->saved_data["relu"] = relu;
    ctx->saved_data["has_res"] = res.defined();
    ctx->save_for_backward({x, weight, o.stats, relu ? o.y : Tensor()});
    return o.y;
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const auto saved = ctx->get_saved_variables();
    const bool relu = ctx->saved_data["relu"].toBool();
    const bool has_res = ctx->saved_data["has_res"].toBool();
    BnGrads o = bn_backward_raw(grads[0].contiguous(), saved[0], saved[3], saved[1], saved[2], relu, has_res,
                                ctx->needs_input_grad(0));
    return {o.dx, o.dres, o.dgamma, o.dbeta, Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

Tensor bn_act(const Tensor& x, const c10::optional<Tensor>& residual, const Tensor& weight, const Tensor& bias,
              const c10::optional<Tensor>& running_mean, const c10::optional<Tensor>& running_var,
              double momentum, double eps, bool relu) {
  return BnAct::apply(x, residual, weight, bias, running_mean, running_var, momentum, eps, relu);
}

// ------------------------------------------------------------------------------------------------
// conv2d: the encoders' bias-free convolutions through MIOpen with everything cached.
// Measured on the MI355X host (tools/probes/miopen_host_cost.cpp, tools/host_op_cost.py): MIOpen itself
// needs 6.7 / 5.3 / 17.6 us of host time for forward / backward-data / backward-weights, but the same
// three calls cost 40 + 86 us through ATen (descriptor objects, algorithm cache keyed on a parameter
// struct, dispatcher, two autograd nodes): with 620 convolutions per step that overhead IS the step.
// Here a shape's descriptors and the algorithms chosen by MIOpen's Find (the same call ATen makes)
// are built once and one C++ autograd node issues the library calls.
// ------------------------------------------------------------------------------------------------
#define HCM_MIOPEN(expr)                                                                   \
  do {                                                                                     \
    miopenStatus_t st__ = (expr);                                                          \
    TORCH_CHECK(st__ == miopenStatusSuccess, #expr, " failed: ", miopenGetErrorString(st__)); \
  } while (0)

struct ConvKey {
  int dev, N, C, H, W, K, R, S, stride, pad;
  bool operator==(const ConvKey& o) const {
    return dev == o.dev && N == o.N && C == o.C && H == o.H && W == o.W && K == o.K && R == o.R && S == o.S &&
           stride == o.stride && pad == o.pad;
  }
};
struct ConvKeyHash {
  size_t operator()(const ConvKey& k) const {
    size_t h = 1469598103934665603ull;
    for (int v : {k.dev, k.N, k.C, k.H, k.W, k.K, k.R, k.S, k.stride, k.pad}) h = (h ^ (size_t)v) * 1099511628211ull;
    return h;
  }
};
struct ConvPlan {
  miopenTensorDescriptor_t xd = nullptr, wd = nullptr, yd = nullptr;
  miopenConvolutionDescriptor_t cd = nullptr;
  int Ho = 0, Wo = 0;
};

// What one MIOpen handle found for one plan.  MIOpen registers a Find's kernels with the handle that
// ran it and two handles may rank the algorithms differently (timings are noisy when several ranks
// share a device), so the choice is per handle, never shared between threads.
struct HandlePlan {
  unsigned found = 0;
  miopenConvFwdAlgorithm_t fwd_algo{};
  miopenConvBwdDataAlgorithm_t bd_algo{};
  miopenConvBwdWeightsAlgorithm_t bw_algo{};
  size_t fwd_ws = 0, bd_ws = 0, bw_ws = 0;
};

std::mutex g_plan_mutex;
std::unordered_map<ConvKey, ConvPlan*, ConvKeyHash> g_plans;

ConvPlan* get_plan(const ConvKey& k) {
  std::lock_guard<std::mutex> lock(g_plan_mutex);
  auto it = g_plans.find(k);
  if (it != g_plans.end()) return it->second;
  auto* p = new ConvPlan();
  p->Ho = (k.H + 2 * k.pad - k.R) / k.stride + 1;
  p->Wo = (k.W + 2 * k.pad - k.S) / k.stride + 1;
  HCM_MIOPEN(miopenCreateTensorDescriptor(&p->xd));
  HCM_MIOPEN(miopenCreateTensorDescriptor(&p->wd));
  HCM_MIOPEN(miopenCreateTensorDescriptor(&p->yd));
  HCM_MIOPEN(miopenSet4dTensorDescriptor(p->xd, miopenFloat, k.N, k.C, k.H, k.W));
  HCM_MIOPEN(miopenSet4dTensorDescriptor(p->wd, miopenFloat, k.K, k.C, k.R, k.S));
  HCM_MIOPEN(miopenSet4dTensorDescriptor(p->yd, miopenFloat, k.N, k.K, p->Ho, p->Wo));
  HCM_MIOPEN(miopenCreateConvolutionDescriptor(&p->cd));
  HCM_MIOPEN(miopenInitConvolutionDescriptor(p->cd, miopenConvolution, k.pad, k.pad, k.stride, k.stride, 1, 1));
  g_plans.emplace(k, p);
  return p;
}

// One MIOpen handle per (thread, stream): forward runs on the caller's thread, backward on the
// autograd engine's device thread or a helper thread, weight gradients possibly on their own stream.
miopenHandle_t thread_handle(int /*dev*/, hipStream_t stream) {
  // a stream belongs to one device, so (thread, stream) identifies the handle; no miopenSetStream per call
  thread_local std::unordered_map<hipStream_t, miopenHandle_t> handles;
  auto it = handles.find(stream);
  if (it == handles.end()) {
    miopenHandle_t h;
    HCM_MIOPEN(miopenCreateWithStream(&h, stream));
    it = handles.emplace(stream, h).first;
  }
  return it->second;
}

// Find results live with the (thread-local) handle that produced them; see HandlePlan.
enum : unsigned { kFoundFwd = 1, kFoundBwdData = 2, kFoundBwdWeights = 4 };
HandlePlan& handle_plan(miopenHandle_t h, ConvPlan* p) {
  thread_local std::unordered_map<const void*, std::unordered_map<ConvPlan*, HandlePlan>> found;
  return found[h][p];
}

// MIOpen workspaces get 2 MiB of slack behind them (r06).  Find BENCHMARKS every applicable solver on the caller's buffers, and on
// small problems some of the library's kernels read past what they were given (r05: a 1x1 solver on x [2, 5, 12, 64]; r06:
// igemm_bwd_gtcx35_nhwc_fp32 behind its layout transposes on 2 x 2 maps, "Memory access fault" on the boxes of the pool where
// the bytes behind a 2 MB allocator segment are unmapped).  The slack also lifts every workspace out of the small-block pool.
inline Tensor workspace(size_t bytes, const Tensor& like) {
  static const size_t slack = [] { const char* v = getenv("HCM_DEBUG_WS_SLACK"); return v ? (size_t)atoll(v) : (size_t)2 << 20; }();
  return at::empty({(int64_t)(bytes + slack)}, like.options().dtype(at::kByte));
}

ConvKey key_of(const Tensor& x, const Tensor& w, int64_t stride, int64_t pad) {
  return ConvKey{(int)x.get_device(), (int)x.size(0), (int)x.size(1), (int)x.size(2), (int)x.size(3),
                 (int)w.size(0), (int)w.size(2), (int)w.size(3), (int)stride, (int)pad};
}

// hcm_conv3x3_forward / _backward_data (csrc/conv.hip) serve the 3x3 stride-1 convolutions of the two
// high-resolution branches (18ch@64-wide, 36ch@32-wide: 14 us against MIOpen's 25 / 21 us); everything else
// stays on MIOpen.
bool own_conv(const Tensor& x, const Tensor& w, int64_t stride, int64_t pad) {
  return stride == 1 && pad == 1 && w.size(2) == 3 && w.size(3) == 3 &&
         hcm_conv3x3_supported((int)x.size(1), (int)w.size(0), (int)x.size(2), (int)x.size(3)) == 1;
}

bool stats_in_conv() { return true; }

// hcm_conv1x1_forward / _backward_data (csrc/conv1x1.hip): the 1x1 convolutions of the PointNet++ shared MLPs on ball tensors
// (maps far larger than an HRNet branch), on the tensors as they lie.
// A ball tensor is tall -- npoint / nsample = 8 .. 256 -- where an HRNet branch is square: the encoders' own 1x1 layers (80 x 80
// maps at a 320 crop) stay where they were tuned.
bool ball_shaped(int64_t H, int64_t W) { return H * W > 4096 && H >= 8 * W; }

bool own_conv1x1(const Tensor& x, const Tensor& w, int64_t stride, int64_t pad) {
  return stride == 1 && pad == 0 && w.size(2) == 1 && w.size(3) == 1 && ball_shaped(x.size(2), x.size(3)) &&
         hcm_conv1x1_supported((int)x.size(1), (int)w.size(0), (int)(x.size(2) * x.size(3))) == 1;
}

Tensor conv_forward_raw(const Tensor& x, const Tensor& w, int64_t stride, int64_t pad) {
  // the shape / device / dtype contract FIRST: own_conv1x1 indexes w.size(2), w.size(3) and hcm_conv1x1_forward trusts
  // w.size(1) == x.size(1) (a weight of another channel count read out of bounds on the fast path)
  TORCH_CHECK(x.is_cuda() && x.scalar_type() == at::kFloat && x.dim() == 4 && w.is_cuda() &&
                  w.scalar_type() == at::kFloat && w.dim() == 4 && w.size(1) == x.size(1) && x.is_contiguous() &&
                  w.is_contiguous(),
              "hcmoco::conv2d needs contiguous fp32 ROCm tensors x [N,C,H,W], w [K,C,R,S] (groups = dilation = 1, no bias)");
  if (own_conv1x1(x, w, stride, pad)) {
    Tensor y = at::empty({x.size(0), w.size(0), x.size(2), x.size(3)}, x.options());
    check_rc(hcm_conv1x1_forward(x.data_ptr<float>(), w.data_ptr<float>(), y.data_ptr<float>(), (int)x.size(0), (int)x.size(1),
                                 (int)w.size(0), (int)(x.size(2) * x.size(3)), current_stream(x)),
             "hcm_conv1x1_forward");
    return y;
  }
  if (own_conv(x, w, stride, pad)) {
    Tensor y = at::empty({x.size(0), w.size(0), x.size(2), x.size(3)}, x.options());
    check_rc(hcm_conv3x3_forward(x.data_ptr<float>(), w.data_ptr<float>(), y.data_ptr<float>(), (int)x.size(0),
                                 (int)x.size(1), (int)w.size(0), (int)x.size(2), (int)x.size(3), current_stream(x)),
             "hcm_conv3x3_forward");
    return y;
  }
  const ConvKey k = key_of(x, w, stride, pad);
  ConvPlan* p = get_plan(k);
  hipStream_t st = (hipStream_t)current_stream(x);
  miopenHandle_t h = thread_handle(k.dev, st);
  Tensor y = at::empty({k.N, k.K, p->Ho, p->Wo}, x.options());
  HandlePlan& hp = handle_plan(h, p);
  if (!(hp.found & kFoundFwd)) {
    size_t need = 0;
    HCM_MIOPEN(miopenConvolutionForwardGetWorkSpaceSize(h, p->wd, p->xd, p->cd, p->yd, &need));
    Tensor ws = workspace(need, x);
    miopenConvAlgoPerf_t perf; int got = 0;
    HCM_MIOPEN(miopenFindConvolutionForwardAlgorithm(h, p->xd, x.data_ptr(), p->wd, w.data_ptr(), p->cd, p->yd,
                                                     y.data_ptr(), 1, &got, &perf, ws.data_ptr(), need, false));
    TORCH_CHECK(got >= 1, "hcmoco::conv2d: MIOpen found no forward algorithm");
    hp.fwd_algo = perf.fwd_algo; hp.fwd_ws = perf.memory; hp.found |= kFoundFwd;
  }
  Tensor ws;                                   // the Winograd kernels HRNet mostly gets need none
  if (hp.fwd_ws) ws = workspace(hp.fwd_ws, x);
  const float one = 1.f, zero = 0.f;
  HCM_MIOPEN(miopenConvolutionForward(h, &one, p->xd, x.data_ptr(), p->wd, w.data_ptr(), p->cd, hp.fwd_algo, &zero,
                                      p->yd, y.data_ptr(), hp.fwd_ws ? ws.data_ptr() : nullptr, hp.fwd_ws));
  return y;
}

// ------------------------------------------------------------------------------------------------
// Deferred weight gradients.  In the backward walk only dX is on the dependency chain; dW feeds
// nothing but the optimizer, yet its MIOpen call is the most expensive one to ISSUE (five launches,
// 17.6 us of host time; 620 of them per step = a third of the autograd thread's work, while the GPU
// idles ~40 % of a batch-32 step).  With set_async_wgrad(true) the autograd thread only allocates dW
// and queues the call; one helper thread issues the queued calls on the SAME HIP stream, in order.
// Same stream => no cross-stream hazards: the task keeps g and x alive until its launches are in
// the stream, so later re-use of their memory is ordered behind them.  Contract for the caller
// (learning/contrast_trainer.py): gradients are reset with set_to_none (AccumulateGrad then only
// stores dW, it launches nothing) and wgrad_join() runs after backward() before anything reads .grad.
// ------------------------------------------------------------------------------------------------
void run_wgrad(ConvPlan* p, const Tensor& g, const Tensor& x, void* dw, const Tensor& like_w, Tensor* cached_ws);

// One helper thread per HIP stream; tasks run in push order with that stream current.
class StreamWorker {
 public:
  explicit StreamWorker(c10::hip::HIPStream stream) : stream_(stream) {
    std::thread([this] { loop(); }).detach();
  }
  void push(std::function<void(Tensor*)>&& fn) {
    {
      std::lock_guard<std::mutex> lk(m_);
      q_.emplace_back(std::move(fn));
      ++pending_;
    }
    cv_.notify_one();
  }
  void wait_idle() {
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [this] { return pending_ == 0; });
    if (!error_.empty()) { std::string e; e.swap(error_); TORCH_CHECK(false, "deferred work failed: ", e); }
  }
  c10::hip::HIPStream stream() const { return stream_; }

 private:
  void loop() {
    for (;;) {
      std::function<void(Tensor*)> fn;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [this] { return !q_.empty(); });
        fn = std::move(q_.front());
        q_.pop_front();
      }
      try {
        c10::hip::HIPStreamGuard guard(stream_);   // device + current stream of this thread
        fn(&ws_);
      } catch (const std::exception& e) {
        std::lock_guard<std::mutex> lk(m_);
        error_ = e.what();
      }
      fn = nullptr;                                // drop captured tensors before reporting idle
      {
        std::lock_guard<std::mutex> lk(m_);
        if (--pending_ == 0) done_.notify_all();
      }
    }
  }
  c10::hip::HIPStream stream_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  std::deque<std::function<void(Tensor*)>> q_;
  size_t pending_ = 0;
  std::string error_;
  Tensor ws_;   // MIOpen workspace of this stream, grown to the largest size seen, used in order
};

std::mutex g_workers_mutex;
std::unordered_map<hipStream_t, StreamWorker*> g_workers;   // leaked on purpose (threads outlive static dtors)

StreamWorker& worker_for(c10::hip::HIPStream stream) {
  std::lock_guard<std::mutex> lk(g_workers_mutex);
  auto it = g_workers.find(stream.stream());
  if (it == g_workers.end()) it = g_workers.emplace(stream.stream(), new StreamWorker(stream)).first;
  return *it->second;
}

// Blocks until every queued task is in its stream, then orders the caller's current stream behind
// the streams the tasks went to (autograd's own end-of-backward stream sync ran before them).
void join_workers() {
  std::vector<StreamWorker*> all;
  {
    std::lock_guard<std::mutex> lk(g_workers_mutex);
    for (auto& kv : g_workers) all.push_back(kv.second);
  }
  for (StreamWorker* w : all) {
    w->wait_idle();
    const auto s = w->stream();
    const auto cur = c10::hip::getCurrentHIPStream(s.device_index());
    if (cur == s) continue;
    hipEvent_t ev;
    TORCH_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess, "hipEventCreate failed");
    TORCH_CHECK(hipEventRecord(ev, s.stream()) == hipSuccess, "hipEventRecord failed");
    TORCH_CHECK(hipStreamWaitEvent(cur.stream(), ev, 0) == hipSuccess, "hipStreamWaitEvent failed");
    (void)hipEventDestroy(ev);
  }
}

std::atomic<bool> g_async_wgrad{false};
// set_deterministic(true) (or HCM_DETERMINISTIC=1 at load time): every convolution weight gradient the own kernels CAN
// compute goes to them (csrc/wgrad.hip: fixed-order partial sums + fixed-order reduction), whatever its size -- MIOpen's
// igemm_wrw...gkgs kernels accumulate with float atomics and are the one source of run-to-run noise in an HRNet step
// (DESIGN 2).  Slower for the wide layers; meant for tests that compare runs bit for bit.
std::atomic<bool> g_deterministic{[] { const char* e = getenv("HCM_DETERMINISTIC"); return e && e[0] != '0'; }()};
std::atomic<bool> g_wgrad_stream{false};
std::atomic<int> g_wgrad_batch{16};

struct ConvGrads { Tensor dx, dw; };

// g must be contiguous.
ConvGrads conv_backward_raw(const Tensor& g, const Tensor& x, const Tensor& w, int64_t stride, int64_t pad, bool need_dx,
                            bool need_dw) {
  TORCH_CHECK(x.is_cuda() && x.scalar_type() == at::kFloat && x.dim() == 4 && w.is_cuda() && w.scalar_type() == at::kFloat &&
                  w.dim() == 4 && w.size(1) == x.size(1) && g.is_cuda() && g.scalar_type() == at::kFloat && g.dim() == 4 &&
                  g.size(0) == x.size(0) && g.size(1) == w.size(0),
              "hcmoco::conv2d backward: fp32 ROCm tensors g [N,K,Ho,Wo], x [N,C,H,W], w [K,C,R,S] expected");
  const ConvKey k = key_of(x, w, stride, pad);
  ConvPlan* p = get_plan(k);
  hipStream_t st = (hipStream_t)current_stream(x);
  miopenHandle_t h = thread_handle(k.dev, st);
  const float one = 1.f, zero = 0.f;
  ConvGrads o;
  if (need_dx && own_conv1x1(x, w, stride, pad)) {
    o.dx = at::empty_like(x);
    check_rc(hcm_conv1x1_backward_data(g.data_ptr<float>(), w.data_ptr<float>(), o.dx.data_ptr<float>(), (int)x.size(0),
                                       (int)x.size(1), (int)w.size(0), (int)(x.size(2) * x.size(3)), st),
             "hcm_conv1x1_backward_data");
  } else if (need_dx && own_conv(x, w, stride, pad)) {
    o.dx = at::empty_like(x);
    check_rc(hcm_conv3x3_backward_data(g.data_ptr<float>(), w.data_ptr<float>(), o.dx.data_ptr<float>(), (int)x.size(0),
                                       (int)x.size(1), (int)w.size(0), (int)x.size(2), (int)x.size(3), st),
             "hcm_conv3x3_backward_data");
  } else if (need_dx) {
    o.dx = at::empty_like(x);
    HandlePlan& hp = handle_plan(h, p);
    if (!(hp.found & kFoundBwdData)) {
      size_t need = 0;
      HCM_MIOPEN(miopenConvolutionBackwardDataGetWorkSpaceSize(h, p->yd, p->wd, p->cd, p->xd, &need));
      Tensor ws = workspace(need, x);
      miopenConvAlgoPerf_t perf; int got = 0;
      HCM_MIOPEN(miopenFindConvolutionBackwardDataAlgorithm(h, p->yd, g.data_ptr(), p->wd, w.data_ptr(), p->cd, p->xd,
                                                            o.dx.data_ptr(), 1, &got, &perf, ws.data_ptr(), need, false));
      TORCH_CHECK(got >= 1, "hcmoco::conv2d: MIOpen found no backward-data algorithm");
      hp.bd_algo = perf.bwd_data_algo; hp.bd_ws = perf.memory; hp.found |= kFoundBwdData;
    }
    Tensor ws;
    if (hp.bd_ws) ws = workspace(hp.bd_ws, x);
    HCM_MIOPEN(miopenConvolutionBackwardData(h, &one, p->yd, g.data_ptr(), p->wd, w.data_ptr(), p->cd, hp.bd_algo, &zero,
                                             p->xd, o.dx.data_ptr(), hp.bd_ws ? ws.data_ptr() : nullptr, hp.bd_ws));
  }
  if (need_dw) {
    o.dw = at::empty_like(w);
    if (g_async_wgrad.load(std::memory_order_relaxed)) {
      // dw travels as a raw pointer so that AccumulateGrad can steal the tensor (use_count == 1)
      void* dw = o.dw.data_ptr();
      worker_for(c10::hip::getCurrentHIPStream(k.dev)).push([p, g, x, w, dw](Tensor* ws) { run_wgrad(p, g, x, dw, w, ws); });
    } else
      run_wgrad(p, g, x, o.dw.data_ptr(), w, nullptr);
  }
  return o;
}

// hcm_conv3x3_wgrad / hcm_conv1x1_wgrad (csrc/wgrad.hip) serve the stride-1 layers where they beat
// MIOpen's five-launch path: 3x3 with at most 48 channels (25 / 22 us against 49 / 38 us) and the 1x1
// convolutions of the fuse layers (9-11 us against 25-47 us) and their 3x3 stride-2 convolutions (14-23 us
// against 27-37 us); two launches, deterministic.  Everything
// else stays on MIOpen.
int own_wgrad(const Tensor& x, const Tensor& w, const Tensor& g) {      // 0: no; 3 / 1: kernel size at stride 1; 2: 3x3 at stride 2; 4: 1x1 on ball tensors
  constexpr int64_t maxc = 48, max1 = 160;
  // the 3-channel stem convolution (3 -> 64, stride 2, 128x128 output) falls on the kernel's generic,
  // non-constant-folded instantiation: 279 us per call against MIOpen's ~60 us (r02 profile)
  constexpr int64_t minc = 8;
  const bool det = g_deterministic.load(std::memory_order_relaxed);
  if ((g.size(3) & 3) != 0 || (!det && w.size(1) < minc)) return 0;
  const int64_t mc = det ? (int64_t)1 << 30 : maxc, m1 = det ? (int64_t)1 << 30 : max1;
  const bool same = g.size(2) == x.size(2) && g.size(3) == x.size(3);
  const bool half = 2 * g.size(2) == x.size(2) && 2 * g.size(3) == x.size(3);
  if (same && w.size(2) == 3 && w.size(3) == 3 && w.size(0) <= mc && w.size(1) <= mc) return 3;
  // 1x1 layers on maps far larger than an HRNet branch -- the shared MLPs of PointNet++ on [B, C, npoint, nsample] ball tensors
  // (16 .. 256 channels, 8 K - 131 K positions per image): hcm_conv1x1_ball_wgrad (csrc/conv1x1.hip), on the NCHW tensors
  // as they lie (MIOpen: two layout transposes + an NHWC implicit GEMM)
  if (same && w.size(2) == 1 && w.size(3) == 1 && ball_shaped(g.size(2), g.size(3)) &&
      hcm_conv1x1_ball_wgrad_workspace_bytes((int)x.size(0), (int)w.size(1), (int)w.size(0), (int)g.size(2), (int)g.size(3)) > 0)
    return 4;
  if (same && w.size(2) == 1 && w.size(3) == 1 && w.size(0) <= m1 && w.size(1) <= m1) {
    // wide 1x1 layers on maps far larger than an HRNet branch -- the shared MLPs of PointNet++ on [B, C, npoint, nsample]
    // ball tensors (64 / 128 channels, 16 K - 131 K positions per image) -- run one or two map rows per unit here
    // (0.8-1.5 ms, 2 TB/s); MIOpen's implicit GEMM is faster on them: HRNetPN 468 -> 480 samples/s (r03)
    constexpr int64_t maxpix = 4096;
    if (!det && std::max(w.size(0), w.size(1)) > maxc && g.size(2) * g.size(3) > maxpix) return 0;
    return 1;
  }
  if (half && w.size(2) == 3 && w.size(3) == 3 && w.size(1) <= mc && w.size(0) <= 2 * mc) return 2;   // stride 2
  return 0;
}

void run_wgrad(ConvPlan* p, const Tensor& g, const Tensor& x, void* dw, const Tensor& w, Tensor* cached_ws) {
  const int dev = (int)x.get_device();
  hipStream_t st = (hipStream_t)current_stream(x);
  if (const int ks = own_wgrad(x, w, g)) {
    // H, W: the OUTPUT map (= the input map at stride 1)
    const int N = (int)x.size(0), C = (int)x.size(1), K = (int)w.size(0), H = (int)g.size(2), W = (int)g.size(3);
    const auto bytes = ks == 3   ? hcm_conv3x3_wgrad_workspace_bytes
                       : ks == 1 ? hcm_conv1x1_wgrad_workspace_bytes
                       : ks == 4 ? hcm_conv1x1_ball_wgrad_workspace_bytes
                                 : hcm_conv3x3s2_wgrad_workspace_bytes;
    const auto run = ks == 3 ? hcm_conv3x3_wgrad : ks == 1 ? hcm_conv1x1_wgrad : ks == 4 ? hcm_conv1x1_ball_wgrad : hcm_conv3x3s2_wgrad;
    const size_t need = bytes(N, C, K, H, W);
    if (need > 0) {
      Tensor local;
      Tensor* ws = cached_ws ? cached_ws : &local;
      if (!ws->defined() || (size_t)ws->numel() < need) *ws = workspace(need, x);
      check_rc(run(x.data_ptr<float>(), g.data_ptr<float>(), N, C, K, H, W, static_cast<float*>(dw), ws->data_ptr(), need, st),
               "hcm_conv_wgrad");
      return;
    }
  }
  miopenHandle_t h = thread_handle(dev, st);
  const float one = 1.f, zero = 0.f;
  HandlePlan& hp = handle_plan(h, p);
  if (!(hp.found & kFoundBwdWeights)) {
    size_t need = 0;
    HCM_MIOPEN(miopenConvolutionBackwardWeightsGetWorkSpaceSize(h, p->yd, p->xd, p->cd, p->wd, &need));
    Tensor ws = workspace(need, x);
    miopenConvAlgoPerf_t perf; int got = 0;
    HCM_MIOPEN(miopenFindConvolutionBackwardWeightsAlgorithm(h, p->yd, g.data_ptr(), p->xd, x.data_ptr(), p->cd, p->wd, dw,
                                                             1, &got, &perf, ws.data_ptr(), need, false));
    TORCH_CHECK(got >= 1, "hcmoco::conv2d: MIOpen found no backward-weights algorithm");
    hp.bw_algo = perf.bwd_weights_algo; hp.bw_ws = perf.memory; hp.found |= kFoundBwdWeights;
  }
  Tensor local;
  Tensor* ws = cached_ws ? cached_ws : &local;
  if (!ws->defined() || (size_t)ws->numel() < hp.bw_ws) *ws = workspace(hp.bw_ws, x);
  HCM_MIOPEN(miopenConvolutionBackwardWeights(h, &one, p->yd, g.data_ptr(), p->xd, x.data_ptr(), p->cd, hp.bw_algo, &zero,
                                              p->wd, dw, ws->data_ptr(), hp.bw_ws));
}

void set_async_wgrad(bool on) {
  if (!on) join_workers();
  g_async_wgrad.store(on);
}
void wgrad_join() { join_workers(); }
void set_deterministic(bool on) { g_deterministic.store(on); }
bool get_deterministic() { return g_deterministic.load(); }
void set_wgrad_stream(bool on, int64_t batch) { g_wgrad_stream.store(on); g_wgrad_batch.store(batch > 0 ? (int)batch : 16); }

struct Conv2d : public torch::autograd::Function<Conv2d> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x_in, const Tensor& w_in, int64_t stride, int64_t pad) {
    Tensor x = x_in.contiguous(), w = w_in.contiguous();
    Tensor y = conv_forward_raw(x, w, stride, pad);
    ctx->save_for_backward({x, w});
    ctx->saved_data["stride"] = stride;
    ctx->saved_data["pad"] = pad;
    return y;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const auto saved = ctx->get_saved_variables();
    ConvGrads o = conv_backward_raw(grads[0].contiguous(), saved[0], saved[1], ctx->saved_data["stride"].toInt(),
                                    ctx->saved_data["pad"].toInt(), ctx->needs_input_grad(0), ctx->needs_input_grad(1));
    return {o.dx, o.dw, Tensor(), Tensor()};
  }
};

Tensor conv2d(const Tensor& x, const Tensor& w, int64_t stride, int64_t pad) { return Conv2d::apply(x, w, stride, pad); }

// conv -> bn [+ residual] [-> relu] as ONE autograd node (one node per layer in the backward walk).
struct ConvBnAct : public torch::autograd::Function<ConvBnAct> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x_in, const Tensor& w_in, int64_t stride, int64_t pad,
                        const c10::optional<Tensor>& residual, const Tensor& gamma, const Tensor& beta,
                        const c10::optional<Tensor>& running_mean, const c10::optional<Tensor>& running_var,
                        double momentum, double eps, bool relu) {
    Tensor x = x_in.contiguous(), w = w_in.contiguous();
    Tensor z = conv_forward_raw(x, w, stride, pad);
    Tensor res = opt_contig(residual);
    BnOut o = bn_forward_raw(z, res, gamma, beta, opt_tensor(running_mean), opt_tensor(running_var), momentum, eps, relu);
    ctx->saved_data["stride"] = stride;
    ctx->saved_data["pad"] = pad;
    ctx->saved_data["relu"] = relu;
    ctx->saved_data["has_res"] = res.defined();
    ctx->save_for_backward({x, w, z, gamma, o.stats, relu ? o.y : Tensor()});
    return o.y;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const auto saved = ctx->get_saved_variables();
    const bool relu = ctx->saved_data["relu"].toBool();
    const bool has_res = ctx->saved_data["has_res"].toBool();
    BnGrads b = bn_backward_raw(grads[0].contiguous(), saved[2], saved[5], saved[3], saved[4], relu, has_res, true);
    ConvGrads c = conv_backward_raw(b.dx, saved[0], saved[1], ctx->saved_data["stride"].toInt(),
                                    ctx->saved_data["pad"].toInt(), ctx->needs_input_grad(0), ctx->needs_input_grad(1));
    return {c.dx, c.dw, Tensor(), Tensor(), b.dres, b.dgamma, b.dbeta, Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

Tensor conv_bn_act(const Tensor& x, const Tensor& w, int64_t stride, int64_t pad, const c10::optional<Tensor>& residual,
                   const Tensor& gamma, const Tensor& beta, const c10::optional<Tensor>& running_mean,
                   const c10::optional<Tensor>& running_var, double momentum, double eps, bool relu) {
  return ConvBnAct::apply(x, w, stride, pad, residual, gamma, beta, running_mean, running_var, momentum, eps, relu);
}

// 1x1 conv -> bn -> relu -> max over the ball as ONE autograd node (r05): the last layer of a PointNet++ SharedMLP and the
// F.max_pool2d that follows it (reference: networks/pointnet2/pointnet2_modules.py:44-55).  x [N, C, np, ns] -> [N, K, np];
// y = relu(bn(z)) is never written, forward or backward (hcm_bn_relu_ballmax_*, csrc/bnact.hip).
struct ConvBnReluBallMax : public torch::autograd::Function<ConvBnReluBallMax> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x_in, const Tensor& w_in, const Tensor& gamma, const Tensor& beta,
                        const c10::optional<Tensor>& running_mean, const c10::optional<Tensor>& running_var,
                        double momentum, double eps) {
    Tensor x = x_in.contiguous(), w = w_in.contiguous();
    TORCH_CHECK(x.is_cuda() && x.scalar_type() == at::kFloat && x.dim() == 4,
                "hcmoco::conv_bn_relu_ballmax needs fp32 NCHW ROCm tensors (no CPU fallback exists)");
    Tensor z = conv_forward_raw(x, w, 1, 0);
    const int N = (int)z.size(0), K = (int)z.size(1), np = (int)z.size(2), ns = (int)z.size(3);
    const size_t nf = hcm_bn_relu_ballmax_stats_floats(N, K, np, ns);
    TORCH_CHECK(nf > 0, "hcmoco::conv_bn_relu_ballmax: unsupported ball shape (nsample in {4,8,16,32,64}, npoint % 4 == 0)");
    Tensor out = at::empty({N, K, np}, z.options()), zsel = at::empty({N, K, np}, z.options());
    Tensor arg = at::empty({N, K, np}, z.options().dtype(at::kInt));
    Tensor stats = at::empty({(int64_t)nf}, z.options());
    Tensor rm = opt_tensor(running_mean), rv = opt_tensor(running_var);
    check_rc(hcm_bn_relu_ballmax_forward(z.data_ptr<float>(), gamma.data_ptr<float>(), beta.data_ptr<float>(), fptr(rm),
                                         fptr(rv), (float)momentum, (float)eps, N, K, np, ns, out.data_ptr<float>(),
                                         arg.data_ptr<int>(), zsel.data_ptr<float>(), stats.data_ptr<float>(),
                                         current_stream(z)),
             "hcm_bn_relu_ballmax_forward");
    ctx->save_for_backward({x, w, z, gamma, stats, out, arg, zsel});
    return out;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const auto saved = ctx->get_saved_variables();
    const Tensor &x = saved[0], &w = saved[1], &z = saved[2], &gamma = saved[3], &stats = saved[4], &out = saved[5],
                 &arg = saved[6], &zsel = saved[7];
    const int N = (int)z.size(0), K = (int)z.size(1), np = (int)z.size(2), ns = (int)z.size(3);
    Tensor g = grads[0].contiguous();
    Tensor dz = at::empty_like(z);
    Tensor gstats = at::empty_like(stats);
    check_rc(hcm_bn_relu_ballmax_backward(g.data_ptr<float>(), out.data_ptr<float>(), arg.data_ptr<int>(),
                                          zsel.data_ptr<float>(), z.data_ptr<float>(), gamma.data_ptr<float>(),
                                          stats.data_ptr<float>(), N, K, np, ns, dz.data_ptr<float>(),
                                          gstats.data_ptr<float>(), current_stream(z)),
             "hcm_bn_relu_ballmax_backward");
    ConvGrads c = conv_backward_raw(dz, x, w, 1, 0, ctx->needs_input_grad(0), ctx->needs_input_grad(1));
    return {c.dx, c.dw, gstats.narrow(0, 0, K), gstats.narrow(0, K, K), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

Tensor conv_bn_relu_ballmax(const Tensor& x, const Tensor& w, const Tensor& gamma, const Tensor& beta,
                            const c10::optional<Tensor>& running_mean, const c10::optional<Tensor>& running_var,
                            double momentum, double eps) {
  return ConvBnReluBallMax::apply(x, w, gamma, beta, running_mean, running_var, momentum, eps);
}

// g [N, C, Ho, Wo] contiguous -> gradient of the [N, C, Hi, Wi] input (gather form, deterministic)
Tensor upsample_backward_raw(const Tensor& g, int64_t N, int64_t C, int64_t Hi, int64_t Wi) {
  Tensor gi = at::empty({N, C, Hi, Wi}, g.options());
  check_rc(hcm_upsample_bilinear2d_backward(g.data_ptr<float>(), (int)(N * C), (int)Hi, (int)Wi, (int)g.size(2),
                                            (int)g.size(3), gi.data_ptr<float>(), current_stream(g)),
           "hcm_upsample_bilinear2d_backward");
  return gi;
}

// F.interpolate(mode='bilinear', align_corners=False) on NCHW maps: forward hcm_upsample_bilinear2d,
// backward hcm_upsample_bilinear2d_backward (official_hrnet.py:231-236, build_backbone.py:247-254).
struct Upsample : public torch::autograd::Function<Upsample> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x_in, int64_t Ho, int64_t Wo) {
    TORCH_CHECK(x_in.is_cuda() && x_in.scalar_type() == at::kFloat && x_in.dim() == 4,
                "hcmoco::upsample_bilinear needs fp32 ROCm maps (no CPU fallback exists)");
    Tensor x = x_in.contiguous();
    const int64_t N = x.size(0), C = x.size(1), Hi = x.size(2), Wi = x.size(3);
    Tensor y = at::empty({N, C, Ho, Wo}, x.options());
    check_rc(hcm_upsample_bilinear2d(x.data_ptr<float>(), (int)(N * C), (int)Hi, (int)Wi, (int)Ho, (int)Wo,
                                     y.data_ptr<float>(), current_stream(x)),
             "hcm_upsample_bilinear2d");
    ctx->saved_data["in"] = std::vector<int64_t>{N, C, Hi, Wi};
    ctx->saved_data["out"] = std::vector<int64_t>{Ho, Wo};
    return y;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const auto in = ctx->saved_data["in"].toIntVector();
    const auto out = ctx->saved_data["out"].toIntVector();
    Tensor gi = upsample_backward_raw(grads[0].contiguous(), in[0], in[1], in[2], in[3]);
    return {gi, Tensor(), Tensor()};
  }
};

Tensor upsample_bilinear(const Tensor& x, int64_t Ho, int64_t Wo) { return Upsample::apply(x, Ho, Wo); }

// ------------------------------------------------------------------------------------------------
// Encoder programs.  An HRNet is ~330 instructions of four kinds (conv+bn[+residual][+relu], add,
// relu, bilinear up-sampling) on value slots; networks/hrnet.py compiles the module tree into that
// list once per input size.  run_encoder executes it as ONE autograd node:
//   * forward: a C++ loop over the raw calls above -- no Python, no dispatcher, no per-layer node;
//   * backward: the reverse loop with its own gradient slots.  The encoder's input is an image
//     (no gradient), so nothing on autograd's dependency chain waits for this node: with
//     set_async_wgrad(true) the WHOLE reverse loop is queued on the helper thread of the node's stream
//     and autograd moves on -- the two HRNets' backward passes are issued by two threads in parallel.
//     Parameter gradients are views of one flat buffer allocated up front; wgrad_join() as before.
//   * encoder_forward_async/_wait run the forward on the helper thread of the current stream as well,
//     so the caller can issue the other encoder meanwhile.
// ------------------------------------------------------------------------------------------------
enum : int64_t { kOpConvBn = 0, kOpAdd = 1, kOpRelu = 2, kOpUpsample = 3, kOpUpsampleAdd = 4 };
constexpr int kInstrInts = 12;   // op dst a b layer stride pad relu out_h out_w stream 0

struct Tape : torch::CustomClassHolder {
  std::vector<int64_t> prog;
  std::vector<Tensor> val;            // value slots (outputs held as detached aliases: no cycle with the node)
  std::vector<Tensor> z, stats;       // per conv+bn layer
  std::vector<Tensor> w, gamma;       // parameters the backward reads
  std::vector<int64_t> layer_off;     // offset of layer L's [dw | dgamma | dbeta] block in the flat gradient buffer
  std::vector<int64_t> scratch_off;   // offset of layer L's partial-sum scratch (hcm_bn_act_backward_ws) in its own buffer
  int64_t flat_numel = 0, scratch_numel = 0;
  int64_t tag = 0;                    // encoder id given by the caller (grad_chunk_wait looks the gradients up by it)
  bool need_dx0 = false;
};

// ------------------------------------------------------------------------------------------------
// Gradient chunks.  The flat parameter-gradient buffer of an encoder is dense ([dw | dgamma | dbeta]
// per layer, layers in program order) and the reverse loop fills it from the back, so it is handed to
// the collective library piecewise WHILE the loop is still running: with set_grad_chunks(n) the loop
// records a HIP event on its stream each time another n-th of the buffer is complete (chunk 0 = the
// LAST layers, finished first) and grad_chunk_wait(tag, k) -- called by the trainer thread after
// backward() returned -- blocks until chunk k's launches are in the stream, orders the CALLER's
// current stream behind that event and returns the chunk as a 1-D view of the flat buffer, ready for
// an in-place RCCL all-reduce (learning/grad_sync.py).  That is DistributedDataParallel's bucket
// overlap (reference: learning/contrast_trainer.py:74) for gradients that are written off the
// autograd thread, where DDP's own hooks cannot see them.
// ------------------------------------------------------------------------------------------------
struct GradChunks {
  Tensor flat;
  std::vector<int64_t> off;           // chunk k = flat[off[k+1], off[k]) -- descending, off[0] = numel, off.back() = 0
  std::vector<int64_t> first_layer;   // chunk k is complete once layer first_layer[k] has been issued
  std::vector<hipEvent_t> ev;
  std::mutex m;
  std::condition_variable cv;
  int ready = 0;                      // chunks [0, ready) have been issued
  std::string error;
  ~GradChunks() { for (auto e : ev) (void)hipEventDestroy(e); }
};
std::atomic<int> g_grad_chunks{0};
std::mutex g_chunks_mutex;
std::unordered_map<int64_t, std::shared_ptr<GradChunks>> g_chunks;

std::shared_ptr<GradChunks> make_chunks(const Tape& T, const Tensor& flat) {
  const int want = g_grad_chunks.load(std::memory_order_relaxed);
  if (want <= 0 || T.tag == 0) return nullptr;
  auto gc = std::make_shared<GradChunks>();
  gc->flat = flat;
  const int64_t layers = (int64_t)T.layer_off.size();
  const int64_t per = (T.flat_numel + want - 1) / want;
  gc->off.push_back(T.flat_numel);
  // walk the layers from the back; cut when the open chunk reached its share
  for (int64_t L = layers - 1; L >= 0; --L) {
    const bool last = L == 0;
    if (last || gc->off.back() - T.layer_off[L] >= per) {
      gc->off.push_back(T.layer_off[L]);
      gc->first_layer.push_back(L);
    }
  }
  gc->ev.resize(gc->first_layer.size());
  for (auto& e : gc->ev) TORCH_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess, "hipEventCreate failed");
  std::lock_guard<std::mutex> lk(g_chunks_mutex);
  g_chunks[T.tag] = gc;
  return gc;
}

void set_grad_chunks(int64_t n) { g_grad_chunks.store((int)n); }

int64_t grad_chunk_count(int64_t tag) {
  std::lock_guard<std::mutex> lk(g_chunks_mutex);
  auto it = g_chunks.find(tag);
  return it == g_chunks.end() ? 0 : (int64_t)it->second->ev.size();
}

Tensor grad_chunk_wait(int64_t tag, int64_t k) {
  std::shared_ptr<GradChunks> gc;
  {
    std::lock_guard<std::mutex> lk(g_chunks_mutex);
    auto it = g_chunks.find(tag);
    TORCH_CHECK(it != g_chunks.end(), "hcmoco::grad_chunk_wait: encoder ", tag, " has no backward pass in flight");
    gc = it->second;
    if (k + 1 == (int64_t)gc->ev.size()) g_chunks.erase(it);      // last chunk handed out: entry is spent
  }
  TORCH_CHECK(k >= 0 && k < (int64_t)gc->ev.size(), "hcmoco::grad_chunk_wait: chunk index out of range");
  {
    std::unique_lock<std::mutex> lk(gc->m);
    gc->cv.wait(lk, [&] { return gc->ready > k || !gc->error.empty(); });
    TORCH_CHECK(gc->error.empty(), "hcmoco: encoder backward failed: ", gc->error);
  }
  const auto cur = c10::hip::getCurrentHIPStream(gc->flat.get_device());
  TORCH_CHECK(hipStreamWaitEvent(cur.stream(), gc->ev[k], 0) == hipSuccess, "hipStreamWaitEvent failed");
  c10::hip::HIPCachingAllocator::recordStream(gc->flat.storage().data_ptr(), cur);
  return gc->flat.narrow(0, gc->off[k + 1], gc->off[k] - gc->off[k + 1]);
}

Tensor upsample_forward_raw(const Tensor& x, int64_t Ho, int64_t Wo) {
  const int64_t N = x.size(0), C = x.size(1), Hi = x.size(2), Wi = x.size(3);
  Tensor y = at::empty({N, C, Ho, Wo}, x.options());
  check_rc(hcm_upsample_bilinear2d(x.data_ptr<float>(), (int)(N * C), (int)Hi, (int)Wi, (int)Ho, (int)Wo,
                                   y.data_ptr<float>(), current_stream(x)),
           "hcm_upsample_bilinear2d");
  return y;
}

// out = relu?(acc + upsample(x)): a term of a fuse layer (and its ReLU when it is the last) in one launch
Tensor upsample_add_raw(const Tensor& x, const Tensor& acc, bool relu) {
  Tensor y = at::empty_like(acc);
  check_rc(hcm_upsample_bilinear2d_add(x.data_ptr<float>(), acc.data_ptr<float>(), relu ? 1 : 0, (int)(x.size(0) * x.size(1)),
                                       (int)x.size(2), (int)x.size(3), (int)acc.size(2), (int)acc.size(3), y.data_ptr<float>(),
                                       current_stream(x)),
           "hcm_upsample_bilinear2d_add");
  return y;
}

// Up to four HIP streams per encoder: [0] = the stream the encoder was called on, [1..3] side streams
// for the parallel HRNet branches (the program stamps a stream id on every instruction).  All
// launches of one pass are issued by ONE thread in program order, so "record an event on the producer
// stream now, make the consumer stream wait for it" is always sufficient (possibly more than needed).
constexpr int kMaxSid = 4;
std::mutex g_side_mutex;
std::unordered_map<hipStream_t, std::vector<c10::hip::HIPStream>> g_side_streams;

struct StreamCtx {
  std::vector<c10::hip::HIPStream> st;
  int cur = 0;
  int64_t epoch[kMaxSid] = {0, 0, 0, 0};              // launches issued per stream so far
  int64_t seen[kMaxSid][kMaxSid] = {};                // seen[from][to]: epoch of `from` that `to` has waited for
  bool used[kMaxSid] = {true, false, false, false};

  explicit StreamCtx(c10::hip::HIPStream base) {
    st.push_back(base);
    std::lock_guard<std::mutex> lk(g_side_mutex);
    auto& side = g_side_streams[base.stream()];
    while ((int)side.size() < kMaxSid - 1) side.push_back(c10::hip::getStreamFromPool(false, base.device_index()));
    for (auto& x : side) st.push_back(x);
  }
  ~StreamCtx() { if (cur != 0) c10::hip::setCurrentHIPStream(st[0]); }
  void wait(int from, int to) {
    if (from == to || seen[from][to] >= epoch[from]) return;
    hipEvent_t ev;
    TORCH_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess, "hipEventCreate failed");
    TORCH_CHECK(hipEventRecord(ev, st[from].stream()) == hipSuccess, "hipEventRecord failed");
    TORCH_CHECK(hipStreamWaitEvent(st[to].stream(), ev, 0) == hipSuccess, "hipStreamWaitEvent failed");
    (void)hipEventDestroy(ev);
    seen[from][to] = epoch[from];
  }
  void enter(int sid) {                               // make `sid` the current stream of this thread
    if (!used[sid]) { used[sid] = true; ++epoch[0]; wait(0, sid); }   // a side stream starts behind the base stream
    if (sid != cur) { c10::hip::setCurrentHIPStream(st[sid]); cur = sid; }
  }
  // tensor t was last written on stream `from`; make it usable on the current stream
  void acquire(const Tensor& t, int from) {
    if (from < 0 || from == cur || !t.defined()) return;
    wait(from, cur);
    c10::hip::HIPCachingAllocator::recordStream(t.storage().data_ptr(), st[cur]);
  }
  void issued() { ++epoch[cur]; }
  void finish() {                                      // the base stream continues behind every side stream
    for (int s = 1; s < kMaxSid; ++s) if (used[s]) { ++epoch[s]; wait(s, 0); }
    enter(0);
  }
};

Tensor& stream_workspace(hipStream_t st) {             // MIOpen workspace per (thread, stream), used in order
  thread_local std::unordered_map<hipStream_t, Tensor> ws;
  return ws[st];
}

// Deferred reductions of the own weight-gradient kernels (hcm_conv_wgrad_partial / hcm_wgrad_reduce_batch): the
// partial sums of a stretch of layers are parked in an arena (per thread and stream, used in order) and reduced by
// ONE launch per 64 layers instead of one tiny launch per layer.
struct WgradArena {
  Tensor buf;
  size_t used = 0;
  std::vector<hcm_wgrad_reduce_desc> red;
};
WgradArena& wgrad_arena(hipStream_t st) {
  thread_local std::unordered_map<hipStream_t, WgradArena> arenas;
  return arenas[st];
}
constexpr size_t kArenaBytes = (size_t)384 << 20;

void flush_wgrad_reductions(hipStream_t st) {
  WgradArena& A = wgrad_arena(st);
  if (A.red.empty()) return;
  check_rc(hcm_wgrad_reduce_batch(A.red.data(), (int)A.red.size(), st), "hcm_wgrad_reduce_batch");
  A.red.clear();
  A.used = 0;
}

// true: the layer's partial sums are parked and its reduction is queued on stream `st` (the current stream)
bool defer_own_wgrad(const Tensor& x, const Tensor& w, const Tensor& g, float* dw, hipStream_t st) {
  const int ks = own_wgrad(x, w, g);
  if (ks == 0 || ks == 4) return false;          // 4: the ball kernel keeps its own partials (run_wgrad)
  const int N = (int)x.size(0), C = (int)x.size(1), K = (int)w.size(0), H = (int)g.size(2), W = (int)g.size(3);
  const size_t need = ks == 3 ? hcm_conv3x3_wgrad_workspace_bytes(N, C, K, H, W)
                    : ks == 1 ? hcm_conv1x1_wgrad_workspace_bytes(N, C, K, H, W) : hcm_conv3x3s2_wgrad_workspace_bytes(N, C, K, H, W);
  if (need == 0 || need > kArenaBytes) return false;
  WgradArena& A = wgrad_arena(st);
  if (!A.buf.defined()) A.buf = at::empty({(int64_t)kArenaBytes}, x.options().dtype(at::kByte));
  if (A.used + need > kArenaBytes || A.red.size() >= 64) flush_wgrad_reductions(st);
  char* slot = static_cast<char*>(A.buf.data_ptr()) + A.used;
  int chunks = 0;
  check_rc(hcm_conv_wgrad_partial(ks, x.data_ptr<float>(), g.data_ptr<float>(), N, C, K, H, W, slot, need, &chunks, st),
           "hcm_conv_wgrad_partial");
  A.red.push_back(hcm_wgrad_reduce_desc{reinterpret_cast<const float*>(slot), dw, (int)w.numel(), chunks});
  A.used += (need + 255) & ~(size_t)255;
  return true;
}

// A gradient slot holds up to two tensors whose sum is the gradient: where the consumer is a conv+bn
// instruction the sum is formed inside hcm_bn_act_backward (no add kernel); anything else resolves it.
struct GradSlot { Tensor t, t2; bool owned = false; int sid = -1, sid2 = -1; };
inline void resolve(StreamCtx& S, GradSlot& s) {            // t <- t + t2 on the current stream
  if (!s.t2.defined()) return;
  S.acquire(s.t, s.sid);
  S.acquire(s.t2, s.sid2);
  if (s.owned) s.t.add_(s.t2);
  else { s.t = s.t + s.t2; s.owned = true; }
  s.t2 = Tensor(); s.sid2 = -1;
  s.sid = S.cur;
}

inline void accumulate(StreamCtx& S, GradSlot& s, const Tensor& t, bool owned) {
  if (!s.t.defined()) { s.t = t; s.owned = owned; s.sid = S.cur; return; }
  if (!s.t2.defined()) { s.t2 = t; s.sid2 = S.cur; return; }     // keep the pair lazy
  resolve(S, s);
  s.t2 = t; s.sid2 = S.cur;
}

// Reverse pass of a tape.  out_grads: gradients of the output slots (on the base stream); flat: the
// parameter-gradient buffer.  Every instruction runs on the stream its forward ran on.
void run_encoder_backward_impl(const c10::intrusive_ptr<Tape>& tape, const std::vector<int64_t>& out_slots,
                               std::vector<Tensor> out_grads, Tensor flat, Tensor scratch, c10::hip::HIPStream base,
                               GradChunks* gc);

void run_encoder_backward(const c10::intrusive_ptr<Tape>& tape, const std::vector<int64_t>& out_slots,
                          std::vector<Tensor> out_grads, Tensor flat, Tensor scratch, c10::hip::HIPStream base,
                          std::shared_ptr<GradChunks> gc) {
  try {
    run_encoder_backward_impl(tape, out_slots, std::move(out_grads), flat, scratch, base, gc.get());
  } catch (const std::exception& e) {
    if (gc) {                                   // wake grad_chunk_wait instead of leaving it blocked
      { std::lock_guard<std::mutex> lk(gc->m); gc->error = e.what(); }
      gc->cv.notify_all();
    }
    throw;
  }
}

void run_encoder_backward_impl(const c10::intrusive_ptr<Tape>& tape, const std::vector<int64_t>& out_slots,
                               std::vector<Tensor> out_grads, Tensor flat, Tensor scratch, c10::hip::HIPStream base,
                               GradChunks* gc) {
  at::NoGradGuard no_grad;
  Tape& T = *tape;
  StreamCtx S(base);
  float* sbase = scratch.data_ptr<float>();
  std::vector<GradSlot> G(T.val.size());
  for (size_t i = 0; i < out_slots.size(); ++i)
    if (out_grads[i].defined()) accumulate(S, G[out_slots[i]], out_grads[i], false);
  out_grads.clear();
  float* fbase = flat.data_ptr<float>();
  bool flat_on[kMaxSid] = {true, false, false, false};
  struct PendingWgrad { ConvPlan* p; Tensor g, x; float* dw; Tensor w; int sid; };
  std::vector<PendingWgrad> pending;
  const int wgrad_batch = g_wgrad_stream.load(std::memory_order_relaxed) ? g_wgrad_batch.load(std::memory_order_relaxed) : 0;
  auto flush_wgrads = [&]() {
    if (pending.empty()) return;
    const int back = S.cur;
    S.issued();
    S.enter(kMaxSid - 1);
    if (!flat_on[S.cur]) { c10::hip::HIPCachingAllocator::recordStream(flat.storage().data_ptr(), S.st[S.cur]); flat_on[S.cur] = true; }
    for (auto& j : pending) {
      S.acquire(j.g, j.sid);                              // one event per producing stream and batch
      c10::hip::HIPCachingAllocator::recordStream(j.x.storage().data_ptr(), S.st[S.cur]);
      run_wgrad(j.p, j.g, j.x, j.dw, j.w, &stream_workspace(S.st[S.cur].stream()));
    }
    pending.clear();
    S.issued();
    S.enter(back);
  };
  // chunk k of the flat buffer is complete once layer gc->first_layer[k] has been walked (issued or skipped)
  auto signal_chunks = [&](int64_t layer_done) {
    if (!gc) return;
    while (gc->ready < (int)gc->first_layer.size() && gc->first_layer[gc->ready] >= layer_done) {
      flush_wgrads();
      const int back = S.cur;
      S.enter(0);
      flush_wgrad_reductions(S.st[0].stream());      // the chunk's dW must be complete before its event
      S.issued();
      // The chunk's event must stand behind everything that wrote the chunk.  With the weight gradients on their side stream
      // (and no branch streams) that is stream 0's reductions AND the side stream's batches: the SIDE stream waits for stream
      // 0 and carries the event -- the reverse chain on stream 0 never waits for weight gradients at a chunk boundary
      // (forced 1-rank collectives: 726 samples/s with the chain waiting, against 747 without collectives).
      constexpr int W = kMaxSid - 1;
      bool side_only = wgrad_batch > 0 && S.used[W];
      for (int s2 = 1; s2 < W; ++s2) side_only = side_only && !S.used[s2];
      if (side_only) {
        ++S.epoch[0];
        S.wait(0, W);
        TORCH_CHECK(hipEventRecord(gc->ev[gc->ready], S.st[W].stream()) == hipSuccess, "hipEventRecord failed");
      } else {
        for (int s2 = 1; s2 < kMaxSid; ++s2) if (S.used[s2]) { ++S.epoch[s2]; S.wait(s2, 0); }
        TORCH_CHECK(hipEventRecord(gc->ev[gc->ready], S.st[0].stream()) == hipSuccess, "hipEventRecord failed");
      }
      S.enter(back);
      { std::lock_guard<std::mutex> lk(gc->m); ++gc->ready; }
      gc->cv.notify_all();
    }
  };
  { WgradArena& A0 = wgrad_arena(S.st[0].stream()); A0.red.clear(); A0.used = 0; }   // nothing left over from a failed pass
  const int64_t n = (int64_t)T.prog.size() / kInstrInts;
  for (int64_t i = n - 1; i >= 0; --i) {
    const int64_t* I = &T.prog[i * kInstrInts];
    const int64_t op = I[0], dst = I[1], a = I[2], b = I[3];
    GradSlot gs = std::move(G[dst]);
    G[dst] = GradSlot();
    if (!gs.t.defined()) {                                       // value does not reach the outputs: its
      T.val[dst] = Tensor();                                     // parameters keep the zeros `flat` starts with
      if (op == kOpConvBn) signal_chunks(I[4]);
      continue;
    }
    S.enter((int)I[10]);
    if (op != kOpConvBn) resolve(S, gs);
    S.acquire(gs.t, gs.sid);
    S.acquire(gs.t2, gs.sid2);
    Tensor g = gs.t.contiguous();
    Tensor g2 = gs.t2.defined() ? gs.t2.contiguous() : Tensor();
    if (op == kOpConvBn) {
      const int64_t L = I[4];
      const bool relu = I[7] != 0, has_res = b >= 0;
      const Tensor& z = T.z[L];
      const int N = (int)z.size(0), C = (int)z.size(1), HW = (int)(z.size(2) * z.size(3));
      if (!flat_on[S.cur]) { c10::hip::HIPCachingAllocator::recordStream(flat.storage().data_ptr(), S.st[S.cur]); flat_on[S.cur] = true; }
      Tensor dzc = at::empty_like(z);
      const bool has_dz = relu || g2.defined();
      Tensor dz = has_dz ? at::empty_like(z) : Tensor();
      float* gstats = fbase + T.layer_off[L] + T.w[L].numel();
      check_rc(hcm_bn_act_backward_ws(g.data_ptr<float>(), fptr(g2), z.data_ptr<float>(),
                                      relu ? T.val[dst].data_ptr<float>() : nullptr, T.gamma[L].data_ptr<float>(),
                                      T.stats[L].data_ptr<float>(), relu ? 1 : 0, N, C, HW, fptr(dz), dzc.data_ptr<float>(),
                                      gstats, sbase + T.scratch_off[L], current_stream(z)),
               "hcm_bn_act_backward_ws");
      if (has_res) accumulate(S, G[b], has_dz ? dz : g, has_dz);
      const Tensor& x = T.val[a];
      const bool need_dx = a != 0 || T.need_dx0;
      ConvPlan* p = get_plan(key_of(x, T.w[L], I[5], I[6]));
      hipStream_t st = (hipStream_t)current_stream(x);
      if (need_dx && own_conv(x, T.w[L], I[5], I[6])) {
        Tensor dx = at::empty_like(x);
        check_rc(hcm_conv3x3_backward_data(dzc.data_ptr<float>(), T.w[L].data_ptr<float>(), dx.data_ptr<float>(),
                                           (int)x.size(0), (int)x.size(1), (int)T.w[L].size(0), (int)x.size(2),
                                           (int)x.size(3), st),
                 "hcm_conv3x3_backward_data");
        accumulate(S, G[a], dx, true);
      } else if (need_dx) {
        miopenHandle_t h = thread_handle((int)x.get_device(), st);
        Tensor dx = at::empty_like(x);
        const float one = 1.f, zero = 0.f;
        HandlePlan& hp = handle_plan(h, p);
    if (!(hp.found & kFoundBwdData)) {
          size_t need = 0;
          HCM_MIOPEN(miopenConvolutionBackwardDataGetWorkSpaceSize(h, p->yd, p->wd, p->cd, p->xd, &need));
          Tensor fws = workspace(need, x);
          miopenConvAlgoPerf_t perf; int got = 0;
          HCM_MIOPEN(miopenFindConvolutionBackwardDataAlgorithm(h, p->yd, dzc.data_ptr(), p->wd, T.w[L].data_ptr(), p->cd, p->xd,
                                                                dx.data_ptr(), 1, &got, &perf, fws.data_ptr(), need, false));
          TORCH_CHECK(got >= 1, "hcmoco: MIOpen found no backward-data algorithm");
          hp.bd_algo = perf.bwd_data_algo; hp.bd_ws = perf.memory; hp.found |= kFoundBwdData;
        }
        Tensor bws;
        if (hp.bd_ws) bws = workspace(hp.bd_ws, x);
        HCM_MIOPEN(miopenConvolutionBackwardData(h, &one, p->yd, dzc.data_ptr(), p->wd, T.w[L].data_ptr(), p->cd, hp.bd_algo,
                                                 &zero, p->xd, dx.data_ptr(), hp.bd_ws ? bws.data_ptr() : nullptr, hp.bd_ws));
        accumulate(S, G[a], dx, true);
      }
      if (wgrad_batch > 0) {
        // dW is off the dependency chain: the calls are collected and issued in batches on the encoder's
        // stream 3 behind ONE event per batch, so the chain (norm backward -> data gradient -> next
        // layer) never queues behind their five small launches each
        pending.push_back(PendingWgrad{p, dzc, x, fbase + T.layer_off[L], T.w[L], S.cur});
        if ((int)pending.size() >= wgrad_batch) flush_wgrads();
      } else if (!(S.cur == 0 && defer_own_wgrad(x, T.w[L], dzc, fbase + T.layer_off[L], st))) {
        run_wgrad(p, dzc, x, fbase + T.layer_off[L], T.w[L], &stream_workspace(st));
      }
      T.z[L] = Tensor(); T.stats[L] = Tensor();           // release activations as the walk passes them
      S.issued();
      T.val[dst] = Tensor();
      signal_chunks(L);
      continue;
    } else if (op == kOpAdd) {
      accumulate(S, G[a], g, false);
      accumulate(S, G[b], g, false);
    } else if (op == kOpRelu) {
      accumulate(S, G[a], at::threshold_backward(g, T.val[dst], 0), true);
    } else if (op == kOpUpsample) {
      const Tensor& x = T.val[a];
      accumulate(S, G[a], upsample_backward_raw(g, x.size(0), x.size(1), x.size(2), x.size(3)), true);
    } else if (op == kOpUpsampleAdd) {
      const Tensor& x = T.val[a];
      if (I[7] != 0) {                                 // fused ReLU: mask by the output, the masked gradient is d acc
        const Tensor& y = T.val[dst];
        Tensor dx = at::empty_like(x), gm = at::empty_like(g);
        const int rc = hcm_upsample_bilinear2d_backward_relu(g.data_ptr<float>(), y.data_ptr<float>(), (int)(x.size(0) * x.size(1)),
                                                             (int)x.size(2), (int)x.size(3), (int)g.size(2), (int)g.size(3),
                                                             dx.data_ptr<float>(), gm.data_ptr<float>(), current_stream(x));
        if (rc != 0) {                                 // plane too large for the LDS form: mask, then the plain backward
          gm = at::threshold_backward(g, y, 0);
          dx = upsample_backward_raw(gm, x.size(0), x.size(1), x.size(2), x.size(3));
        }
        accumulate(S, G[a], dx, true);
        accumulate(S, G[b], gm, true);
      } else {
        accumulate(S, G[a], upsample_backward_raw(g, x.size(0), x.size(1), x.size(2), x.size(3)), true);
        accumulate(S, G[b], g, false);
      }
    }
    S.issued();
    T.val[dst] = Tensor();
  }
  flush_wgrads();
  signal_chunks(0);
  S.enter(0);
  flush_wgrad_reductions(S.st[0].stream());
  S.issued();
  S.finish();
}

// The forward pass of a program without any autograd bookkeeping: the launches, the tape the reverse loop reads, the
// output tensors.  Called inside EncoderFn::forward (run_encoder) or, for encoder_forward_async, on a helper thread
// AHEAD of the node's creation: the node is then made by encoder_forward_wait on the caller's thread, which gives it
// the caller's position in autograd's execution order (sequence numbers are per thread; a node made on the helper
// thread carries a tiny number and is the LAST ready node the engine picks -- r05: the HRNetPN model's HRNet backward
// started only after autograd had walked the whole cloud branch).
struct ForwardResult { c10::intrusive_ptr<Tape> tape; variable_list outs; int64_t layers = 0; };

ForwardResult encoder_forward_compute(const Tensor& x_in, at::TensorList params, at::TensorList buffers,
                                      std::vector<int64_t> prog, const std::vector<int64_t>& out_slots, int64_t n_values,
                                      double momentum, double eps, int64_t tag) {
  {
    TORCH_CHECK(prog.size() % kInstrInts == 0 && params.size() % 3 == 0 && buffers.size() * 3 == params.size() * 2,
                "hcmoco::run_encoder: malformed program");
    TORCH_CHECK(x_in.is_cuda(), "hcmoco::run_encoder needs ROCm tensors (no CPU fallback exists)");
    auto tape = c10::make_intrusive<Tape>();
    Tape& T = *tape;
    const int64_t layers = (int64_t)params.size() / 3;
    T.val.resize(n_values);
    T.z.resize(layers); T.stats.resize(layers); T.w.resize(layers); T.gamma.resize(layers); T.layer_off.resize(layers);
    T.scratch_off.resize(layers);
    T.tag = tag;
    T.val[0] = x_in.contiguous();
    T.need_dx0 = x_in.requires_grad();
    StreamCtx S(c10::hip::getCurrentHIPStream(x_in.get_device()));
    std::vector<int8_t> vsid(n_values, 0);
    const int64_t n = (int64_t)prog.size() / kInstrInts;
    for (int64_t i = 0; i < n; ++i) {
      const int64_t* I = &prog[i * kInstrInts];
      const int64_t op = I[0], dst = I[1], a = I[2], b = I[3];
      TORCH_CHECK(I[10] >= 0 && I[10] < kMaxSid, "hcmoco::run_encoder: stream id out of range");
      S.enter((int)I[10]);
      S.acquire(T.val[a], vsid[a]);
      if (b >= 0) S.acquire(T.val[b], vsid[b]);
      if (op == kOpConvBn) {
        const int64_t L = I[4];
        const Tensor& w = params[3 * L];
        Tensor z;
        BnOut o;
        if (stats_in_conv() && own_conv(T.val[a], w, I[5], I[6])) {
          // conv.hip leaves the per-channel sums of its output in its epilogue: no statistics launch, no re-read
          const Tensor& xin = T.val[a];
          const int N = (int)xin.size(0), Cc = (int)xin.size(1), Kc = (int)w.size(0), H = (int)xin.size(2), W = (int)xin.size(3);
          const int slots = hcm_conv3x3_stats_slots(N, H);
          z = at::empty({N, Kc, H, W}, xin.options());
          // sums about the running mean (shifted: no cancellation when |mean| >> std); the kernel leaves the shift in the last row
          Tensor part = at::empty({(2 * (int64_t)slots + 1) * Kc}, xin.options());
          check_rc(hcm_conv3x3_forward_stats(xin.data_ptr<float>(), w.data_ptr<float>(), z.data_ptr<float>(), N, Cc, Kc, H, W,
                                             fptr(buffers[2 * L]), part.data_ptr<float>(), current_stream(xin)),
                   "hcm_conv3x3_forward_stats");
          const Tensor res = b >= 0 ? T.val[b] : Tensor();
          o.y = at::empty_like(z);
          o.stats = at::empty({(int64_t)hcm_bn_act_stats_floats(N, Kc, H * W)}, z.options());
          check_rc(hcm_bn_act_forward_pre(z.data_ptr<float>(), fptr(res), params[3 * L + 1].data_ptr<float>(),
                                          params[3 * L + 2].data_ptr<float>(), fptr(buffers[2 * L]), fptr(buffers[2 * L + 1]),
                                          (float)momentum, (float)eps, I[7] != 0 ? 1 : 0, N, Kc, H * W, o.y.data_ptr<float>(),
                                          o.stats.data_ptr<float>(), part.data_ptr<float>(), slots, current_stream(z)),
                   "hcm_bn_act_forward_pre");
        } else {
          z = conv_forward_raw(T.val[a], w, I[5], I[6]);
          o = bn_forward_raw(z, b >= 0 ? T.val[b] : Tensor(), params[3 * L + 1], params[3 * L + 2], buffers[2 * L],
                             buffers[2 * L + 1], momentum, eps, I[7] != 0);
        }
        T.val[dst] = o.y; T.z[L] = z; T.stats[L] = o.stats; T.w[L] = w; T.gamma[L] = params[3 * L + 1];
        const int64_t C2 = 2 * params[3 * L + 1].numel();
        T.layer_off[L] = T.flat_numel;
        T.flat_numel += w.numel() + C2;
        T.scratch_off[L] = T.scratch_numel;
        T.scratch_numel += std::max<int64_t>(o.stats.numel() - C2, 0);
      } else if (op == kOpAdd) {
        T.val[dst] = at::add(T.val[a], T.val[b]);
      } else if (op == kOpRelu) {
        T.val[dst] = at::relu(T.val[a]);
      } else if (op == kOpUpsample) {
        T.val[dst] = upsample_forward_raw(T.val[a], I[8], I[9]);
      } else if (op == kOpUpsampleAdd) {
        T.val[dst] = upsample_add_raw(T.val[a], T.val[b], I[7] != 0);
      } else {
        TORCH_CHECK(false, "hcmoco::run_encoder: unknown opcode ", op);
      }
      vsid[dst] = (int8_t)S.cur;
      S.issued();
    }
    S.finish();
    variable_list outs;
    for (int64_t s : out_slots) {
      if (vsid[s] != 0) c10::hip::HIPCachingAllocator::recordStream(T.val[s].storage().data_ptr(), S.st[0]);
      outs.push_back(T.val[s]);
      T.val[s] = T.val[s].detach();       // the tape keeps an alias without autograd history
    }
    T.val[0] = T.val[0].detach();
    T.prog = std::move(prog);
    ForwardResult r;
    r.tape = tape; r.outs = std::move(outs); r.layers = layers;
    return r;
  }
}

std::mutex g_precomputed_mutex;
std::unordered_map<int64_t, ForwardResult> g_precomputed;   // encoder_forward_async results waiting for their node

struct EncoderFn : public torch::autograd::Function<EncoderFn> {
  // params: 3 per layer (w, gamma, beta); buffers: 2 per layer (running_mean, running_var)
  // pre_id != 0: the launches were issued ahead (encoder_forward_async); this call only makes the node
  static variable_list forward(AutogradContext* ctx, const Tensor& x_in, at::TensorList params, at::TensorList buffers,
                               std::vector<int64_t> prog, std::vector<int64_t> out_slots, int64_t n_values,
                               double momentum, double eps, int64_t tag, int64_t pre_id) {
    ForwardResult r;
    if (pre_id != 0) {
      std::lock_guard<std::mutex> lk(g_precomputed_mutex);
      auto it = g_precomputed.find(pre_id);
      TORCH_CHECK(it != g_precomputed.end(), "hcmoco::run_encoder: no forward pass was issued under handle ", pre_id);
      r = std::move(it->second);
      g_precomputed.erase(it);
    } else {
      r = encoder_forward_compute(x_in, params, buffers, std::move(prog), out_slots, n_values, momentum, eps, tag);
    }
    ctx->saved_data["tape"] = c10::IValue::make_capsule(r.tape);
    ctx->saved_data["outs"] = out_slots;
    ctx->saved_data["layers"] = r.layers;
    return r.outs;
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    TORCH_CHECK(ctx->saved_data.count("tape"),
                "hcmoco::run_encoder: the activations were released by the first backward pass "
                "(retain_graph / a second backward through the same forward is not supported)");
    auto tape = c10::static_intrusive_pointer_cast<Tape>(ctx->saved_data["tape"].toCapsule());
    std::vector<int64_t> out_slots = ctx->saved_data["outs"].toIntVector();
    const int64_t layers = ctx->saved_data["layers"].toInt();
    ctx->saved_data.erase("tape");
    Tape& T = *tape;
    // zeros, not empty: a layer whose value reaches no used output is skipped by the reverse loop and its
    // slice must still be a valid (zero) gradient for the optimizer
    Tensor flat = at::zeros({T.flat_numel}, T.w[0].options());
    Tensor scratch = at::empty({std::max<int64_t>(T.scratch_numel, 1)}, T.w[0].options());
    std::shared_ptr<GradChunks> gc = make_chunks(T, flat);
    variable_list out(1 + 5 * layers + 7);           // x, params, buffers, then the seven non-tensor arguments
    for (int64_t L = 0; L < layers; ++L) {
      const int64_t C = T.gamma[L].numel(), off = T.layer_off[L], wn = T.w[L].numel();
      out[1 + 3 * L] = flat.narrow(0, off, wn).view(T.w[L].sizes());
      out[1 + 3 * L + 1] = flat.narrow(0, off + wn, C);
      out[1 + 3 * L + 2] = flat.narrow(0, off + wn + C, C);
    }
    for (auto& g : grads) if (g.defined()) g = g.contiguous();
    auto stream = c10::hip::getCurrentHIPStream(T.val[0].get_device());
    if (g_async_wgrad.load(std::memory_order_relaxed) && !T.need_dx0) {
      worker_for(stream).push([tape, out_slots, grads, flat, scratch, stream, gc](Tensor*) mutable {
        run_encoder_backward(tape, out_slots, std::move(grads), flat, scratch, stream, gc);
      });
    } else {
      run_encoder_backward(tape, out_slots, std::move(grads), flat, scratch, stream, gc);
    }
    return out;
  }
};

std::vector<Tensor> run_encoder(const Tensor& x, at::TensorList params, at::TensorList buffers, std::vector<int64_t> prog,
                                std::vector<int64_t> out_slots, int64_t n_values, double momentum, double eps, int64_t tag) {
  return EncoderFn::apply(x, params, buffers, std::move(prog), std::move(out_slots), n_values, momentum, eps, tag, (int64_t)0);
}

// Forward on the helper thread of the current stream; the caller collects the outputs with _wait, which is also where
// the autograd node is made (on the caller's thread, under the stream the forward ran on).
struct PendingForward {
  std::mutex m; std::condition_variable cv; bool done = false; std::string error;
  ForwardResult result;
  Tensor x; std::vector<Tensor> params, buffers; std::vector<int64_t> prog, out_slots;
  int64_t n_values = 0, tag = 0; double momentum = 0, eps = 0; bool grad = false, node_at_wait = false;
  std::vector<Tensor> outs;     // node_at_wait == false: the node was made on the helper thread, these are its outputs
  c10::hip::HIPStream stream = c10::hip::getDefaultHIPStream();
};
std::mutex g_pending_mutex;
std::unordered_map<int64_t, std::shared_ptr<PendingForward>> g_pending;
int64_t g_pending_next = 1;

// node_at_wait: false -- the autograd node is made on the helper thread, right behind the launches (its sequence number is
// that thread's: the engine picks it LAST among ready nodes).  That is the order the two-HRNet model wants: measured on one
// box, alternating runs, r05: 704-707 samples/s with encoder2's node made here against 680-684 with it made at _wait (where
// it outranks encoder1's and its reverse loop is queued first) -- a 3.5 % difference from the ORDER of two pushes;
// true -- encoder_forward_wait makes it on the caller's thread, i.e. AFTER every node the caller made meanwhile (HRNetPN: the
// HRNet's reverse loop then starts before autograd walks the cloud branch).
int64_t encoder_forward_async(const Tensor& x, at::TensorList params, at::TensorList buffers, std::vector<int64_t> prog,
                              std::vector<int64_t> out_slots, int64_t n_values, double momentum, double eps, int64_t tag,
                              bool node_at_wait) {
  auto pend = std::make_shared<PendingForward>();
  int64_t id;
  {
    std::lock_guard<std::mutex> lk(g_pending_mutex);
    id = g_pending_next++;
    g_pending[id] = pend;
  }
  pend->x = x;
  pend->params.assign(params.begin(), params.end());
  pend->buffers.assign(buffers.begin(), buffers.end());
  pend->prog = std::move(prog);
  pend->out_slots = std::move(out_slots);
  pend->n_values = n_values; pend->momentum = momentum; pend->eps = eps; pend->tag = tag;
  pend->grad = at::GradMode::is_enabled();
  pend->node_at_wait = node_at_wait;
  pend->stream = c10::hip::getCurrentHIPStream(x.get_device());
  worker_for(pend->stream).push([pend](Tensor*) mutable {
    ForwardResult r;
    std::string err;
    try {
      if (pend->node_at_wait || !pend->grad) {
        at::AutoGradMode mode(false);         // raw launches only; the node is made by encoder_forward_wait
        r = encoder_forward_compute(pend->x, at::TensorList(pend->params), at::TensorList(pend->buffers), pend->prog,
                                    pend->out_slots, pend->n_values, pend->momentum, pend->eps, pend->tag);
      } else {
        at::AutoGradMode mode(true);
        pend->outs = EncoderFn::apply(pend->x, at::TensorList(pend->params), at::TensorList(pend->buffers), pend->prog,
                                      pend->out_slots, pend->n_values, pend->momentum, pend->eps, pend->tag, (int64_t)0);
      }
    } catch (const std::exception& e) { err = e.what(); }
    {
      std::lock_guard<std::mutex> lk(pend->m);
      pend->result = std::move(r); pend->error = err; pend->done = true;
    }
    pend->cv.notify_all();
  });
  return id;
}

std::vector<Tensor> encoder_forward_wait(int64_t id) {
  std::shared_ptr<PendingForward> pend;
  {
    std::lock_guard<std::mutex> lk(g_pending_mutex);
    auto it = g_pending.find(id);
    TORCH_CHECK(it != g_pending.end(), "hcmoco::encoder_forward_wait: unknown handle ", id);
    pend = it->second;
    g_pending.erase(it);
  }
  {
    std::unique_lock<std::mutex> lk(pend->m);
    pend->cv.wait(lk, [&] { return pend->done; });
  }
  TORCH_CHECK(pend->error.empty(), "hcmoco::encoder_forward_async failed: ", pend->error);
  if (pend->grad && !pend->node_at_wait) return std::move(pend->outs);
  if (!pend->grad || !at::GradMode::is_enabled()) return std::move(pend->result.outs);
  {
    std::lock_guard<std::mutex> lk(g_precomputed_mutex);
    g_precomputed[id] = std::move(pend->result);
  }
  // EncoderFn::forward takes the entry out again; if apply() throws before it gets there (argument checks of the autograd
  // machinery), the parked tape -- every saved activation of an HRNet forward -- must not stay in the map
  struct Unpark {
    int64_t id;
    ~Unpark() {
      std::lock_guard<std::mutex> lk(g_precomputed_mutex);
      g_precomputed.erase(id);
    }
  } unpark{id};
  // the node remembers the stream that is current while it is made: backward replays on the forward's stream
  c10::hip::HIPStreamGuard guard(pend->stream);
  return EncoderFn::apply(pend->x, at::TensorList(pend->params), at::TensorList(pend->buffers), std::move(pend->prog),
                          std::move(pend->out_slots), pend->n_values, pend->momentum, pend->eps, pend->tag, id);
}

}  // namespace

TORCH_LIBRARY(hcmoco, m) {
  m.def("conv2d(Tensor x, Tensor weight, int stride, int pad) -> Tensor", &conv2d);
  m.def("conv_bn_act(Tensor x, Tensor weight, int stride, int pad, Tensor? residual, Tensor gamma, Tensor beta, "
        "Tensor? running_mean, Tensor? running_var, float momentum, float eps, bool relu) -> Tensor", &conv_bn_act);
  m.def("conv_bn_relu_ballmax(Tensor x, Tensor weight, Tensor gamma, Tensor beta, Tensor? running_mean, "
        "Tensor? running_var, float momentum, float eps) -> Tensor", &conv_bn_relu_ballmax);
  m.def("upsample_bilinear(Tensor x, int out_h, int out_w) -> Tensor", &upsample_bilinear);
  m.def("run_encoder(Tensor x, Tensor[] params, Tensor[] buffers, int[] program, int[] outputs, int n_values, "
        "float momentum, float eps, int tag=0) -> Tensor[]", &run_encoder);
  m.def("encoder_forward_async(Tensor x, Tensor[] params, Tensor[] buffers, int[] program, int[] outputs, int n_values, "
        "float momentum, float eps, int tag=0, bool node_at_wait=False) -> int", &encoder_forward_async);
  m.def("set_grad_chunks(int n) -> ()", &set_grad_chunks);
  m.def("grad_chunk_count(int tag) -> int", &grad_chunk_count);
  m.def("grad_chunk_wait(int tag, int k) -> Tensor", &grad_chunk_wait);
  m.def("encoder_forward_wait(int handle) -> Tensor[]", &encoder_forward_wait);
  m.def("set_async_wgrad(bool on) -> ()", &set_async_wgrad);
  m.def("wgrad_join() -> ()", &wgrad_join);
  m.def("set_wgrad_stream(bool on, int batch) -> ()", &set_wgrad_stream);
  m.def("set_deterministic(bool on) -> ()", &set_deterministic);
  m.def("get_deterministic() -> bool", &get_deterministic);
  m.def("bn_act(Tensor x, Tensor? residual, Tensor weight, Tensor bias, Tensor? running_mean, "
        "Tensor? running_var, float momentum, float eps, bool relu) -> Tensor", &bn_act);
}
