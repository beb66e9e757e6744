// Host enqueue cost of one MIOpen convolution call (the encoders' 620 convolutions cost ~125 us of
// host time each through ATen): Find-mode API vs immediate-mode API, descriptors created once.
// Build: g++ -O2 miopen_host_cost.cpp -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -L/opt/rocm/lib -lMIOpen -lamdhip64
#include <hip/hip_runtime.h>
#include <miopen/miopen.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { auto s_ = (x); if (s_ != 0) { printf("fail %s -> %d\n", #x, (int)s_); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  int N = 32, C = argc > 1 ? atoi(argv[1]) : 18, H = argc > 2 ? atoi(argv[2]) : 64, K = C, R = 3;
  miopenHandle_t h; CK(miopenCreate(&h));
  hipStream_t st; CK(hipStreamCreate(&st)); CK(miopenSetStream(h, st));
  miopenTensorDescriptor_t xd, wd, yd; miopenConvolutionDescriptor_t cd;
  CK(miopenCreateTensorDescriptor(&xd)); CK(miopenCreateTensorDescriptor(&wd)); CK(miopenCreateTensorDescriptor(&yd));
  CK(miopenSet4dTensorDescriptor(xd, miopenFloat, N, C, H, H));
  CK(miopenSet4dTensorDescriptor(wd, miopenFloat, K, C, R, R));
  CK(miopenSet4dTensorDescriptor(yd, miopenFloat, N, K, H, H));
  CK(miopenCreateConvolutionDescriptor(&cd));
  CK(miopenInitConvolutionDescriptor(cd, miopenConvolution, 1, 1, 1, 1, 1, 1));
  size_t nx = (size_t)N * C * H * H, nw = (size_t)K * C * R * R;
  float *x, *w, *y, *dx, *dw; void* ws; size_t wsz = 512u << 20;
  CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&y, nx * 4)); CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&w, nw * 4)); CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&ws, wsz));
  CK(hipMemset(x, 0, nx * 4)); CK(hipMemset(w, 0, nw * 4)); CK(hipMemset(y, 0, nx * 4));
  float one = 1.f, zero = 0.f;
  // --- find mode (what ATen does)
  miopenConvAlgoPerf_t perf[4]; int got = 0;
  CK(miopenFindConvolutionForwardAlgorithm(h, xd, x, wd, w, cd, yd, y, 4, &got, perf, ws, wsz, false));
  miopenConvAlgoPerf_t pb[4], pw[4]; int gb = 0, gw = 0;
  CK(miopenFindConvolutionBackwardDataAlgorithm(h, yd, y, wd, w, cd, xd, dx, 4, &gb, pb, ws, wsz, false));
  CK(miopenFindConvolutionBackwardWeightsAlgorithm(h, yd, y, xd, x, cd, wd, dw, 4, &gw, pw, ws, wsz, false));
  const int IT = 500;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipStreamSynchronize(st)); double t0 = now();
    for (int i = 0; i < IT; ++i) CK(miopenConvolutionForward(h, &one, xd, x, wd, w, cd, perf[0].fwd_algo, &zero, yd, y, ws, wsz));
    double t1 = now();
    for (int i = 0; i < IT; ++i) CK(miopenConvolutionBackwardData(h, &one, yd, y, wd, w, cd, pb[0].bwd_data_algo, &zero, xd, dx, ws, wsz));
    double t2 = now();
    for (int i = 0; i < IT; ++i) CK(miopenConvolutionBackwardWeights(h, &one, yd, y, xd, x, cd, pw[0].bwd_weights_algo, &zero, wd, dw, ws, wsz));
    double t3 = now(); CK(hipStreamSynchronize(st)); double t4 = now();
    if (rep) printf("find-mode API  C=%d H=%d: host us/call fwd %.1f  bwd-data %.1f  bwd-weights %.1f   (drain %.1f ms)\n", C, H, (t1 - t0) / IT * 1e6, (t2 - t1) / IT * 1e6, (t3 - t2) / IT * 1e6, (t4 - t3) * 1e3);
  }
  // --- immediate mode
  size_t cnt = 0; CK(miopenConvolutionForwardGetSolutionCount(h, wd, xd, cd, yd, &cnt));
  std::vector<miopenConvSolution_t> sol(cnt); size_t ret = 0;
  CK(miopenConvolutionForwardGetSolution(h, wd, xd, cd, yd, cnt, &ret, sol.data()));
  CK(miopenConvolutionForwardCompileSolution(h, wd, xd, cd, yd, sol[0].solution_id));
  size_t cb = 0; CK(miopenConvolutionBackwardDataGetSolutionCount(h, yd, wd, cd, xd, &cb));
  std::vector<miopenConvSolution_t> sb(cb); CK(miopenConvolutionBackwardDataGetSolution(h, yd, wd, cd, xd, cb, &ret, sb.data()));
  CK(miopenConvolutionBackwardDataCompileSolution(h, yd, wd, cd, xd, sb[0].solution_id));
  size_t cw = 0; CK(miopenConvolutionBackwardWeightsGetSolutionCount(h, yd, xd, cd, wd, &cw));
  std::vector<miopenConvSolution_t> sw(cw); CK(miopenConvolutionBackwardWeightsGetSolution(h, yd, xd, cd, wd, cw, &ret, sw.data()));
  CK(miopenConvolutionBackwardWeightsCompileSolution(h, yd, xd, cd, wd, sw[0].solution_id));
  printf("immediate solutions: fwd id %llu (%.3f ms, ws %zu)  bwd-data id %llu (%.3f ms)  bwd-weights id %llu (%.3f ms, ws %zu)\n",
         (unsigned long long)sol[0].solution_id, sol[0].time, sol[0].workspace_size, (unsigned long long)sb[0].solution_id, sb[0].time,
         (unsigned long long)sw[0].solution_id, sw[0].time, sw[0].workspace_size);
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipStreamSynchronize(st)); double t0 = now();
    for (int i = 0; i < IT; ++i) CK(miopenConvolutionForwardImmediate(h, wd, w, xd, x, cd, yd, y, ws, wsz, sol[0].solution_id));
    double t1 = now();
    for (int i = 0; i < IT; ++i) CK(miopenConvolutionBackwardDataImmediate(h, yd, y, wd, w, cd, xd, dx, ws, wsz, sb[0].solution_id));
    double t2 = now();
    for (int i = 0; i < IT; ++i) CK(miopenConvolutionBackwardWeightsImmediate(h, yd, y, xd, x, cd, wd, dw, ws, wsz, sw[0].solution_id));
    double t3 = now(); CK(hipStreamSynchronize(st)); double t4 = now();
    if (rep) printf("immediate API  C=%d H=%d: host us/call fwd %.1f  bwd-data %.1f  bwd-weights %.1f   (drain %.1f ms)\n", C, H, (t1 - t0) / IT * 1e6, (t2 - t1) / IT * 1e6, (t3 - t2) / IT * 1e6, (t4 - t3) * 1e3);
  }
  // every backward-weights / backward-data / forward solution MIOpen offers: host enqueue vs GPU time
  {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (size_t i = 0; i < sw.size() && i < 12; ++i) {
      if (sw[i].workspace_size > wsz) continue;
      if (miopenConvolutionBackwardWeightsCompileSolution(h, yd, xd, cd, wd, sw[i].solution_id) != 0) continue;
      for (int k = 0; k < 3; ++k) miopenConvolutionBackwardWeightsImmediate(h, yd, y, xd, x, cd, wd, dw, ws, wsz, sw[i].solution_id);
      CK(hipStreamSynchronize(st)); double t0 = now(); CK(hipEventRecord(e0, st));
      for (int k = 0; k < 100; ++k) miopenConvolutionBackwardWeightsImmediate(h, yd, y, xd, x, cd, wd, dw, ws, wsz, sw[i].solution_id);
      double t1 = now(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("  wrw solution %llu algo %d: est %.3f ms, ws %zu  host %.1f us/call  wall %.1f us/call\n", (unsigned long long)sw[i].solution_id, (int)sw[i].algorithm, sw[i].time, sw[i].workspace_size, (t1 - t0) * 1e4, ms * 10);
    }
    for (size_t i = 0; i < sol.size() && i < 8; ++i) {
      if (sol[i].workspace_size > wsz) continue;
      if (miopenConvolutionForwardCompileSolution(h, wd, xd, cd, yd, sol[i].solution_id) != 0) continue;
      for (int k = 0; k < 3; ++k) miopenConvolutionForwardImmediate(h, wd, w, xd, x, cd, yd, y, ws, wsz, sol[i].solution_id);
      CK(hipStreamSynchronize(st)); double t0 = now(); CK(hipEventRecord(e0, st));
      for (int k = 0; k < 100; ++k) miopenConvolutionForwardImmediate(h, wd, w, xd, x, cd, yd, y, ws, wsz, sol[i].solution_id);
      double t1 = now(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("  fwd solution %llu algo %d: est %.3f ms, ws %zu  host %.1f us/call  wall %.1f us/call\n", (unsigned long long)sol[i].solution_id, (int)sol[i].algorithm, sol[i].time, sol[i].workspace_size, (t1 - t0) * 1e4, ms * 10);
    }
  }
  // GPU time per call of each (events)
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); float ms;
  CK(hipEventRecord(a, st)); for (int i = 0; i < 100; ++i) CK(miopenConvolutionForwardImmediate(h, wd, w, xd, x, cd, yd, y, ws, wsz, sol[0].solution_id));
  CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); printf("wall us/call (100 back-to-back): fwd %.1f", ms * 10);
  CK(hipEventRecord(a, st)); for (int i = 0; i < 100; ++i) CK(miopenConvolutionBackwardDataImmediate(h, yd, y, wd, w, cd, xd, dx, ws, wsz, sb[0].solution_id));
  CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); printf("  bwd-data %.1f", ms * 10);
  CK(hipEventRecord(a, st)); for (int i = 0; i < 100; ++i) CK(miopenConvolutionBackwardWeightsImmediate(h, yd, y, xd, x, cd, wd, dw, ws, wsz, sw[0].solution_id));
  CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); printf("  bwd-weights %.1f\n", ms * 10);
  return 0;
}
