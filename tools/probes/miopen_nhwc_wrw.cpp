// Does MIOpen's backward-weights run without its layout transposes when x and dy are handed over in
// NHWC?  (NCHW: three batched_transpose launches + a clear + the implicit-GEMM kernel per call.)
// Build: g++ -O2 miopen_nhwc_wrw.cpp -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -L/opt/rocm/lib -lMIOpen -lamdhip64
#include <hip/hip_runtime.h>
#include <miopen/miopen.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { auto s_ = (x); if (s_ != 0) { printf("fail %s -> %d\n", #x, (int)s_); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  int N = 32, C = argc > 1 ? atoi(argv[1]) : 18, H = argc > 2 ? atoi(argv[2]) : 64, K = C, R = 3;
  int wl = argc > 3 ? atoi(argv[3]) : 0;   // 0: weights NCHW (KCRS), 1: weights NHWC (KRSC)
  miopenHandle_t h; CK(miopenCreate(&h));
  hipStream_t st; CK(hipStreamCreate(&st)); CK(miopenSetStream(h, st));
  miopenTensorDescriptor_t xd, wd, yd; miopenConvolutionDescriptor_t cd;
  CK(miopenCreateTensorDescriptor(&xd)); CK(miopenCreateTensorDescriptor(&wd)); CK(miopenCreateTensorDescriptor(&yd));
  int xl[4] = {N, C, H, H}, yl[4] = {N, K, H, H}, wlens[4] = {K, C, R, R};
  CK(miopenSetNdTensorDescriptorWithLayout(xd, miopenFloat, miopenTensorNHWC, xl, 4));
  CK(miopenSetNdTensorDescriptorWithLayout(yd, miopenFloat, miopenTensorNHWC, yl, 4));
  if (wl) CK(miopenSetNdTensorDescriptorWithLayout(wd, miopenFloat, miopenTensorNHWC, wlens, 4));
  else    CK(miopenSet4dTensorDescriptor(wd, miopenFloat, K, C, R, R));
  CK(miopenCreateConvolutionDescriptor(&cd));
  CK(miopenInitConvolutionDescriptor(cd, miopenConvolution, 1, 1, 1, 1, 1, 1));
  size_t nx = (size_t)N * C * H * H, nw = (size_t)K * C * R * R;
  float *x, *y, *dw; void* ws; size_t wsz = 512u << 20;
  CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&y, nx * 4)); CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&ws, wsz));
  CK(hipMemset(x, 0, nx * 4)); CK(hipMemset(y, 0, nx * 4));
  miopenConvAlgoPerf_t pw[4]; int gw = 0;
  miopenStatus_t fs = miopenFindConvolutionBackwardWeightsAlgorithm(h, yd, y, xd, x, cd, wd, dw, 4, &gw, pw, ws, wsz, false);
  if (fs != 0 || gw < 1) { printf("C=%d H=%d weights-layout %d: Find failed (%d, %d results)\n", C, H, wl, (int)fs, gw); return 0; }
  float one = 1.f, zero = 0.f;
  const int IT = 300;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); float ms;
  for (int i = 0; i < 10; ++i) CK(miopenConvolutionBackwardWeights(h, &one, yd, y, xd, x, cd, pw[0].bwd_weights_algo, &zero, wd, dw, ws, wsz));
  CK(hipStreamSynchronize(st)); double t0 = now();
  CK(hipEventRecord(a, st));
  for (int i = 0; i < IT; ++i) CK(miopenConvolutionBackwardWeights(h, &one, yd, y, xd, x, cd, pw[0].bwd_weights_algo, &zero, wd, dw, ws, wsz));
  double t1 = now();
  CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
  printf("C=%d H=%d x,dy NHWC, weights-layout %d: algo %d ws %zu  host %.1f us/call  wall %.1f us/call\n", C, H, wl,
         (int)pw[0].bwd_weights_algo, pw[0].memory, (t1 - t0) / IT * 1e6, ms / IT * 1e3);
  return 0;
}
