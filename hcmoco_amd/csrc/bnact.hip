// Fused BatchNorm2d (+ residual add) (+ ReLU) for the HRNet encoders, gfx950.
//
// Every convolution of the two HRNets (pycontrast/networks/official_hrnet/official_hrnet.py:40-105
// BasicBlock / Bottleneck, :287-347 transitions, :161-207 fuse layers) is followed by
// BatchNorm2d, usually a residual add, usually a ReLU: 618 normalisations per step on activations of
// 0.3-134 MB.  The stock path runs them as 3 kernels forward and 3 backward, and the library's
// spatial batch-norm assigns ONE workgroup per channel -- with C = 18 channels (the high-resolution
// branch) that is 18 workgroups on a 256-CU part (r01 profile: 19-21 us per launch for a 9.4 MB map).
//
// Here a layer is two kernels per direction, both on a (channel, slice) grid so that even C = 18 is
// several hundred workgroups (PMC: HBM traffic = the algorithmic passes below, profiles/r01_bnact_pmc.json):
//   forward : stats  (per-slice shifted sums  S1 = sum(x-k), S2 = sum((x-k)^2), k = x[0,c,0,0])
//             apply  (merge slices in fixed order -> mean, invstd, running stats;
//                     y = relu(x*sc + sh + residual))
//   backward: reduce (dz = (dy + dy2?) * [y > 0]; per-slice sum(dz), sum(dz*(x-mean)); dz doubles as d residual)
//             apply  (dgamma, dbeta; dx = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)))
// Maps of <= 8192 values per channel take the one-kernel-per-direction path further down.
// Sums are plain additions of per-slice partials in slice order: deterministic, no atomics.
// HBM-bound: forward moves 3 (4 with residual) tensor passes, backward 7; see DESIGN.md 4.7.
#include "hcm_common.h"
#include "../../include/hcmoco_hip.h"

namespace {

using namespace hcm;

constexpr int kBT = 256;          // threads per workgroup
constexpr int kVec = kBT * 4;     // floats per workgroup iteration
constexpr int kMaxSplit = 64;     // slices per channel (merged by one wave)

struct Geo {
  int C, HW, M;      // M = N*HW elements per channel
  int per, split;    // slice length (multiple of kVec) and slice count
  int shift;         // log2(HW) or -1
};

Geo make_geo(int N, int C, int HW) {
  Geo g;
  g.C = C; g.HW = HW; g.M = N * HW;
  int want = 1024 / C;
  if (want < 1) want = 1;
  if (want > kMaxSplit) want = kMaxSplit;
  int per = (g.M + want - 1) / want;
  per = ((per + kVec - 1) / kVec) * kVec;
  g.per = per;
  g.split = (g.M + per - 1) / per;
  g.shift = -1;
  if ((HW & (HW - 1)) == 0) { int s = 0; while ((1 << s) < HW) ++s; g.shift = s; }
  return g;
}

__device__ __forceinline__ size_t elem_offset(const Geo& g, int c, int f) {
  const int n = g.shift >= 0 ? (f >> g.shift) : (f / g.HW);
  return ((size_t)n * g.C + c) * (size_t)g.HW + (size_t)(f - n * g.HW);
}

// Sum two values over the workgroup; result valid in every thread.
__device__ __forceinline__ void block_sum2(float& a, float& b) {
  __shared__ float sh[2][kBT / 64];
  a = wave_sum(a);
  b = wave_sum(b);
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = a; sh[1][threadIdx.x >> 6] = b; }
  __syncthreads();
  a = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
  b = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
}

// ReLU that keeps a NaN (ATen's relu / clamp_min propagate it; fmaxf(NaN, 0) = 0 would hide a diverged layer)
__device__ __forceinline__ float relu_nan(float v) { return v < 0.f ? 0.f : v; }

// Merge the per-slice partials of channel c (slice order, one lane per slice).
__device__ __forceinline__ void merge_partials(const float* part, const Geo& g, int c, float& p1, float& p2) {
  const int lane = threadIdx.x & 63;
  p1 = lane < g.split ? part[(size_t)(2 * lane) * g.C + c] : 0.f;
  p2 = lane < g.split ? part[(size_t)(2 * lane + 1) * g.C + c] : 0.f;
  p1 = wave_sum(p1);
  p2 = wave_sum(p2);
}

__global__ __launch_bounds__(kBT) void bn_stats_kernel(const float* __restrict__ x, Geo g,
                                                       float* __restrict__ part) {
  const int c = blockIdx.x, s = blockIdx.y;
  const int beg = s * g.per, end = min(g.M, beg + g.per);
  const float k = x[(size_t)c * g.HW];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll 4
  for (int f = beg + threadIdx.x * 4; f < end; f += kVec) {
    const float4 v = *reinterpret_cast<const float4*>(x + elem_offset(g, c, f));
    const float a = v.x - k, b = v.y - k, cc = v.z - k, d = v.w - k;
    s1 += (a + b) + (cc + d);
    s2 = fmaf(a, a, s2); s2 = fmaf(b, b, s2); s2 = fmaf(cc, cc, s2); s2 = fmaf(d, d, s2);
  }
  block_sum2(s1, s2);
  if (threadIdx.x == 0) {
    part[(size_t)(2 * s) * g.C + c] = s1;
    part[(size_t)(2 * s + 1) * g.C + c] = s2;
  }
}

// PRE: the statistics pass was done by the PRODUCER of x (conv.hip's epilogue): `part` holds `nslots` partial sums
// (sum (x - k), sum (x - k)^2) per channel, [2*slot][C], and the shift k[C] the producer used in row 2*nslots (the
// running mean as it stood BEFORE this kernel updates it, or zeros); they are added in a fixed order (thread t:
// slots t, t + 256, ...; then the block tree).
template <bool RELU, bool RES, bool PRE>
__global__ __launch_bounds__(kBT) void bn_apply_kernel(
    const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ part, Geo g, float eps, float momentum,
    float* __restrict__ rmean, float* __restrict__ rvar, float* __restrict__ stats, float* __restrict__ y, int nslots) {
  const int c = blockIdx.x, s = blockIdx.y;
  float p1, p2;
  if (PRE) {
    p1 = 0.f; p2 = 0.f;
    for (int sl = threadIdx.x; sl < nslots; sl += kBT) {
      p1 += part[(size_t)(2 * sl) * g.C + c];
      p2 += part[(size_t)(2 * sl + 1) * g.C + c];
    }
    block_sum2(p1, p2);
  } else {
    merge_partials(part, g, c, p1, p2);
  }
  const float k = PRE ? part[(size_t)(2 * nslots) * g.C + c] : x[(size_t)c * g.HW];
  const float invM = 1.f / (float)g.M;
  const float m1 = p1 * invM;
  const float var = fmaxf(fmaf(-m1, m1, p2 * invM), 0.f);
  const float mean = k + m1;
  const float invstd = 1.f / sqrtf(var + eps);
  if (s == 0 && threadIdx.x == 0) {
    stats[c] = mean;
    stats[g.C + c] = invstd;
    if (rmean != nullptr) {
      const float unbiased = g.M > 1 ? var * ((float)g.M / (float)(g.M - 1)) : var;
      rmean[c] = fmaf(momentum, mean - rmean[c], rmean[c]);
      rvar[c] = fmaf(momentum, unbiased - rvar[c], rvar[c]);
    }
  }
  const float sc = gamma[c] * invstd;
  const float bt = beta[c];
  const int beg = s * g.per, end = min(g.M, beg + g.per);
#pragma unroll 4
  for (int f = beg + threadIdx.x * 4; f < end; f += kVec) {
    const size_t o = elem_offset(g, c, f);
    const float4 v = *reinterpret_cast<const float4*>(x + o);
    // (x - mean) * (gamma invstd) + beta, ATen's order: the subtraction is exact near the mean, where x * sc + (beta - mean sc)
    // keeps eps * |mean| / std of round-off in the normalised value (r06; 8 such layers in a row in the PointNet++ FP chain)
    float4 r = make_float4(fmaf(v.x - mean, sc, bt), fmaf(v.y - mean, sc, bt), fmaf(v.z - mean, sc, bt), fmaf(v.w - mean, sc, bt));
    if (RES) {
      const float4 q = *reinterpret_cast<const float4*>(res + o);
      r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w;
    }
    if (RELU) { r.x = relu_nan(r.x); r.y = relu_nan(r.y); r.z = relu_nan(r.z); r.w = relu_nan(r.w); }
    *reinterpret_cast<float4*>(y + o) = r;
  }
}

// dy2 (optional): a second gradient of the same tensor -- the sum a residual block's input receives
// (skip path + convolution path) is formed here instead of in a separate add kernel.
template <bool RELU, bool TWO>
__global__ __launch_bounds__(kBT) void bn_bwd_reduce_kernel(
    const float* __restrict__ dy, const float* __restrict__ dy2, const float* __restrict__ x,
    const float* __restrict__ y, const float* __restrict__ stats, Geo g, float* __restrict__ dz,
    float* __restrict__ part) {
  const int c = blockIdx.x, s = blockIdx.y;
  const int beg = s * g.per, end = min(g.M, beg + g.per);
  const float mean = stats[c];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll 4
  for (int f = beg + threadIdx.x * 4; f < end; f += kVec) {
    const size_t o = elem_offset(g, c, f);
    float4 d = *reinterpret_cast<const float4*>(dy + o);
    const float4 v = *reinterpret_cast<const float4*>(x + o);
    if (TWO) {
      const float4 e = *reinterpret_cast<const float4*>(dy2 + o);
      d.x += e.x; d.y += e.y; d.z += e.z; d.w += e.w;
    }
    if (RELU) {
      const float4 out = *reinterpret_cast<const float4*>(y + o);
      d.x = out.x > 0.f ? d.x : 0.f; d.y = out.y > 0.f ? d.y : 0.f;
      d.z = out.z > 0.f ? d.z : 0.f; d.w = out.w > 0.f ? d.w : 0.f;
    }
    if (RELU || TWO) *reinterpret_cast<float4*>(dz + o) = d;
    s1 += (d.x + d.y) + (d.z + d.w);
    s2 = fmaf(d.x, v.x - mean, s2); s2 = fmaf(d.y, v.y - mean, s2);
    s2 = fmaf(d.z, v.z - mean, s2); s2 = fmaf(d.w, v.w - mean, s2);
  }
  block_sum2(s1, s2);
  if (threadIdx.x == 0) {
    part[(size_t)(2 * s) * g.C + c] = s1;
    part[(size_t)(2 * s + 1) * g.C + c] = s2;
  }
}

__global__ __launch_bounds__(kBT) void bn_bwd_apply_kernel(
    const float* __restrict__ dz, const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ stats, const float* __restrict__ part, Geo g, float* __restrict__ gstats,
    float* __restrict__ dx) {
  const int c = blockIdx.x, s = blockIdx.y;
  float p1, p2;
  merge_partials(part, g, c, p1, p2);
  const float mean = stats[c], invstd = stats[g.C + c];
  if (s == 0 && threadIdx.x == 0) {
    gstats[c] = p2 * invstd;     // d gamma
    gstats[g.C + c] = p1;        // d beta
  }
  if (dx == nullptr) return;
  const float invM = 1.f / (float)g.M;
  const float a = gamma[c] * invstd;
  const float b = p1 * invM;
  const float q = p2 * invstd * invstd * invM;
  const int beg = s * g.per, end = min(g.M, beg + g.per);
#pragma unroll 4
  for (int f = beg + threadIdx.x * 4; f < end; f += kVec) {
    const size_t o = elem_offset(g, c, f);
    const float4 d = *reinterpret_cast<const float4*>(dz + o);
    const float4 v = *reinterpret_cast<const float4*>(x + o);
    float4 r;
    r.x = a * (d.x - b - (v.x - mean) * q);
    r.y = a * (d.y - b - (v.y - mean) * q);
    r.z = a * (d.z - b - (v.z - mean) * q);
    r.w = a * (d.w - b - (v.w - mean) * q);
    *reinterpret_cast<float4*>(dx + o) = r;
  }
}

// ---------------------------------------------------------------------------------------------
// Small maps (N*H*W <= 8192 per channel: the two coarse HRNet branches, ~40 % of the layers): one
// workgroup owns a whole channel, its slice lives in registers (<= 2 float4 per thread), so each
// direction is ONE kernel that reads every tensor once.  For these maps a launch is ~4.5 us of
// latency whatever it does, so halving the launch count is the whole gain.  (Extending this to the
// 32768-value channels of the 36-channel branch -- 36 workgroups streaming 128 KB each -- was measured
// slower: 565 vs 586 samples/s.)
// ---------------------------------------------------------------------------------------------
constexpr int kSmallM = 8192;

__device__ __forceinline__ void block_sum2_n(float& a, float& b) {   // up to 16 waves
  __shared__ float sh[2][16];
  a = wave_sum(a);
  b = wave_sum(b);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) { sh[0][w] = a; sh[1][w] = b; }
  __syncthreads();
  float ra = 0.f, rb = 0.f;
  for (int i = 0; i < nw; ++i) { ra += sh[0][i]; rb += sh[1][i]; }
  a = ra; b = rb;
  __syncthreads();
}

template <bool RELU, bool RES>
__global__ __launch_bounds__(1024) void bn_small_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ gamma,
    const float* __restrict__ beta, Geo g, float eps, float momentum, float* __restrict__ rmean,
    float* __restrict__ rvar, float* __restrict__ stats, float* __restrict__ y) {
  const int c = blockIdx.x;
  const float k = x[(size_t)c * g.HW];
  float4 v[2];
  size_t o[2];
  bool ok[2];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int f = (threadIdx.x + i * blockDim.x) * 4;
    ok[i] = f < g.M;
    if (ok[i]) {
      o[i] = elem_offset(g, c, f);
      v[i] = *reinterpret_cast<const float4*>(x + o[i]);
      const float a = v[i].x - k, b = v[i].y - k, cc = v[i].z - k, d = v[i].w - k;
      s1 += (a + b) + (cc + d);
      s2 = fmaf(a, a, s2); s2 = fmaf(b, b, s2); s2 = fmaf(cc, cc, s2); s2 = fmaf(d, d, s2);
    }
  }
  block_sum2_n(s1, s2);
  const float invM = 1.f / (float)g.M;
  const float m1 = s1 * invM;
  const float mean = k + m1;
  // the channel sits in registers: second pass for the variance (sum (x - mean)^2, no E[x^2] - E[x]^2 cancellation);
  // the first-order correction term (sum (x - mean))^2 / M removes what the rounding of `mean` leaves (r06)
  float q1 = 0.f, q2 = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (!ok[i]) continue;
    const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
    q1 += (a + b) + (cc + d);
    q2 = fmaf(a, a, q2); q2 = fmaf(b, b, q2); q2 = fmaf(cc, cc, q2); q2 = fmaf(d, d, q2);
  }
  block_sum2_n(q1, q2);
  const float var = fmaxf((q2 - q1 * q1 * invM) * invM, 0.f);
  const float invstd = 1.f / sqrtf(var + eps);
  if (threadIdx.x == 0) {
    stats[c] = mean;
    stats[g.C + c] = invstd;
    if (rmean != nullptr) {
      const float unbiased = g.M > 1 ? var * ((float)g.M / (float)(g.M - 1)) : var;
      rmean[c] = fmaf(momentum, mean - rmean[c], rmean[c]);
      rvar[c] = fmaf(momentum, unbiased - rvar[c], rvar[c]);
    }
  }
  const float sc = gamma[c] * invstd;
  const float bt = beta[c];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (!ok[i]) continue;
    float4 r = make_float4(fmaf(v[i].x - mean, sc, bt), fmaf(v[i].y - mean, sc, bt), fmaf(v[i].z - mean, sc, bt),
                           fmaf(v[i].w - mean, sc, bt));
    if (RES) {
      const float4 q = *reinterpret_cast<const float4*>(res + o[i]);
      r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w;
    }
    if (RELU) { r.x = relu_nan(r.x); r.y = relu_nan(r.y); r.z = relu_nan(r.z); r.w = relu_nan(r.w); }
    *reinterpret_cast<float4*>(y + o[i]) = r;
  }
}

template <bool RELU, bool TWO>
__global__ __launch_bounds__(1024) void bn_small_bwd_kernel(
    const float* __restrict__ dy, const float* __restrict__ dy2, const float* __restrict__ x,
    const float* __restrict__ y, const float* __restrict__ gamma, const float* __restrict__ stats, Geo g,
    float* __restrict__ dz, float* __restrict__ dx, float* __restrict__ gstats) {
  const int c = blockIdx.x;
  const float mean = stats[c], invstd = stats[g.C + c];
  float4 d[2], v[2];
  size_t o[2];
  bool ok[2];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int f = (threadIdx.x + i * blockDim.x) * 4;
    ok[i] = f < g.M;
    if (ok[i]) {
      o[i] = elem_offset(g, c, f);
      d[i] = *reinterpret_cast<const float4*>(dy + o[i]);
      v[i] = *reinterpret_cast<const float4*>(x + o[i]);
      if (TWO) {
        const float4 e = *reinterpret_cast<const float4*>(dy2 + o[i]);
        d[i].x += e.x; d[i].y += e.y; d[i].z += e.z; d[i].w += e.w;
      }
      if (RELU) {
        const float4 out = *reinterpret_cast<const float4*>(y + o[i]);
        d[i].x = out.x > 0.f ? d[i].x : 0.f; d[i].y = out.y > 0.f ? d[i].y : 0.f;
        d[i].z = out.z > 0.f ? d[i].z : 0.f; d[i].w = out.w > 0.f ? d[i].w : 0.f;
      }
      if (RELU || TWO) *reinterpret_cast<float4*>(dz + o[i]) = d[i];
      s1 += (d[i].x + d[i].y) + (d[i].z + d[i].w);
      s2 = fmaf(d[i].x, v[i].x - mean, s2); s2 = fmaf(d[i].y, v[i].y - mean, s2);
      s2 = fmaf(d[i].z, v[i].z - mean, s2); s2 = fmaf(d[i].w, v[i].w - mean, s2);
    }
  }
  block_sum2_n(s1, s2);
  if (threadIdx.x == 0) {
    gstats[c] = s2 * invstd;     // d gamma
    gstats[g.C + c] = s1;        // d beta
  }
  if (dx == nullptr) return;
  const float invM = 1.f / (float)g.M;
  const float a = gamma[c] * invstd;
  const float b = s1 * invM;
  const float q = s2 * invstd * invstd * invM;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (!ok[i]) continue;
    float4 r;
    r.x = a * (d[i].x - b - (v[i].x - mean) * q);
    r.y = a * (d[i].y - b - (v[i].y - mean) * q);
    r.z = a * (d[i].z - b - (v[i].z - mean) * q);
    r.w = a * (d[i].w - b - (v[i].w - mean) * q);
    *reinterpret_cast<float4*>(dx + o[i]) = r;
  }
}

inline bool small_map(const Geo& g) { return g.M <= kSmallM; }
inline int small_threads(const Geo& g) {           // each thread holds up to two float4 of the channel
  int t = ((g.M / 4 + 1) / 2 + 63) / 64 * 64;
  if (t < 64) t = 64;
  if (t > 1024) t = 1024;
  return t;
}

bool bad_shape(int N, int C, int HW) {
  return N <= 0 || C <= 0 || HW <= 0 || (HW & 3) != 0 || (long long)N * HW > 0x7fffffffLL - kVec ||
         C > 65535;
}

// ---------------------------------------------------------------------------------------------
// BatchNorm2d + ReLU + max over the ball (r05).  The last layer of every PointNet++ SharedMLP is followed by
// F.max_pool2d(y, [1, nsample]) (networks/pointnet2/pointnet2_modules.py:44-55 of the reference): of the [B, C, npoint,
// nsample] tensor y = relu(bn(z)) only one value in nsample is ever used, forward or backward, so y is not written at
// all.  Forward = the statistics pass over z (bn_stats_kernel) + ONE pass that normalises in registers and keeps, per
// (n, c, ball), the maximum of y, the FIRST index that attains it (ATen's max_pool2d rule, evaluated on y itself:
// ties among clamped zeros included) and the z value there.  Backward: the gradient that reaches y is non-zero at ONE
// element per ball, so the two sums of the BatchNorm backward come from the [N, C, npoint] tensors alone and one pass
// over z writes dz.  4 passes over z-sized tensors where stats + apply + max + max-backward + reduce + apply take 12
// (0.27 - 1.07 GB each at the BASELINE config-4 shapes).
// ---------------------------------------------------------------------------------------------
template <int L>      // lanes per ball row = nsample / 4 (1, 2, 4, 8, 16)
__global__ __launch_bounds__(kBT) void bn_relu_ballmax_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ part, Geo g, float eps, float momentum, float* __restrict__ rmean, float* __restrict__ rvar,
    float* __restrict__ stats, float* __restrict__ out, int* __restrict__ arg, float* __restrict__ zsel, int np) {
  const int c = blockIdx.x, s = blockIdx.y;
  float p1, p2;
  merge_partials(part, g, c, p1, p2);
  const float k = x[(size_t)c * g.HW];
  const float invM = 1.f / (float)g.M;
  const float m1 = p1 * invM;
  const float var = fmaxf(fmaf(-m1, m1, p2 * invM), 0.f);
  const float mean = k + m1;
  const float invstd = 1.f / sqrtf(var + eps);
  if (s == 0 && threadIdx.x == 0) {
    stats[c] = mean;
    stats[g.C + c] = invstd;
    if (rmean != nullptr) {
      const float unbiased = g.M > 1 ? var * ((float)g.M / (float)(g.M - 1)) : var;
      rmean[c] = fmaf(momentum, mean - rmean[c], rmean[c]);
      rvar[c] = fmaf(momentum, unbiased - rvar[c], rvar[c]);
    }
  }
  const float sc = gamma[c] * invstd;
  const float bt = beta[c];
  constexpr int ns = 4 * L;
  const int beg = s * g.per, end = min(g.M, beg + g.per);
  // Four float4 per thread and trip, requested together and UNCONDITIONALLY (r06): a guarded load is a branch and a full
  // s_waitcnt, and one 16-byte load per wave and trip left the latency cover to occupancy alone (0.49 of the HBM rate in the
  // r06 bench line).  A quad past the slice is clamped to the slice's last one; it is computed and never stored.
  constexpr int U = 4;
  for (int f0 = beg; f0 < end; f0 += U * kVec) {
    float4 v[U];
    int nn[U], ww[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int f = f0 + u * kVec + threadIdx.x * 4;
      ok[u] = f < end;                             // a ball row never straddles `end` (per, M are multiples of ns)
      const int fc = ok[u] ? f : end - 4;
      nn[u] = g.shift >= 0 ? (fc >> g.shift) : (fc / g.HW);
      ww[u] = fc - nn[u] * g.HW;                   // offset inside the (n, c) plane
      v[u] = *reinterpret_cast<const float4*>(x + ((size_t)nn[u] * g.C + c) * (size_t)g.HW + ww[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float y0 = relu_nan(fmaf(v[u].x - mean, sc, bt)), y1 = relu_nan(fmaf(v[u].y - mean, sc, bt));
      const float y2 = relu_nan(fmaf(v[u].z - mean, sc, bt)), y3 = relu_nan(fmaf(v[u].w - mean, sc, bt));
      const int j0 = ww[u] & (ns - 1);
      float bv = y0, bz = v[u].x;
      int bj = j0;
      // ATen's max_pool2d: a later value wins if it is larger OR NaN while the running maximum is not (a NaN sticks)
      if (y1 > bv || (y1 != y1 && bv == bv)) { bv = y1; bz = v[u].y; bj = j0 + 1; }
      if (y2 > bv || (y2 != y2 && bv == bv)) { bv = y2; bz = v[u].z; bj = j0 + 2; }
      if (y3 > bv || (y3 != y3 && bv == bv)) { bv = y3; bz = v[u].w; bj = j0 + 3; }
#pragma unroll
      for (int off = 1; off < L; off <<= 1) {      // the L lanes of a row are adjacent and aligned
        const float ov = __shfl_xor(bv, off);
        const float oz = __shfl_xor(bz, off);
        const int oj = __shfl_xor(bj, off);
        const bool onan = ov != ov, bnan = bv != bv;
        const bool take = (onan && !bnan) || (onan == bnan && (ov > bv || ((ov == bv || onan) && oj < bj)));
        bv = take ? ov : bv; bz = take ? oz : bz; bj = take ? oj : bj;
      }
      if (ok[u] && (threadIdx.x & (L - 1)) == 0) {
        const size_t r = ((size_t)nn[u] * g.C + c) * (size_t)np + (size_t)(ww[u] / ns);
        out[r] = bv;
        arg[r] = bj;
        zsel[r] = bz;
      }
    }
  }
}

// sums of the BatchNorm backward from the sparse gradient: g = dout where out > 0 (ReLU active at the maximum), else 0;
// S1 = sum g, S2 = sum g (z_sel - mean), over the N * npoint balls of channel c.  Same slice / partial layout as the
// dense kernels (Geo of the [N, C, npoint] tensors).
__global__ __launch_bounds__(kBT) void ballmax_bwd_reduce_kernel(
    const float* __restrict__ dout, const float* __restrict__ out, const float* __restrict__ zsel,
    const float* __restrict__ stats, Geo gr, float* __restrict__ part) {
  const int c = blockIdx.x, s = blockIdx.y;
  const int beg = s * gr.per, end = min(gr.M, beg + gr.per);
  const float mean = stats[c];
  float s1 = 0.f, s2 = 0.f;
  for (int f = beg + threadIdx.x * 4; f < end; f += kVec) {
    const size_t o = elem_offset(gr, c, f);
    const float4 d = *reinterpret_cast<const float4*>(dout + o);
    const float4 y = *reinterpret_cast<const float4*>(out + o);
    const float4 z = *reinterpret_cast<const float4*>(zsel + o);
    const float g0 = y.x > 0.f ? d.x : 0.f, g1 = y.y > 0.f ? d.y : 0.f;
    const float g2 = y.z > 0.f ? d.z : 0.f, g3 = y.w > 0.f ? d.w : 0.f;
    s1 += (g0 + g1) + (g2 + g3);
    s2 = fmaf(g0, z.x - mean, s2); s2 = fmaf(g1, z.y - mean, s2);
    s2 = fmaf(g2, z.z - mean, s2); s2 = fmaf(g3, z.w - mean, s2);
  }
  block_sum2(s1, s2);
  if (threadIdx.x == 0) {
    part[(size_t)(2 * s) * gr.C + c] = s1;
    part[(size_t)(2 * s + 1) * gr.C + c] = s2;
  }
}

// dz[n, c, i, j] = gamma invstd ( g[n, c, i] [j == arg] - mean(g) - xhat mean(g xhat) ): one pass over z
template <int L>
__global__ __launch_bounds__(kBT) void ballmax_bwd_apply_kernel(
    const float* __restrict__ dout, const float* __restrict__ out, const int* __restrict__ arg,
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ stats,
    const float* __restrict__ part, Geo g, int split_r, int np, float* __restrict__ gstats, float* __restrict__ dz) {
  const int c = blockIdx.x, s = blockIdx.y;
  Geo gr = g;
  gr.split = split_r;
  float p1, p2;
  merge_partials(part, gr, c, p1, p2);
  const float mean = stats[c], invstd = stats[g.C + c];
  if (s == 0 && threadIdx.x == 0) {
    gstats[c] = p2 * invstd;     // d gamma
    gstats[g.C + c] = p1;        // d beta
  }
  const float invM = 1.f / (float)g.M;
  const float a = gamma[c] * invstd;
  const float b = p1 * invM;
  const float q = p2 * invstd * invstd * invM;
  constexpr int ns = 4 * L;
  const int beg = s * g.per, end = min(g.M, beg + g.per);
#pragma unroll 2
  for (int f = beg + threadIdx.x * 4; f < end; f += kVec) {
    const int n = g.shift >= 0 ? (f >> g.shift) : (f / g.HW);
    const int w = f - n * g.HW;
    const size_t o = ((size_t)n * g.C + c) * (size_t)g.HW + w;
    const size_t r = ((size_t)n * g.C + c) * (size_t)np + (size_t)(w / ns);
    const float4 v = *reinterpret_cast<const float4*>(x + o);
    const float gsel = out[r] > 0.f ? dout[r] : 0.f;
    const int js = arg[r] - (w & (ns - 1));          // 0..3 when the maximum sits in this thread's four values
    float4 d;
    d.x = a * ((js == 0 ? gsel : 0.f) - b - (v.x - mean) * q);
    d.y = a * ((js == 1 ? gsel : 0.f) - b - (v.y - mean) * q);
    d.z = a * ((js == 2 ? gsel : 0.f) - b - (v.z - mean) * q);
    d.w = a * ((js == 3 ? gsel : 0.f) - b - (v.w - mean) * q);
    *reinterpret_cast<float4*>(dz + o) = d;
  }
}


// ---------------------------------------------------------------------------------------------
// First layer of a PointNet++ SharedMLP without the grouped tensor (r05; arithmetic restated in r06).  QueryAndGroup builds
// x[b, :, i, j] = [xyz[idx] - centre_i ; features[idx]] ([B, 3 + C, npoint, nsample], up to 415 MB) and the first 1x1
// convolution multiplies it by W = [W_xyz | W_f] (networks/pointnet2/pointnet2_utils.py:231-268, pytorch_utils.py:5-33 of the
// reference).  The FEATURE half of the convolution commutes with the gather:  W_f features[idx] = P[b, :, idx[b, i, j]] with
// P = W_f features ([B, C1, N]: nsample x fewer products, hcm_conv1x1_forward on the host side).  The COORDINATE half is
// evaluated here from the relative offsets D[b, :, i, j] = xyz[idx] - centre_i ([B, 3, np, ns], the reference's own
// grouped_xyz, 3 channels: geometry, computed ahead of the feature path):
//     z[b, c, i, j] = P[b, c, idx[b, i, j]] + W_xyz[c, :] . D[b, :, i, j].
// r05 also commuted the coordinate half (z = P'[idx] - Q, P' = W [xyz ; features], Q = W_xyz centre): a difference of two
// O(|xyz|) numbers whose value is O(radius) -- 40 x the round-off of the reference's subtract-then-multiply for 2.5 cm balls in
// a 1 m cloud, 50 x the reference's own fp32 error on the first level's output (tests/test_pn_reference_gpu.py, r06).
// These kernels take z as that implicit tensor: statistics and normalisation read it through the gather (P's row of one
// (b, c) is N floats: cache resident) plus three coalesced reads of D; the backward hands dz out once for the planned scatter
// into dP and reduces dW_xyz[c, :] = sum dz D in the same pass.  Neither x nor z nor their gradients exist in memory.
// ---------------------------------------------------------------------------------------------
struct BallGeo {
  int N, np, ns;      // source points, centres, ball size
};

// z of four consecutive ball members of (n, c): o = n * HW + w indexes idx, Dn = D + n * 3 * HW
template <bool HASP>
__device__ __forceinline__ float4 ball_z(const float* __restrict__ Prow, const int* __restrict__ idx, size_t o,
                                         const float* __restrict__ Dn, int HW, int w, float w0, float w1, float w2) {
  const float4 d0 = *reinterpret_cast<const float4*>(Dn + w);
  const float4 d1 = *reinterpret_cast<const float4*>(Dn + (size_t)HW + w);
  const float4 d2 = *reinterpret_cast<const float4*>(Dn + 2 * (size_t)HW + w);
  float4 z = make_float4(fmaf(w2, d2.x, fmaf(w1, d1.x, w0 * d0.x)), fmaf(w2, d2.y, fmaf(w1, d1.y, w0 * d0.y)),
                         fmaf(w2, d2.z, fmaf(w1, d1.z, w0 * d0.z)), fmaf(w2, d2.w, fmaf(w1, d1.w, w0 * d0.w)));
  if (HASP) {
    const int4 id = *reinterpret_cast<const int4*>(idx + o);
    z.x += Prow[id.x]; z.y += Prow[id.y]; z.z += Prow[id.z]; z.w += Prow[id.w];
  }
  return z;
}

// z of the channel's very first element (the shift of the one-pass variance)
template <bool HASP>
__device__ __forceinline__ float ball_z0(const float* __restrict__ P, const int* __restrict__ idx, const float* __restrict__ D,
                                         int c, int N, int HW, float w0, float w1, float w2) {
  const float z = fmaf(w2, D[2 * (size_t)HW], fmaf(w1, D[(size_t)HW], w0 * D[0]));
  return HASP ? z + P[(size_t)c * N + idx[0]] : z;
}

// slice sums of (z - k), (z - k)^2; k = z of the channel's first element.  Geo g describes [B, C, np * ns].
template <bool HASP>
__global__ __launch_bounds__(kBT) void ball_stats_kernel(const float* __restrict__ P, const float* __restrict__ D,
                                                         const float* __restrict__ Wxyz, const int* __restrict__ idx, Geo g,
                                                         BallGeo bg, float* __restrict__ part) {
  const int c = blockIdx.x, s = blockIdx.y;
  const int beg = s * g.per, end = min(g.M, beg + g.per);
  const float w0 = Wxyz[3 * c], w1 = Wxyz[3 * c + 1], w2 = Wxyz[3 * c + 2];
  const float k = ball_z0<HASP>(P, idx, D, c, bg.N, g.HW, w0, w1, w2);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll 2
  for (int f = beg + threadIdx.x * 4; f < end; f += kVec) {
    const int n = g.shift >= 0 ? (f >> g.shift) : (f / g.HW);
    const int w = f - n * g.HW;
    const float4 v = ball_z<HASP>(HASP ? P + ((size_t)n * g.C + c) * bg.N : nullptr, idx, (size_t)n * g.HW + w,
                                  D + (size_t)n * 3 * g.HW, g.HW, w, w0, w1, w2);
    const float a = v.x - k, b = v.y - k, cc = v.z - k, d = v.w - k;
    s1 += (a + b) + (cc + d);
    s2 = fmaf(a, a, s2); s2 = fmaf(b, b, s2); s2 = fmaf(cc, cc, s2); s2 = fmaf(d, d, s2);
  }
  block_sum2(s1, s2);
  if (threadIdx.x == 0) {
    part[(size_t)(2 * s) * g.C + c] = s1;
    part[(size_t)(2 * s + 1) * g.C + c] = s2;
  }
}

template <bool RELU, bool HASP>
__global__ __launch_bounds__(kBT) void ball_apply_kernel(
    const float* __restrict__ P, const float* __restrict__ D, const float* __restrict__ Wxyz, const int* __restrict__ idx,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ part, Geo g, BallGeo bg,
    float eps, float momentum, float* __restrict__ rmean, float* __restrict__ rvar, float* __restrict__ stats,
    float* __restrict__ y) {
  const int c = blockIdx.x, s = blockIdx.y;
  float p1, p2;
  merge_partials(part, g, c, p1, p2);
  const float w0 = Wxyz[3 * c], w1 = Wxyz[3 * c + 1], w2 = Wxyz[3 * c + 2];
  const float k = ball_z0<HASP>(P, idx, D, c, bg.N, g.HW, w0, w1, w2);
  const float invM = 1.f / (float)g.M;
  const float m1 = p1 * invM;
  const float var = fmaxf(fmaf(-m1, m1, p2 * invM), 0.f);
  const float mean = k + m1;
  const float invstd = 1.f / sqrtf(var + eps);
  if (s == 0 && threadIdx.x == 0) {
    stats[c] = mean;
    stats[g.C + c] = invstd;
    if (rmean != nullptr) {
      const float unbiased = g.M > 1 ? var * ((float)g.M / (float)(g.M - 1)) : var;
      rmean[c] = fmaf(momentum, mean - rmean[c], rmean[c]);
      rvar[c] = fmaf(momentum, unbiased - rvar[c], rvar[c]);
    }
  }
  const float sc = gamma[c] * invstd;
  const float bt = beta[c];
  const int beg = s * g.per, end = min(g.M, beg + g.per);
#pragma unroll 2
  for (int f = beg + threadIdx.x * 4; f < end; f += kVec) {
    const int n = g.shift >= 0 ? (f >> g.shift) : (f / g.HW);
    const int w = f - n * g.HW;
    const float4 v = ball_z<HASP>(HASP ? P + ((size_t)n * g.C + c) * bg.N : nullptr, idx, (size_t)n * g.HW + w,
                                  D + (size_t)n * 3 * g.HW, g.HW, w, w0, w1, w2);
    // (z - mean) first: exact where z is close to the mean, no eps * |mean| / std left in the normalised value
    float4 r = make_float4(fmaf(v.x - mean, sc, bt), fmaf(v.y - mean, sc, bt), fmaf(v.z - mean, sc, bt), fmaf(v.w - mean, sc, bt));
    if (RELU) { r.x = relu_nan(r.x); r.y = relu_nan(r.y); r.z = relu_nan(r.z); r.w = relu_nan(r.w); }
    *reinterpret_cast<float4*>(y + ((size_t)n * g.C + c) * (size_t)g.HW + w) = r;
  }
}

template <bool RELU, bool HASP>
__global__ __launch_bounds__(kBT) void ball_bwd_reduce_kernel(
    const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ P, const float* __restrict__ D,
    const float* __restrict__ Wxyz, const int* __restrict__ idx, const float* __restrict__ stats, Geo g, BallGeo bg,
    float* __restrict__ part) {
  const int c = blockIdx.x, s = blockIdx.y;
  const int beg = s * g.per, end = min(g.M, beg + g.per);
  const float mean = stats[c];
  const float w0 = Wxyz[3 * c], w1 = Wxyz[3 * c + 1], w2 = Wxyz[3 * c + 2];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll 2
  for (int f = beg + threadIdx.x * 4; f < end; f += kVec) {
    const int n = g.shift >= 0 ? (f >> g.shift) : (f / g.HW);
    const int w = f - n * g.HW;
    const size_t o = ((size_t)n * g.C + c) * (size_t)g.HW + w;
    float4 d = *reinterpret_cast<const float4*>(dy + o);
    if (RELU) {
      const float4 out = *reinterpret_cast<const float4*>(y + o);
      d.x = out.x > 0.f ? d.x : 0.f; d.y = out.y > 0.f ? d.y : 0.f;
      d.z = out.z > 0.f ? d.z : 0.f; d.w = out.w > 0.f ? d.w : 0.f;
    }
    const float4 v = ball_z<HASP>(HASP ? P + ((size_t)n * g.C + c) * bg.N : nullptr, idx, (size_t)n * g.HW + w,
                                  D + (size_t)n * 3 * g.HW, g.HW, w, w0, w1, w2);
    s1 += (d.x + d.y) + (d.z + d.w);
    s2 = fmaf(d.x, v.x - mean, s2); s2 = fmaf(d.y, v.y - mean, s2);
    s2 = fmaf(d.z, v.z - mean, s2); s2 = fmaf(d.w, v.w - mean, s2);
  }
  block_sum2(s1, s2);
  if (threadIdx.x == 0) {
    part[(size_t)(2 * s) * g.C + c] = s1;
    part[(size_t)(2 * s + 1) * g.C + c] = s2;
  }
}

// dz (for the scatter into dP) and this slice's share of dW_xyz[c, :] = sum dz D  ->  wpart[s][c][3]
template <bool RELU, bool HASP>
__global__ __launch_bounds__(kBT) void ball_bwd_apply_kernel(
    const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ P, const float* __restrict__ D,
    const float* __restrict__ Wxyz, const int* __restrict__ idx, const float* __restrict__ gamma,
    const float* __restrict__ stats, const float* __restrict__ part, Geo g, BallGeo bg, float* __restrict__ gstats,
    float* __restrict__ dz, float* __restrict__ wpart) {
  const int c = blockIdx.x, s = blockIdx.y;
  float p1, p2;
  merge_partials(part, g, c, p1, p2);
  const float mean = stats[c], invstd = stats[g.C + c];
  if (s == 0 && threadIdx.x == 0) {
    gstats[c] = p2 * invstd;     // d gamma
    gstats[g.C + c] = p1;        // d beta
  }
  const float w0 = Wxyz[3 * c], w1 = Wxyz[3 * c + 1], w2 = Wxyz[3 * c + 2];
  const float invM = 1.f / (float)g.M;
  const float a = gamma[c] * invstd;
  const float b = p1 * invM;
  const float q = p2 * invstd * invstd * invM;
  const int beg = s * g.per, end = min(g.M, beg + g.per);
  float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll 2
  for (int f = beg + threadIdx.x * 4; f < end; f += kVec) {
    const int n = g.shift >= 0 ? (f >> g.shift) : (f / g.HW);
    const int w = f - n * g.HW;
    const size_t o = ((size_t)n * g.C + c) * (size_t)g.HW + w;
    float4 d = *reinterpret_cast<const float4*>(dy + o);
    if (RELU) {
      const float4 out = *reinterpret_cast<const float4*>(y + o);
      d.x = out.x > 0.f ? d.x : 0.f; d.y = out.y > 0.f ? d.y : 0.f;
      d.z = out.z > 0.f ? d.z : 0.f; d.w = out.w > 0.f ? d.w : 0.f;
    }
    const float* Dn = D + (size_t)n * 3 * g.HW;
    const float4 v = ball_z<HASP>(HASP ? P + ((size_t)n * g.C + c) * bg.N : nullptr, idx, (size_t)n * g.HW + w, Dn, g.HW, w,
                                  w0, w1, w2);
    float4 r;
    r.x = a * (d.x - b - (v.x - mean) * q);
    r.y = a * (d.y - b - (v.y - mean) * q);
    r.z = a * (d.z - b - (v.z - mean) * q);
    r.w = a * (d.w - b - (v.w - mean) * q);
    *reinterpret_cast<float4*>(dz + o) = r;
    const float4 d0 = *reinterpret_cast<const float4*>(Dn + w);
    const float4 d1 = *reinterpret_cast<const float4*>(Dn + (size_t)g.HW + w);
    const float4 d2 = *reinterpret_cast<const float4*>(Dn + 2 * (size_t)g.HW + w);
    t0 = fmaf(r.x, d0.x, t0); t0 = fmaf(r.y, d0.y, t0); t0 = fmaf(r.z, d0.z, t0); t0 = fmaf(r.w, d0.w, t0);
    t1 = fmaf(r.x, d1.x, t1); t1 = fmaf(r.y, d1.y, t1); t1 = fmaf(r.z, d1.z, t1); t1 = fmaf(r.w, d1.w, t1);
    t2 = fmaf(r.x, d2.x, t2); t2 = fmaf(r.y, d2.y, t2); t2 = fmaf(r.z, d2.z, t2); t2 = fmaf(r.w, d2.w, t2);
  }
  __shared__ float sh3[3][kBT / 64];
  t0 = wave_sum(t0); t1 = wave_sum(t1); t2 = wave_sum(t2);
  if ((threadIdx.x & 63) == 0) { sh3[0][threadIdx.x >> 6] = t0; sh3[1][threadIdx.x >> 6] = t1; sh3[2][threadIdx.x >> 6] = t2; }
  __syncthreads();
  if (threadIdx.x < 3)
    wpart[((size_t)s * g.C + c) * 3 + threadIdx.x] =
        (sh3[threadIdx.x][0] + sh3[threadIdx.x][1]) + (sh3[threadIdx.x][2] + sh3[threadIdx.x][3]);
}

// dW_xyz[c, d] = sum over the slices, in slice order (one thread per (c, d))
__global__ void ball_wxyz_merge_kernel(const float* __restrict__ wpart, int C, int split, float* __restrict__ dW) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 3 * C) return;
  float acc = 0.f;
  for (int s = 0; s < split; ++s) acc += wpart[(size_t)s * 3 * C + t];
  dW[t] = acc;
}

// ---------------------------------------------------------------------------------------------
// The first SA level has NO point features (P == nullptr, z = W_xyz D): the two largest tensors of the network (y: 134 + 537 MB
// at B = 32) and nothing to gather.  Two specialisations (r06):
//
// FORWARD, all channels per workgroup.  The (channel, slice) kernels above re-read D for every channel: C x 50 MB through L2 per
// pass for a 50 MB tensor (C = 16 / 32), twice.  Here a workgroup owns a run of ball members, reads its D ONCE and loops over
// the CC channels in registers: z is three FMAs.  ball_all_stats_kernel leaves per-block partial sums (2 CC accumulators per
// thread, summed through LDS), ball_all_finish_kernel merges them in block order -> mean / invstd / running statistics,
// ball_all_apply_kernel normalises and writes y: per pass D is read once, y written once = the algorithmic bytes.
//
// BACKWARD, one pass and no dz.  dz exists to be scattered into dP; without point features nothing needs it.  What is left --
// dgamma, dbeta, dW_xyz -- are sums: with d = dy [y > 0], zc = z - mean,
//     dbeta = S1 = sum d,   dgamma = invstd S2,  S2 = sum d zc,
//     dW_xyz[c, k] = sum r D_k  with  r = a (d - b - q zc),  a = gamma invstd, b = S1 / M, q = S2 invstd^2 / M
//                  = a (A_k - b B_k - q C_k),   A_k = sum d D_k,  B_k = sum D_k,  C_k = sum zc D_k
// so ONE pass over (dy, y, D) accumulates S1, S2, A, C per channel (and B once), and a finish kernel combines them: 2 M floats
// read instead of 4 M read + M written (ball_bwd_reduce + ball_bwd_apply), partial sums merged in slice order (deterministic).
// ---------------------------------------------------------------------------------------------
constexpr int kAllBlocks = 512;      // forward workgroups (partials merged by one block of the finish kernel)

template <int CC>
__global__ __launch_bounds__(kBT) void ball_all_stats_kernel(const float* __restrict__ D, const float* __restrict__ Wxyz,
                                                             int HW, int total4, float* __restrict__ part) {
  // total4 = B * HW / 4 float4 positions; position p -> image n = 4p / HW, offset w = 4p - n HW
  __shared__ float red[2 * CC][kBT / 64];
  __shared__ float4 wk[CC];                     // (w0, w1, w2, shift) per channel: LDS broadcast reads instead of 128 SGPRs
  float s1[CC], s2[CC];
  // shift of the one-pass variance: z of element 0 (any value near the mean does)
  if (threadIdx.x < CC) {
    const int c = threadIdx.x;
    const float w0 = Wxyz[3 * c], w1 = Wxyz[3 * c + 1], w2 = Wxyz[3 * c + 2];
    wk[c] = make_float4(w0, w1, w2, fmaf(w2, D[2 * (size_t)HW], fmaf(w1, D[(size_t)HW], w0 * D[0])));
  }
#pragma unroll
  for (int c = 0; c < CC; ++c) { s1[c] = 0.f; s2[c] = 0.f; }
  __syncthreads();
  for (int p = blockIdx.x * kBT + threadIdx.x; p < total4; p += gridDim.x * kBT) {
    const int f = 4 * p, n = f / HW, w = f - n * HW;
    const float* Dn = D + (size_t)n * 3 * HW + w;
    const float4 d0 = *reinterpret_cast<const float4*>(Dn);
    const float4 d1 = *reinterpret_cast<const float4*>(Dn + (size_t)HW);
    const float4 d2 = *reinterpret_cast<const float4*>(Dn + 2 * (size_t)HW);
#pragma unroll
    for (int c = 0; c < CC; ++c) {
      const float4 q = wk[c];
      const float a = fmaf(q.z, d2.x, fmaf(q.y, d1.x, q.x * d0.x)) - q.w, b = fmaf(q.z, d2.y, fmaf(q.y, d1.y, q.x * d0.y)) - q.w;
      const float e = fmaf(q.z, d2.z, fmaf(q.y, d1.z, q.x * d0.z)) - q.w, h = fmaf(q.z, d2.w, fmaf(q.y, d1.w, q.x * d0.w)) - q.w;
      s1[c] += (a + b) + (e + h);
      s2[c] = fmaf(a, a, s2[c]); s2[c] = fmaf(b, b, s2[c]); s2[c] = fmaf(e, e, s2[c]); s2[c] = fmaf(h, h, s2[c]);
    }
  }
#pragma unroll
  for (int c = 0; c < CC; ++c) {
    const float a = wave_sum(s1[c]), b = wave_sum(s2[c]);
    if ((threadIdx.x & 63) == 0) { red[2 * c][threadIdx.x >> 6] = a; red[2 * c + 1][threadIdx.x >> 6] = b; }
  }
  __syncthreads();
  if (threadIdx.x < 2 * CC)
    part[(size_t)blockIdx.x * 2 * CC + threadIdx.x] =
        (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// one workgroup (two waves) per channel: wave 0 merges the sum partials, wave 1 the square partials, each lane its blocks in
// order, in DOUBLE (1024 partials added one after the other in fp32 cost 2e-6 of the mean offset: r06, first version), then a
// fixed-order lane tree -> stats = [mean C][invstd C], running statistics
__global__ __launch_bounds__(128) void ball_all_finish_kernel(const float* __restrict__ part, int nblocks, int C,
                                                              const float* __restrict__ D, const float* __restrict__ Wxyz, int HW,
                                                              float M, float eps, float momentum, float* __restrict__ rmean,
                                                              float* __restrict__ rvar, float* __restrict__ stats) {
  __shared__ double sh[2];
  const int c = blockIdx.x, lane = threadIdx.x & 63, which = threadIdx.x >> 6;
  double acc = 0.0;
  for (int b = lane; b < nblocks; b += 64) acc += (double)part[(size_t)b * 2 * C + 2 * c + which];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) sh[which] = acc;
  __syncthreads();
  if (threadIdx.x != 0) return;
  const double invM = 1.0 / (double)M;
  const double m1 = sh[0] * invM;
  double var = sh[1] * invM - m1 * m1;
  var = var > 0.0 ? var : 0.0;
  const float k = fmaf(Wxyz[3 * c + 2], D[2 * (size_t)HW], fmaf(Wxyz[3 * c + 1], D[(size_t)HW], Wxyz[3 * c] * D[0]));
  const float mean = (float)((double)k + m1), varf = (float)var;
  stats[c] = mean;
  stats[C + c] = 1.f / sqrtf(varf + eps);
  if (rmean != nullptr) {
    const float unbiased = M > 1.f ? varf * (M / (M - 1.f)) : varf;
    rmean[c] = fmaf(momentum, mean - rmean[c], rmean[c]);
    rvar[c] = fmaf(momentum, unbiased - rvar[c], rvar[c]);
  }
}

template <int CC, bool RELU>
__global__ __launch_bounds__(kBT) void ball_all_apply_kernel(const float* __restrict__ D, const float* __restrict__ Wxyz,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ stats, int HW, int total4,
                                                             float* __restrict__ y) {
  __shared__ float4 wk[CC];                     // (w0, w1, w2, mean)
  __shared__ float2 sb[CC];                     // (gamma invstd, beta)
  if (threadIdx.x < CC) {
    const int c = threadIdx.x;
    wk[c] = make_float4(Wxyz[3 * c], Wxyz[3 * c + 1], Wxyz[3 * c + 2], stats[c]);
    sb[c] = make_float2(gamma[c] * stats[CC + c], beta[c]);
  }
  __syncthreads();
  for (int p = blockIdx.x * kBT + threadIdx.x; p < total4; p += gridDim.x * kBT) {
    const int f = 4 * p, n = f / HW, w = f - n * HW;
    const float* Dn = D + (size_t)n * 3 * HW + w;
    const float4 d0 = *reinterpret_cast<const float4*>(Dn);
    const float4 d1 = *reinterpret_cast<const float4*>(Dn + (size_t)HW);
    const float4 d2 = *reinterpret_cast<const float4*>(Dn + 2 * (size_t)HW);
    float* yn = y + (size_t)n * CC * HW + w;
#pragma unroll 8
    for (int c = 0; c < CC; ++c) {
      const float4 q = wk[c];
      const float2 t = sb[c];
      float4 r;
      r.x = fmaf(fmaf(q.z, d2.x, fmaf(q.y, d1.x, q.x * d0.x)) - q.w, t.x, t.y);
      r.y = fmaf(fmaf(q.z, d2.y, fmaf(q.y, d1.y, q.x * d0.y)) - q.w, t.x, t.y);
      r.z = fmaf(fmaf(q.z, d2.z, fmaf(q.y, d1.z, q.x * d0.z)) - q.w, t.x, t.y);
      r.w = fmaf(fmaf(q.z, d2.w, fmaf(q.y, d1.w, q.x * d0.w)) - q.w, t.x, t.y);
      if (RELU) { r.x = relu_nan(r.x); r.y = relu_nan(r.y); r.z = relu_nan(r.z); r.w = relu_nan(r.w); }
      *reinterpret_cast<float4*>(yn + (size_t)c * HW) = r;
    }
  }
}

// backward without point features: ONE pass, per (channel, slice) partial sums [S1, S2, A0..2, C0..2] (+ B0..2 by channel 0)
template <bool RELU>
__global__ __launch_bounds__(kBT) void ball_bwd_nop_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                           const float* __restrict__ D, const float* __restrict__ Wxyz,
                                                           const float* __restrict__ stats, Geo g, float* __restrict__ part8,
                                                           float* __restrict__ partB) {
  const int c = blockIdx.x, s = blockIdx.y;
  const int beg = s * g.per, end = min(g.M, beg + g.per);
  const float mean = stats[c];
  const float w0 = Wxyz[3 * c], w1 = Wxyz[3 * c + 1], w2 = Wxyz[3 * c + 2];
  float acc[11];
#pragma unroll
  for (int k = 0; k < 11; ++k) acc[k] = 0.f;
#pragma unroll 2
  for (int f = beg + threadIdx.x * 4; f < end; f += kVec) {
    const int n = g.shift >= 0 ? (f >> g.shift) : (f / g.HW);
    const int w = f - n * g.HW;
    const size_t o = ((size_t)n * g.C + c) * (size_t)g.HW + w;
    float4 d = *reinterpret_cast<const float4*>(dy + o);
    if (RELU) {
      const float4 out = *reinterpret_cast<const float4*>(y + o);
      d.x = out.x > 0.f ? d.x : 0.f; d.y = out.y > 0.f ? d.y : 0.f;
      d.z = out.z > 0.f ? d.z : 0.f; d.w = out.w > 0.f ? d.w : 0.f;
    }
    const float* Dn = D + (size_t)n * 3 * g.HW + w;
    const float4 d0 = *reinterpret_cast<const float4*>(Dn);
    const float4 d1 = *reinterpret_cast<const float4*>(Dn + (size_t)g.HW);
    const float4 d2 = *reinterpret_cast<const float4*>(Dn + 2 * (size_t)g.HW);
    const float zx = fmaf(w2, d2.x, fmaf(w1, d1.x, w0 * d0.x)) - mean, zy = fmaf(w2, d2.y, fmaf(w1, d1.y, w0 * d0.y)) - mean;
    const float zz = fmaf(w2, d2.z, fmaf(w1, d1.z, w0 * d0.z)) - mean, zw = fmaf(w2, d2.w, fmaf(w1, d1.w, w0 * d0.w)) - mean;
    acc[0] += (d.x + d.y) + (d.z + d.w);
    acc[1] = fmaf(d.x, zx, acc[1]); acc[1] = fmaf(d.y, zy, acc[1]); acc[1] = fmaf(d.z, zz, acc[1]); acc[1] = fmaf(d.w, zw, acc[1]);
    acc[2] = fmaf(d.x, d0.x, acc[2]); acc[2] = fmaf(d.y, d0.y, acc[2]); acc[2] = fmaf(d.z, d0.z, acc[2]); acc[2] = fmaf(d.w, d0.w, acc[2]);
    acc[3] = fmaf(d.x, d1.x, acc[3]); acc[3] = fmaf(d.y, d1.y, acc[3]); acc[3] = fmaf(d.z, d1.z, acc[3]); acc[3] = fmaf(d.w, d1.w, acc[3]);
    acc[4] = fmaf(d.x, d2.x, acc[4]); acc[4] = fmaf(d.y, d2.y, acc[4]); acc[4] = fmaf(d.z, d2.z, acc[4]); acc[4] = fmaf(d.w, d2.w, acc[4]);
    acc[5] = fmaf(zx, d0.x, acc[5]); acc[5] = fmaf(zy, d0.y, acc[5]); acc[5] = fmaf(zz, d0.z, acc[5]); acc[5] = fmaf(zw, d0.w, acc[5]);
    acc[6] = fmaf(zx, d1.x, acc[6]); acc[6] = fmaf(zy, d1.y, acc[6]); acc[6] = fmaf(zz, d1.z, acc[6]); acc[6] = fmaf(zw, d1.w, acc[6]);
    acc[7] = fmaf(zx, d2.x, acc[7]); acc[7] = fmaf(zy, d2.y, acc[7]); acc[7] = fmaf(zz, d2.z, acc[7]); acc[7] = fmaf(zw, d2.w, acc[7]);
    if (c == 0) {          // block-uniform
      acc[8] += (d0.x + d0.y) + (d0.z + d0.w);
      acc[9] += (d1.x + d1.y) + (d1.z + d1.w);
      acc[10] += (d2.x + d2.y) + (d2.z + d2.w);
    }
  }
  __shared__ float red[11][kBT / 64];
#pragma unroll
  for (int k = 0; k < 11; ++k) {
    const float v = wave_sum(acc[k]);
    if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < 8)
    part8[((size_t)s * g.C + c) * 8 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
  if (c == 0 && threadIdx.x >= 8 && threadIdx.x < 11)
    partB[s * 3 + threadIdx.x - 8] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// one block, one thread per channel: slices in order -> dgamma, dbeta, dW_xyz
__global__ void ball_bwd_nop_finish_kernel(const float* __restrict__ part8, const float* __restrict__ partB, int split, int C,
                                           const float* __restrict__ gamma, const float* __restrict__ stats, float M,
                                           float* __restrict__ gstats, float* __restrict__ dW) {
  const int c = threadIdx.x;
  if (c >= C) return;
  double t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bsum[3] = {0, 0, 0};          // <= 64 slices, merged in slice order in double
  for (int s = 0; s < split; ++s) {
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] += (double)part8[((size_t)s * C + c) * 8 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) bsum[k] += (double)partB[s * 3 + k];
  }
  const double invstd = (double)stats[C + c];
  gstats[c] = (float)(t[1] * invstd);       // d gamma
  gstats[C + c] = (float)t[0];              // d beta
  const double a = (double)gamma[c] * invstd, b = t[0] / (double)M, q = t[1] * invstd * invstd / (double)M;
#pragma unroll
  for (int k = 0; k < 3; ++k) dW[3 * c + k] = (float)(a * (t[2 + k] - b * bsum[k] - q * t[5 + k]));
}

// ---------------------------------------------------------------------------------------------
// Levels 2-4 (point features present), the gathered row in LDS (r06).  tools/probes/ball_levels.py put the (channel, slice) kernels
// at 1.4-1.9 TB/s of their algorithmic bytes.  Not the re-read of idx / D per channel (a sixteen-channels-per-workgroup form that
// reads them once per sixteen channels was SLOWER: 210 / 400 us against 145 / 269 at level 2 -- its gathers leave the one-row
// working set of a workgroup), but the gather itself: P[b, c, idx] is one 4-byte access per ball member through the texture
// path, 64 different lines per wave instruction = 64 cycles, 67 M of them per pass at level 2 / ns = 32 = 0.11 ms on 256 CUs --
// twice per direction.  A workgroup now owns (channel, image, sub-slice), copies the image's row of P (N floats <= 16 KB at every
// level) into LDS with coalesced loads and gathers from THERE (64 lanes over 32 banks: a few cycles).  Partial sums per
// (image, sub-slice) slot, merged slot by slot in order (lanes stride over the slots, fixed lane tree: deterministic).
// ---------------------------------------------------------------------------------------------
struct RowGeo {
  int C, HW, N, np, ns;   // channels, members per image (np * ns), source points
  int B, sub, per;        // images, sub-slices per image, members per sub-slice (multiple of 4)
  int M;                  // B * HW
};
RowGeo make_row_geo(int B, int C, int N, int np, int ns) {
  RowGeo r;
  r.C = C; r.HW = np * ns; r.N = N; r.np = np; r.ns = ns; r.B = B; r.M = B * r.HW;
  // enough workgroups for the chip, at least 4 float4 per thread where the image allows it
  int sub = 1;
  while ((long long)(C / 4 + 1) * B * sub < 2048 && r.HW / (sub * 2) >= kVec * 2) sub *= 2;
  int per = (r.HW + sub - 1) / sub;
  per = (per + 3) & ~3;
  r.sub = (r.HW + per - 1) / per;
  r.per = per;
  return r;
}
__device__ __forceinline__ void stage_row(float* __restrict__ row, const float* __restrict__ src, int N) {
  if ((N & 3) == 0) {
    for (int e = threadIdx.x; e < (N >> 2); e += kBT) reinterpret_cast<float4*>(row)[e] = reinterpret_cast<const float4*>(src)[e];
  } else {
    for (int e = threadIdx.x; e < N; e += kBT) row[e] = src[e];
  }
}
// merge the slot partials of channel c: lanes stride over the slots in order, then the fixed lane tree
__device__ __forceinline__ void merge_slots(const float* part, int nslots, int C, int c, float& p1, float& p2) {
  const int lane = threadIdx.x & 63;
  p1 = 0.f; p2 = 0.f;
  for (int sl = lane; sl < nslots; sl += 64) { p1 += part[(size_t)(2 * sl) * C + c]; p2 += part[(size_t)(2 * sl + 1) * C + c]; }
  p1 = wave_sum(p1);
  p2 = wave_sum(p2);
}
// shift of the one-pass variance: z of the channel's first member (image 0)
__device__ __forceinline__ float ball_k_row(const float* P, const int* idx, const float* D, int c, int N, int HW, float w0, float w1,
                                            float w2) {
  return P[(size_t)c * N + idx[0]] + fmaf(w2, D[2 * (size_t)HW], fmaf(w1, D[(size_t)HW], w0 * D[0]));
}

// CG channels per workgroup (their CG rows of P in LDS): idx and D are read once per CG channels (with one channel per workgroup
// the row kernels ran at ~17 TB/s of L2 reads for 2.2 TB/s of output, r06)
template <int CG>
struct RowCtx {
  float w0[CG], w1[CG], w2[CG];
};
template <int CG>
__device__ __forceinline__ void row_setup(float* __restrict__ rows, const float* __restrict__ P, const float* __restrict__ Wxyz,
                                          const RowGeo& r, int c0, int n, RowCtx<CG>& x) {
#pragma unroll
  for (int u = 0; u < CG; ++u) {
    stage_row(rows + (size_t)u * r.N, P + ((size_t)n * r.C + c0 + u) * r.N, r.N);
    x.w0[u] = Wxyz[3 * (c0 + u)]; x.w1[u] = Wxyz[3 * (c0 + u) + 1]; x.w2[u] = Wxyz[3 * (c0 + u) + 2];
  }
}
struct Member4 {
  int4 id;
  float4 d0, d1, d2;
};
__device__ __forceinline__ Member4 load_members(const int* __restrict__ idn, const float* __restrict__ Dn, int HW, int w) {
  Member4 m;
  m.id = *reinterpret_cast<const int4*>(idn + w);
  m.d0 = *reinterpret_cast<const float4*>(Dn + w);
  m.d1 = *reinterpret_cast<const float4*>(Dn + (size_t)HW + w);
  m.d2 = *reinterpret_cast<const float4*>(Dn + 2 * (size_t)HW + w);
  return m;
}
__device__ __forceinline__ float4 z_of(const float* __restrict__ row, const Member4& m, float w0, float w1, float w2) {
  return make_float4(row[m.id.x] + fmaf(w2, m.d2.x, fmaf(w1, m.d1.x, w0 * m.d0.x)), row[m.id.y] + fmaf(w2, m.d2.y, fmaf(w1, m.d1.y, w0 * m.d0.y)),
                     row[m.id.z] + fmaf(w2, m.d2.z, fmaf(w1, m.d1.z, w0 * m.d0.z)), row[m.id.w] + fmaf(w2, m.d2.w, fmaf(w1, m.d1.w, w0 * m.d0.w)));
}
// workgroup sums of 2 CG values; slot k's total lands in thread k
template <int NV>
__device__ __forceinline__ float block_sum_nv(float (&v)[NV]) {
  __shared__ float red[NV][kBT / 64];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const float xx = wave_sum(v[k]);
    if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = xx;
  }
  __syncthreads();
  const int t = threadIdx.x < NV ? threadIdx.x : 0;
  return (red[t][0] + red[t][1]) + (red[t][2] + red[t][3]);
}

template <int CG>
__global__ __launch_bounds__(kBT) void ball_stats_row_kernel(const float* __restrict__ P, const float* __restrict__ D,
                                                             const float* __restrict__ Wxyz, const int* __restrict__ idx, RowGeo r,
                                                             float* __restrict__ part) {
  extern __shared__ float rows[];
  const int c0 = blockIdx.x * CG, n = blockIdx.y / r.sub, j = blockIdx.y - n * r.sub;
  RowCtx<CG> x;
  row_setup<CG>(rows, P, Wxyz, r, c0, n, x);
  float k[CG], acc[2 * CG];
#pragma unroll
  for (int u = 0; u < CG; ++u) {
    k[u] = ball_k_row(P, idx, D, c0 + u, r.N, r.HW, x.w0[u], x.w1[u], x.w2[u]);
    acc[2 * u] = 0.f; acc[2 * u + 1] = 0.f;
  }
  __syncthreads();
  const int beg = j * r.per, end = min(r.HW, beg + r.per);
  const int* idn = idx + (size_t)n * r.HW;
  const float* Dn = D + (size_t)n * 3 * r.HW;
#pragma unroll 2
  for (int w = beg + threadIdx.x * 4; w < end; w += kVec) {
    const Member4 m = load_members(idn, Dn, r.HW, w);
#pragma unroll
    for (int u = 0; u < CG; ++u) {
      const float4 v = z_of(rows + (size_t)u * r.N, m, x.w0[u], x.w1[u], x.w2[u]);
      const float a = v.x - k[u], b = v.y - k[u], cc = v.z - k[u], d = v.w - k[u];
      acc[2 * u] += (a + b) + (cc + d);
      acc[2 * u + 1] = fmaf(a, a, acc[2 * u + 1]); acc[2 * u + 1] = fmaf(b, b, acc[2 * u + 1]);
      acc[2 * u + 1] = fmaf(cc, cc, acc[2 * u + 1]); acc[2 * u + 1] = fmaf(d, d, acc[2 * u + 1]);
    }
  }
  const float v = block_sum_nv<2 * CG>(acc);
  if (threadIdx.x < 2 * CG) part[(size_t)(2 * blockIdx.y + (threadIdx.x & 1)) * r.C + c0 + (threadIdx.x >> 1)] = v;
}

template <int CG, bool RELU>
__global__ __launch_bounds__(kBT) void ball_apply_row_kernel(
    const float* __restrict__ P, const float* __restrict__ D, const float* __restrict__ Wxyz, const int* __restrict__ idx,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ part, RowGeo r, float eps,
    float momentum, float* __restrict__ rmean, float* __restrict__ rvar, float* __restrict__ stats, float* __restrict__ y) {
  extern __shared__ float rows[];
  const int c0 = blockIdx.x * CG, n = blockIdx.y / r.sub, j = blockIdx.y - n * r.sub;
  RowCtx<CG> x;
  row_setup<CG>(rows, P, Wxyz, r, c0, n, x);
  float mean[CG], sc[CG], bt[CG];
  const float invM = 1.f / (float)r.M;
#pragma unroll
  for (int u = 0; u < CG; ++u) {
    const int c = c0 + u;
    float p1, p2;
    merge_slots(part, r.B * r.sub, r.C, c, p1, p2);
    const float k = ball_k_row(P, idx, D, c, r.N, r.HW, x.w0[u], x.w1[u], x.w2[u]);
    const float m1 = p1 * invM;
    const float var = fmaxf(fmaf(-m1, m1, p2 * invM), 0.f);
    mean[u] = k + m1;
    const float invstd = 1.f / sqrtf(var + eps);
    if (blockIdx.y == 0 && threadIdx.x == 0) {
      stats[c] = mean[u];
      stats[r.C + c] = invstd;
      if (rmean != nullptr) {
        const float unbiased = r.M > 1 ? var * ((float)r.M / (float)(r.M - 1)) : var;
        rmean[c] = fmaf(momentum, mean[u] - rmean[c], rmean[c]);
        rvar[c] = fmaf(momentum, unbiased - rvar[c], rvar[c]);
      }
    }
    sc[u] = gamma[c] * invstd;
    bt[u] = beta[c];
  }
  __syncthreads();
  const int beg = j * r.per, end = min(r.HW, beg + r.per);
  const int* idn = idx + (size_t)n * r.HW;
  const float* Dn = D + (size_t)n * 3 * r.HW;
  float* yn = y + ((size_t)n * r.C + c0) * (size_t)r.HW;
#pragma unroll 2
  for (int w = beg + threadIdx.x * 4; w < end; w += kVec) {
    const Member4 m = load_members(idn, Dn, r.HW, w);
#pragma unroll
    for (int u = 0; u < CG; ++u) {
      const float4 v = z_of(rows + (size_t)u * r.N, m, x.w0[u], x.w1[u], x.w2[u]);
      float4 o = make_float4(fmaf(v.x - mean[u], sc[u], bt[u]), fmaf(v.y - mean[u], sc[u], bt[u]), fmaf(v.z - mean[u], sc[u], bt[u]),
                             fmaf(v.w - mean[u], sc[u], bt[u]));
      if (RELU) { o.x = relu_nan(o.x); o.y = relu_nan(o.y); o.z = relu_nan(o.z); o.w = relu_nan(o.w); }
      *reinterpret_cast<float4*>(yn + (size_t)u * r.HW + w) = o;
    }
  }
}

template <int CG, bool RELU>
__global__ __launch_bounds__(kBT) void ball_bwd_reduce_row_kernel(
    const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ P, const float* __restrict__ D,
    const float* __restrict__ Wxyz, const int* __restrict__ idx, const float* __restrict__ stats, RowGeo r,
    float* __restrict__ part) {
  extern __shared__ float rows[];
  const int c0 = blockIdx.x * CG, n = blockIdx.y / r.sub, j = blockIdx.y - n * r.sub;
  RowCtx<CG> x;
  row_setup<CG>(rows, P, Wxyz, r, c0, n, x);
  float mean[CG], acc[2 * CG];
#pragma unroll
  for (int u = 0; u < CG; ++u) { mean[u] = stats[c0 + u]; acc[2 * u] = 0.f; acc[2 * u + 1] = 0.f; }
  __syncthreads();
  const int beg = j * r.per, end = min(r.HW, beg + r.per);
  const int* idn = idx + (size_t)n * r.HW;
  const float* Dn = D + (size_t)n * 3 * r.HW;
  const size_t o0 = ((size_t)n * r.C + c0) * (size_t)r.HW;
#pragma unroll 2
  for (int w = beg + threadIdx.x * 4; w < end; w += kVec) {
    const Member4 m = load_members(idn, Dn, r.HW, w);
#pragma unroll
    for (int u = 0; u < CG; ++u) {
      float4 d = *reinterpret_cast<const float4*>(dy + o0 + (size_t)u * r.HW + w);
      if (RELU) {
        const float4 out = *reinterpret_cast<const float4*>(y + o0 + (size_t)u * r.HW + w);
        d.x = out.x > 0.f ? d.x : 0.f; d.y = out.y > 0.f ? d.y : 0.f;
        d.z = out.z > 0.f ? d.z : 0.f; d.w = out.w > 0.f ? d.w : 0.f;
      }
      const float4 v = z_of(rows + (size_t)u * r.N, m, x.w0[u], x.w1[u], x.w2[u]);
      acc[2 * u] += (d.x + d.y) + (d.z + d.w);
      acc[2 * u + 1] = fmaf(d.x, v.x - mean[u], acc[2 * u + 1]); acc[2 * u + 1] = fmaf(d.y, v.y - mean[u], acc[2 * u + 1]);
      acc[2 * u + 1] = fmaf(d.z, v.z - mean[u], acc[2 * u + 1]); acc[2 * u + 1] = fmaf(d.w, v.w - mean[u], acc[2 * u + 1]);
    }
  }
  const float v = block_sum_nv<2 * CG>(acc);
  if (threadIdx.x < 2 * CG) part[(size_t)(2 * blockIdx.y + (threadIdx.x & 1)) * r.C + c0 + (threadIdx.x >> 1)] = v;
}

template <int CG, bool RELU>
__global__ __launch_bounds__(kBT) void ball_bwd_apply_row_kernel(
    const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ P, const float* __restrict__ D,
    const float* __restrict__ Wxyz, const int* __restrict__ idx, const float* __restrict__ gamma,
    const float* __restrict__ stats, const float* __restrict__ part, RowGeo r, float* __restrict__ gstats,
    float* __restrict__ dz, float* __restrict__ wpart) {
  extern __shared__ float rows[];
  const int c0 = blockIdx.x * CG, n = blockIdx.y / r.sub, j = blockIdx.y - n * r.sub;
  RowCtx<CG> x;
  row_setup<CG>(rows, P, Wxyz, r, c0, n, x);
  float mean[CG], a[CG], b[CG], q[CG], t[3 * CG];
  const float invM = 1.f / (float)r.M;
#pragma unroll
  for (int u = 0; u < CG; ++u) {
    const int c = c0 + u;
    float p1, p2;
    merge_slots(part, r.B * r.sub, r.C, c, p1, p2);
    mean[u] = stats[c];
    const float invstd = stats[r.C + c];
    if (blockIdx.y == 0 && threadIdx.x == 0) {
      gstats[c] = p2 * invstd;     // d gamma
      gstats[r.C + c] = p1;        // d beta
    }
    a[u] = gamma[c] * invstd; b[u] = p1 * invM; q[u] = p2 * invstd * invstd * invM;
    t[3 * u] = 0.f; t[3 * u + 1] = 0.f; t[3 * u + 2] = 0.f;
  }
  __syncthreads();
  const int beg = j * r.per, end = min(r.HW, beg + r.per);
  const int* idn = idx + (size_t)n * r.HW;
  const float* Dn = D + (size_t)n * 3 * r.HW;
  const size_t o0 = ((size_t)n * r.C + c0) * (size_t)r.HW;
#pragma unroll 2
  for (int w = beg + threadIdx.x * 4; w < end; w += kVec) {
    const Member4 m = load_members(idn, Dn, r.HW, w);
#pragma unroll
    for (int u = 0; u < CG; ++u) {
      float4 d = *reinterpret_cast<const float4*>(dy + o0 + (size_t)u * r.HW + w);
      if (RELU) {
        const float4 out = *reinterpret_cast<const float4*>(y + o0 + (size_t)u * r.HW + w);
        d.x = out.x > 0.f ? d.x : 0.f; d.y = out.y > 0.f ? d.y : 0.f;
        d.z = out.z > 0.f ? d.z : 0.f; d.w = out.w > 0.f ? d.w : 0.f;
      }
      const float4 v = z_of(rows + (size_t)u * r.N, m, x.w0[u], x.w1[u], x.w2[u]);
      float4 g4;
      g4.x = a[u] * (d.x - b[u] - (v.x - mean[u]) * q[u]);
      g4.y = a[u] * (d.y - b[u] - (v.y - mean[u]) * q[u]);
      g4.z = a[u] * (d.z - b[u] - (v.z - mean[u]) * q[u]);
      g4.w = a[u] * (d.w - b[u] - (v.w - mean[u]) * q[u]);
      *reinterpret_cast<float4*>(dz + o0 + (size_t)u * r.HW + w) = g4;
      t[3 * u] = fmaf(g4.x, m.d0.x, t[3 * u]); t[3 * u] = fmaf(g4.y, m.d0.y, t[3 * u]);
      t[3 * u] = fmaf(g4.z, m.d0.z, t[3 * u]); t[3 * u] = fmaf(g4.w, m.d0.w, t[3 * u]);
      t[3 * u + 1] = fmaf(g4.x, m.d1.x, t[3 * u + 1]); t[3 * u + 1] = fmaf(g4.y, m.d1.y, t[3 * u + 1]);
      t[3 * u + 1] = fmaf(g4.z, m.d1.z, t[3 * u + 1]); t[3 * u + 1] = fmaf(g4.w, m.d1.w, t[3 * u + 1]);
      t[3 * u + 2] = fmaf(g4.x, m.d2.x, t[3 * u + 2]); t[3 * u + 2] = fmaf(g4.y, m.d2.y, t[3 * u + 2]);
      t[3 * u + 2] = fmaf(g4.z, m.d2.z, t[3 * u + 2]); t[3 * u + 2] = fmaf(g4.w, m.d2.w, t[3 * u + 2]);
    }
  }
  const float v = block_sum_nv<3 * CG>(t);
  if (threadIdx.x < 3 * CG) wpart[((size_t)blockIdx.y * r.C + c0) * 3 + threadIdx.x] = v;      // [slot][c][k]
}

constexpr int kRowMaxN = 32768;      // 128 KB of LDS for the row
// the row kernels pay where an image holds many members per channel (levels 2 / 3 of Pointnet2MSG: 4096-32768); the coarsest
// level (1024 / 2048 members per image and channel: thousands of two-iteration workgroups) stays on the per-channel kernels
inline bool row_ok(const RowGeo& r) { return r.N <= kRowMaxN && (r.HW & 3) == 0 && r.B * r.sub <= 65535 && r.HW >= 8192; }
inline int row_group(const RowGeo& r) { return (r.C % 4 == 0 && (size_t)4 * r.N * sizeof(float) <= 64 * 1024) ? 4 : 1; }

bool bad_ball(int N, int C, int np, int ns) {
  if (!(ns == 4 || ns == 8 || ns == 16 || ns == 32 || ns == 64) || np <= 0 || (np & 3) != 0) return true;
  return bad_shape(N, C, np * ns) || bad_shape(N, C, np);
}

}  // namespace

extern "C" {

size_t hcm_bn_act_stats_floats(int N, int C, int HW) {
  if (bad_shape(N, C, HW)) return 0;
  const Geo g = make_geo(N, C, HW);
  return (size_t)(2 + 2 * g.split) * (size_t)C;
}

int hcm_bn_act_forward(const float* x, const float* residual, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, float momentum, float eps, int relu,
                       int N, int C, int HW, float* y, float* stats, hcm_stream_t stream) {
  if (bad_shape(N, C, HW) || !x || !gamma || !beta || !y || !stats || (running_mean == nullptr) != (running_var == nullptr))
    return (int)hipErrorInvalidValue;
  const Geo g = make_geo(N, C, HW);
  hipStream_t st = (hipStream_t)stream;
  if (small_map(g)) {
    const int th = small_threads(g);
#define HCM_BN_SMALL(R, S)                                                                                \
  bn_small_fwd_kernel<R, S><<<C, th, 0, st>>>(x, residual, gamma, beta, g, eps, momentum, running_mean,   \
                                              running_var, stats, y)
    if (relu) { if (residual) HCM_BN_SMALL(true, true); else HCM_BN_SMALL(true, false); }
    else      { if (residual) HCM_BN_SMALL(false, true); else HCM_BN_SMALL(false, false); }
#undef HCM_BN_SMALL
    HCM_CHECK_LAUNCH();
    return 0;
  }
  const dim3 grid(C, g.split);
  float* part = stats + 2 * (size_t)C;
  bn_stats_kernel<<<grid, kBT, 0, st>>>(x, g, part);
  HCM_CHECK_LAUNCH();
#define HCM_BN_APPLY(R, S)                                                                           \
  bn_apply_kernel<R, S, false><<<grid, kBT, 0, st>>>(x, residual, gamma, beta, part, g, eps, momentum, \
                                                     running_mean, running_var, stats, y, 0)
  if (relu) { if (residual) HCM_BN_APPLY(true, true); else HCM_BN_APPLY(true, false); }
  else      { if (residual) HCM_BN_APPLY(false, true); else HCM_BN_APPLY(false, false); }
#undef HCM_BN_APPLY
  HCM_CHECK_LAUNCH();
  return 0;
}

int hcm_bn_act_forward_pre(const float* x, const float* residual, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float momentum, float eps, int relu,
                           int N, int C, int HW, float* y, float* stats, const float* partial_sums, int nslots,
                           hcm_stream_t stream) {
  if (bad_shape(N, C, HW) || !x || !gamma || !beta || !y || !stats || !partial_sums || nslots <= 0 ||
      (running_mean == nullptr) != (running_var == nullptr))
    return (int)hipErrorInvalidValue;
  const Geo g = make_geo(N, C, HW);
  if (small_map(g)) return (int)hipErrorInvalidValue;       // the one-workgroup form has no separate statistics pass
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(C, g.split);
#define HCM_BN_APPLY(R, S)                                                                                   \
  bn_apply_kernel<R, S, true><<<grid, kBT, 0, st>>>(x, residual, gamma, beta, partial_sums, g, eps, momentum, \
                                                    running_mean, running_var, stats, y, nslots)
  if (relu) { if (residual) HCM_BN_APPLY(true, true); else HCM_BN_APPLY(true, false); }
  else      { if (residual) HCM_BN_APPLY(false, true); else HCM_BN_APPLY(false, false); }
#undef HCM_BN_APPLY
  HCM_CHECK_LAUNCH();
  return 0;
}

int hcm_bn_act_backward(const float* dy, const float* dy2, const float* x, const float* y, const float* gamma,
                        const float* stats, int relu, int N, int C, int HW, float* dz, float* dx,
                        float* gstats, hcm_stream_t stream) {
  return hcm_bn_act_backward_ws(dy, dy2, x, y, gamma, stats, relu, N, C, HW, dz, dx, gstats,
                                gstats ? gstats + 2 * (size_t)(C > 0 ? C : 0) : nullptr, stream);
}

int hcm_bn_act_backward_ws(const float* dy, const float* dy2, const float* x, const float* y, const float* gamma,
                           const float* stats, int relu, int N, int C, int HW, float* dz, float* dx,
                           float* gstats, float* scratch, hcm_stream_t stream) {
  const bool needs_dz = relu || dy2 != nullptr;
  if (bad_shape(N, C, HW) || !dy || !x || !gamma || !stats || !gstats || !scratch || (relu && !y) || (needs_dz && !dz))
    return (int)hipErrorInvalidValue;
  const Geo g = make_geo(N, C, HW);
  hipStream_t st = (hipStream_t)stream;
  if (small_map(g)) {
    const int th = small_threads(g);
    if (relu) {
      if (dy2) bn_small_bwd_kernel<true, true><<<C, th, 0, st>>>(dy, dy2, x, y, gamma, stats, g, dz, dx, gstats);
      else     bn_small_bwd_kernel<true, false><<<C, th, 0, st>>>(dy, nullptr, x, y, gamma, stats, g, dz, dx, gstats);
    } else {
      if (dy2) bn_small_bwd_kernel<false, true><<<C, th, 0, st>>>(dy, dy2, x, nullptr, gamma, stats, g, dz, dx, gstats);
      else     bn_small_bwd_kernel<false, false><<<C, th, 0, st>>>(dy, nullptr, x, nullptr, gamma, stats, g, nullptr, dx, gstats);
    }
    HCM_CHECK_LAUNCH();
    return 0;
  }
  const dim3 grid(C, g.split);
  float* part = scratch;
  if (relu) {
    if (dy2) bn_bwd_reduce_kernel<true, true><<<grid, kBT, 0, st>>>(dy, dy2, x, y, stats, g, dz, part);
    else     bn_bwd_reduce_kernel<true, false><<<grid, kBT, 0, st>>>(dy, nullptr, x, y, stats, g, dz, part);
  } else {
    if (dy2) bn_bwd_reduce_kernel<false, true><<<grid, kBT, 0, st>>>(dy, dy2, x, nullptr, stats, g, dz, part);
    else     bn_bwd_reduce_kernel<false, false><<<grid, kBT, 0, st>>>(dy, nullptr, x, nullptr, stats, g, nullptr, part);
  }
  HCM_CHECK_LAUNCH();
  bn_bwd_apply_kernel<<<grid, kBT, 0, st>>>(needs_dz ? dz : dy, x, gamma, stats, part, g, gstats, dx);
  HCM_CHECK_LAUNCH();
  return 0;
}


size_t hcm_bn_relu_ballmax_stats_floats(int N, int C, int np, int ns) {
  if (bad_ball(N, C, np, ns)) return 0;
  const Geo g = make_geo(N, C, np * ns), gr = make_geo(N, C, np);
  const int sp = g.split > gr.split ? g.split : gr.split;
  return (size_t)(2 + 2 * sp) * (size_t)C;
}

int hcm_bn_relu_ballmax_forward(const float* z, const float* gamma, const float* beta, float* running_mean,
                                float* running_var, float momentum, float eps, int N, int C, int np, int ns, float* out,
                                int32_t* arg, float* zsel, float* stats, hcm_stream_t stream) {
  if (bad_ball(N, C, np, ns) || !z || !gamma || !beta || !out || !arg || !zsel || !stats ||
      (running_mean == nullptr) != (running_var == nullptr))
    return (int)hipErrorInvalidValue;
  const Geo g = make_geo(N, C, np * ns);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(C, g.split);
  float* part = stats + 2 * (size_t)C;
  const double M = (double)N * C * np * ns;
  hcm::ProfSpan span(HCM_PROF_BALLMAX_FWD, st, 4.0 * (2.0 * M + 3.0 * M / ns));     // z twice; out, arg, zsel written
  bn_stats_kernel<<<grid, kBT, 0, st>>>(z, g, part);
  HCM_CHECK_LAUNCH();
#define HCM_BALLMAX(LL)                                                                                              \
  bn_relu_ballmax_kernel<LL><<<grid, kBT, 0, st>>>(z, gamma, beta, part, g, eps, momentum, running_mean, running_var, \
                                                   stats, out, arg, zsel, np)
  switch (ns) {
    case 4: HCM_BALLMAX(1); break;
    case 8: HCM_BALLMAX(2); break;
    case 16: HCM_BALLMAX(4); break;
    case 32: HCM_BALLMAX(8); break;
    default: HCM_BALLMAX(16); break;
  }
#undef HCM_BALLMAX
  HCM_CHECK_LAUNCH();
  return 0;
}

int hcm_bn_relu_ballmax_backward(const float* dout, const float* out, const int32_t* arg, const float* zsel,
                                 const float* z, const float* gamma, const float* stats, int N, int C, int np, int ns,
                                 float* dz, float* gstats, hcm_stream_t stream) {
  if (bad_ball(N, C, np, ns) || !dout || !out || !arg || !zsel || !z || !gamma || !stats || !dz || !gstats)
    return (int)hipErrorInvalidValue;
  const Geo g = make_geo(N, C, np * ns), gr = make_geo(N, C, np);
  hipStream_t st = (hipStream_t)stream;
  float* part = gstats + 2 * (size_t)C;
  const double M = (double)N * C * np * ns;
  hcm::ProfSpan span(HCM_PROF_BALLMAX_BWD, st, 4.0 * (2.0 * M + 6.0 * M / ns));     // z read, dz written; the [N, C, np] tensors twice
  ballmax_bwd_reduce_kernel<<<dim3(C, gr.split), kBT, 0, st>>>(dout, out, zsel, stats, gr, part);
  HCM_CHECK_LAUNCH();
  const dim3 grid(C, g.split);
#define HCM_BALLMAX(LL)                                                                                        \
  ballmax_bwd_apply_kernel<LL><<<grid, kBT, 0, st>>>(dout, out, arg, z, gamma, stats, part, g, gr.split, np, gstats, dz)
  switch (ns) {
    case 4: HCM_BALLMAX(1); break;
    case 8: HCM_BALLMAX(2); break;
    case 16: HCM_BALLMAX(4); break;
    case 32: HCM_BALLMAX(8); break;
    default: HCM_BALLMAX(16); break;
  }
#undef HCM_BALLMAX
  HCM_CHECK_LAUNCH();
  return 0;
}


size_t hcm_ball_project_stats_floats(int B, int C, int np, int ns) {
  if (bad_ball(B, C, np, ns)) return 0;
  const Geo g = make_geo(B, C, np * ns);
  // [2C results][scratch]: with point features 2 S C partial sums + 3 S C dW_xyz partials (S = the slice count of the
  // per-channel kernels or the (image, sub-slice) slots of the row kernels, whichever is larger); without (the all-channel forward / one-pass backward) kAllBlocks * 2C
  // forward partials or split * (8C + 3) backward partials
  size_t scratch = (size_t)5 * g.split * C;
  {
    const RowGeo r = make_row_geo(B, C, 1, np, ns);          // (the slot count does not depend on N)
    const size_t rows = (size_t)5 * r.B * r.sub * C;
    if (rows > scratch) scratch = rows;
  }
  const size_t fwd_all = (size_t)kAllBlocks * 2 * C, bwd_nop = (size_t)g.split * (8 * (size_t)C + 3);
  if (fwd_all > scratch) scratch = fwd_all;
  if (bwd_nop > scratch) scratch = bwd_nop;
  return 2 * (size_t)C + scratch;
}

int hcm_ball_project_forward(const float* P, const float* D, const float* Wxyz, const int32_t* idx, const float* gamma,
                             const float* beta, float* running_mean, float* running_var, float momentum, float eps, int relu,
                             int B, int C, int N, int np, int ns, float* y, float* stats, hcm_stream_t stream) {
  if (bad_ball(B, C, np, ns) || (P && (N <= 0 || !idx)) || !D || !Wxyz || !gamma || !beta || !y || !stats ||
      (running_mean == nullptr) != (running_var == nullptr))
    return (int)hipErrorInvalidValue;
  const Geo g = make_geo(B, C, np * ns);
  const BallGeo bg = {N, np, ns};
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(C, g.split);
  float* part = stats + 2 * (size_t)C;
  // algorithmic bytes: y written once; idx, D and P each read once per pass (P's rows and D are re-read per channel out of L2)
  const double uniq = (double)B * np * ns * (P ? 4.0 : 3.0) + (P ? (double)B * C * N : 0.0);
  hcm::ProfSpan span(HCM_PROF_BALL_FWD, st, 4.0 * ((double)B * C * np * ns + 2.0 * uniq));
  if (P == nullptr && (C == 16 || C == 32) && ((long long)B * g.HW) % 4 == 0) {
    // no point features (first SA level): all channels per workgroup, D read once per pass
    const int total4 = (int)(((long long)B * g.HW) / 4);
    int blocks = (total4 + kBT - 1) / kBT;
    if (blocks > kAllBlocks) blocks = kAllBlocks;
    if (C == 16) ball_all_stats_kernel<16><<<blocks, kBT, 0, st>>>(D, Wxyz, g.HW, total4, part);
    else ball_all_stats_kernel<32><<<blocks, kBT, 0, st>>>(D, Wxyz, g.HW, total4, part);
    HCM_CHECK_LAUNCH();
    ball_all_finish_kernel<<<C, 128, 0, st>>>(part, blocks, C, D, Wxyz, g.HW, (float)g.M, eps, momentum, running_mean,
                                              running_var, stats);
    HCM_CHECK_LAUNCH();
    const int ablocks = (total4 + kBT - 1) / kBT < 4096 ? (total4 + kBT - 1) / kBT : 4096;
    if (C == 16) {
      if (relu) ball_all_apply_kernel<16, true><<<ablocks, kBT, 0, st>>>(D, Wxyz, gamma, beta, stats, g.HW, total4, y);
      else ball_all_apply_kernel<16, false><<<ablocks, kBT, 0, st>>>(D, Wxyz, gamma, beta, stats, g.HW, total4, y);
    } else {
      if (relu) ball_all_apply_kernel<32, true><<<ablocks, kBT, 0, st>>>(D, Wxyz, gamma, beta, stats, g.HW, total4, y);
      else ball_all_apply_kernel<32, false><<<ablocks, kBT, 0, st>>>(D, Wxyz, gamma, beta, stats, g.HW, total4, y);
    }
    HCM_CHECK_LAUNCH();
    return 0;
  }
  if (P != nullptr) {
    const RowGeo r = make_row_geo(B, C, N, np, ns);
    if (row_ok(r) && hcm_ball_project_stats_floats(B, C, np, ns) >= (size_t)(2 + 5 * (size_t)r.B * r.sub) * C) {
      const int cg = row_group(r);
      const dim3 gr(C / cg, r.B * r.sub);
      const size_t lds = (size_t)cg * r.N * sizeof(float);
#define HCM_BALL_FWD_ROW(CG, R)                                                                                                    \
  do {                                                                                                                             \
    hipFuncSetAttribute(reinterpret_cast<const void*>(ball_stats_row_kernel<CG>), hipFuncAttributeMaxDynamicSharedMemorySize,      \
                        (int)lds);                                                                                                 \
    ball_stats_row_kernel<CG><<<gr, kBT, lds, st>>>(P, D, Wxyz, idx, r, part);                                                    \
    HCM_CHECK_LAUNCH();                                                                                                            \
    hipFuncSetAttribute(reinterpret_cast<const void*>(ball_apply_row_kernel<CG, R>), hipFuncAttributeMaxDynamicSharedMemorySize,   \
                        (int)lds);                                                                                                 \
    ball_apply_row_kernel<CG, R><<<gr, kBT, lds, st>>>(P, D, Wxyz, idx, gamma, beta, part, r, eps, momentum, running_mean,         \
                                                       running_var, stats, y);                                                    \
  } while (0)
      if (cg == 4) { if (relu) HCM_BALL_FWD_ROW(4, true); else HCM_BALL_FWD_ROW(4, false); }
      else         { if (relu) HCM_BALL_FWD_ROW(1, true); else HCM_BALL_FWD_ROW(1, false); }
#undef HCM_BALL_FWD_ROW
      HCM_CHECK_LAUNCH();
      return 0;
    }
  }
#define HCM_BALL_FWD(R, HP)                                                                                          \
  do {                                                                                                               \
    ball_stats_kernel<HP><<<grid, kBT, 0, st>>>(P, D, Wxyz, idx, g, bg, part);                                      \
    HCM_CHECK_LAUNCH();                                                                                              \
    ball_apply_kernel<R, HP><<<grid, kBT, 0, st>>>(P, D, Wxyz, idx, gamma, beta, part, g, bg, eps, momentum,        \
                                                   running_mean, running_var, stats, y);                             \
  } while (0)
  if (relu) { if (P) HCM_BALL_FWD(true, true); else HCM_BALL_FWD(true, false); }
  else      { if (P) HCM_BALL_FWD(false, true); else HCM_BALL_FWD(false, false); }
#undef HCM_BALL_FWD
  HCM_CHECK_LAUNCH();
  return 0;
}

int hcm_ball_project_backward(const float* dy, const float* y, const float* P, const float* D, const float* Wxyz,
                              const int32_t* idx, const float* gamma, const float* stats, int relu, int B, int C, int N,
                              int np, int ns, float* dz, float* dWxyz, float* gstats, hcm_stream_t stream) {
  if (bad_ball(B, C, np, ns) || !dy || (relu && !y) || (P && (N <= 0 || !idx || !dz)) || !D || !Wxyz || !gamma || !stats ||
      !dWxyz || !gstats)
    return (int)hipErrorInvalidValue;
  const Geo g = make_geo(B, C, np * ns);
  const BallGeo bg = {N, np, ns};
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(C, g.split);
  float* part = gstats + 2 * (size_t)C;
  float* wpart = part + 2 * (size_t)g.split * C;
  const double uniq = (double)B * np * ns * (P ? 4.0 : 3.0) + (P ? (double)B * C * N : 0.0);
  if (P == nullptr && C <= 1024) {
    // no point features: nobody needs dz (dz == NULL allowed) -- one pass over (dy, y, D), see ball_bwd_nop_kernel
    hcm::ProfSpan span(HCM_PROF_BALL_BWD, st, 4.0 * (2.0 * (double)B * C * np * ns + uniq));       // dy, y once
    float* part8 = gstats + 2 * (size_t)C;
    float* partB = part8 + (size_t)g.split * C * 8;
    if (relu) ball_bwd_nop_kernel<true><<<grid, kBT, 0, st>>>(dy, y, D, Wxyz, stats, g, part8, partB);
    else ball_bwd_nop_kernel<false><<<grid, kBT, 0, st>>>(dy, y, D, Wxyz, stats, g, part8, partB);
    HCM_CHECK_LAUNCH();
    ball_bwd_nop_finish_kernel<<<1, 1024, 0, st>>>(part8, partB, g.split, C, gamma, stats, (float)g.M, gstats, dWxyz);
    HCM_CHECK_LAUNCH();
    return 0;
  }
  hcm::ProfSpan span(HCM_PROF_BALL_BWD, st, 4.0 * (5.0 * (double)B * C * np * ns + 2.0 * uniq));   // dy, y twice; dz written
  if (P != nullptr) {
    const RowGeo r = make_row_geo(B, C, N, np, ns);
    if (row_ok(r) && hcm_ball_project_stats_floats(B, C, np, ns) >= (size_t)(2 + 5 * (size_t)r.B * r.sub) * C) {
      const int cg = row_group(r);
      const dim3 gr(C / cg, r.B * r.sub);
      const size_t lds = (size_t)cg * r.N * sizeof(float);
      float* wp = part + 2 * (size_t)r.B * r.sub * C;
#define HCM_BALL_BWD_ROW(CG, R)                                                                                                        \
  do {                                                                                                                                 \
    hipFuncSetAttribute(reinterpret_cast<const void*>(ball_bwd_reduce_row_kernel<CG, R>), hipFuncAttributeMaxDynamicSharedMemorySize,  \
                        (int)lds);                                                                                                     \
    ball_bwd_reduce_row_kernel<CG, R><<<gr, kBT, lds, st>>>(dy, y, P, D, Wxyz, idx, stats, r, part);                                  \
    HCM_CHECK_LAUNCH();                                                                                                                \
    hipFuncSetAttribute(reinterpret_cast<const void*>(ball_bwd_apply_row_kernel<CG, R>), hipFuncAttributeMaxDynamicSharedMemorySize,   \
                        (int)lds);                                                                                                     \
    ball_bwd_apply_row_kernel<CG, R><<<gr, kBT, lds, st>>>(dy, y, P, D, Wxyz, idx, gamma, stats, part, r, gstats, dz, wp);            \
  } while (0)
      if (cg == 4) { if (relu) HCM_BALL_BWD_ROW(4, true); else HCM_BALL_BWD_ROW(4, false); }
      else         { if (relu) HCM_BALL_BWD_ROW(1, true); else HCM_BALL_BWD_ROW(1, false); }
#undef HCM_BALL_BWD_ROW
      HCM_CHECK_LAUNCH();
      ball_wxyz_merge_kernel<<<(3 * C + 255) / 256, 256, 0, st>>>(wp, C, r.B * r.sub, dWxyz);
      HCM_CHECK_LAUNCH();
      return 0;
    }
  }
#define HCM_BALL_BWD(R, HP)                                                                                          \
  do {                                                                                                               \
    ball_bwd_reduce_kernel<R, HP><<<grid, kBT, 0, st>>>(dy, y, P, D, Wxyz, idx, stats, g, bg, part);                \
    HCM_CHECK_LAUNCH();                                                                                              \
    ball_bwd_apply_kernel<R, HP><<<grid, kBT, 0, st>>>(dy, y, P, D, Wxyz, idx, gamma, stats, part, g, bg, gstats,   \
                                                       dz, wpart);                                                   \
  } while (0)
  if (relu) { if (P) HCM_BALL_BWD(true, true); else HCM_BALL_BWD(true, false); }
  else      { if (P) HCM_BALL_BWD(false, true); else HCM_BALL_BWD(false, false); }
#undef HCM_BALL_BWD
  HCM_CHECK_LAUNCH();
  ball_wxyz_merge_kernel<<<(3 * C + 255) / 256, 256, 0, st>>>(wpart, C, g.split, dWxyz);
  HCM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
