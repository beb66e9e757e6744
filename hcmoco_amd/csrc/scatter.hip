// Deterministic backward of the three gather-type PointNet++ ops:
//     grad_points[b, c, j] = sum_{q : idx[b, q] == j} coef[b, q] * grad_out[b, c, q / div]
// (group_points_grad: coef = 1, div = 1, q over npoint * nsample; gather_points_grad: coef = 1, div = 1;
//  three_interpolate_grad: coef = weight, div = 3, q = 3 n + t).   Reference: one global float atomicAdd per (channel,
// contribution) in arbitrary order (src/group_points_gpu.cu:8-25, src/sampling_gpu.cu:46-63, src/interpolate_gpu.cu:120-142).
//
// Source order, targets in LDS, no atomics of any kind, every sum in an order fixed by idx alone:
//   hcm_scatter_plan   ONCE per index tensor (forward's idx and weights are reused by every backward that needs them).
//                      The contributions are cut into STEPS of 256 sources (lane l of a wave owns sources 4l .. 4l+3 of
//                      the step) and, inside a step, into DOMAINS: the `div` contributions of source 4l+i of all 64
//                      lanes.  Contributions of one domain that hit the same target are the only ones that can collide in
//                      the kernel below, so each gets its RANK among them (order: t, then lane).  Targets with more than
//                      4 contributions in a domain are flagged HEAVY (an empty-mask image sends every pixel of pts2depth
//                      to points 0, 1, 2; background pixels of a row share their nearest silhouette points).
//                      plan[b][step][t][lane][i] = target | rank << 16 | heavy << 18 (-1: past the end), the weights
//                      re-laid the same way.
//   hcm_scatter_add_planned   workgroup = (b, block of CBL channels): the accumulators acc[CBL][m] of ALL targets live in
//                      LDS; wave w OWNS channels w*CPW .. w*CPW+CPW-1 -- no other wave touches their rows, so there is
//                      nothing to synchronise -- and streams their grad_out rows once, coalesced (one float4 per lane,
//                      channel and step; a ring of loads in flight).  Per domain: round r = the lanes of rank r do a
//                      plain LDS read - add - write (ranks make the addresses of one round distinct, the in-order LDS
//                      pipe orders the rounds); a heavy target is summed across the wave first (masked DPP tree) and
//                      added by one lane.  Sum order of a target: step, source-in-lane, round / tree -- a function of
//                      idx only: bit-reproducible.
// History: r02 hcm_scatter_add_lds (LDS float atomics, one dependent load per contribution and channel: latency bound,
// 1.0 TB/s in tools/bench_pointnet2.py's unit, not reproducible); r03a target-sorted contributions + segmented wave scans
// (deterministic, but one 4-byte gather and 12 cross-lane moves per contribution and channel: 13 ms on pts2depth).
#include "hcm_common.h"
#include "../../include/hcmoco_hip.h"

namespace {

using namespace hcm;

constexpr int kSPL = 4;                 // sources per lane and step
constexpr int kStep = 64 * kSPL;        // sources per step
constexpr int kMaxAcc = 36 * 1024;      // accumulators per workgroup (144 KB of LDS)
constexpr int kLight = 4;               // up to this many contributions per target and domain go by rounds

__device__ __forceinline__ float comp(const float4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

// one wave per (b, step)
template <int DIV>
__global__ __launch_bounds__(256) void plan_kernel(const int* __restrict__ idx, const float* __restrict__ coef, int B, int Qsrc,
                                                   int m, int steps, int* __restrict__ plan, float* __restrict__ wq) {
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (gw >= B * steps) return;
  const int b = gw / steps, s = gw - b * steps;
#pragma unroll
  for (int i = 0; i < kSPL; ++i) {
    const int src = kStep * s + kSPL * lane + i;
    const bool valid = src < Qsrc;
    int tgt[DIV], rank[DIV], gs[DIV];
    float w[DIV];
#pragma unroll
    for (int t = 0; t < DIV; ++t) {
      tgt[t] = -1 - (t * 64 + lane);                      // past the end: matches nobody
      w[t] = 0.f;
      if (valid) {
        const int64_t q = ((int64_t)b * Qsrc + src) * DIV + t;
        const int j = idx[q];
        tgt[t] = j < 0 ? 0 : (j >= m ? m - 1 : j);        // like a raw scatter the caller guarantees 0 <= idx < m
        w[t] = coef != nullptr ? coef[q] : 1.f;
      }
      rank[t] = 0;
      gs[t] = 0;
    }
    // rank / size of every contribution's group inside the domain, order (t, lane)
#pragma unroll
    for (int t2 = 0; t2 < DIV; ++t2)
      for (int l = 0; l < 64; ++l) {
        const int other = __builtin_amdgcn_readlane(tgt[t2], l);
#pragma unroll
        for (int t = 0; t < DIV; ++t) {
          const bool same = other == tgt[t];
          gs[t] += same;
          rank[t] += same && (t2 < t || (t2 == t && l < lane));
        }
      }
#pragma unroll
    for (int t = 0; t < DIV; ++t) {
      const bool heavy = gs[t] > kLight;
      const int p = valid ? (tgt[t] | (heavy ? (1 << 18) : (rank[t] << 16))) : -1;
      const int64_t at = ((((int64_t)b * steps + s) * DIV + t) * 64 + lane) * kSPL + i;
      plan[at] = p;
      if (wq != nullptr) wq[at] = w[t];
    }
  }
}

__device__ __forceinline__ int comp(const int4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

// One domain: DIV slots (p, w, val[CPW]) per lane.
template <int DIV, int CPW>
__device__ __forceinline__ void domain(float* __restrict__ accw, int m, int lane, const int (&p)[DIV], const float (&w)[DIV],
                                       const float (&val)[CPW]) {
  bool valid[DIV], light[DIV];
  int tgt[DIV], rank[DIV];
  bool any_heavy = false;
#pragma unroll
  for (int t = 0; t < DIV; ++t) {
    valid[t] = p[t] >= 0;
    tgt[t] = p[t] & 0xFFFF;
    rank[t] = (p[t] >> 16) & 3;
    light[t] = valid[t] && !(p[t] & (1 << 18));
    any_heavy |= valid[t] && !light[t];
  }
#pragma unroll
  for (int r = 0; r < kLight; ++r) {
    bool mine[DIV], some = false;
#pragma unroll
    for (int t = 0; t < DIV; ++t) { mine[t] = light[t] && rank[t] == r; some |= mine[t]; }
    if (r > 0 && !__any(some)) break;                      // ranks are dense: nobody at r => nobody above
    float cur[DIV][CPW];
#pragma unroll
    for (int t = 0; t < DIV; ++t)
      if (mine[t]) {
#pragma unroll
        for (int k = 0; k < CPW; ++k) cur[t][k] = accw[(size_t)k * m + tgt[t]];
      }
#pragma unroll
    for (int t = 0; t < DIV; ++t)
      if (mine[t]) {
#pragma unroll
        for (int k = 0; k < CPW; ++k) accw[(size_t)k * m + tgt[t]] = cur[t][k] + w[t] * val[k];
      }
  }
  if (__any(any_heavy)) {
    unsigned long long hm[DIV];
#pragma unroll
    for (int t = 0; t < DIV; ++t) hm[t] = __ballot(valid[t] && !light[t]);
#pragma unroll
    for (int t = 0; t < DIV; ++t)
      while (hm[t]) {
        const int L = __ffsll((long long)hm[t]) - 1;
        const int T = __builtin_amdgcn_readlane(tgt[t], L);
        float x[CPW];
#pragma unroll
        for (int k = 0; k < CPW; ++k) x[k] = 0.f;
#pragma unroll
        for (int t2 = 0; t2 < DIV; ++t2) {
          const bool in = valid[t2] && tgt[t2] == T;
#pragma unroll
          for (int k = 0; k < CPW; ++k) x[k] += in ? w[t2] * val[k] : 0.f;
          hm[t2] &= ~__ballot(in);
        }
#pragma unroll
        for (int k = 0; k < CPW; ++k) {
          const float sum = wave_sum(x[k]);
          if (lane == L) accw[(size_t)k * m + T] += sum;
        }
      }
  }
}

// grid (channel blocks, B); 64 * ceil(CBL / CPW) threads; dynamic LDS acc[CBL][m]
template <int DIV, int CPW, bool WEIGHTED, int D>
__global__ __launch_bounds__(1024) void scatter_planned_kernel(const float* __restrict__ grad_out, const int* __restrict__ plan,
                                                                const float* __restrict__ wq, int C, int Qsrc, int m, int steps,
                                                                int CBL, float* __restrict__ grad_points) {
  extern __shared__ __attribute__((aligned(16))) float acc[];      // [CBL][m]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, c0 = blockIdx.x * CBL;
  const int nc = min(CBL, C - c0);
  for (int e = tid; e < CBL * m; e += blockDim.x) acc[e] = 0.f;
  __syncthreads();
  const int wch = wave * CPW;
  if (wch < nc) {
    const bool vec = (Qsrc & 3) == 0;
    const float* rows[CPW];
#pragma unroll
    for (int k = 0; k < CPW; ++k)                                 // channels past the block's end: a copy into a spare row
      rows[k] = grad_out + ((int64_t)b * C + c0 + min(wch + k, nc - 1)) * Qsrc;
    float* accw = acc + (size_t)wch * m;
    const int4* pl = reinterpret_cast<const int4*>(plan) + (int64_t)b * steps * (DIV * 64) + lane;
    const float4* wl = WEIGHTED ? reinterpret_cast<const float4*>(wq) + (int64_t)b * steps * (DIV * 64) + lane : nullptr;
    float4 ring[D][CPW];
    int4 Pb[2][DIV];
    float4 Wb[2][DIV];
    auto load_rows = [&](float4 (&dst)[CPW], int s) {
      const int src = kStep * s + kSPL * lane;
#pragma unroll
      for (int k = 0; k < CPW; ++k) {
        if (vec) {
          dst[k] = src < Qsrc ? *reinterpret_cast<const float4*>(rows[k] + src) : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
          dst[k].x = src < Qsrc ? rows[k][src] : 0.f;
          dst[k].y = src + 1 < Qsrc ? rows[k][src + 1] : 0.f;
          dst[k].z = src + 2 < Qsrc ? rows[k][src + 2] : 0.f;
          dst[k].w = src + 3 < Qsrc ? rows[k][src + 3] : 0.f;
        }
      }
    };
    auto load_plan = [&](int4 (&P)[DIV], float4 (&W)[DIV], int s) {
#pragma unroll
      for (int t = 0; t < DIV; ++t) {
        P[t] = pl[((int64_t)s * DIV + t) * 64];
        if (WEIGHTED) W[t] = wl[((int64_t)s * DIV + t) * 64];
      }
    };
    load_plan(Pb[0], Wb[0], 0);
#pragma unroll
    for (int d = 0; d < D; ++d)
      if (d < steps) load_rows(ring[d], d);
    static_assert(D % 2 == 0, "the plan double buffer alternates with the unrolled step");
    for (int s0 = 0; s0 < steps; s0 += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int s = s0 + d;
        if (s < steps) {
          if (s + 1 < steps) load_plan(Pb[(d + 1) & 1], Wb[(d + 1) & 1], s + 1);
          float4 cur[CPW];
#pragma unroll
          for (int k = 0; k < CPW; ++k) cur[k] = ring[d][k];
          if (s + D < steps) load_rows(ring[d], s + D);
#pragma unroll
          for (int i = 0; i < kSPL; ++i) {
            float val[CPW], w[DIV];
            int p[DIV];
#pragma unroll
            for (int k = 0; k < CPW; ++k) val[k] = comp(cur[k], i);
#pragma unroll
            for (int t = 0; t < DIV; ++t) {
              p[t] = comp(Pb[d & 1][t], i);
              w[t] = WEIGHTED ? comp(Wb[d & 1][t], i) : 1.f;
            }
            domain<DIV, CPW>(accw, m, lane, p, w, val);
          }
        }
      }
    }
  }
  __syncthreads();
  float* out = grad_points + ((int64_t)b * C + c0) * m;
  for (int e = tid; e < nc * m; e += blockDim.x) out[e] = acc[e];
}

template <int DIV, bool WEIGHTED>
int launch_planned(const float* grad_out, const int* plan, const float* wq, int B, int C, int Qsrc, int m, float* grad_points,
                   hipStream_t st) {
  int CBL = kMaxAcc / m;
  if (CBL < 1) return (int)hipErrorInvalidConfiguration;           // the target axis does not fit LDS: atomic kernels
  if (CBL > (DIV == 3 ? 32 : 64)) CBL = DIV == 3 ? 32 : 64;   // (DIV 3 with 4 channels per wave does not fit 128 VGPRs)
  if (CBL > C) CBL = C;
  // keep the device busy: at least ~2 workgroups per CU when the channels allow it
  while (CBL > 1 && (long long)B * ((C + CBL - 1) / CBL) < 512) CBL = (CBL + 1) / 2;
  const int cpw = CBL > 32 ? 4 : (CBL > 16 ? 2 : 1);
  const int waves = (CBL + cpw - 1) / cpw;
  const int steps = (Qsrc + kStep - 1) / kStep;
  const size_t ldsb = (size_t)CBL * m * sizeof(float);
  const dim3 grid((C + CBL - 1) / CBL, B);
#define HCM_PLANNED(CPW, D)                                                                                              \
  do {                                                                                                                   \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(scatter_planned_kernel<DIV, CPW, WEIGHTED, D>),      \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);                           \
    if (e != hipSuccess) return (int)e;                                                                                  \
    scatter_planned_kernel<DIV, CPW, WEIGHTED, D><<<grid, 64 * waves, ldsb, st>>>(grad_out, plan, wq, C, Qsrc, m, steps, \
                                                                                 CBL, grad_points);                      \
  } while (0)
  if (cpw == 1) HCM_PLANNED(1, 6);
  else if (cpw == 2) HCM_PLANNED(2, 4);
  else if (DIV == 1) HCM_PLANNED(4, 2);
  else return (int)hipErrorInvalidConfiguration;
#undef HCM_PLANNED
  HCM_CHECK_LAUNCH();
  return 0;
}

inline bool plan_shape_ok(int B, int Qsrc, int div, int m) {
  return B > 0 && Qsrc > 0 && m > 0 && m <= 65535 && (div == 1 || div == 3) &&
         (int64_t)B * ((Qsrc + kStep - 1) / kStep) * kStep * div < ((int64_t)1 << 31);
}

}  // namespace

extern "C" {

size_t hcm_scatter_plan_elems(int B, int Qsrc, int div, int m) {
  if (!plan_shape_ok(B, Qsrc, div, m)) return 0;
  return (size_t)B * ((Qsrc + kStep - 1) / kStep) * kStep * div;
}

int hcm_scatter_plan(const int* idx, const float* coef, int B, int Qsrc, int div, int m, int* plan, float* plan_coef,
                     hcm_stream_t stream) {
  if (!plan_shape_ok(B, Qsrc, div, m) || !idx || !plan || ((coef != nullptr) != (plan_coef != nullptr)))
    return (int)hipErrorInvalidValue;
  const int steps = (Qsrc + kStep - 1) / kStep;
  const int waves = B * steps;
  if (div == 1) plan_kernel<1><<<(waves + 3) / 4, 256, 0, (hipStream_t)stream>>>(idx, coef, B, Qsrc, m, steps, plan, plan_coef);
  else plan_kernel<3><<<(waves + 3) / 4, 256, 0, (hipStream_t)stream>>>(idx, coef, B, Qsrc, m, steps, plan, plan_coef);
  HCM_CHECK_LAUNCH();
  return 0;
}

int hcm_scatter_add_planned(const float* grad_out, const int* plan, const float* plan_coef, int B, int C, int Qsrc, int m,
                            int div, float* grad_points, hcm_stream_t stream) {
  if (!plan_shape_ok(B, Qsrc, div, m) || C <= 0 || !grad_out || !plan || !grad_points) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  if (div == 1) {
    return plan_coef ? launch_planned<1, true>(grad_out, plan, plan_coef, B, C, Qsrc, m, grad_points, st)
                     : launch_planned<1, false>(grad_out, plan, nullptr, B, C, Qsrc, m, grad_points, st);
  }
  return plan_coef ? launch_planned<3, true>(grad_out, plan, plan_coef, B, C, Qsrc, m, grad_points, st)
                   : launch_planned<3, false>(grad_out, plan, nullptr, B, C, Qsrc, m, grad_points, st);
}

}  // extern "C"
