// Deterministic backward of the three gather-type PointNet++ ops:
//     grad_points[b, c, j] = sum_{q : idx[b, q] == j} coef[b, q] * grad_out[b, c, q / div]
// (group_points_grad: coef = 1, div = 1, q over npoint * nsample; gather_points_grad: coef = 1, div = 1;
//  three_interpolate_grad: coef = weight, div = 3, q = 3 n + t).   Reference: one global float atomicAdd per (channel,
// contribution) in arbitrary order (src/group_points_gpu.cu:8-25, src/sampling_gpu.cu:46-63, src/interpolate_gpu.cu:120-142).
//
// Source order, targets in LDS, no atomics of any kind, every sum in an order fixed by idx alone:
//   hcm_scatter_plan   ONCE per index tensor (forward's idx and weights are reused by every backward that needs them).
//                      Lane l of a wave owns the contiguous piece [l R, (l+1) R) of the source axis and walks it four
//                      sources a STEP.  The div (target, weight) pairs of a source are ordered by target.  Consecutive
//                      sources of a lane that name the same target in the same slot form a RUN (adjacent pixels of
//                      pts2depth share their nearest points; background pixels all see the same silhouette points; an
//                      empty-mask image sends everything to points 0, 1, 2; a sparse ball repeats its first index): a run
//                      is summed in registers and only its last contribution -- a FLUSH -- touches LDS.  The flushes of
//                      one DOMAIN (source i of the step, all 64 lanes, all slots) are the only things that can collide, so
//                      each gets its RANK among the flushes to the same target (order: slot, lane); more than 4 flushes to
//                      one target in a domain are flagged HEAVY.
//                      plan[b][step][slot][lane][i] = target | class << 16 | any-flush-in-this-source << 20 (class 0-3:
//                      rank of a flush, 4: heavy flush, 5: the run goes on; -1: past the end), the weights re-laid alike.
//   hcm_scatter_add_planned   workgroup = (b, block of CBL channels): the accumulators acc[CBL][m] of ALL targets live in
//                      LDS; wave w OWNS channels w*CPW .. w*CPW+CPW-1 -- no other wave touches their rows, so there is
//                      nothing to synchronise -- and streams their grad_out rows once (one float4 per lane, channel and
//                      step; three steps of data + plan + weights in flight, waits exact: see the loop).  Per domain:
//                      run += w * g; if nobody flushes (one compare + ballot) that is all.  Otherwise round r = the
//                      flushing lanes of rank r do a plain LDS read - add - write (ranks make the addresses of one round
//                      distinct, the in-order LDS pipe orders the rounds; a lane that sits a round out works on a spare
//                      slot of its own, so there is no exec masking), a heavy target is summed across the wave first
//                      (masked DPP tree) and added by one lane.  Sum order of a target: lane-local source order inside a
//                      run; runs by step, source, round / tree -- a function of idx only: bit-reproducible.
// Measured on the pts2depth backward (B=32, c=128, 65 536 pixels -> 4096 points): 1.3 ms on random indices (LDS pipe 66 %
// busy, 59 % of it bank conflicts of the random read-add-writes; VALU 55 %), 1.9 ms on true nearest neighbours of a pixel
// grid whose points sit in one quarter of the image, 1.8 ms in the HRNetPN step (was 4.1 ms / 9.2 ms without runs).
// History: r02 hcm_scatter_add_lds (LDS float atomics, one dependent load per contribution and channel: latency bound,
// 4.1 ms, not reproducible); r03a target-sorted contributions + segmented wave scans (deterministic, but one 4-byte gather
// and 12 cross-lane moves per contribution and channel: 13 ms); r03b source order with ranks but lane l = sources 4l..4l+3
// and no runs (1.7 ms random, 5.1 ms on nearest neighbours, 9.2 ms in the step: most domains needed 3-4 rounds + heavy sums).
#include "hcm_common.h"
#include "../../include/hcmoco_hip.h"

namespace {

using namespace hcm;

constexpr int kSPL = 4;                 // sources per lane and step
constexpr int kStep = 64 * kSPL;        // sources per step
// R = region_len(Qsrc): the 64 lanes of a wave are always R sources apart.  The price: a wave's float4 loads of grad_out
// touch 64 different lines, each used up over the following steps (L2 -> L1 traffic 4x the data, measured harmless).
inline __host__ __device__ int region_len(int Qsrc) { return ((Qsrc + 63) / 64 + kSPL - 1) & ~(kSPL - 1); }
inline __host__ __device__ int step_count(int Qsrc) { return region_len(Qsrc) / kSPL; }
constexpr int kMaxAcc = 37 * 1024;      // accumulators (+ 64 spare slots per row) per workgroup: 148 KB of LDS
constexpr int kLight = 4;               // up to this many contributions per target and domain go by rounds
constexpr int kAnyFlush = 1 << 20;


// The DIV (target, weight) pairs of one source, ordered by target (ties: original slot): neighbouring pixels of pts2depth
// list the same three points in different nearest / second / third roles; ordered, their slots line up and the runs
// below get long.
template <int DIV>
__device__ __forceinline__ void load_source(const int* __restrict__ idx, const float* __restrict__ coef, int64_t q0, int m,
                                            int (&tgt)[DIV], float (&w)[DIV]) {
#pragma unroll
  for (int t = 0; t < DIV; ++t) {
    const int j = idx[q0 + t];
    tgt[t] = j < 0 ? 0 : (j >= m ? m - 1 : j);            // like a raw scatter the caller guarantees 0 <= idx < m
    w[t] = coef != nullptr ? coef[q0 + t] : 1.f;
  }
  if (DIV == 3) {                                         // three compare-exchanges, stable
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
      const int a = pass == 1 ? 1 : 0, c = a + 1;
      if (tgt[c] < tgt[a]) {
        const int tj = tgt[a]; tgt[a] = tgt[c]; tgt[c] = tj;
        const float tw = w[a]; w[a] = w[c]; w[c] = tw;
      }
    }
  }
}

// one wave per (b, step)
template <int DIV>
__global__ __launch_bounds__(256) void plan_kernel(const int* __restrict__ idx, const float* __restrict__ coef, int B, int Qsrc,
                                                   int m, int steps, int* __restrict__ plan, float* __restrict__ wq) {
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (gw >= B * steps) return;
  const int b = gw / steps, s = gw - b * steps;
  const int region = steps * kSPL;
  const int first = lane * region + kSPL * s;
  const int end = min(Qsrc, (lane + 1) * region);          // this lane's piece of the source axis ends here
  int tgt[kSPL + 1][DIV];
  float w[kSPL + 1][DIV];
#pragma unroll
  for (int i = 0; i <= kSPL; ++i) {                        // the step's four sources and the one after them
#pragma unroll
    for (int t = 0; t < DIV; ++t) { tgt[i][t] = -1; w[i][t] = 0.f; }
    if (first + i < end) load_source<DIV>(idx, coef, ((int64_t)b * Qsrc + first + i) * DIV, m, tgt[i], w[i]);
  }
#pragma unroll
  for (int i = 0; i < kSPL; ++i) {
    const bool valid = first + i < end;
    int key[DIV], rank[DIV], gs[DIV];
#pragma unroll
    for (int t = 0; t < DIV; ++t) {
      // a run ends where the lane's next source names another target in this slot (or there is no next source)
      const bool flush = valid && tgt[i][t] != tgt[i + 1][t];
      key[t] = flush ? tgt[i][t] : -1 - (t * 64 + lane);  // only flushes meet in LDS; the rest match nobody
      rank[t] = 0;
      gs[t] = 0;
    }
    // rank / size of every flush's group inside the domain, order (t, lane)
#pragma unroll
    for (int t2 = 0; t2 < DIV; ++t2)
      for (int l = 0; l < 64; ++l) {
        const int other = __builtin_amdgcn_readlane(key[t2], l);
#pragma unroll
        for (int t = 0; t < DIV; ++t) {
          const bool same = other == key[t];
          gs[t] += same;
          rank[t] += same && (t2 < t || (t2 == t && l < lane));
        }
      }
    bool any_flush = false;
#pragma unroll
    for (int t = 0; t < DIV; ++t) any_flush |= key[t] >= 0;
#pragma unroll
    for (int t = 0; t < DIV; ++t) {
      const int cls = key[t] < 0 ? 5 : (gs[t] > kLight ? 4 : rank[t]);
      // slot 0 also says whether ANY slot of this source closes a run (bit 20): one compare decides the common case
      const int p = valid ? (tgt[i][t] | (cls << 16) | ((t == 0 && any_flush) ? kAnyFlush : 0)) : -1;
      const int64_t at = ((((int64_t)b * steps + s) * DIV + t) * 64 + lane) * kSPL + i;
      plan[at] = p;
      if (wq != nullptr) wq[at] = w[i][t];
    }
  }
}

__device__ __forceinline__ float comp(const float4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }
__device__ __forceinline__ int comp(const int4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

// One domain: DIV slots (p, w) and the CPW channel values of one source per lane.  run[t][k]: the lane's open run of slot t
// (contributions to one target, summed in registers in source order); a contribution whose class is <= 4 closes the run
// and sends it to LDS.  accw: this wave's rows, row stride ms = m + 64: the last 64 floats of a row are per-lane spare
// slots, so a lane that sits a round out reads and writes its own spare instead of being masked off -- no exec juggling,
// no branches inside a round.
template <int DIV, int CPW>
__device__ __forceinline__ void domain(float* __restrict__ accw, int m, int ms, int lane, const int (&p)[DIV],
                                       const float (&w)[DIV], const float (&val)[CPW], float (&run)[DIV][CPW]) {
#pragma unroll
  for (int t = 0; t < DIV; ++t)
#pragma unroll
    for (int k = 0; k < CPW; ++k) run[t][k] = fmaf(w[t], val[k], run[t][k]);
  if (!__any(p[0] >= 0 && (p[0] & kAnyFlush))) return;     // nobody closes a run here: the common case on image-like indices
  int cls[DIV];                                            // 0..3 rank of a light flush, 4 heavy flush, 5 run goes on, 7 past the end
#pragma unroll
  for (int t = 0; t < DIV; ++t) cls[t] = (p[t] >> 16) & 7;
  int tgt[DIV];
  bool heavy = false;
#pragma unroll
  for (int t = 0; t < DIV; ++t) {
    tgt[t] = p[t] & 0xFFFF;
    heavy |= cls[t] == 4;
  }
#pragma unroll
  for (int r = 0; r < kLight; ++r) {
    if (r > 0) {
      bool some = false;
#pragma unroll
      for (int t = 0; t < DIV; ++t) some |= cls[t] == r;
      if (!__any(some)) break;                             // ranks are dense: nobody at r => nobody above
    }
    int off[DIV];
    float cur[DIV][CPW];
#pragma unroll
    for (int t = 0; t < DIV; ++t) {
      off[t] = cls[t] == r ? tgt[t] : m + lane;
#pragma unroll
      for (int k = 0; k < CPW; ++k) cur[t][k] = accw[k * ms + off[t]];
    }
#pragma unroll
    for (int t = 0; t < DIV; ++t)
#pragma unroll
      for (int k = 0; k < CPW; ++k) accw[k * ms + off[t]] = __fadd_rn(cur[t][k], run[t][k]);
  }
  if (__any(heavy)) {
    unsigned long long hm[DIV];
#pragma unroll
    for (int t = 0; t < DIV; ++t) hm[t] = __ballot(cls[t] == 4);
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
          const bool in = cls[t2] == 4 && tgt[t2] == T;
#pragma unroll
          for (int k = 0; k < CPW; ++k) x[k] = __fadd_rn(x[k], in ? run[t2][k] : 0.f);
          hm[t2] &= ~__ballot(in);
        }
#pragma unroll
        for (int k = 0; k < CPW; ++k) {
          const float sum = wave_sum(x[k]);
          if (lane == L) accw[k * ms + T] += sum;
        }
      }
  }
#pragma unroll
  for (int t = 0; t < DIV; ++t)
#pragma unroll
    for (int k = 0; k < CPW; ++k) run[t][k] = cls[t] <= 4 ? 0.f : run[t][k];
}

template <int DIV, int CPW, bool WEIGHTED>
struct Stage {
  float4 v[CPW];
  int4 P[DIV];
  float4 W[WEIGHTED ? DIV : 1];
};

// grid (channel blocks, B); 64 * ceil(CBL / CPW) threads (<= MAXT); dynamic LDS acc[CBL][m + 64]
template <int DIV, int CPW, bool WEIGHTED, bool VEC, int D, int MAXT>
__global__ __launch_bounds__(MAXT) void scatter_planned_kernel(const float* __restrict__ grad_out, const int* __restrict__ plan,
                                                                const float* __restrict__ wq, int C, int Qsrc, int m, int steps,
                                                                int CBL, float* __restrict__ grad_points) {
  extern __shared__ __attribute__((aligned(16))) float acc[];      // [CBL][ms]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, c0 = blockIdx.x * CBL;
  const int nc = min(CBL, C - c0);
  const int ms = m + 64;
  for (int e = tid; e < CBL * ms; e += blockDim.x) acc[e] = 0.f;
  __syncthreads();
  const int wch = wave * CPW;
  if (wch < nc) {
    const float* rows[CPW];
#pragma unroll
    for (int k = 0; k < CPW; ++k)                                 // channels past the block's end: a copy into a spare row
      rows[k] = grad_out + ((int64_t)b * C + c0 + min(wch + k, nc - 1)) * Qsrc;
    float* accw = acc + (size_t)wch * ms;
    const int4* pl = reinterpret_cast<const int4*>(plan) + (int64_t)b * steps * (DIV * 64) + lane;
    const float4* wl = WEIGHTED ? reinterpret_cast<const float4*>(wq) + (int64_t)b * steps * (DIV * 64) + lane : nullptr;
    Stage<DIV, CPW, WEIGHTED> ring[D];
    float run[DIV][CPW];
#pragma unroll
    for (int t = 0; t < DIV; ++t)
#pragma unroll
      for (int k = 0; k < CPW; ++k) run[t][k] = 0.f;
    auto load = [&](Stage<DIV, CPW, WEIGHTED>& st, int s) {
      const int src = lane * (steps * kSPL) + kSPL * s;
#pragma unroll
      for (int t = 0; t < DIV; ++t) {
        st.P[t] = pl[((int64_t)s * DIV + t) * 64];
        if (WEIGHTED) st.W[t] = wl[((int64_t)s * DIV + t) * 64];
      }
      // past the end: re-read valid floats (no zero-fill, no branch -- the plan sends those lanes to their spare slot)
#pragma unroll
      for (int k = 0; k < CPW; ++k) {
        if (VEC) {
          st.v[k] = *reinterpret_cast<const float4*>(rows[k] + min(src, Qsrc - 4));
        } else {
          st.v[k].x = rows[k][min(src, Qsrc - 1)];
          st.v[k].y = rows[k][min(src + 1, Qsrc - 1)];
          st.v[k].z = rows[k][min(src + 2, Qsrc - 1)];
          st.v[k].w = rows[k][min(src + 3, Qsrc - 1)];
        }
      }
    };
    // Every load below is unconditional (past either end: the nearest step again), so that the number of loads in flight
    // behind the one a step needs is the same on every path and the compiler's s_waitcnt vmcnt(N) stays exact; and the ring
    // is filled by the loop's own first trip (s0 = -D: nothing to process yet) rather than by a prologue, so each ring
    // register has ONE defining load and no copy of a register with a load in flight appears on the back edge.
    for (int s0 = -D; s0 < steps; s0 += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int s = s0 + d;
        if (s >= 0 && s < steps) {
#pragma unroll
          for (int i = 0; i < kSPL; ++i) {
            float val[CPW], w[DIV];
            int p[DIV];
#pragma unroll
            for (int k = 0; k < CPW; ++k) val[k] = comp(ring[d].v[k], i);
#pragma unroll
            for (int t = 0; t < DIV; ++t) {
              p[t] = comp(ring[d].P[t], i);
              w[t] = WEIGHTED ? comp(ring[d].W[t], i) : 1.f;
            }
            domain<DIV, CPW>(accw, m, ms, lane, p, w, val, run);
          }
        }
        asm volatile("" ::: "memory");
        load(ring[d], min(s + D, steps - 1));
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  __syncthreads();
  float* out = grad_points + ((int64_t)b * C + c0) * m;
  for (int e = tid; e < nc * m; e += blockDim.x) {
    const int ch = e / m, j = e - ch * m;
    out[e] = acc[ch * ms + j];
  }
}

template <int DIV, bool WEIGHTED>
int launch_planned(const float* grad_out, const int* plan, const float* wq, int B, int C, int Qsrc, int m, float* grad_points,
                   hipStream_t st) {
  const int ms = m + 64;
  int CBL = kMaxAcc / ms;
  if (CBL < 1) return (int)hipErrorInvalidConfiguration;           // the target axis does not fit LDS: atomic kernels
  constexpr int kCap = DIV == 3 ? 32 : 64;                         // (DIV 3 with 4 channels per wave does not fit 128 VGPRs)
  if (CBL > kCap) CBL = kCap;
  if (CBL > C) CBL = C;
  // keep the device busy: at least ~2 workgroups per CU when the channels allow it
  while (CBL > 1 && (long long)B * ((C + CBL - 1) / CBL) < 512) CBL = (CBL + 1) / 2;
  const int cpw = CBL > 32 ? 4 : (CBL > 8 ? 2 : 1);
  const int waves = (CBL + cpw - 1) / cpw;
  const int steps = step_count(Qsrc);
  const size_t ldsb = (size_t)CBL * ms * sizeof(float);
  const dim3 grid((C + CBL - 1) / CBL, B);
#define HCM_PLANNED1(CPW, D, MAXT, VEC)                                                                                           \
  do {                                                                                                                       \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(scatter_planned_kernel<DIV, CPW, WEIGHTED, VEC, D, MAXT>),    \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);                               \
    if (e != hipSuccess) return (int)e;                                                                                      \
    scatter_planned_kernel<DIV, CPW, WEIGHTED, VEC, D, MAXT><<<grid, 64 * waves, ldsb, st>>>(grad_out, plan, wq, C, Qsrc, m,      \
                                                                                       steps, CBL, grad_points);             \
  } while (0)
#define HCM_PLANNED(CPW, D, MAXT) do { if ((Qsrc & 3) == 0) HCM_PLANNED1(CPW, D, MAXT, true); else HCM_PLANNED1(CPW, D, MAXT, false); } while (0)
  // <= 8 waves of one channel each: 256 VGPRs, three steps = twelve sources (data + plan + weights) in flight; more waves:
  // 128 VGPRs, two steps
  if (cpw == 1) HCM_PLANNED(1, 3, 512);
  else if (cpw == 2) HCM_PLANNED(2, 2, 1024);
  else if constexpr (DIV == 1) HCM_PLANNED(4, 2, 1024);
  else return (int)hipErrorInvalidConfiguration;
#undef HCM_PLANNED
#undef HCM_PLANNED1
  HCM_CHECK_LAUNCH();
  return 0;
}

inline bool plan_shape_ok(int B, int Qsrc, int div, int m) {
  return B > 0 && Qsrc > 0 && m > 0 && m <= 65535 && (div == 1 || div == 3) &&
         (int64_t)B * step_count(Qsrc) * kStep * div < ((int64_t)1 << 31);
}

}  // namespace

extern "C" {

size_t hcm_scatter_plan_elems(int B, int Qsrc, int div, int m) {
  if (!plan_shape_ok(B, Qsrc, div, m)) return 0;
  return (size_t)B * step_count(Qsrc) * kStep * div;
}

int hcm_scatter_plan(const int* idx, const float* coef, int B, int Qsrc, int div, int m, int* plan, float* plan_coef,
                     hcm_stream_t stream) {
  if (!plan_shape_ok(B, Qsrc, div, m) || !idx || !plan || ((coef != nullptr) != (plan_coef != nullptr)))
    return (int)hipErrorInvalidValue;
  const int steps = step_count(Qsrc);
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
