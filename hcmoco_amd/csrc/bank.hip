// Memory-bank kernels for gfx950 (MI355X): alias draw, fused bank-NCE forward+backward,
// materialised-logits API mode, momentum update, MoCo queue.
//
// Reference behaviour: pycontrast/memory/{alias_multinomial,mem_bank,mem_moco}.py and
// pycontrast/learning/contrast_trainer.py:212-253.
//
// Design (DESIGN.md "bank NCE"): the contraction is a batched gather-GEMV -- every sample
// has its own K+1 randomly drawn 512-byte rows in each of three banks -- so it is bound by
// HBM/Infinity-Cache bandwidth, not by MFMA.  One pass over the gathered rows produces all
// six logit sets, their online-softmax statistics and the six softmax-weighted row sums the
// backward needs; nothing of size B*(K+1)*D is ever written.
//
// Lane layout: a 64-lane wave is four DPP rows of 16 lanes.  Each 16-lane row owns one bank
// row at a time: lane t holds floats [4t,4t+4) and [64+4t,64+4t+4) of it (two 16-byte loads
// that coalesce into 256-byte segments), so the dot products reduce with four DPP adds and
// never touch LDS.  A 256-thread workgroup therefore runs 16 independent "streams", each
// with its own running max / sum / weighted-row accumulators, merged once at the end.
#include <cstdlib>
#include <utility>
#include <vector>

#include <mutex>
#include <vector>

#include "hcm_common.h"
#include "bank_lean.h"
#include "../../include/hcmoco_hip.h"

namespace {

using namespace hcm;

constexpr int kWG = 256;
constexpr int kStreams = 16;
constexpr float kNegBig = -1.0e30f;   // "minus infinity" that survives a subtraction
constexpr float kInvalid = -3.0e30f;  // logit of a padded row: exp2(kInvalid - m) == 0

enum Mode { kFused = 0, kLogitsFwd = 1, kLogitsBwd = 2 };

// pair p -> bank it gathers from (0:M1 1:M2 2:M3); the query modality of the pairs is
// x1: 0,4  x2: 1,2  x3: 3,5.  Order 12,21,23,32,13,31 (mem_bank.py:186-191).
__device__ __constant__ const int kPairBank[6] = {1, 0, 2, 1, 2, 0};

__host__ __device__ inline int rows_per_wg(int B, int K1) {
  // 512 rows = 32 iterations per stream: the prologue (first gathered rows: a full HBM round trip) and the merge are
  // amortised over twice the work of r02's 256 -- the lever the r03 sweep found: in the training step 0.1244-0.1266 ms
  // against 0.1279-0.1297 ms (0.80-0.81 of the 8 TB/s peak against 0.78-0.79), 5.92 against 5.58 TB/s with 1.6 GB of banks
  int R = 512;
  while ((long long)B * ((K1 + R - 1) / R) > 8192 && R < (1 << 20)) R <<= 1;
  return R;
}

template <int NV>
struct Row3 {
  float4 v[3][NV];
};

// ---- bank storage types -----------------------------------------------------------------
// float: the reference's arithmetic.  bf16_t (BASELINE config 5): rows stored as bfloat16
// (256 B instead of 512 B per row -> half the gather traffic), every product/sum still fp32.
typedef uint16_t bf16_t;
__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const bf16_t* p) { return __uint_as_float((uint32_t)(*p) << 16); }
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }
__device__ __forceinline__ void stf(bf16_t* p, float v) {  // round to nearest even
  uint32_t u = __float_as_uint(v);
  if ((u & 0x7fffffffu) > 0x7f800000u) { *p = (bf16_t)((u >> 16) | 0x40); return; }  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  *p = (bf16_t)(u >> 16);
}
// first column held by float4 slot v of lane t (t = lane within its 16-lane row):
//   float: [4t,4t+4) and [64+4t,64+4t+4)   (two 256-byte coalesced segments per 16 lanes)
//   bf16 : [8t,8t+4) and [8t+4,8t+8)       (one 16-byte load = 8 contiguous bf16 per lane)
template <class T> __device__ __forceinline__ int colbase(int t, int v);
template <> __device__ __forceinline__ int colbase<float>(int t, int v) { return 64 * v + 4 * t; }
template <> __device__ __forceinline__ int colbase<bf16_t>(int t, int v) { return 8 * t + 4 * v; }

template <int NV>
__device__ __forceinline__ void load_rows(Row3<NV>& r, const float* __restrict__ b1,
                                          const float* __restrict__ b2,
                                          const float* __restrict__ b3, int64_t row, int t) {
  constexpr int D = 64 * NV;
  const int64_t off = row * D + 4 * t;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    r.v[0][v] = *reinterpret_cast<const float4*>(b1 + off + 64 * v);
    r.v[1][v] = *reinterpret_cast<const float4*>(b2 + off + 64 * v);
    r.v[2][v] = *reinterpret_cast<const float4*>(b3 + off + 64 * v);
  }
}
__device__ __forceinline__ void unpack8(const uint4& q, float4& lo, float4& hi) {
  lo = make_float4(__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u),
                   __uint_as_float(q.y << 16), __uint_as_float(q.y & 0xffff0000u));
  hi = make_float4(__uint_as_float(q.z << 16), __uint_as_float(q.z & 0xffff0000u),
                   __uint_as_float(q.w << 16), __uint_as_float(q.w & 0xffff0000u));
}
template <int NV>
__device__ __forceinline__ void load_rows(Row3<NV>& r, const bf16_t* __restrict__ b1,
                                          const bf16_t* __restrict__ b2,
                                          const bf16_t* __restrict__ b3, int64_t row, int t) {
  static_assert(NV == 2, "bf16 banks need D == 128");
  const int64_t off = row * 128 + 8 * t;
  const uint4 q1 = *reinterpret_cast<const uint4*>(b1 + off);
  const uint4 q2 = *reinterpret_cast<const uint4*>(b2 + off);
  const uint4 q3 = *reinterpret_cast<const uint4*>(b3 + off);
  unpack8(q1, r.v[0][0], r.v[0][1]);
  unpack8(q2, r.v[1][0], r.v[1][1]);
  unpack8(q3, r.v[2][0], r.v[2][1]);
}

// A gathered row triple as it sits in the prefetch ring: still packed (bf16: 12 VGPRs instead of 24),
// unpacked to fp32 only when its turn to be consumed comes.
template <class T, int NV> struct Packed;
template <int NV> struct Packed<float, NV> { Row3<NV> r; };
template <> struct Packed<bf16_t, 2> { uint4 q[3]; };

template <int NV>
__device__ __forceinline__ void load_packed(Packed<float, NV>& p, const float* b1, const float* b2,
                                            const float* b3, int64_t row, int t) {
  load_rows<NV>(p.r, b1, b2, b3, row, t);
}
__device__ __forceinline__ void load_packed(Packed<bf16_t, 2>& p, const bf16_t* b1, const bf16_t* b2,
                                            const bf16_t* b3, int64_t row, int t) {
  const int64_t off = row * 128 + 8 * t;
  p.q[0] = *reinterpret_cast<const uint4*>(b1 + off);
  p.q[1] = *reinterpret_cast<const uint4*>(b2 + off);
  p.q[2] = *reinterpret_cast<const uint4*>(b3 + off);
}
template <int NV>
__device__ __forceinline__ void unpack(const Packed<float, NV>& p, Row3<NV>& r) { r = p.r; }
__device__ __forceinline__ void unpack(const Packed<bf16_t, 2>& p, Row3<2>& r) {
  unpack8(p.q[0], r.v[0][0], r.v[0][1]);
  unpack8(p.q[1], r.v[1][0], r.v[1][1]);
  unpack8(p.q[2], r.v[2][0], r.v[2][1]);
}

template <int NV>
__device__ __forceinline__ float dotv(const float4 (&a)[NV], const float4 (&b)[NV]) {
  float d = dot4(a[0], b[0]);
#pragma unroll
  for (int v = 1; v < NV; ++v) d += dot4(a[v], b[v]);
  return d;
}

// ---------------------------------------------------------------------------------------
// Pass 1: one workgroup = (sample b, chunk of rows).  MODE selects what is done with the
// six dot products of each gathered row triple.
//   kFused     : online softmax + weighted row sums -> per-workgroup partials
//   kLogitsFwd : logits[p][b][k] = dot/T
//   kLogitsBwd : acc[p] += (grad_logits[p][b][k]/T) * row  -> per-workgroup partials
// ---------------------------------------------------------------------------------------
template <class T, int NV, int MODE, int NPF = 1, int MINW = 1>
__global__ __launch_bounds__(kWG, MINW) void bank_pass_kernel(
    const T* __restrict__ b1, const T* __restrict__ b2, const T* __restrict__ b3,
    const int64_t* __restrict__ idx, const float* __restrict__ x1, const float* __restrict__ x2,
    const float* __restrict__ x3, const float* __restrict__ glogits, int B, int K1, int R,
    float scale, float* __restrict__ part_m, float* __restrict__ part_s,
    float* __restrict__ part_acc, float* __restrict__ l0_out, float* __restrict__ logits_out) {
  constexpr int D = 64 * NV;
  const int b = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = lane & 15, g = lane >> 4;
  const int s = wave * 4 + g;
  const int kbeg = chunk * R;
  const int kend = min(K1, kbeg + R);
  const int niter = (kend - kbeg + kStreams - 1) / kStreams;
  const int64_t* __restrict__ idxb = idx + (int64_t)b * K1;

  float4 xq[3][NV];
  if (MODE != kLogitsBwd) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      xq[0][v] = *reinterpret_cast<const float4*>(x1 + (int64_t)b * D + colbase<T>(t, v));
      xq[1][v] = *reinterpret_cast<const float4*>(x2 + (int64_t)b * D + colbase<T>(t, v));
      xq[2][v] = *reinterpret_cast<const float4*>(x3 + (int64_t)b * D + colbase<T>(t, v));
    }
  }

  float m[6], ssum[6];
  float4 acc[6][NV];
#pragma unroll
  for (int p = 0; p < 6; ++p) {
    m[p] = kNegBig;
    ssum[p] = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[p][v] = make_float4(0.f, 0.f, 0.f, 0.f);
  }

  // software pipeline: a ring of NPF gathered (still packed) row triples in flight per stream, row
  // indices one further round ahead.  In-flight bytes per wave = NPF x 4 rows x 3 banks x row size:
  // this is what Little's law prices.
  auto ld_idx = [&](int it) -> int64_t {
    const int k = kbeg + it * kStreams + s;
    return (it < niter && k < kend) ? idxb[k] : (int64_t)0;
  };
  Packed<T, NV> ring[NPF];
  int64_t ridx[NPF];
#pragma unroll
  for (int j = 0; j < NPF; ++j) load_packed(ring[j], b1, b2, b3, ld_idx(j), t);
#pragma unroll
  for (int j = 0; j < NPF; ++j) ridx[j] = ld_idx(NPF + j);

  for (int it0 = 0; it0 < niter; it0 += NPF) {
#pragma unroll
   for (int j = 0; j < NPF; ++j) {
    const int it = it0 + j;
    if (it >= niter) break;  // workgroup-uniform
    Row3<NV> cur;
    unpack(ring[j], cur);
    if (it + NPF < niter) load_packed(ring[j], b1, b2, b3, ridx[j], t);
    ridx[j] = ld_idx(it + 2 * NPF);
    const int k = kbeg + it * kStreams + s;
    const bool valid = k < kend;

    float d[6];
    if (MODE != kLogitsBwd) {
      d[1] = dotv<NV>(xq[1], cur.v[0]);  // x2 . M1
      d[5] = dotv<NV>(xq[2], cur.v[0]);  // x3 . M1
      d[0] = dotv<NV>(xq[0], cur.v[1]);  // x1 . M2
      d[3] = dotv<NV>(xq[2], cur.v[1]);  // x3 . M2
      d[2] = dotv<NV>(xq[1], cur.v[2]);  // x2 . M3
      d[4] = dotv<NV>(xq[0], cur.v[2]);  // x1 . M3
#pragma unroll
      for (int p = 0; p < 6; ++p) d[p] = row16_sum(d[p]) * scale;
    }

    if (MODE == kFused) {
      if (k == 0 && t == 0) {
#pragma unroll
        for (int p = 0; p < 6; ++p) l0_out[b * 6 + p] = d[p];
      }
#pragma unroll
      for (int p = 0; p < 6; ++p) {
        const float l = valid ? d[p] : kInvalid;
        if (__any(l > m[p])) {  // rare after the first rows: ~ln(#rows) record highs per stream
          const float mn = fmaxf(m[p], l);
          const float a = fast_exp2(m[p] - mn);
          ssum[p] *= a;
#pragma unroll
          for (int v = 0; v < NV; ++v) scale4(acc[p][v], a);
          m[p] = mn;
        }
        const float pr = fast_exp2(l - m[p]);
        ssum[p] += pr;
        const int c = (p == 1 || p == 5) ? 0 : ((p == 0 || p == 3) ? 1 : 2);
#pragma unroll
        for (int v = 0; v < NV; ++v) fma4(acc[p][v], pr, cur.v[c][v]);
      }
    } else if (MODE == kLogitsFwd) {
      if (valid && t == 0) {
#pragma unroll
        for (int p = 0; p < 6; ++p) logits_out[((int64_t)p * B + b) * K1 + k] = d[p];
      }
    } else {  // kLogitsBwd
#pragma unroll
      for (int p = 0; p < 6; ++p) {
        const float wgt = valid ? glogits[((int64_t)p * B + b) * K1 + k] * scale : 0.f;
        const int c = (p == 1 || p == 5) ? 0 : ((p == 0 || p == 3) ? 1 : 2);
#pragma unroll
        for (int v = 0; v < NV; ++v) fma4(acc[p][v], wgt, cur.v[c][v]);
      }
    }
   }
  }
  if (MODE == kLogitsFwd) return;

  // ---- merge the 4 lane-rows of the wave (lanes ^16, ^32) ----
#pragma unroll
  for (int p = 0; p < 6; ++p) {
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
      float a = 1.f, bs = 1.f;
      if (MODE == kFused) {
        const float mo = __shfl_xor(m[p], off, 64);
        const float so = __shfl_xor(ssum[p], off, 64);
        const float mn = fmaxf(m[p], mo);
        a = fast_exp2(m[p] - mn);
        bs = fast_exp2(mo - mn);
        ssum[p] = ssum[p] * a + so * bs;
        m[p] = mn;
      }
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        float4 o;
        o.x = __shfl_xor(acc[p][v].x, off, 64);
        o.y = __shfl_xor(acc[p][v].y, off, 64);
        o.z = __shfl_xor(acc[p][v].z, off, 64);
        o.w = __shfl_xor(acc[p][v].w, off, 64);
        acc[p][v].x = acc[p][v].x * a + o.x * bs;
        acc[p][v].y = acc[p][v].y * a + o.y * bs;
        acc[p][v].z = acc[p][v].z * a + o.z * bs;
        acc[p][v].w = acc[p][v].w * a + o.w * bs;
      }
    }
  }

  // ---- merge the 4 waves through LDS, write one partial per workgroup ----
  __shared__ float lds_acc[4][6][D];
  __shared__ float lds_m[4][6];
  __shared__ float lds_s[4][6];
  if (g == 0) {
#pragma unroll
    for (int p = 0; p < 6; ++p) {
#pragma unroll
      for (int v = 0; v < NV; ++v)
        *reinterpret_cast<float4*>(&lds_acc[wave][p][colbase<T>(t, v)]) = acc[p][v];
      if (t == 0) {
        lds_m[wave][p] = m[p];
        lds_s[wave][p] = ssum[p];
      }
    }
  }
  __syncthreads();
  const int64_t pbase = ((int64_t)b * nchunks + chunk) * 6;
  for (int e = threadIdx.x; e < 6 * D; e += kWG) {
    const int p = e / D, col = e - p * D;
    float out;
    if (MODE == kFused) {
      const float M = fmaxf(fmaxf(lds_m[0][p], lds_m[1][p]), fmaxf(lds_m[2][p], lds_m[3][p]));
      float sc[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) sc[w] = fast_exp2(lds_m[w][p] - M);
      out = lds_acc[0][p][col] * sc[0] + lds_acc[1][p][col] * sc[1] + lds_acc[2][p][col] * sc[2] +
            lds_acc[3][p][col] * sc[3];
      if (col == 0) {
        part_m[pbase + p] = M;
        part_s[pbase + p] = lds_s[0][p] * sc[0] + lds_s[1][p] * sc[1] + lds_s[2][p] * sc[2] +
                            lds_s[3][p] * sc[3];
      }
    } else {
      out = lds_acc[0][p][col] + lds_acc[1][p][col] + lds_acc[2][p][col] + lds_acc[3][p][col];
    }
    part_acc[(pbase + p) * D + col] = out;
  }
}

// ---------------------------------------------------------------------------------------
// Row selection of _compute_loss_accuracy (contrast_trainer.py:223-250), evaluated by every
// workgroup that needs it (B is tiny).  cnt[p] = |R_p|, 0 marks a degenerate set.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ bool row_selected(int p, int b, const int32_t* use_depth,
                                             const int32_t* use_rgb, bool any_sel) {
  if (use_rgb != nullptr) {
    if (!any_sel) return p >= 4;
    return use_depth[b] == 1 && use_rgb[b] == 1;
  }
  if (use_depth != nullptr) {
    if (!any_sel) return p >= 4;
    return p >= 4 ? true : (use_depth[b] == 1);
  }
  return true;
}

// Pass 2 (fused): one workgroup per sample merges the chunk partials of that sample.
template <class T, int D>
__global__ __launch_bounds__(kWG) void bank_finish_kernel(
    const T* __restrict__ b1, const T* __restrict__ b2, const T* __restrict__ b3,
    const int64_t* __restrict__ idx, const int32_t* __restrict__ use_depth,
    const int32_t* __restrict__ use_rgb, int B, int K1, int nchunks, float invT,
    const float* __restrict__ part_m, const float* __restrict__ part_s,
    const float* __restrict__ part_acc, const float* __restrict__ l0,
    float* __restrict__ ps_loss, float* __restrict__ ps_correct, float* __restrict__ gx1,
    float* __restrict__ gx2, float* __restrict__ gx3) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];  // scale[nchunks][6]
  __shared__ float sM[6], sS[6];
  __shared__ float sG[6][D];
  __shared__ int sCntSel;
  const int b = blockIdx.x, tid = threadIdx.x;

  if (tid == 0) sCntSel = 0;
  __syncthreads();
  {
    int sel = 0;
    for (int i = tid; i < B; i += kWG) {
      bool v = true;
      if (use_rgb != nullptr) v = use_depth[i] == 1 && use_rgb[i] == 1;
      else if (use_depth != nullptr) v = use_depth[i] == 1;
      sel += v ? 1 : 0;
    }
    if (sel) atomicAdd(&sCntSel, sel);
  }
  if (tid < 192) {  // 32 threads per pair: strided partial max / sum, then a fixed 32-lane tree
    const int p = tid >> 5, j = tid & 31;
    float M = kNegBig;
    for (int c = j; c < nchunks; c += 32) M = fmaxf(M, part_m[((int64_t)b * nchunks + c) * 6 + p]);
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) M = fmaxf(M, __shfl_xor(M, off, 32));
    float S = 0.f;
    for (int c = j; c < nchunks; c += 32) {
      const float sc = fast_exp2(part_m[((int64_t)b * nchunks + c) * 6 + p] - M);
      dyn[c * 6 + p] = sc;
      S += part_s[((int64_t)b * nchunks + c) * 6 + p] * sc;
    }
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) S += __shfl_xor(S, off, 32);
    if (j == 0) { sM[p] = M; sS[p] = S; }
  }
  __syncthreads();
  const int cnt_sel = sCntSel;
  const bool any_sel = cnt_sel > 0;
  const bool masked = use_depth != nullptr;  // use_rgb implies use_depth

  const int64_t r0 = idx[(int64_t)b * K1];
  for (int e = tid; e < 6 * D; e += kWG) {
    const int p = e / D, col = e - p * D;
    float a = 0.f;
    for (int c = 0; c < nchunks; ++c)
      a += part_acc[(((int64_t)b * nchunks + c) * 6 + p) * D + col] * dyn[c * 6 + p];
    const T* bank = (kPairBank[p] == 0) ? b1 : (kPairBank[p] == 1 ? b2 : b3);
    const float row0 = ldf(bank + r0 * D + col);
    // |R_p| : sets 0-3 follow the mask, sets 4-5 use every row unless use_rgb is given
    int cnt;
    if (!masked) cnt = B;
    else if (use_rgb != nullptr) cnt = any_sel ? cnt_sel : (p >= 4 ? B : 0);
    else cnt = (p >= 4) ? B : cnt_sel;
    const bool selrow = row_selected(p, b, use_depth, use_rgb, any_sel) && cnt > 0;
    sG[p][col] = selrow ? (a / sS[p] - row0) * (invT / (float)cnt) : 0.f;
  }
  if (tid < 6) {
    const int p = tid;
    const float lse2 = sM[p] + fast_log2(sS[p]);
    const float l02 = l0[b * 6 + p];
    ps_loss[b * 6 + p] = (lse2 - l02) * HCM_LN2;
    ps_correct[b * 6 + p] = (l02 >= sM[p]) ? 1.f : 0.f;
  }
  __syncthreads();
  for (int e = tid; e < 3 * D; e += kWG) {
    const int a = e / D, col = e - a * D;
    // x1: pairs 0,4   x2: pairs 1,2   x3: pairs 3,5
    const float v = (a == 0) ? sG[0][col] + sG[4][col]
                             : (a == 1 ? sG[1][col] + sG[2][col] : sG[3][col] + sG[5][col]);
    float* out = (a == 0) ? gx1 : (a == 1 ? gx2 : gx3);
    out[(int64_t)b * D + col] = v;
  }
}

// Pass 3 (fused): fixed-shape (deterministic) reduction over the batch -> 6 losses, 6 accuracies.
// One wave per pair: lane i takes samples i, i+64, ...; wave_sum is a fixed butterfly.
__global__ __launch_bounds__(384) void bank_reduce_kernel(const float* __restrict__ ps_loss,
                                                          const float* __restrict__ ps_correct,
                                                          const int32_t* __restrict__ use_depth,
                                                          const int32_t* __restrict__ use_rgb, int B,
                                                          float* __restrict__ losses6,
                                                          float* __restrict__ accs6) {
  const int p = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float fsel = 0.f;
  for (int i = lane; i < B; i += 64) {
    bool v = true;
    if (use_rgb != nullptr) v = use_depth[i] == 1 && use_rgb[i] == 1;
    else if (use_depth != nullptr) v = use_depth[i] == 1;
    fsel += v ? 1.f : 0.f;
  }
  const bool any_sel = wave_sum(fsel) > 0.f;
  float sl = 0.f, sc = 0.f, cnt = 0.f;
  for (int i = lane; i < B; i += 64) {
    if (row_selected(p, i, use_depth, use_rgb, any_sel)) {
      sl += ps_loss[i * 6 + p];
      sc += ps_correct[i * 6 + p];
      cnt += 1.f;
    }
  }
  sl = wave_sum(sl); sc = wave_sum(sc); cnt = wave_sum(cnt);
  if (lane == 0) {
    const bool degenerate = (use_depth != nullptr) && !any_sel && p < 4;
    losses6[p] = (degenerate || cnt == 0.f) ? 0.f : sl / cnt;
    accs6[p] = (degenerate || cnt == 0.f) ? 0.f : 100.f * sc / cnt;
  }
}

// Pass 2 (API-mode backward): gx_a[b] = sum over chunks and over the two pairs of a.
template <int D>
__global__ __launch_bounds__(kWG) void bank_logits_bwd_finish_kernel(
    int nchunks, const float* __restrict__ part_acc, float* __restrict__ gx1,
    float* __restrict__ gx2, float* __restrict__ gx3) {
  const int b = blockIdx.x;
  for (int e = threadIdx.x; e < 3 * D; e += kWG) {
    const int a = e / D, col = e - a * D;
    const int p0 = (a == 0) ? 0 : (a == 1 ? 1 : 3);
    const int p1 = (a == 0) ? 4 : (a == 1 ? 2 : 5);
    float v = 0.f;
    for (int c = 0; c < nchunks; ++c) {
      const int64_t base = ((int64_t)b * nchunks + c) * 6;
      v += part_acc[(base + p0) * D + col] + part_acc[(base + p1) * D + col];
    }
    float* out = (a == 0) ? gx1 : (a == 1 ? gx2 : gx3);
    out[(int64_t)b * D + col] = v;
  }
}

// ---------------------------------------------------------------------------------------
// Row 3: momentum update.  grid (BW, 3 banks), one wave per (j, bank).
// ---------------------------------------------------------------------------------------
template <class T, int D>
__global__ __launch_bounds__(64) void bank_update_kernel(T* __restrict__ b1,
                                                         T* __restrict__ b2,
                                                         T* __restrict__ b3,
                                                         const float* __restrict__ x1,
                                                         const float* __restrict__ x2,
                                                         const float* __restrict__ x3,
                                                         const int64_t* __restrict__ y, int BW,
                                                         float mom, float one_minus_mom, int64_t ldx,
                                                         int64_t n, int* __restrict__ oob) {
  const int j = blockIdx.x, which = blockIdx.y, lane = threadIdx.x;
  int64_t row = y[j];
  // checked form (oob != nullptr): indices are clamped into [0, n) -- nothing outside the banks is touched -- and
  // a sticky flag records that a clamp changed something (the reference's index_copy_ would device-assert)
  if (oob != nullptr) {
    const int64_t safe = row < 0 ? 0 : (row >= n ? n - 1 : row);
    if (safe != row && which == 0 && lane == 0) atomicOr(oob, 1);
    row = safe;
  }
  // last occurrence wins (torch CPU index_copy_ order)
  bool later_dup = false;
  for (int jj = j + 1 + lane; jj < BW; jj += 64) {
    int64_t other = y[jj];
    if (oob != nullptr) other = other < 0 ? 0 : (other >= n ? n - 1 : other);
    later_dup |= (other == row);
  }
  if (__any(later_dup)) return;
  T* bank = which == 0 ? b1 : (which == 1 ? b2 : b3);
  const float* x = which == 0 ? x1 : (which == 1 ? x2 : x3);
  constexpr int PER = D / 64;
  float w[PER];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int col = lane + 64 * i;
    const float old = ldf(bank + row * D + col);
    const float xv = x[(int64_t)j * ldx + col];
    w[i] = __fadd_rn(__fmul_rn(old, mom), __fmul_rn(xv, one_minus_mom));
    ss = fmaf(w[i], w[i], ss);
  }
  ss = wave_sum(ss);
  const float denom = fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
  for (int i = 0; i < PER; ++i) stf(bank + row * D + lane + 64 * i, w[i] / denom);
}

// ---------------------------------------------------------------------------------------
// Row 1: alias draw with Philox4x32-10.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWG) void alias_draw_kernel(const float* __restrict__ prob,
                                                         const int64_t* __restrict__ alias,
                                                         int64_t n, const int64_t* __restrict__ y,
                                                         int K1, int64_t total, uint64_t seed,
                                                         uint64_t offset, int64_t* __restrict__ idx,
                                                         int* __restrict__ oob) {
  for (int64_t e = (int64_t)blockIdx.x * kWG + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * kWG) {
    if (y != nullptr && (e % K1) == 0) {
      int64_t v = y[e / K1];
      if (oob != nullptr) {        // checked form: clamp the positive's row index, remember that it happened
        const int64_t safe = v < 0 ? 0 : (v >= n ? n - 1 : v);
        if (safe != v) atomicOr(oob, 1);
        v = safe;
      }
      idx[e] = v;
      continue;
    }
    uint32_t c[4] = {(uint32_t)e, (uint32_t)((uint64_t)e >> 32), (uint32_t)offset,
                     (uint32_t)(offset >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint64_t r64 = ((uint64_t)c[0] << 32) | c[1];
    const int64_t kk = (int64_t)(r64 % (uint64_t)n);
    const float u = (float)(c[2] >> 8) * 5.9604644775390625e-08f;  // 2^-24
    idx[e] = (u < prob[kk]) ? kk : alias[kk];
  }
}

// ---------------------------------------------------------------------------------------
// MoCo queue (mem_moco.py:15-49): logits [B, K+1], enqueue.
// One 16-lane row per queue row, queries staged in LDS.
// ---------------------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(kWG) void moco_logits_kernel(const float* __restrict__ q,
                                                          const float* __restrict__ kpos,
                                                          const float* __restrict__ queue, int B,
                                                          int K, float invT,
                                                          float* __restrict__ logits) {
  constexpr int D = 64 * NV;
  extern __shared__ __attribute__((aligned(16))) float sq[];  // [B][D]
  for (int e = threadIdx.x; e < B * D; e += kWG) sq[e] = q[e];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, t = lane & 15, g = lane >> 4;
  const int rows_total = K + B;  // K queue rows, then B "positive" rows (one per sample)
  for (int r = (blockIdx.x * 4 + wave) * 4 + g; r < rows_total; r += gridDim.x * 16) {
    float4 row[NV];
    const float* src = (r < K) ? queue + (int64_t)r * D : kpos + (int64_t)(r - K) * D;
#pragma unroll
    for (int v = 0; v < NV; ++v) row[v] = *reinterpret_cast<const float4*>(src + 64 * v + 4 * t);
    if (r < K) {
      for (int b = 0; b < B; ++b) {
        float d = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v)
          d += dot4(row[v], *reinterpret_cast<const float4*>(&sq[b * D + 64 * v + 4 * t]));
        d = row16_sum(d);
        if (t == 0) logits[(int64_t)b * (K + 1) + 1 + r] = d * invT;
      }
    } else {
      const int b = r - K;
      float d = 0.f;
#pragma unroll
      for (int v = 0; v < NV; ++v)
        d += dot4(row[v], *reinterpret_cast<const float4*>(&sq[b * D + 64 * v + 4 * t]));
      d = row16_sum(d);
      if (t == 0) logits[(int64_t)b * (K + 1)] = d * invT;
    }
  }
}

__global__ void moco_enqueue_kernel(float* __restrict__ queue, const float* __restrict__ all_k,
                                    int n_new, int K, int D, int64_t index) {
  // rows (index + j) % K ; if n_new > K later rows overwrite earlier ones (index_copy_ order)
  const int j = blockIdx.x;
  if (j + K < n_new) return;  // a later j' = j + K writes the same slot and wins
  const int64_t dst = (index + j) % K;
  for (int c = threadIdx.x; c < D; c += blockDim.x) queue[dst * D + c] = all_k[(int64_t)j * D + c];
}

inline bool dim_ok(int D) { return D == 64 || D == 128; }

struct FusedWs {
  float *part_m, *part_s, *part_acc, *l0, *ps_loss, *ps_correct;
  size_t bytes;
};
inline FusedWs carve(void* ws, int B, int K1, int D) {
  const int R = rows_per_wg(B, K1);
  const size_t nch = (size_t)((K1 + R - 1) / R);
  size_t off = 0;  // in floats
  auto take = [&](size_t n) {
    const size_t o = off;
    off += (n + 3) & ~(size_t)3;  // keep every region 16-byte aligned
    return o;
  };
  const size_t o_m = take((size_t)B * nch * 6), o_s = take((size_t)B * nch * 6);
  const size_t o_l0 = take((size_t)B * 6), o_pl = take((size_t)B * 6), o_pc = take((size_t)B * 6);
  const size_t o_acc = take((size_t)B * nch * 6 * D);
  FusedWs o;
  float* base = reinterpret_cast<float*>(ws);
  o.part_m = base ? base + o_m : nullptr;
  o.part_s = base ? base + o_s : nullptr;
  o.l0 = base ? base + o_l0 : nullptr;
  o.ps_loss = base ? base + o_pl : nullptr;
  o.ps_correct = base ? base + o_pc : nullptr;
  o.part_acc = base ? base + o_acc : nullptr;
  o.bytes = off * sizeof(float);
  return o;
}

// ---- optional in-library timing of selected kernels (bench.py roofline objects) ----------------
// The only process-global state of the library; off by default; mutex-protected.
constexpr int kProfTags = HCM_PROF_NTAGS;
struct ProfState {
  std::mutex mu;
  bool on = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> spans[kProfTags];
  double work[kProfTags] = {};
};
ProfState& prof() {
  static ProfState p;
  return p;
}

}  // namespace

namespace hcm {
ProfSpan::ProfSpan(int tag_, hipStream_t s, double work) : st(s), tag(tag_) {
  bool on;
  {
    std::lock_guard<std::mutex> lk(prof().mu);
    on = prof().on && tag >= 0 && tag < kProfTags && prof().spans[tag].size() < 65536;
    if (on) prof().work[tag] += work;
  }
  if (on && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess)
    hipEventRecord(e0, st);
  else
    e0 = nullptr;
}
void ProfSpan::add_work(double work) {
  if (e0 != nullptr) {
    std::lock_guard<std::mutex> lk(prof().mu);
    prof().work[tag] += work;
  }
}
void ProfSpan::stop() {
  if (e0 != nullptr) {
    hipEventRecord(e1, st);
    std::lock_guard<std::mutex> lk(prof().mu);
    prof().spans[tag].emplace_back(e0, e1);
    e0 = nullptr;
  }
}
}  // namespace hcm

// ---- host launch logic, shared by the fp32 and bf16 entry points -----------------------------
namespace {

template <class T>
int fused_impl(const T* bank1, const T* bank2, const T* bank3, const int64_t* idx, const float* x1,
               const float* x2, const float* x3, const int32_t* use_depth, const int32_t* use_rgb,
               int B, int K1, int D, float T_, float* losses6, float* accs6, float* gx1, float* gx2,
               float* gx3, void* workspace, size_t workspace_bytes, hcm_stream_t stream) {
  constexpr bool kBf16 = sizeof(T) == 2;
  if (!dim_ok(D) || (kBf16 && D != 128) || B <= 0 || K1 <= 0 || !(T_ > 0.f))
    return (int)hipErrorInvalidValue;
  if (use_rgb != nullptr && use_depth == nullptr) return (int)hipErrorInvalidValue;
  const FusedWs ws = carve(workspace, B, K1, D);
  if (workspace == nullptr || workspace_bytes < ws.bytes) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  const int R = rows_per_wg(B, K1);
  const int nch = (K1 + R - 1) / R;
  const float invT = (float)(1.0 / (double)T_);
  const float scale2 = (float)((double)HCM_LOG2E / (double)T_);
  dim3 grid(nch, B);
  hcm::ProfSpan span(HCM_PROF_BANK_PASS, st);  // brackets the dominant kernel only
  if (D == 128) {
    // D = 128: the instruction-lean kernel of csrc/bank_lean.hip -- ring depth 2 for fp32 rows (6.22 / 5.90 / 6.22 TB/s on the
    // HBM-resident cells 1M rows K=16384, 4M rows K=16384, 4M rows K=65536; 0.83-0.85 of the 8 TB/s peak inside the training
    // step), ring depth 4 for bf16 rows (5.66 / 5.05 / 5.76 TB/s, 7.58 at K = 131072; rings 5 and 6 spill at two waves per
    // SIMD).  Everything else that was built and measured on the way -- register rings of depth 1-6 in the general kernel
    // below, an LDS-DMA ring (global_load ... lds) of 2-6 stages, a 32-lanes-per-row bf16 kernel -- is recorded in DESIGN 4.1 /
    // 4.6 with its numbers; the variants and their HCM_BANK_VARIANT / HCM_BANK_ROWS switches were removed in r05.
    {   // the helper returns (and clears) the launch status itself: HCM_CHECK_LAUNCH below would see hipSuccess
      const int rc = hcm::bank_pass_lean_launch(kBf16 ? 1 : 0, kBf16 ? 4 : 2, bank1, bank2, bank3, idx, x1, x2, x3, B, K1, R,
                                                scale2, ws.part_m, ws.part_s, ws.part_acc, ws.l0, stream);
      if (rc != 0) { span.stop(); return rc; }
    }
    span.stop();
    HCM_CHECK_LAUNCH();
    bank_finish_kernel<T, 128><<<B, kWG, (size_t)nch * 6 * sizeof(float), st>>>(
        bank1, bank2, bank3, idx, use_depth, use_rgb, B, K1, nch, invT, ws.part_m, ws.part_s,
        ws.part_acc, ws.l0, ws.ps_loss, ws.ps_correct, gx1, gx2, gx3);
  } else {
    if constexpr (!kBf16) {
      bank_pass_kernel<T, 1, kFused><<<grid, kWG, 0, st>>>(bank1, bank2, bank3, idx, x1, x2, x3,
                                                           nullptr, B, K1, R, scale2, ws.part_m,
                                                           ws.part_s, ws.part_acc, ws.l0, nullptr);
      span.stop();
      HCM_CHECK_LAUNCH();
      bank_finish_kernel<T, 64><<<B, kWG, (size_t)nch * 6 * sizeof(float), st>>>(
          bank1, bank2, bank3, idx, use_depth, use_rgb, B, K1, nch, invT, ws.part_m, ws.part_s,
          ws.part_acc, ws.l0, ws.ps_loss, ws.ps_correct, gx1, gx2, gx3);
    }
  }
  HCM_CHECK_LAUNCH();
  bank_reduce_kernel<<<1, 384, 0, st>>>(ws.ps_loss, ws.ps_correct, use_depth, use_rgb, B, losses6,
                                       accs6);
  HCM_CHECK_LAUNCH();
  return 0;
}

template <class T>
int logits_fwd_impl(const T* bank1, const T* bank2, const T* bank3, const int64_t* idx,
                    const float* x1, const float* x2, const float* x3, int B, int K1, int D, float T_,
                    float* logits, hcm_stream_t stream) {
  constexpr bool kBf16 = sizeof(T) == 2;
  if (!dim_ok(D) || (kBf16 && D != 128) || B <= 0 || K1 <= 0 || !(T_ > 0.f))
    return (int)hipErrorInvalidValue;
  const int R = rows_per_wg(B, K1);
  dim3 grid((K1 + R - 1) / R, B);
  const float invT = (float)(1.0 / (double)T_);
  hipStream_t st = (hipStream_t)stream;
  if (D == 128) {
    bank_pass_kernel<T, 2, kLogitsFwd><<<grid, kWG, 0, st>>>(bank1, bank2, bank3, idx, x1, x2, x3,
                                                             nullptr, B, K1, R, invT, nullptr,
                                                             nullptr, nullptr, nullptr, logits);
  } else {
    if constexpr (!kBf16)
      bank_pass_kernel<T, 1, kLogitsFwd><<<grid, kWG, 0, st>>>(bank1, bank2, bank3, idx, x1, x2, x3,
                                                               nullptr, B, K1, R, invT, nullptr,
                                                               nullptr, nullptr, nullptr, logits);
  }
  HCM_CHECK_LAUNCH();
  return 0;
}

template <class T>
int logits_bwd_impl(const T* bank1, const T* bank2, const T* bank3, const int64_t* idx,
                    const float* grad_logits, int B, int K1, int D, float T_, float* gx1, float* gx2,
                    float* gx3, void* workspace, size_t workspace_bytes, hcm_stream_t stream) {
  constexpr bool kBf16 = sizeof(T) == 2;
  if (!dim_ok(D) || (kBf16 && D != 128) || B <= 0 || K1 <= 0 || !(T_ > 0.f))
    return (int)hipErrorInvalidValue;
  const FusedWs ws = carve(workspace, B, K1, D);
  if (workspace == nullptr || workspace_bytes < ws.bytes) return (int)hipErrorInvalidValue;
  const int R = rows_per_wg(B, K1);
  const int nch = (K1 + R - 1) / R;
  dim3 grid(nch, B);
  const float invT = (float)(1.0 / (double)T_);
  hipStream_t st = (hipStream_t)stream;
  if (D == 128) {
    bank_pass_kernel<T, 2, kLogitsBwd><<<grid, kWG, 0, st>>>(bank1, bank2, bank3, idx, nullptr,
                                                             nullptr, nullptr, grad_logits, B, K1, R,
                                                             invT, nullptr, nullptr, ws.part_acc,
                                                             nullptr, nullptr);
    HCM_CHECK_LAUNCH();
    bank_logits_bwd_finish_kernel<128><<<B, kWG, 0, st>>>(nch, ws.part_acc, gx1, gx2, gx3);
  } else {
    if constexpr (!kBf16) {
      bank_pass_kernel<T, 1, kLogitsBwd><<<grid, kWG, 0, st>>>(bank1, bank2, bank3, idx, nullptr,
                                                               nullptr, nullptr, grad_logits, B, K1,
                                                               R, invT, nullptr, nullptr, ws.part_acc,
                                                               nullptr, nullptr);
      HCM_CHECK_LAUNCH();
      bank_logits_bwd_finish_kernel<64><<<B, kWG, 0, st>>>(nch, ws.part_acc, gx1, gx2, gx3);
    }
  }
  HCM_CHECK_LAUNCH();
  return 0;
}

template <class T>
int update_impl(T* bank1, T* bank2, T* bank3, const float* all_x1, const float* all_x2,
                const float* all_x3, const int64_t* all_y, int BW, int D, float momentum,
                hcm_stream_t stream, int64_t ldx = 0, int64_t n = 0, int* oob = nullptr) {
  constexpr bool kBf16 = sizeof(T) == 2;
  if (!dim_ok(D) || (kBf16 && D != 128) || BW <= 0) return (int)hipErrorInvalidValue;
  if (ldx == 0) ldx = D;
  if (ldx < D || (oob != nullptr && n <= 0)) return (int)hipErrorInvalidValue;
  const float omm = (float)(1.0 - (double)momentum);
  dim3 grid(BW, 3);
  hipStream_t st = (hipStream_t)stream;
  if (D == 128)
    bank_update_kernel<T, 128><<<grid, 64, 0, st>>>(bank1, bank2, bank3, all_x1, all_x2, all_x3,
                                                    all_y, BW, momentum, omm, ldx, n, oob);
  else
    bank_update_kernel<T, 64><<<grid, 64, 0, st>>>(bank1, bank2, bank3, all_x1, all_x2, all_x3,
                                                   all_y, BW, momentum, omm, ldx, n, oob);
  HCM_CHECK_LAUNCH();
  return 0;
}

template <class T>
int fused_timed_impl(const T* bank1, const T* bank2, const T* bank3, const int64_t* idx,
                     const float* x1, const float* x2, const float* x3, const int32_t* use_depth,
                     const int32_t* use_rgb, int B, int K1, int D, float T_, float* losses6,
                     float* accs6, float* gx1, float* gx2, float* gx3, void* workspace,
                     size_t workspace_bytes, hcm_stream_t stream, int reps, float* ms_per_pass_host) {
  if (reps <= 0 || ms_per_pass_host == nullptr) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t e0, e1;
  hipError_t e = hipEventCreate(&e0);
  if (e != hipSuccess) return (int)e;
  e = hipEventCreate(&e1);
  if (e != hipSuccess) return (int)e;
  int rc = 0;
  hipEventRecord(e0, st);
  for (int i = 0; i < reps && rc == 0; ++i)
    rc = fused_impl<T>(bank1, bank2, bank3, idx, x1, x2, x3, use_depth, use_rgb, B, K1, D, T_,
                       losses6, accs6, gx1, gx2, gx3, workspace, workspace_bytes, stream);
  hipEventRecord(e1, st);
  e = hipEventSynchronize(e1);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  if (rc != 0) return rc;
  if (e != hipSuccess) return (int)e;
  *ms_per_pass_host = ms / (float)reps;
  return 0;
}

}  // namespace

extern "C" {

int hcm_prof_enable(int enable) {
  std::lock_guard<std::mutex> lk(prof().mu);
  for (auto& v : prof().spans) {
    for (auto& sp : v) {
      hipEventDestroy(sp.first);
      hipEventDestroy(sp.second);
    }
    v.clear();
  }
  for (auto& w : prof().work) w = 0.0;
  prof().on = enable != 0;
  return 0;
}

int hcm_prof_read_tag(int tag, double* total_ms_host, int64_t* launches_host) {
  if (tag < 0 || tag >= kProfTags) return (int)hipErrorInvalidValue;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> spans;
  {
    std::lock_guard<std::mutex> lk(prof().mu);
    spans = prof().spans[tag];
  }
  double total = 0.0;
  int64_t n = 0;
  for (auto& sp : spans) {
    hipError_t e = hipEventSynchronize(sp.second);
    if (e != hipSuccess) return (int)e;
    float ms = 0.f;
    e = hipEventElapsedTime(&ms, sp.first, sp.second);
    if (e != hipSuccess) return (int)e;
    total += ms;
    ++n;
  }
  if (total_ms_host) *total_ms_host = total;
  if (launches_host) *launches_host = n;
  return 0;
}

int hcm_prof_read_work(int tag, double* work_host) {
  if (tag < 0 || tag >= kProfTags || !work_host) return (int)hipErrorInvalidValue;
  std::lock_guard<std::mutex> lk(prof().mu);
  *work_host = prof().work[tag];
  return 0;
}

int hcm_prof_read(double* total_ms_host, int64_t* launches_host) {
  return hcm_prof_read_tag(HCM_PROF_BANK_PASS, total_ms_host, launches_host);
}

int hcm_abi_version(void) { return HCM_ABI_VERSION; }
const char* hcm_error_string(int err) { return hipGetErrorString((hipError_t)err); }

size_t hcm_bank_nce_workspace_bytes(int B, int K1, int D) { return carve(nullptr, B, K1, D).bytes; }

int hcm_bank_nce_fused(const float* bank1, const float* bank2, const float* bank3, int64_t n,
                       const int64_t* idx, const float* x1, const float* x2, const float* x3,
                       const int32_t* use_depth, const int32_t* use_rgb, int B, int K1, int D,
                       float T, float* losses6, float* accs6, float* gx1, float* gx2, float* gx3,
                       void* workspace, size_t workspace_bytes, hcm_stream_t stream) {
  (void)n;
  return fused_impl<float>(bank1, bank2, bank3, idx, x1, x2, x3, use_depth, use_rgb, B, K1, D, T,
                           losses6, accs6, gx1, gx2, gx3, workspace, workspace_bytes, stream);
}
int hcm_bank_nce_fused_bf16(const uint16_t* bank1, const uint16_t* bank2, const uint16_t* bank3,
                            int64_t n, const int64_t* idx, const float* x1, const float* x2,
                            const float* x3, const int32_t* use_depth, const int32_t* use_rgb, int B,
                            int K1, int D, float T, float* losses6, float* accs6, float* gx1,
                            float* gx2, float* gx3, void* workspace, size_t workspace_bytes,
                            hcm_stream_t stream) {
  (void)n;
  return fused_impl<bf16_t>(bank1, bank2, bank3, idx, x1, x2, x3, use_depth, use_rgb, B, K1, D, T,
                            losses6, accs6, gx1, gx2, gx3, workspace, workspace_bytes, stream);
}

int hcm_bank_nce_fused_timed(const float* bank1, const float* bank2, const float* bank3, int64_t n,
                             const int64_t* idx, const float* x1, const float* x2, const float* x3,
                             const int32_t* use_depth, const int32_t* use_rgb, int B, int K1, int D,
                             float T, float* losses6, float* accs6, float* gx1, float* gx2,
                             float* gx3, void* workspace, size_t workspace_bytes,
                             hcm_stream_t stream, int reps, float* ms_per_pass_host) {
  (void)n;
  return fused_timed_impl<float>(bank1, bank2, bank3, idx, x1, x2, x3, use_depth, use_rgb, B, K1, D,
                                 T, losses6, accs6, gx1, gx2, gx3, workspace, workspace_bytes, stream,
                                 reps, ms_per_pass_host);
}
int hcm_bank_nce_fused_timed_bf16(const uint16_t* bank1, const uint16_t* bank2,
                                  const uint16_t* bank3, int64_t n, const int64_t* idx,
                                  const float* x1, const float* x2, const float* x3,
                                  const int32_t* use_depth, const int32_t* use_rgb, int B, int K1,
                                  int D, float T, float* losses6, float* accs6, float* gx1,
                                  float* gx2, float* gx3, void* workspace, size_t workspace_bytes,
                                  hcm_stream_t stream, int reps, float* ms_per_pass_host) {
  (void)n;
  return fused_timed_impl<bf16_t>(bank1, bank2, bank3, idx, x1, x2, x3, use_depth, use_rgb, B, K1, D,
                                  T, losses6, accs6, gx1, gx2, gx3, workspace, workspace_bytes,
                                  stream, reps, ms_per_pass_host);
}

int hcm_bank_logits_fwd(const float* bank1, const float* bank2, const float* bank3, int64_t n,
                        const int64_t* idx, const float* x1, const float* x2, const float* x3,
                        int B, int K1, int D, float T, float* logits, hcm_stream_t stream) {
  (void)n;
  return logits_fwd_impl<float>(bank1, bank2, bank3, idx, x1, x2, x3, B, K1, D, T, logits, stream);
}
int hcm_bank_logits_fwd_bf16(const uint16_t* bank1, const uint16_t* bank2, const uint16_t* bank3,
                             int64_t n, const int64_t* idx, const float* x1, const float* x2,
                             const float* x3, int B, int K1, int D, float T, float* logits,
                             hcm_stream_t stream) {
  (void)n;
  return logits_fwd_impl<bf16_t>(bank1, bank2, bank3, idx, x1, x2, x3, B, K1, D, T, logits, stream);
}

int hcm_bank_logits_bwd(const float* bank1, const float* bank2, const float* bank3, int64_t n,
                        const int64_t* idx, const float* grad_logits, int B, int K1, int D, float T,
                        float* gx1, float* gx2, float* gx3, void* workspace,
                        size_t workspace_bytes, hcm_stream_t stream) {
  (void)n;
  return logits_bwd_impl<float>(bank1, bank2, bank3, idx, grad_logits, B, K1, D, T, gx1, gx2, gx3,
                                workspace, workspace_bytes, stream);
}
int hcm_bank_logits_bwd_bf16(const uint16_t* bank1, const uint16_t* bank2, const uint16_t* bank3,
                             int64_t n, const int64_t* idx, const float* grad_logits, int B, int K1,
                             int D, float T, float* gx1, float* gx2, float* gx3, void* workspace,
                             size_t workspace_bytes, hcm_stream_t stream) {
  (void)n;
  return logits_bwd_impl<bf16_t>(bank1, bank2, bank3, idx, grad_logits, B, K1, D, T, gx1, gx2, gx3,
                                 workspace, workspace_bytes, stream);
}

int hcm_bank_update(float* bank1, float* bank2, float* bank3, int64_t n, const float* all_x1,
                    const float* all_x2, const float* all_x3, const int64_t* all_y, int BW, int D,
                    float momentum, hcm_stream_t stream) {
  (void)n;
  return update_impl<float>(bank1, bank2, bank3, all_x1, all_x2, all_x3, all_y, BW, D, momentum, stream);
}
int hcm_bank_update_bf16(uint16_t* bank1, uint16_t* bank2, uint16_t* bank3, int64_t n,
                         const float* all_x1, const float* all_x2, const float* all_x3,
                         const int64_t* all_y, int BW, int D, float momentum, hcm_stream_t stream) {
  (void)n;
  return update_impl<bf16_t>(bank1, bank2, bank3, all_x1, all_x2, all_x3, all_y, BW, D, momentum, stream);
}

int hcm_alias_build(const float* probs_host, int64_t n, float* prob_out_host,
                    int64_t* alias_out_host) {
  // memory/alias_multinomial.py:7-42, fp32 arithmetic, LIFO work lists.
  if (n <= 0 || probs_host == nullptr || prob_out_host == nullptr || alias_out_host == nullptr)
    return (int)hipErrorInvalidValue;
  int64_t* stack_small = new int64_t[n];
  int64_t* stack_large = new int64_t[n];
  int64_t ns = 0, nl = 0;
  for (int64_t k = 0; k < n; ++k) {
    prob_out_host[k] = (float)n * probs_host[k];
    alias_out_host[k] = 0;
    if (prob_out_host[k] < 1.0f) stack_small[ns++] = k;
    else stack_large[nl++] = k;
  }
  while (ns > 0 && nl > 0) {
    const int64_t small = stack_small[--ns];
    const int64_t large = stack_large[--nl];
    alias_out_host[small] = large;
    volatile float tmp = prob_out_host[large] - 1.0f;  // two separately rounded fp32 ops
    prob_out_host[large] = tmp + prob_out_host[small];
    if (prob_out_host[large] < 1.0f) stack_small[ns++] = large;
    else stack_large[nl++] = large;
  }
  for (int64_t i = 0; i < ns; ++i) prob_out_host[stack_small[i]] = 1.0f;
  for (int64_t i = 0; i < nl; ++i) prob_out_host[stack_large[i]] = 1.0f;
  delete[] stack_small;
  delete[] stack_large;
  return 0;
}

int hcm_alias_draw(const float* prob, const int64_t* alias, int64_t n, const int64_t* y, int B,
                   int K1, uint64_t seed, uint64_t offset, int64_t* idx, hcm_stream_t stream) {
  if (n <= 0 || B <= 0 || K1 <= 0) return (int)hipErrorInvalidValue;
  const int64_t total = (int64_t)B * K1;
  int blocks = (int)((total + kWG - 1) / kWG);
  if (blocks > 4096) blocks = 4096;
  alias_draw_kernel<<<blocks, kWG, 0, (hipStream_t)stream>>>(prob, alias, n, y, K1, total, seed,
                                                            offset, idx, nullptr);
  HCM_CHECK_LAUNCH();
  return 0;
}

int hcm_alias_draw_checked(const float* prob, const int64_t* alias, int64_t n, const int64_t* y, int B,
                           int K1, uint64_t seed, uint64_t offset, int64_t* idx, int32_t* oob_flag,
                           hcm_stream_t stream) {
  if (n <= 0 || B <= 0 || K1 <= 0 || oob_flag == nullptr) return (int)hipErrorInvalidValue;
  const int64_t total = (int64_t)B * K1;
  int blocks = (int)((total + kWG - 1) / kWG);
  if (blocks > 4096) blocks = 4096;
  alias_draw_kernel<<<blocks, kWG, 0, (hipStream_t)stream>>>(prob, alias, n, y, K1, total, seed,
                                                            offset, idx, oob_flag);
  HCM_CHECK_LAUNCH();
  return 0;
}

int hcm_bank_update_checked(float* bank1, float* bank2, float* bank3, int64_t n, const float* all_x1,
                            const float* all_x2, const float* all_x3, int64_t ldx, const int64_t* all_y,
                            int BW, int D, float momentum, int32_t* oob_flag, hcm_stream_t stream) {
  if (oob_flag == nullptr) return (int)hipErrorInvalidValue;
  return update_impl<float>(bank1, bank2, bank3, all_x1, all_x2, all_x3, all_y, BW, D, momentum, stream, ldx, n,
                            oob_flag);
}
int hcm_bank_update_checked_bf16(uint16_t* bank1, uint16_t* bank2, uint16_t* bank3, int64_t n,
                                 const float* all_x1, const float* all_x2, const float* all_x3, int64_t ldx,
                                 const int64_t* all_y, int BW, int D, float momentum, int32_t* oob_flag,
                                 hcm_stream_t stream) {
  if (oob_flag == nullptr) return (int)hipErrorInvalidValue;
  return update_impl<bf16_t>(bank1, bank2, bank3, all_x1, all_x2, all_x3, all_y, BW, D, momentum, stream, ldx, n,
                             oob_flag);
}

int hcm_moco_logits(const float* q, const float* k, const float* queue, int B, int K, int D, float T,
                    float* logits, hcm_stream_t stream) {
  if (!dim_ok(D) || B <= 0 || K <= 0 || !(T > 0.f)) return (int)hipErrorInvalidValue;
  const size_t lds = (size_t)B * D * sizeof(float);
  if (lds > 128 * 1024) return (int)hipErrorInvalidValue;
  const float invT = (float)(1.0 / (double)T);
  int blocks = (K + B + 15) / 16;
  if (blocks > 2048) blocks = 2048;
  hipStream_t st = (hipStream_t)stream;
  if (D == 128) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(moco_logits_kernel<2>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    moco_logits_kernel<2><<<blocks, kWG, lds, st>>>(q, k, queue, B, K, invT, logits);
  } else {
    hipFuncSetAttribute(reinterpret_cast<const void*>(moco_logits_kernel<1>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    moco_logits_kernel<1><<<blocks, kWG, lds, st>>>(q, k, queue, B, K, invT, logits);
  }
  HCM_CHECK_LAUNCH();
  return 0;
}

int hcm_moco_enqueue(float* queue, const float* all_k, int n_new, int K, int D, int64_t index,
                     hcm_stream_t stream) {
  if (n_new <= 0 || K <= 0 || D <= 0) return (int)hipErrorInvalidValue;
  moco_enqueue_kernel<<<n_new, 128, 0, (hipStream_t)stream>>>(queue, all_k, n_new, K, D, index);
  HCM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
