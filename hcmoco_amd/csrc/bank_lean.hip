// Bank pass (memory/mem_bank.py:172-193: gather K+1 rows of three banks per sample, six dot products per row triple,
// softmax statistics and the probability-weighted row sums of the backward), written for the NUMBER OF INSTRUCTIONS per
// gathered row.  csrc/bank.hip holds the general kernel (every mode, D = 64 / 128, both storage types) and everything
// around the pass; this file is the fused mode at D = 128 only.
//
// Why it exists (r04, profiles/r04_bf16_variants.txt): the bf16 pass is bound by instruction issue, not by
// bytes in flight -- a 256-byte row costs the same bookkeeping as a 512-byte one.  One ring slot (four rows x three
// banks per wave) of the general kernel executes ~270 instructions; the same work here takes ~150:
//   * every product / sum is written on v2f (v_pk_mul_f32 / v_pk_fma_f32) with ONE pairing throughout -- a 32-bit word of
//     a bf16 row unpacks to the pair (w << 16, w & 0xffff0000) = two adjacent columns, the queries and accumulators are
//     paired the same way -- so no register moves re-pair operands (the vectoriser's pairing differed between the dot
//     products and the accumulation: ~45 v_mov per slot).  The file is compiled with -fno-slp-vectorize: the 24 DPP row
//     sums stay v_add_f32_dpp (fused) instead of two v_mov_dpp + one v_pk_add per pair;
//   * the queries are pre-multiplied by log2(e) / T (six v_mul per slot gone);
//   * the softmax reference point of a stream does not chase the running maximum (with 32 rows per stream and chunk a
//     new maximum arrives in one round out of two: compare, branch, v_exp, four packed rescales, per pair).  It starts
//     at the Cauchy-Schwarz bound 1.01 |x| log2(e) / T -- the banks hold unit rows -- which no logit exceeds, so the
//     rescale path is a single, almost never taken branch per slot (taken, it is the exact online-softmax update, so
//     rows of any norm stay correct; a query whose bound is so large that exp2 could underflow every term starts at
//     "minus infinity" and the branch does the chasing).  The TRUE maximum, which the accuracy needs, costs one v_max
//     per pair and round; the partial is re-referenced to it once, before the merge, so the workspace carries the same
//     (m, s, acc) triple as the general kernel's.
// Numerics: fp32 sums in a different order than the general kernel (pairwise halves, pre-scaled queries): agreement to
// rounding, same tolerances against the oracle (tests/test_bank_gpu.py).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>

#include "hcm_common.h"
#include "bank_lean.h"

namespace {

using namespace hcm;

typedef float v2f __attribute__((ext_vector_type(2)));
typedef uint16_t bf16_t;

constexpr int kWG = 256;
constexpr int kStreams = 16;
constexpr float kNegBig = -1.0e30f;
constexpr float kInvalid = -3.0e30f;
constexpr int kD = 128;

__device__ __forceinline__ v2f pkfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat(float a) { return v2f{a, a}; }

// bank of pair p (0:M1 1:M2 2:M3) and query of pair p (0:x1 1:x2 2:x3); order 12,21,23,32,13,31 (mem_bank.py:186-191)
__device__ __forceinline__ constexpr int pair_bank(int p) { return (p == 1 || p == 5) ? 0 : ((p == 0 || p == 3) ? 1 : 2); }
__device__ __forceinline__ constexpr int pair_query(int p) { return (p == 0 || p == 4) ? 0 : ((p == 1 || p == 2) ? 1 : 2); }

// ---- a gathered row triple: as it sits in the ring, and as four column pairs per bank ------------------------------
template <class T> struct Packed;
template <> struct Packed<bf16_t> { uint4 q[3]; };
template <> struct Packed<float> { float4 q[3][2]; };

// first column of pair w (0..3) of lane t
template <class T> __device__ __forceinline__ int pair_col(int t, int w);
template <> __device__ __forceinline__ int pair_col<bf16_t>(int t, int w) { return 8 * t + 2 * w; }
template <> __device__ __forceinline__ int pair_col<float>(int t, int w) { return (w < 2 ? 0 : 64) + 4 * t + 2 * (w & 1); }

__device__ __forceinline__ void load_packed(Packed<bf16_t>& p, const bf16_t* b1, const bf16_t* b2, const bf16_t* b3,
                                            int64_t row, int t) {
  const int64_t off = row * kD + 8 * t;
  p.q[0] = *reinterpret_cast<const uint4*>(b1 + off);
  p.q[1] = *reinterpret_cast<const uint4*>(b2 + off);
  p.q[2] = *reinterpret_cast<const uint4*>(b3 + off);
}
__device__ __forceinline__ void load_packed(Packed<float>& p, const float* b1, const float* b2, const float* b3,
                                            int64_t row, int t) {
  const int64_t off = row * kD + 4 * t;
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    p.q[0][v] = *reinterpret_cast<const float4*>(b1 + off + 64 * v);
    p.q[1][v] = *reinterpret_cast<const float4*>(b2 + off + 64 * v);
    p.q[2][v] = *reinterpret_cast<const float4*>(b3 + off + 64 * v);
  }
}
__device__ __forceinline__ v2f bf_pair(uint32_t w) { return v2f{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)}; }
__device__ __forceinline__ void unpack(const Packed<bf16_t>& p, v2f (&r)[3][4]) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    r[c][0] = bf_pair(p.q[c].x); r[c][1] = bf_pair(p.q[c].y); r[c][2] = bf_pair(p.q[c].z); r[c][3] = bf_pair(p.q[c].w);
  }
}
__device__ __forceinline__ void unpack(const Packed<float>& p, v2f (&r)[3][4]) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    r[c][0] = v2f{p.q[c][0].x, p.q[c][0].y}; r[c][1] = v2f{p.q[c][0].z, p.q[c][0].w};
    r[c][2] = v2f{p.q[c][1].x, p.q[c][1].y}; r[c][3] = v2f{p.q[c][1].z, p.q[c][1].w};
  }
}

template <class T, int NPF>
__global__ __launch_bounds__(kWG, 2) void bank_pass_lean_kernel(
    const T* __restrict__ b1, const T* __restrict__ b2, const T* __restrict__ b3, const int64_t* __restrict__ idx,
    const float* __restrict__ x1, const float* __restrict__ x2, const float* __restrict__ x3, int B, int K1, int R,
    float scale, float* __restrict__ part_m, float* __restrict__ part_s, float* __restrict__ part_acc,
    float* __restrict__ l0_out) {
  const int b = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = lane & 15, g = lane >> 4;
  const int s = wave * 4 + g;
  const int kbeg = chunk * R;
  const int kend = min(K1, kbeg + R);
  const int niter = (kend - kbeg + kStreams - 1) / kStreams;
  const int64_t* __restrict__ idxb = idx + (int64_t)b * K1;

  // queries, pre-scaled by log2(e) / T, paired like the rows
  v2f xq[3][4];
  float mstart[3];
  {
    const float* xs[3] = {x1, x2, x3};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      float n2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float2 v = *reinterpret_cast<const float2*>(xs[q] + (int64_t)b * kD + pair_col<T>(t, w));
        xq[q][w] = v2f{v.x * scale, v.y * scale};
        n2 = fmaf(xq[q][w].x, xq[q][w].x, fmaf(xq[q][w].y, xq[q][w].y, n2));
      }
      const float bound = 1.01f * __builtin_sqrtf(row16_sum(n2));      // >= every logit of a unit row (log2 units)
      mstart[q] = bound < 60.f ? bound : kNegBig;                      // NaN / huge: chase the maximum instead
    }
  }

  v2f acc[6][4], mref[3], ssum[3];      // pairs (0,1) (2,3) (4,5) share a v2f where they are combined packed
  float mx[6];
#pragma unroll
  for (int p = 0; p < 6; ++p) {
    mx[p] = kNegBig;
#pragma unroll
    for (int w = 0; w < 4; ++w) acc[p][w] = splat(0.f);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    mref[i] = v2f{mstart[pair_query(2 * i)], mstart[pair_query(2 * i + 1)]};
    ssum[i] = splat(0.f);
  }

  // the row index of round `it` of this lane's stream; past the end it repeats the chunk's last row (never gathered: the
  // loads are guarded by the round count)
  auto ld_idx = [&](int it) -> int64_t { return idxb[min(kbeg + it * kStreams + s, kend - 1)]; };
  Packed<T> ring[NPF];
  int64_t ridx[NPF];
#pragma unroll
  for (int j = 0; j < NPF; ++j) load_packed(ring[j], b1, b2, b3, j < niter ? ld_idx(j) : (int64_t)0, t);
#pragma unroll
  for (int j = 0; j < NPF; ++j) ridx[j] = ld_idx(NPF + j);

  // one round: consume ring[j] (round `it`), refill it with round it + NPF.  DRAIN: the refill may be past the end (guarded)
  // and the round may be the last, partial one (validity mask).
  auto round = [&](auto drain, int j, int it) {
    constexpr bool DRAIN = decltype(drain)::value;
    v2f r[3][4];
    unpack(ring[j], r);
    if (!DRAIN || it + NPF < niter) load_packed(ring[j], b1, b2, b3, ridx[j], t);
    ridx[j] = ld_idx(it + 2 * NPF);
    // six dot products, stage by stage across the pairs: consecutive packed instructions are independent
    v2f sacc[6];
#pragma unroll
    for (int p = 0; p < 6; ++p) sacc[p] = xq[pair_query(p)][0] * r[pair_bank(p)][0];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
#pragma unroll
      for (int p = 0; p < 6; ++p) sacc[p] = pkfma(xq[pair_query(p)][w], r[pair_bank(p)][w], sacc[p]);
    }
    float d[6];
#pragma unroll
    for (int p = 0; p < 6; ++p) d[p] = sacc[p].x + sacc[p].y;
    // 16-lane sums, a DPP step of all six at a time (a DPP read of a register written by the previous instruction costs
    // wait states)
#pragma unroll
    for (int p = 0; p < 6; ++p) d[p] += dpp_mov<0xB1>(d[p]);
#pragma unroll
    for (int p = 0; p < 6; ++p) d[p] += dpp_mov<0x4E>(d[p]);
#pragma unroll
    for (int p = 0; p < 6; ++p) d[p] += dpp_mov<0x141>(d[p]);
#pragma unroll
    for (int p = 0; p < 6; ++p) d[p] += dpp_mov<0x140>(d[p]);
    if (chunk == 0 && it == 0) {       // k = 0 is the positive: its logits go out as they are (mem_bank.py:176-180)
      if (s == 0 && t == 0) {
#pragma unroll
        for (int p = 0; p < 6; ++p) l0_out[b * 6 + p] = d[p];
      }
    }
    if (DRAIN) {
      const bool valid = kbeg + it * kStreams + s < kend;
#pragma unroll
      for (int p = 0; p < 6; ++p) d[p] = valid ? d[p] : kInvalid;
    }
#pragma unroll
    for (int p = 0; p < 6; ++p) mx[p] = fmaxf(mx[p], d[p]);
    v2f e[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) e[i] = v2f{d[2 * i], d[2 * i + 1]} - mref[i];
    const float top = fmaxf(fmaxf(fmaxf(e[0].x, e[0].y), e[1].x), fmaxf(fmaxf(e[1].y, e[2].x), e[2].y));
    if (__any(top > 0.f)) {      // a logit above the reference point: the online-softmax update, all six pairs
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const v2f l = v2f{d[2 * i], d[2 * i + 1]};
        const v2f mn = v2f{fmaxf(mref[i].x, l.x), fmaxf(mref[i].y, l.y)};
        const v2f a = v2f{fast_exp2(mref[i].x - mn.x), fast_exp2(mref[i].y - mn.y)};
        ssum[i] *= a;
#pragma unroll
        for (int w = 0; w < 4; ++w) { acc[2 * i][w] *= splat(a.x); acc[2 * i + 1][w] *= splat(a.y); }
        mref[i] = mn;
        e[i] = l - mn;
      }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const v2f pr = v2f{fast_exp2(e[i].x), fast_exp2(e[i].y)};
      ssum[i] += pr;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        acc[2 * i][w] = pkfma(splat(pr.x), r[pair_bank(2 * i)][w], acc[2 * i][w]);
        acc[2 * i + 1][w] = pkfma(splat(pr.y), r[pair_bank(2 * i + 1)][w], acc[2 * i + 1][w]);
      }
    }
  };

  // steady state: NPF rounds at a time while every refill is in range (and therefore every row valid); then the drain
  int it0 = 0;
  for (; it0 + 2 * NPF <= niter; it0 += NPF) {
#pragma unroll
    for (int j = 0; j < NPF; ++j) round(std::false_type{}, j, it0 + j);
  }
  for (; it0 < niter; it0 += NPF) {
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
      if (it0 + j < niter) round(std::true_type{}, j, it0 + j);      // workgroup-uniform
    }
  }

  // re-reference the partial to the true maximum (the reference point was a bound, or a chased maximum = mx)
  float m[6], sm[6];
#pragma unroll
  for (int p = 0; p < 6; ++p) {
    const float mr = (p & 1) ? mref[p >> 1].y : mref[p >> 1].x;
    const float a = mx[p] > 0.5f * kNegBig ? fast_exp2(mr - mx[p]) : 0.f;
    m[p] = mx[p];
    sm[p] = ((p & 1) ? ssum[p >> 1].y : ssum[p >> 1].x) * a;
#pragma unroll
    for (int w = 0; w < 4; ++w) acc[p][w] *= splat(a);
  }

  // ---- merge the 4 lane-rows of the wave (lanes ^16, ^32) ----
#pragma unroll
  for (int p = 0; p < 6; ++p) {
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
      const float mo = __shfl_xor(m[p], off, 64);
      const float so = __shfl_xor(sm[p], off, 64);
      const float mn = fmaxf(m[p], mo);
      const float a = fast_exp2(m[p] - mn), bs = fast_exp2(mo - mn);
      sm[p] = sm[p] * a + so * bs;
      m[p] = mn;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const v2f o = v2f{__shfl_xor(acc[p][w].x, off, 64), __shfl_xor(acc[p][w].y, off, 64)};
        acc[p][w] = acc[p][w] * splat(a) + o * splat(bs);
      }
    }
  }

  // ---- merge the 4 waves through LDS, write one partial per workgroup ----
  __shared__ float lds_acc[4][6][kD];
  __shared__ float lds_m[4][6];
  __shared__ float lds_s[4][6];
  if (g == 0) {
#pragma unroll
    for (int p = 0; p < 6; ++p) {
#pragma unroll
      for (int w = 0; w < 4; ++w)
        *reinterpret_cast<float2*>(&lds_acc[wave][p][pair_col<T>(t, w)]) = make_float2(acc[p][w].x, acc[p][w].y);
      if (t == 0) {
        lds_m[wave][p] = m[p];
        lds_s[wave][p] = sm[p];
      }
    }
  }
  __syncthreads();
  const int64_t pbase = ((int64_t)b * nchunks + chunk) * 6;
  for (int e = threadIdx.x; e < 6 * kD; e += kWG) {
    const int p = e / kD, col = e - p * kD;
    const float M = fmaxf(fmaxf(lds_m[0][p], lds_m[1][p]), fmaxf(lds_m[2][p], lds_m[3][p]));
    float sc[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) sc[w] = fast_exp2(lds_m[w][p] - M);
    part_acc[(pbase + p) * kD + col] = lds_acc[0][p][col] * sc[0] + lds_acc[1][p][col] * sc[1] +
                                       lds_acc[2][p][col] * sc[2] + lds_acc[3][p][col] * sc[3];
    if (col == 0) {
      part_m[pbase + p] = M;
      part_s[pbase + p] = lds_s[0][p] * sc[0] + lds_s[1][p] * sc[1] + lds_s[2][p] * sc[2] + lds_s[3][p] * sc[3];
    }
  }
}

}  // namespace

int hcm::bank_pass_lean_launch(int is_bf16, int ring, const void* b1, const void* b2, const void* b3, const int64_t* idx,
                                const float* x1, const float* x2, const float* x3, int B, int K1, int R, float scale,
                                float* part_m, float* part_s, float* part_acc, float* l0, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((K1 + R - 1) / R, B);
#define HCM_LEAN(T, NPF)                                                                                          \
  bank_pass_lean_kernel<T, NPF><<<grid, kWG, 0, st>>>((const T*)b1, (const T*)b2, (const T*)b3, idx, x1, x2, x3, B, \
                                                      K1, R, scale, part_m, part_s, part_acc, l0)
  if (is_bf16) {
    switch (ring) {
      case 4: HCM_LEAN(bf16_t, 4); break;
      case 5: HCM_LEAN(bf16_t, 5); break;
      case 8: HCM_LEAN(bf16_t, 8); break;
      default: HCM_LEAN(bf16_t, 6); break;
    }
  } else {
    switch (ring) {
      case 2: HCM_LEAN(float, 2); break;
      case 4: HCM_LEAN(float, 4); break;
      default: HCM_LEAN(float, 3); break;
    }
  }
#undef HCM_LEAN
  return (int)hipGetLastError();
}
