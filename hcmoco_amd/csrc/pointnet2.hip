// PointNet++ point ops for gfx950.
//
// Same contract as the reference's launcher layer (pycontrast/networks/pointnet2/src/*_gpu.h):
// fp32 data, int32 indices, caller-allocated and caller-initialised outputs, asynchronous on the
// given stream.  Launch failures are returned instead of exit(-1).
//
// Arithmetic contract of the four kernels that do fp32 arithmetic (FPS, ball query, three-NN,
// three_interpolate).  The reference writes  a*a + b*b + c*c  (sampling_gpu.cu:131, ball_query_gpu.cu:33,
// interpolate_gpu.cu:39, :96) and is built with `nvcc -O2` (networks/pointnet2/setup.py:20), whose
// default is --fmad=true: the compiler contracts the expression.  Every LLVM-family and GNU compiler
// available here contracts that source form the same way (checked on the ISA: amdgcn, x86 clang, x86
// gcc): the MIDDLE product is rounded on its own, then two fused multiply-adds,
//     fma(c, c, fma(a, a, b*b)).
// NVVM is the same LLVM DAG combiner, so that is taken as the reference's real arithmetic
// (kContractFma, the default).  kContractIeee is the un-fused evaluation ((a*a + b*b) + c*c) -- what the
// same source gives with --fmad=false, and what a CPU port of the reference would compute.
// Near-ties are routine (clouds are sampled WITH replacement, build_backbone.py:427), so the two
// contracts can pick different FPS points / ball members / third neighbours; both are bit-exact against
// oracle/pointnet2_oracle.c in the same mode.  This translation unit is compiled with
// -ffp-contract=off so that only the fmaf() calls written below fuse.
//
// MI355X notes: the three search kernels stage the scanned cloud through LDS in coalesced tiles
// and read it back as wave-wide broadcasts; gather/scatter kernels load each index once and walk
// a block of channels with it (the reference re-reads the index for every channel); FPS keeps
// the whole cloud and its running min-distances in registers/LDS for all m serial rounds.
#include <cstdlib>

#include "hcm_common.h"
#include "../../include/hcmoco_hip.h"

namespace {

constexpr int kT = 256;

template <bool FMA>
__device__ __forceinline__ float sqdist(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  if (FMA) return fmaf(dz, dz, fmaf(dx, dx, dy * dy));
  return (dx * dx + dy * dy) + dz * dz;
}
// Two squared distances at once (two unknown points against one known point) on the packed fp32 lanes
// (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32): element-wise IEEE operations, so each half is bit-identical to sqdist.
typedef float v2f __attribute__((ext_vector_type(2)));
template <bool FMA>
__device__ __forceinline__ v2f sqdist2(v2f ax, v2f ay, v2f az, float bx, float by, float bz) {
  const v2f dx = ax - bx, dy = ay - by, dz = az - bz;
  if (FMA) return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));
  return (dx * dx + dy * dy) + dz * dz;
}
template <bool FMA>
__device__ __forceinline__ float dot3(float a0, float b0, float a1, float b1, float a2, float b2) {
  if (FMA) return fmaf(a2, b2, fmaf(a0, b0, a1 * b1));
  return (a0 * b0 + a1 * b1) + a2 * b2;
}

// ------------------------------------------------------------------------------------------
// gather / group : out[b,c,q] = points[b,c,idx[b,q]]        (q = m  or  q = npoints*nsample)
// sampling_gpu.cu:8-24, group_points_gpu.cu:47-66
// ------------------------------------------------------------------------------------------
constexpr int kCB = 16;  // channels walked per loaded index

__global__ __launch_bounds__(kT) void gather_rows_kernel(int c, int n, int q,
                                                         const float* __restrict__ points,
                                                         const int* __restrict__ idx,
                                                         float* __restrict__ out) {
  const int b = blockIdx.z, pos = blockIdx.x * kT + threadIdx.x;
  if (pos >= q) return;
  const int src = idx[(int64_t)b * q + pos];
  const int c0 = blockIdx.y * kCB, c1 = min(c, c0 + kCB);
  const float* p = points + ((int64_t)b * c + c0) * n + src;
  float* o = out + ((int64_t)b * c + c0) * q + pos;
  for (int ch = c0; ch < c1; ++ch, p += n, o += q) *o = *p;
}

// grad: grad_points[b,c,idx[b,q]] += grad_out[b,c,q]   (atomic; order-dependent fp32 sums as in
// the reference, group_points_gpu.cu:8-25 / sampling_gpu.cu:46-63)
__global__ __launch_bounds__(kT) void scatter_rows_kernel(int c, int n, int q,
                                                          const float* __restrict__ grad_out,
                                                          const int* __restrict__ idx,
                                                          float* __restrict__ grad_points) {
  const int b = blockIdx.z, pos = blockIdx.x * kT + threadIdx.x;
  if (pos >= q) return;
  const int dst = idx[(int64_t)b * q + pos];
  const int c0 = blockIdx.y * kCB, c1 = min(c, c0 + kCB);
  const float* g = grad_out + ((int64_t)b * c + c0) * q + pos;
  float* o = grad_points + ((int64_t)b * c + c0) * n + dst;
  for (int ch = c0; ch < c1; ++ch, g += q, o += n) atomicAdd(o, *g);
}

// ------------------------------------------------------------------------------------------
// three_interpolate: out[b,c,n] = sum_j w[b,n,j] * points[b,c,idx[b,n,j]]
// interpolate_gpu.cu:77-97 / :120-142
// ------------------------------------------------------------------------------------------
template <bool FMA>
__global__ __launch_bounds__(kT) void three_interp_kernel(int c, int m, int n,
                                                          const float* __restrict__ points,
                                                          const int* __restrict__ idx,
                                                          const float* __restrict__ weight,
                                                          float* __restrict__ out) {
  const int b = blockIdx.z, pos = blockIdx.x * kT + threadIdx.x;
  if (pos >= n) return;
  const int64_t t3 = ((int64_t)b * n + pos) * 3;
  const int i0 = idx[t3], i1 = idx[t3 + 1], i2 = idx[t3 + 2];
  const float w0 = weight[t3], w1 = weight[t3 + 1], w2 = weight[t3 + 2];
  const int c0 = blockIdx.y * kCB, c1 = min(c, c0 + kCB);
  const float* p = points + ((int64_t)b * c + c0) * m;
  float* o = out + ((int64_t)b * c + c0) * n + pos;
  for (int ch = c0; ch < c1; ++ch, p += m, o += n) *o = dot3<FMA>(w0, p[i0], w1, p[i1], w2, p[i2]);
}

// three_interpolate with the source rows in LDS (r04).  The kernel above gathers three 4-byte words per output element
// from rows that live in L2: at c = 128, n = 65 536, m = 4096 (pts2depth, build_backbone.py:448-455) and c = 256,
// n = m = 4096 it runs at 2.4-2.7 TB/s of useful bytes.  Here a workgroup keeps a block of CBL channels of ALL m source
// points in LDS, point-major ([m][CBL]: one ds_read_b128 fetches four channels of a point), and streams its share of the
// n positions: idx / weight are read once per position and channel block, every output row is written coalesced.
// Same expression per element (dot3<FMA>), so the results are bit-identical to the gather kernel's.
constexpr int kTIThreads = 1024;
template <bool FMA>
__global__ __launch_bounds__(kTIThreads) void three_interp_lds_kernel(int c, int m, int n, int cbl, int npos,
                                                                      const float* __restrict__ points,
                                                                      const int* __restrict__ idx,
                                                                      const float* __restrict__ weight,
                                                                      float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float src[];     // [m][cbl], cbl % 4 == 0
  const int b = blockIdx.z, c0 = blockIdx.y * cbl, nc = min(cbl, c - c0);
  const int p0 = blockIdx.x * npos, p1 = min(n, p0 + npos);
  const float* pb = points + ((int64_t)b * c + c0) * m;
  // fill: consecutive threads read consecutive points of one channel (coalesced); channels past the end stay zero
  for (int e = threadIdx.x; e < cbl * m; e += kTIThreads) {
    const int k = e / m, j = e - k * m;
    src[j * cbl + k] = k < nc ? pb[(int64_t)k * m + j] : 0.f;
  }
  __syncthreads();
  float* ob = out + ((int64_t)b * c + c0) * n;
  for (int pos = p0 + threadIdx.x; pos < p1; pos += kTIThreads) {
    const int64_t t3 = ((int64_t)b * n + pos) * 3;
    const int i0 = idx[t3], i1 = idx[t3 + 1], i2 = idx[t3 + 2];
    const float w0 = weight[t3], w1 = weight[t3 + 1], w2 = weight[t3 + 2];
    const float4* r0 = reinterpret_cast<const float4*>(src + i0 * cbl);
    const float4* r1 = reinterpret_cast<const float4*>(src + i1 * cbl);
    const float4* r2 = reinterpret_cast<const float4*>(src + i2 * cbl);
    for (int k4 = 0; 4 * k4 < nc; ++k4) {
      const float4 a = r0[k4], bq = r1[k4], d = r2[k4];
      float* o = ob + (int64_t)(4 * k4) * n + pos;
      o[0] = dot3<FMA>(w0, a.x, w1, bq.x, w2, d.x);
      if (4 * k4 + 1 < nc) o[n] = dot3<FMA>(w0, a.y, w1, bq.y, w2, d.y);
      if (4 * k4 + 2 < nc) o[2 * (int64_t)n] = dot3<FMA>(w0, a.z, w1, bq.z, w2, d.z);
      if (4 * k4 + 3 < nc) o[3 * (int64_t)n] = dot3<FMA>(w0, a.w, w1, bq.w, w2, d.w);
    }
  }
}

__global__ __launch_bounds__(kT) void three_interp_grad_kernel(int c, int n, int m,
                                                               const float* __restrict__ grad_out,
                                                               const int* __restrict__ idx,
                                                               const float* __restrict__ weight,
                                                               float* __restrict__ grad_points) {
  const int b = blockIdx.z, pos = blockIdx.x * kT + threadIdx.x;
  if (pos >= n) return;
  const int64_t t3 = ((int64_t)b * n + pos) * 3;
  const int i0 = idx[t3], i1 = idx[t3 + 1], i2 = idx[t3 + 2];
  const float w0 = weight[t3], w1 = weight[t3 + 1], w2 = weight[t3 + 2];
  const int c0 = blockIdx.y * kCB, c1 = min(c, c0 + kCB);
  const float* g = grad_out + ((int64_t)b * c + c0) * n + pos;
  float* o = grad_points + ((int64_t)b * c + c0) * m;
  for (int ch = c0; ch < c1; ++ch, g += n, o += m) {
    const float gv = *g;
    atomicAdd(o + i0, gv * w0);
    atomicAdd(o + i1, gv * w1);
    atomicAdd(o + i2, gv * w2);
  }
}

// ------------------------------------------------------------------------------------------
// LDS-resident scatter-add (preferred backward of group_points / gather_points / three_interpolate):
//     grad_points[b, c, j] = sum_{q : idx[b, q] == j} coef[b, q] * grad_out[b, c, q / div]
// The target axis is short in PointNet++ (m <= 4096 points), so a workgroup keeps the WHOLE target
// row of a block of channels in LDS (m x CBL floats <= 140 KB of the CU's 160 KB), streams
// grad_out / idx / coef once, fully coalesced, accumulates with ds_add_f32 and finally stores the
// rows.  No global atomics, no index inversion, no zero-fill; the reference's float atomicAdd to
// HBM-resident rows runs at ~120 GB/s on MI355X, this runs at the streaming rate of its inputs.
// ------------------------------------------------------------------------------------------
constexpr int kLdsScatterThreads = 1024;
constexpr int kLdsScatterBytes = 140 * 1024;

__global__ __launch_bounds__(kLdsScatterThreads) void lds_scatter_kernel(
    const float* __restrict__ grad_out, const float* __restrict__ coef, const int* __restrict__ idx,
    int C, int Qsrc, int Q, int m, int div, int cbl, float* __restrict__ grad_points) {
  extern __shared__ __attribute__((aligned(16))) float acc[];  // [cbl][m]
  const int b = blockIdx.y, c0 = blockIdx.x * cbl;
  const int nc = min(cbl, C - c0);
  for (int e = threadIdx.x; e < nc * m; e += kLdsScatterThreads) acc[e] = 0.f;
  __syncthreads();
  const int* ix = idx + (int64_t)b * Q;
  const float* cf = coef ? coef + (int64_t)b * Q : nullptr;
  const float* g = grad_out + ((int64_t)b * C + c0) * Qsrc;
  for (int q = threadIdx.x; q < Q; q += kLdsScatterThreads) {
    const int j = ix[q];
    const float w = cf ? cf[q] : 1.f;
    const int src = q / div;
    for (int k = 0; k < nc; ++k) atomicAdd(&acc[k * m + j], w * g[(int64_t)k * Qsrc + src]);
  }
  __syncthreads();
  float* out = grad_points + ((int64_t)b * C + c0) * m;
  for (int e = threadIdx.x; e < nc * m; e += kLdsScatterThreads) out[e] = acc[e];
}

// ------------------------------------------------------------------------------------------
// ball query (ball_query_gpu.cu:9-45): first `nsample` points, in index order, with d2 < r^2;
// the first hit pre-fills all nsample slots; centres with no hit keep the caller's zeros.
// One thread per centre; the scanned cloud goes through LDS in tiles of kTile points.
// ------------------------------------------------------------------------------------------
constexpr int kTile = 1024;

// Hits are collected in LDS ([slot][thread], conflict-free) and written out once at the end: a global
// store inside the scan loop is issued by ~half of all iterations (any of the 64 lanes hitting) and
// paces the kernel otherwise.
template <bool FMA>
__global__ __launch_bounds__(kT) void ball_query_kernel(int n, int m, float radius2, int nsample,
                                                        const float* __restrict__ new_xyz,
                                                        const float* __restrict__ xyz,
                                                        int* __restrict__ idx) {
  __shared__ float4 tile[kTile];
  extern __shared__ __attribute__((aligned(16))) int hits[];  // [nsample][blockDim.x]
  const int b = blockIdx.y, pt = blockIdx.x * blockDim.x + threadIdx.x;
  const int nt = blockDim.x;
  const bool live = pt < m;
  float cx = 0.f, cy = 0.f, cz = 0.f;
  if (live) {
    const float* c = new_xyz + ((int64_t)b * m + pt) * 3;
    cx = c[0]; cy = c[1]; cz = c[2];
  }
  const float* cloud = xyz + (int64_t)b * n * 3;
  int cnt = live ? 0 : nsample;
  for (int base = 0; base < n; base += kTile) {
    const int len = min(kTile, n - base);
    __syncthreads();
    for (int e = threadIdx.x; e < len; e += nt) {
      const float* c = cloud + (int64_t)(base + e) * 3;
      tile[e] = make_float4(c[0], c[1], c[2], 0.f);
    }
    __syncthreads();
    if (cnt < nsample) {
      // chunks of 8: the LDS reads and distance evaluations of a chunk are independent (issued
      // back to back); hits are then recorded strictly in index order, as the reference scans
      for (int k0 = 0; k0 < len && cnt < nsample; k0 += 8) {
        float d2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float4 p = tile[min(k0 + j, len - 1)];
          d2[j] = (k0 + j < len) ? sqdist<FMA>(cx, cy, cz, p.x, p.y, p.z) : __builtin_inff();
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (d2[j] < radius2 && cnt < nsample) {
            hits[cnt * nt + threadIdx.x] = base + k0 + j;
            ++cnt;
          }
        }
      }
    }
    if (__syncthreads_count(cnt < nsample) == 0) break;
  }
  if (live && cnt > 0) {   // no hit: the caller's zeros survive (ball_query_gpu.cu: idx untouched)
    int* out = idx + ((int64_t)b * m + pt) * nsample;
    const int first = hits[threadIdx.x];
    for (int l = 0; l < nsample; ++l) out[l] = l < cnt ? hits[l * nt + threadIdx.x] : first;
  }
}

// ------------------------------------------------------------------------------------------
// ball query, the scan split across the lanes of a wave (r04; the open item of DESIGN 4.3c).  The kernel above gives every
// centre one thread that walks the cloud serially: with m = 1024 centres per cloud there is half a wave per SIMD and
// each lane pays ~100 ns per scanned point (0.385 ms for 32 x 1024 centres -- as long as 32 x 4096).  Here a WAVE owns
// a centre at a time: lane l evaluates point base + l, the ballot of the hits is the hit set in index order, and lane l
// knows its rank among them from a prefix popcount -- "the first nsample points in index order" (ball_query_gpu.cu:31-43)
// without any serial dependence, bit-exact.  The cloud sits in LDS once per workgroup (n x 16 bytes), a wave walks kCPW
// centres, two at a time against the same ds_read_b128 of the points.
// ------------------------------------------------------------------------------------------
constexpr int kBQWaves = 16;       // waves per workgroup: they share one LDS copy of the cloud, and 2 x 16 waves per CU are
                                   // what hides the ds_read -> compare -> ballot -> scalar bookkeeping chain of a step
constexpr int kBQCentres = 8;      // centres per wave

// one 64-point step of one centre: the hits of this step, in index order, go to slots cnt, cnt + 1, ...
__device__ __forceinline__ void bq_record(unsigned long long mask, int base, int lane, unsigned long long below,
                                          int nsample, int& cnt, int& first, int* __restrict__ out) {
  if (mask != 0ull && cnt < nsample) {
    if (cnt == 0) first = base + __builtin_ctzll(mask);
    const int slot = cnt + __popcll(mask & below);
    if (((mask >> lane) & 1ull) && slot < nsample) out[slot] = base + lane;
    cnt += __popcll(mask);
  }
}

template <bool FMA>
__global__ __launch_bounds__(kBQWaves * 64) void ball_query_wave_kernel(int n, int m, float radius2, int nsample,
                                                                       const float* __restrict__ new_xyz,
                                                                       const float* __restrict__ xyz,
                                                                       int* __restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) float4 cloud4[];   // [n rounded up to 128]
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* cloud = xyz + (int64_t)b * n * 3;
  const int npad = (n + 127) & ~127;
  for (int e = tid; e < npad; e += kBQWaves * 64) {
    float4 p = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), 0.f);    // padding: never inside a ball
    if (e < n) p = make_float4(cloud[3 * e], cloud[3 * e + 1], cloud[3 * e + 2], 0.f);
    cloud4[e] = p;
  }
  __syncthreads();
  const int c0 = (blockIdx.x * kBQWaves + wave) * kBQCentres;
  const unsigned long long below = (1ull << lane) - 1ull;
  for (int cc = 0; cc < kBQCentres; cc += 2) {
    const int pa = c0 + cc, pb = pa + 1;
    if (pa >= m) break;
    const bool hasb = pb < m;
    const float* ca = new_xyz + ((int64_t)b * m + pa) * 3;
    const float* cb = new_xyz + ((int64_t)b * m + (hasb ? pb : pa)) * 3;
    const float ax = ca[0], ay = ca[1], az = ca[2], bx = cb[0], by = cb[1], bz = cb[2];
    int* outa = idx + ((int64_t)b * m + pa) * nsample;
    int* outb = idx + ((int64_t)b * m + pb) * nsample;
    int cnta = 0, cntb = hasb ? 0 : nsample, firsta = 0, firstb = 0;
    // two steps (128 points) x two centres per trip: four independent distance / ballot chains, then the bookkeeping
    // in index order
    for (int base = 0; base < npad && (cnta < nsample || cntb < nsample); base += 128) {
      const float4 p = cloud4[base + lane], q = cloud4[base + 64 + lane];
      const unsigned long long ma0 = __ballot(sqdist<FMA>(ax, ay, az, p.x, p.y, p.z) < radius2);
      const unsigned long long mb0 = __ballot(sqdist<FMA>(bx, by, bz, p.x, p.y, p.z) < radius2);
      const unsigned long long ma1 = __ballot(sqdist<FMA>(ax, ay, az, q.x, q.y, q.z) < radius2);
      const unsigned long long mb1 = __ballot(sqdist<FMA>(bx, by, bz, q.x, q.y, q.z) < radius2);
      bq_record(ma0, base, lane, below, nsample, cnta, firsta, outa);
      bq_record(ma1, base + 64, lane, below, nsample, cnta, firsta, outa);
      bq_record(mb0, base, lane, below, nsample, cntb, firstb, outb);
      bq_record(mb1, base + 64, lane, below, nsample, cntb, firstb, outb);
    }
    // the first hit fills the unused slots; a centre without any hit keeps the caller's zeros (ball_query_gpu.cu:36-40)
    if (cnta > 0)
      for (int l = cnta + lane; l < nsample; l += 64) outa[l] = firsta;
    if (hasb && cntb > 0)
      for (int l = cntb + lane; l < nsample; l += 64) outb[l] = firstb;
  }
}

// ------------------------------------------------------------------------------------------
// three_nn (interpolate_gpu.cu:9-52): three smallest squared distances, strict '<', first wins.
// ------------------------------------------------------------------------------------------
// Each thread owns TWO unknown points (halves the LDS traffic per distance) and the known cloud is
// staged as float4 (x,y,z,-) so one ds_read_b128 broadcast feeds both.  The insertion cascade is
// entered only when d beats the current third best.
struct Top3 {
  float b1, b2, b3;
  int i1, i2, i3;
  __device__ __forceinline__ void init() {
    // The reference keeps the bests in doubles initialised to 1e40 and compares the float d against
    // them: every finite d wins over 1e40, +inf / NaN never do, and an unset slot is stored back
    // as +inf.  Float trackers initialised to +inf behave identically.
    b1 = b2 = b3 = __builtin_inff();
    i1 = i2 = i3 = 0;
  }
  __device__ __forceinline__ void push(float d, int k) {
    if (d < b3) {
      if (d < b1) {
        b3 = b2; i3 = i2;
        b2 = b1; i2 = i1;
        b1 = d; i1 = k;
      } else if (d < b2) {
        b3 = b2; i3 = i2;
        b2 = d; i2 = k;
      } else {
        b3 = d; i3 = k;
      }
    }
  }
};

template <bool FMA>
__global__ __launch_bounds__(kT) void three_nn_kernel(int n, int m,
                                                      const float* __restrict__ unknown,
                                                      const float* __restrict__ known,
                                                      float* __restrict__ dist2,
                                                      int* __restrict__ idx) {
  __shared__ float4 tile[kTile];
  const int b = blockIdx.y;
  const int pt0 = blockIdx.x * (2 * kT) + threadIdx.x, pt1 = pt0 + kT;
  float ux0 = 0.f, uy0 = 0.f, uz0 = 0.f, ux1 = 0.f, uy1 = 0.f, uz1 = 0.f;
  if (pt0 < n) {
    const float* u = unknown + ((int64_t)b * n + pt0) * 3;
    ux0 = u[0]; uy0 = u[1]; uz0 = u[2];
  }
  if (pt1 < n) {
    const float* u = unknown + ((int64_t)b * n + pt1) * 3;
    ux1 = u[0]; uy1 = u[1]; uz1 = u[2];
  }
  Top3 t0, t1;
  t0.init();
  t1.init();
  const float* cloud = known + (int64_t)b * m * 3;
  for (int base = 0; base < m; base += kTile) {
    const int len = min(kTile, m - base);
    __syncthreads();
    for (int e = threadIdx.x; e < len; e += kT) {
      const float* c = cloud + (int64_t)(base + e) * 3;
      tile[e] = make_float4(c[0], c[1], c[2], 0.f);
    }
    __syncthreads();
    // four known points per trip (their LDS reads and eight distances are independent), two unknowns per packed
    // instruction; the insertion cascade is entered only by a lane whose distance beats its current third best
    const v2f vx = {ux0, ux1}, vy = {uy0, uy1}, vz = {uz0, uz1};
    int k = 0;
    for (; k + 3 < len; k += 4) {
      const float4 p0 = tile[k], p1 = tile[k + 1], p2 = tile[k + 2], p3 = tile[k + 3];
      const v2f d0 = sqdist2<FMA>(vx, vy, vz, p0.x, p0.y, p0.z), d1 = sqdist2<FMA>(vx, vy, vz, p1.x, p1.y, p1.z);
      const v2f d2 = sqdist2<FMA>(vx, vy, vz, p2.x, p2.y, p2.z), d3 = sqdist2<FMA>(vx, vy, vz, p3.x, p3.y, p3.z);
      if (fminf(fminf(d0.x, d1.x), fminf(d2.x, d3.x)) < t0.b3) {
        t0.push(d0.x, base + k); t0.push(d1.x, base + k + 1); t0.push(d2.x, base + k + 2); t0.push(d3.x, base + k + 3);
      }
      if (fminf(fminf(d0.y, d1.y), fminf(d2.y, d3.y)) < t1.b3) {
        t1.push(d0.y, base + k); t1.push(d1.y, base + k + 1); t1.push(d2.y, base + k + 2); t1.push(d3.y, base + k + 3);
      }
    }
    for (; k < len; ++k) {
      const float4 p = tile[k];
      t0.push(sqdist<FMA>(ux0, uy0, uz0, p.x, p.y, p.z), base + k);
      t1.push(sqdist<FMA>(ux1, uy1, uz1, p.x, p.y, p.z), base + k);
    }
  }
  if (pt0 < n) {
    const int64_t o = ((int64_t)b * n + pt0) * 3;
    dist2[o] = t0.b1; dist2[o + 1] = t0.b2; dist2[o + 2] = t0.b3;
    idx[o] = t0.i1; idx[o + 1] = t0.i2; idx[o + 2] = t0.i3;
  }
  if (pt1 < n) {
    const int64_t o = ((int64_t)b * n + pt1) * 3;
    dist2[o] = t1.b1; dist2[o + 1] = t1.b2; dist2[o + 2] = t1.b3;
    idx[o] = t1.i1; idx[o + 1] = t1.i2; idx[o + 2] = t1.i3;
  }
}

// three_nn with the scan of one unknown split across the 16 lanes of a DPP row (r04; the open item of DESIGN 4.3c).
// The kernel above walks all m known points per thread: at the feature-propagation levels of Pointnet2MSG there are only
// 32 x {256, 1024, 4096} unknowns -- 64 to 1024 waves on 1024 SIMDs -- and a thread pays ~100 ns per scanned point.
// Here lane j of a row takes known points j, j + 16, ... (ascending, so the strict '<' of interpolate_gpu.cu:41-47 keeps
// the first of equal distances inside a lane) for kNU unknowns at once, and the 16 partial top-3 lists are merged by a
// DPP butterfly under the order (distance, index) -- exactly the order the reference's sequential insertion realises:
// among equal distances the smaller index ranks first.  Unset slots are (+inf, 0) in every lane and compare equal.
struct Top3L : Top3 {
  __device__ __forceinline__ static bool lt(float d, int i, float e, int j) { return d < e || (d == e && i < j); }
  __device__ __forceinline__ void push_lex(float d, int k) {
    if (lt(d, k, b3, i3)) {
      if (lt(d, k, b1, i1)) {
        b3 = b2; i3 = i2;
        b2 = b1; i2 = i1;
        b1 = d; i1 = k;
      } else if (lt(d, k, b2, i2)) {
        b3 = b2; i3 = i2;
        b2 = d; i2 = k;
      } else {
        b3 = d; i3 = k;
      }
    }
  }
  template <int CTRL>
  __device__ __forceinline__ void merge_dpp() {
    const float o1 = hcm::dpp_mov<CTRL>(b1), o2 = hcm::dpp_mov<CTRL>(b2), o3 = hcm::dpp_mov<CTRL>(b3);
    const int j1 = __builtin_amdgcn_update_dpp(0, i1, CTRL, 0xF, 0xF, true);
    const int j2 = __builtin_amdgcn_update_dpp(0, i2, CTRL, 0xF, 0xF, true);
    const int j3 = __builtin_amdgcn_update_dpp(0, i3, CTRL, 0xF, 0xF, true);
    push_lex(o1, j1);
    push_lex(o2, j2);
    push_lex(o3, j3);
  }
};
constexpr int kNU = 4;      // unknowns per 16-lane row

template <bool FMA>
__global__ __launch_bounds__(kT) void three_nn_split_kernel(int n, int m, const float* __restrict__ unknown,
                                                            const float* __restrict__ known,
                                                            float* __restrict__ dist2, int* __restrict__ idx) {
  __shared__ float4 tile[kTile];
  const int b = blockIdx.y, l16 = threadIdx.x & 15, row = threadIdx.x >> 4;
  const int u0 = (blockIdx.x * (kT / 16) + row) * kNU;
  float ux[kNU], uy[kNU], uz[kNU];
  Top3L t[kNU];
#pragma unroll
  for (int q = 0; q < kNU; ++q) {
    const int u = min(u0 + q, n - 1);
    const float* p = unknown + ((int64_t)b * n + u) * 3;
    ux[q] = p[0]; uy[q] = p[1]; uz[q] = p[2];
    t[q].init();
  }
  const float* cloud = known + (int64_t)b * m * 3;
  for (int base = 0; base < m; base += kTile) {
    const int len = min(kTile, m - base);
    __syncthreads();
    for (int e = threadIdx.x; e < len; e += kT) {
      const float* c = cloud + (int64_t)(base + e) * 3;
      tile[e] = make_float4(c[0], c[1], c[2], 0.f);
    }
    __syncthreads();
    for (int k = l16; k < len; k += 16) {
      const float4 p = tile[k];
#pragma unroll
      for (int q = 0; q < kNU; ++q) t[q].push(sqdist<FMA>(ux[q], uy[q], uz[q], p.x, p.y, p.z), base + k);
    }
  }
#pragma unroll
  for (int q = 0; q < kNU; ++q) {
    t[q].template merge_dpp<0xB1>();       // quad_perm [1,0,3,2]
    t[q].template merge_dpp<0x4E>();       // quad_perm [2,3,0,1]
    t[q].template merge_dpp<0x141>();      // row_half_mirror
    t[q].template merge_dpp<0x140>();      // row_mirror
    if (l16 == 0 && u0 + q < n) {
      const int64_t o = ((int64_t)b * n + u0 + q) * 3;
      dist2[o] = t[q].b1; dist2[o + 1] = t[q].b2; dist2[o + 2] = t[q].b3;
      idx[o] = t[q].i1; idx[o + 1] = t[q].i2; idx[o + 2] = t[q].i3;
    }
  }
}

// ------------------------------------------------------------------------------------------
// furthest point sampling (sampling_gpu.cu:93-209).
// Virtual thread id of point k is (k % bs), bs = largest power of two <= n, capped at 1024
// (cuda_utils.h:10-14).  Per thread the reference keeps the first strict maximum over ascending
// k; its LDS tree then folds slot t+s into slot t for s = bs/2 .. 1, keeping slot t on ties
// (:86-91, :140-200).  A candidate of thread T therefore beats an equal-valued candidate of
// thread T' iff bitreverse_{log2 bs}(T) < bitreverse_{log2 bs}(T'): the winner is the maximum
// under the total order (value desc, bit-reversed virtual tid asc), and any reduction shape that
// honours this order is bit-identical to the reference tree.
// One workgroup of bs threads per cloud; thread t owns points t, t+bs, ... in registers.
// ------------------------------------------------------------------------------------------
// Candidates are packed into ONE 64-bit unsigned key so that the whole arg-max is a max-reduction:
//   high 32 bits: the distance's float bits (distances are >= +0, so their bit patterns order like
//                 unsigned integers);
//   low  32 bits: 0xFFFFFFFF - ((bitrev(tid) << 21) | j)  -> among equal distances the smaller
//                 bit-reversed thread id wins; j (< 2^21) says which of the thread's points it was.
// A serial round then costs 4 DPP steps + 8 v_readlane + one LDS exchange and ONE barrier
// (the per-wave results are double-buffered and every wave finishes the reduction redundantly).
__device__ __forceinline__ unsigned long long umax64(unsigned long long a, unsigned long long b) {
  return a > b ? a : b;
}
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_mov64(unsigned long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, 0xF, 0xF, true);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, 0xF, 0xF, true);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long row16_umax64(unsigned long long v) {
  v = umax64(v, dpp_mov64<0xB1>(v));
  v = umax64(v, dpp_mov64<0x4E>(v));
  v = umax64(v, dpp_mov64<0x141>(v));
  v = umax64(v, dpp_mov64<0x140>(v));
  return v;
}
__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int l) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
}
// Every point k has the unique key (distance, ~(bitrev(k % bs) << 21 | k / bs)); the global maximum
// of that key IS the reference winner whatever physical thread evaluated the point, so the physical
// workgroup size P is a pure performance choice (fewer waves = cheaper barrier, more points each).
//
// General form (clouds that do not fit the register / LDS form below): points and running minima in global memory,
// per-wave 64-bit candidates exchanged through LDS slots.
template <bool FMA>
__global__ __launch_bounds__(1024) void fps_generic_kernel(int n, int m, int bs, int log2bs,
                                                           const float* __restrict__ dataset,
                                                           float* __restrict__ temp, int* __restrict__ idxs) {
  __shared__ unsigned long long red[2][16];
  if (m <= 0) return;
  const int b = blockIdx.x, tid = threadIdx.x, P = blockDim.x;
  const float* cloud = dataset + (int64_t)b * n * 3;
  float* tmp = temp + (int64_t)b * n;
  int* out = idxs + (int64_t)b * m;
  const int wave = tid >> 6;
  const int shift = 32 - log2bs;
  auto point_key = [&](int k) -> unsigned {   // low word of the candidate key of point k
    const unsigned vt = (unsigned)k & (unsigned)(bs - 1);
    const unsigned brev = (log2bs == 0) ? 0u : (__brev(vt) >> shift);
    return 0xFFFFFFFFu - ((brev << 21) | (unsigned)(k >> log2bs));
  };
  if (tid < 32) red[tid >> 4][tid & 15] = 0ull;  // slots of absent waves stay "lowest"
  if (tid == 0) out[0] = 0;
  __syncthreads();
  int old = 0;
  for (int r = 1; r < m; ++r) {
    const float ox = cloud[3 * old], oy = cloud[3 * old + 1], oz = cloud[3 * old + 2];
    unsigned long long c = 0ull;
    for (int k = tid; k < n; k += P) {
      const float d = sqdist<FMA>(cloud[3 * k], cloud[3 * k + 1], cloud[3 * k + 2], ox, oy, oz);
      const float d2 = fminf(d, tmp[k]);
      tmp[k] = d2;
      c = umax64(c, ((unsigned long long)__float_as_uint(d2) << 32) | point_key(k));
    }
    c = row16_umax64(c);
    c = umax64(umax64(readlane64(c, 0), readlane64(c, 16)), umax64(readlane64(c, 32), readlane64(c, 48)));
    const int par = r & 1;
    if ((tid & 63) == 0) red[par][wave] = c;
    __syncthreads();
    unsigned long long w = red[par][tid & 15];
    w = row16_umax64(w);
    const unsigned low = 0xFFFFFFFFu - (unsigned)w;
    const unsigned wkey = low >> 21, wj = low & 0x1FFFFFu;
    const unsigned wvt = (log2bs == 0) ? 0u : (__brev(wkey) >> shift);
    old = (int)wvt + (int)(wj << log2bs);
    if (tid == 0) out[r] = old;
  }
}

// Register form: thread t owns points t, t+P, ... (coordinates, running minima and key low words in registers), the
// cloud also sits in LDS for the look-up of the picked point.
// r04: a round is a SERIAL chain -- a wave issues one instruction per four cycles whatever its kind, and every wave
// walks the same chain (profiles/r04_fps_counters.txt: 220 instructions per wave and round = half of the round,
// most of the rest at the LDS exchange; n = 256 costs 0.4 us a round, n = 4096 0.72) -- so the round is written for
// the LENGTH of that chain:
//   * two points per packed instruction (v_pk_add / v_pk_mul / v_pk_fma: element-wise IEEE, bit-identical);
//   * running minima are kept as BIT PATTERNS and compared as unsigned integers (distances are >= +0, where the two
//     orders agree): v_min_u32 / v_max3_u32 need no canonicalising v_max_f32 x, x in front of them and the DPP steps
//     fuse into v_max_u32_dpp.  Empty slots (k >= n) carry the minimum 0 and the key 0: they never raise a
//     maximum, and key 0 loses to every point;
//   * per point only distance, v_min and half a v_max3 -- no 64-bit compare-and-select.  The wave reduces the VALUE
//     alone (six v_max_u32_dpp, result in a scalar register); only the lanes that hold it (one, unless points
//     coincide) form a key, max over their matching points' low words;
//   * the workgroup's winner is ONE LDS word: the holding lane of each wave posts its 64-bit key with ds_max_u64 (when
//     several lanes of a wave hold the maximum -- duplicated points, the constant clouds of empty-mask images --
//     their low words are reduced first so that a wave never posts more than once), and after the barrier every lane
//     reads that word: no per-wave slots, no second reduction.  Three words rotate: the one of round r+1 is cleared
//     in round r, after barrier r-1 (its readers, round r-2, have consumed it to get there) and before barrier r
//     (its writers, round r+1, come after).
template <int CTRL, int ROWS>
__device__ __forceinline__ unsigned dpp_umax(unsigned v) {
  return max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWS, 0xF, ROWS == 0xF));
}
__device__ __forceinline__ unsigned wave_umax(unsigned v) {     // scalar result: the maximum over the 64 lanes
  v = dpp_umax<0xB1, 0xF>(v);      // quad_perm [1,0,3,2]
  v = dpp_umax<0x4E, 0xF>(v);      // quad_perm [2,3,0,1]
  v = dpp_umax<0x141, 0xF>(v);     // row_half_mirror
  v = dpp_umax<0x140, 0xF>(v);     // row_mirror
  v = dpp_umax<0x142, 0xA>(v);     // row_bcast:15 into rows 1 and 3
  v = dpp_umax<0x143, 0xC>(v);     // row_bcast:31 into rows 2 and 3
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_umax_step(unsigned v, int step) {   // step is a compile-time constant at every call
  switch (step) {
    case 0: return dpp_umax<0xB1, 0xF>(v);
    case 1: return dpp_umax<0x4E, 0xF>(v);
    case 2: return dpp_umax<0x141, 0xF>(v);
    case 3: return dpp_umax<0x140, 0xF>(v);
    case 4: return dpp_umax<0x142, 0xA>(v);
    default: return dpp_umax<0x143, 0xC>(v);
  }
}
template <int PER, bool FMA>
__global__ __launch_bounds__(1024) void fps_kernel(int n, int m, int bs, int log2bs,
                                                   const float* __restrict__ dataset,
                                                   float* __restrict__ temp,
                                                   int* __restrict__ idxs) {
  extern __shared__ __attribute__((aligned(16))) float sxyz[];  // [n*3]
  __shared__ unsigned long long win[3];
  if (m <= 0) return;
  const int b = blockIdx.x, tid = threadIdx.x, P = blockDim.x;
  const float* cloud = dataset + (int64_t)b * n * 3;
  float* tmp = temp + (int64_t)b * n;
  int* out = idxs + (int64_t)b * m;
  const int shift = 32 - log2bs;
  auto point_key = [&](int k) -> unsigned {   // low word of the candidate key of point k
    const unsigned vt = (unsigned)k & (unsigned)(bs - 1);
    const unsigned brev = (log2bs == 0) ? 0u : (__brev(vt) >> shift);
    return 0xFFFFFFFFu - ((brev << 21) | (unsigned)(k >> log2bs));
  };

  constexpr int NP = (PER + 1) / 2;                   // point pairs per thread: slots 2*jj and 2*jj+1
  v2f px[NP], py[NP], pz[NP];
  unsigned pt[2 * NP];                                // running minima, as bit patterns
  unsigned pk[2 * NP];
  for (int e = tid; e < n * 3; e += P) sxyz[e] = cloud[e];
#pragma unroll
  for (int j = 0; j < 2 * NP; ++j) {
    const int k = tid + j * P;
    float x = 0.f, y = 0.f, z = 0.f, t = 0.f;
    unsigned key = 0u;
    if (j < PER && k < n) { x = cloud[3 * k]; y = cloud[3 * k + 1]; z = cloud[3 * k + 2]; t = tmp[k]; key = point_key(k); }
    if (j & 1) { px[j >> 1].y = x; py[j >> 1].y = y; pz[j >> 1].y = z; }
    else { px[j >> 1].x = x; py[j >> 1].x = y; pz[j >> 1].x = z; }
    pt[j] = __float_as_uint(t);
    pk[j] = key;
  }
  if (tid < 3) win[tid] = 0ull;
  if (tid == 0) out[0] = 0;
  __syncthreads();

  int old = 0, slot = 1;       // slot = r % 3
  for (int r = 1; r < m; ++r) {
    const float ox = sxyz[3 * old], oy = sxyz[3 * old + 1], oz = sxyz[3 * old + 2];
    // stage by stage over the pairs: consecutive packed instructions are independent (a dependent pair costs a wait state)
    v2f dx[NP], dy[NP], dz[NP], acc[NP];
#pragma unroll
    for (int jj = 0; jj < NP; ++jj) { dx[jj] = px[jj] - ox; dy[jj] = py[jj] - oy; dz[jj] = pz[jj] - oz; }
#pragma unroll
    for (int jj = 0; jj < NP; ++jj) acc[jj] = dy[jj] * dy[jj];
#pragma unroll
    for (int jj = 0; jj < NP; ++jj) acc[jj] = FMA ? __builtin_elementwise_fma(dx[jj], dx[jj], acc[jj]) : dx[jj] * dx[jj] + acc[jj];
#pragma unroll
    for (int jj = 0; jj < NP; ++jj) acc[jj] = FMA ? __builtin_elementwise_fma(dz[jj], dz[jj], acc[jj]) : acc[jj] + dz[jj] * dz[jj];
    unsigned best = 0u;
#pragma unroll
    for (int jj = 0; jj < NP; ++jj) {
      pt[2 * jj] = min(__float_as_uint(acc[jj].x), pt[2 * jj]);
      pt[2 * jj + 1] = min(__float_as_uint(acc[jj].y), pt[2 * jj + 1]);
      best = max(best, max(pt[2 * jj], pt[2 * jj + 1]));
    }
    // every lane's own candidate, a step of it between two DPP steps of the wave's maximum (each fills the other's wait
    // states: v_cmp -> v_cndmask on the mask and DPP on a register just written both want two)
    unsigned v = 0u, wred = best;
#pragma unroll
    for (int j = 0; j < 2 * NP; ++j) {
      v = max(v, pt[j] == best ? pk[j] : 0u);
      if (j < 6) wred = wave_umax_step(wred, j);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 2 * NP; j < 6; ++j) wred = wave_umax_step(wred, j);
    const unsigned wbits = (unsigned)__builtin_amdgcn_readlane((int)wred, 63);
    const bool mine = best == wbits;
    bool post = mine;
    if (__popcll(__builtin_amdgcn_ballot_w64(mine)) != 1) {       // coincident points: one post per wave all the same
      v = wave_umax(mine ? v : 0u);
      post = (tid & 63) == 0;
    }
    const int next = slot == 2 ? 0 : slot + 1;
    if (post) atomicMax(&win[slot], ((unsigned long long)wbits << 32) | v);
    if (tid == 0) win[next] = 0ull;
    __syncthreads();
    const unsigned low = 0xFFFFFFFFu - (unsigned)win[slot];
    const unsigned wkey = low >> 21, wj = low & 0x1FFFFFu;
    const unsigned wvt = (log2bs == 0) ? 0u : (__brev(wkey) >> shift);
    old = (int)wvt + (int)(wj << log2bs);
    if (tid == 0) out[r] = old;
    slot = next;
  }
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int k = tid + j * P;
    if (k < n) tmp[k] = __uint_as_float(pt[j]);
  }
}

inline int opt_n_threads(int work) {  // cuda_utils.h:10-14
  int p = 1;
  while ((p << 1) <= work && (p << 1) <= 1024) p <<= 1;
  return p;
}

inline dim3 grid3(int q, int c, int b) { return dim3((q + kT - 1) / kT, (c + kCB - 1) / kCB, b); }


// ---------------------------------------------------------------------------------------------
// Max over the ball: y[r] = max_j x[r][j], arg[r] = first j attaining it  (F.max_pool2d with kernel
// [1, nsample] in PointnetSAModuleMSG.forward, pointnet2_modules.py:60-63; ATen's NCHW pooling kernel
// walks one output per thread with a stride of nsample floats: 0.9 ms for a 268 MB tensor).
// LPR lanes share a row (consecutive floats -> coalesced), segmented (max, first index) butterfly.
// ---------------------------------------------------------------------------------------------
template <int LPR>
__global__ __launch_bounds__(256) void rowmax_fwd_kernel(const float* __restrict__ x, long long rows, int ns,
                                                         float* __restrict__ y, int* __restrict__ arg) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long r = t / LPR;
  const int l = (int)(t - r * LPR);
  float best = -INFINITY;
  int bi = 0x7fffffff;
  if (r < rows) {
    const float* row = x + r * ns;
    for (int j = l; j < ns; j += LPR) {
      const float v = row[j];
      if (v > best || (bi == 0x7fffffff)) { best = v; bi = j; }   // strict >: first index wins (NaN never beats)
    }
  }
#pragma unroll
  for (int off = LPR / 2; off > 0; off >>= 1) {
    const float ov = __shfl_xor(best, off, 64);
    const int oi = __shfl_xor(bi, off, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (r < rows && l == 0) { y[r] = best; arg[r] = bi; }
}

// nsample = 16 / 32 (every level of Pointnet2MSG): a row is 4 / 8 lanes of one float4 each; the lane's own four
// candidates in order, then quad_perm / row_half_mirror DPP exchanges of (value, index) -- no LDS crossbar.  The
// generic kernel above spends ten ds_bpermute per 256 bytes of input on its butterfly: 0.40 ms for a 268 MB ball
// tensor (0.67 TB/s).
template <int CTRL>
__device__ __forceinline__ void rowmax_merge(float& best, int& bi) {
  const float ov = hcm::dpp_mov<CTRL>(best);
  const int oi = __builtin_amdgcn_update_dpp(0, bi, CTRL, 0xF, 0xF, true);
  if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
}
template <int NS>
__global__ __launch_bounds__(256) void rowmax_fwd_vec_kernel(const float4* __restrict__ x, long long rows,
                                                             float* __restrict__ y, int* __restrict__ arg) {
  constexpr int LQ = NS / 4;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long r = t / LQ;
  const int l = (int)(t & (LQ - 1));
  float best = -INFINITY;
  int bi = 0x7fffffff;
  if (r < rows) {
    const float4 v = x[t];
    best = v.x; bi = 4 * l;                                 // the first candidate is taken as it is (like the scan above)
    if (v.y > best) { best = v.y; bi = 4 * l + 1; }
    if (v.z > best) { best = v.z; bi = 4 * l + 2; }
    if (v.w > best) { best = v.w; bi = 4 * l + 3; }
  }
  rowmax_merge<0xB1>(best, bi);                             // quad_perm [1,0,3,2]
  rowmax_merge<0x4E>(best, bi);                             // quad_perm [2,3,0,1]
  if (LQ == 8) rowmax_merge<0x141>(best, bi);               // row_half_mirror: lane i <-> 7 - i of each 8
  if (r < rows && l == 0) { y[r] = best; arg[r] = bi; }
}

// four consecutive elements of dx per thread (nsample % 4 == 0): one float4 store
__global__ __launch_bounds__(256) void rowmax_bwd_vec_kernel(const float* __restrict__ dy, const int* __restrict__ arg,
                                                             long long total4, int ns, float4* __restrict__ dx) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const long long e = 4 * i, r = e / ns;
  const int j = (int)(e - r * ns) , a = arg[r] - j;
  const float d = dy[r];
  dx[i] = make_float4(a == 0 ? d : 0.f, a == 1 ? d : 0.f, a == 2 ? d : 0.f, a == 3 ? d : 0.f);
}

__global__ __launch_bounds__(256) void rowmax_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ arg,
                                                         long long total, int ns, float* __restrict__ dx) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long long r = i / ns;
  const int j = (int)(i - r * ns);
  dx[i] = arg[r] == j ? dy[r] : 0.f;
}
}  // namespace

extern "C" {

int hcm_gather_points(int b, int c, int n, int npoints, const float* points, const int* idx,
                      float* out, hcm_stream_t stream) {
  if (b <= 0 || c <= 0 || npoints <= 0) return b < 0 || c < 0 || npoints < 0 ? (int)hipErrorInvalidValue : 0;
  gather_rows_kernel<<<grid3(npoints, c, b), kT, 0, (hipStream_t)stream>>>(c, n, npoints, points, idx, out);
  HCM_CHECK_LAUNCH();
  return 0;
}
int hcm_gather_points_grad(int b, int c, int n, int npoints, const float* grad_out, const int* idx,
                           float* grad_points, hcm_stream_t stream) {
  if (b <= 0 || c <= 0 || npoints <= 0) return b < 0 || c < 0 || npoints < 0 ? (int)hipErrorInvalidValue : 0;
  scatter_rows_kernel<<<grid3(npoints, c, b), kT, 0, (hipStream_t)stream>>>(c, n, npoints, grad_out, idx, grad_points);
  HCM_CHECK_LAUNCH();
  return 0;
}
int hcm_group_points(int b, int c, int n, int npoints, int nsample, const float* points,
                     const int* idx, float* out, hcm_stream_t stream) {
  const int q = npoints * nsample;
  if (b <= 0 || c <= 0 || q <= 0) return b < 0 || c < 0 || q < 0 ? (int)hipErrorInvalidValue : 0;
  gather_rows_kernel<<<grid3(q, c, b), kT, 0, (hipStream_t)stream>>>(c, n, q, points, idx, out);
  HCM_CHECK_LAUNCH();
  return 0;
}
int hcm_group_points_grad(int b, int c, int n, int npoints, int nsample, const float* grad_out,
                          const int* idx, float* grad_points, hcm_stream_t stream) {
  const int q = npoints * nsample;
  if (b <= 0 || c <= 0 || q <= 0) return b < 0 || c < 0 || q < 0 ? (int)hipErrorInvalidValue : 0;
  scatter_rows_kernel<<<grid3(q, c, b), kT, 0, (hipStream_t)stream>>>(c, n, q, grad_out, idx, grad_points);
  HCM_CHECK_LAUNCH();
  return 0;
}
int hcm_three_interpolate(int b, int c, int m, int n, const float* points, const int* idx,
                          const float* weight, float* out, hcm_stream_t stream) {
  return hcm_three_interpolate_contract(b, c, m, n, points, idx, weight, out, HCM_CONTRACT_FMA, stream);
}
int hcm_three_interpolate_contract(int b, int c, int m, int n, const float* points, const int* idx,
                                   const float* weight, float* out, int contract, hcm_stream_t stream) {
  if (contract != HCM_CONTRACT_FMA && contract != HCM_CONTRACT_IEEE) return (int)hipErrorInvalidValue;
  if (b <= 0 || c <= 0 || n <= 0) return b < 0 || c < 0 || n < 0 ? (int)hipErrorInvalidValue : 0;
  // source rows in LDS when a useful block of channels of all m points fits (m <= 8960 points at 4 channels) and there
  // is enough work per workgroup to pay for the fill; the gather kernel otherwise
  int cbl = m > 0 ? (int)((140 * 1024) / ((size_t)m * sizeof(float))) & ~3 : 0;
  if (cbl > 32) cbl = 32;
  if (cbl > ((c + 3) & ~3)) cbl = (c + 3) & ~3;
  if (cbl >= 4 && n >= 2048) {
    const int nblk = (c + cbl - 1) / cbl;
    // positions per workgroup: all of them unless that leaves the GPU short of workgroups (fill cost ~ m positions)
    int nsplit = 1;
    while ((long long)b * nblk * nsplit < 512 && n / (nsplit * 2) >= 4 * m && n / (nsplit * 2) >= 4096) nsplit *= 2;
    const int npos = (n + nsplit - 1) / nsplit;
    const size_t lds = (size_t)cbl * m * sizeof(float);
    const void* fn = contract == HCM_CONTRACT_FMA ? reinterpret_cast<const void*>(three_interp_lds_kernel<true>)
                                                  : reinterpret_cast<const void*>(three_interp_lds_kernel<false>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    dim3 grid(nsplit, nblk, b);
    if (contract == HCM_CONTRACT_FMA)
      three_interp_lds_kernel<true><<<grid, kTIThreads, lds, (hipStream_t)stream>>>(c, m, n, cbl, npos, points, idx, weight, out);
    else
      three_interp_lds_kernel<false><<<grid, kTIThreads, lds, (hipStream_t)stream>>>(c, m, n, cbl, npos, points, idx, weight, out);
    HCM_CHECK_LAUNCH();
    return 0;
  }
  if (contract == HCM_CONTRACT_FMA)
    three_interp_kernel<true><<<grid3(n, c, b), kT, 0, (hipStream_t)stream>>>(c, m, n, points, idx, weight, out);
  else
    three_interp_kernel<false><<<grid3(n, c, b), kT, 0, (hipStream_t)stream>>>(c, m, n, points, idx, weight, out);
  HCM_CHECK_LAUNCH();
  return 0;
}
int hcm_three_interpolate_grad(int b, int c, int n, int m, const float* grad_out, const int* idx,
                               const float* weight, float* grad_points, hcm_stream_t stream) {
  if (b <= 0 || c <= 0 || n <= 0) return b < 0 || c < 0 || n < 0 ? (int)hipErrorInvalidValue : 0;
  three_interp_grad_kernel<<<grid3(n, c, b), kT, 0, (hipStream_t)stream>>>(c, n, m, grad_out, idx, weight, grad_points);
  HCM_CHECK_LAUNCH();
  return 0;
}
int hcm_ball_query(int b, int n, int m, float radius, int nsample, const float* new_xyz,
                   const float* xyz, int* idx, hcm_stream_t stream) {
  return hcm_ball_query_contract(b, n, m, radius, nsample, new_xyz, xyz, idx, HCM_CONTRACT_FMA, stream);
}
int hcm_ball_query_contract(int b, int n, int m, float radius, int nsample, const float* new_xyz,
                            const float* xyz, int* idx, int contract, hcm_stream_t stream) {
  if (contract != HCM_CONTRACT_FMA && contract != HCM_CONTRACT_IEEE) return (int)hipErrorInvalidValue;
  if (b <= 0 || m <= 0 || nsample <= 0) return b < 0 || m < 0 || nsample < 0 ? (int)hipErrorInvalidValue : 0;
  hcm::ProfSpan span(HCM_PROF_BALL_QUERY, (hipStream_t)stream, (double)b * m * (double)n);
  // the cloud fits LDS (every level of Pointnet2MSG: n <= 4096 points = 64 KB): one WAVE per centre, lanes over the points
  const size_t cloud_lds = (size_t)((n + 127) & ~127) * sizeof(float4);
  if (n > 0 && cloud_lds <= 72 * 1024) {          // two workgroups per CU
    const void* fw = contract == HCM_CONTRACT_FMA ? reinterpret_cast<const void*>(ball_query_wave_kernel<true>)
                                                  : reinterpret_cast<const void*>(ball_query_wave_kernel<false>);
    hipError_t e2 = hipFuncSetAttribute(fw, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cloud_lds);
    if (e2 != hipSuccess) return (int)e2;
    dim3 gridw((m + kBQWaves * kBQCentres - 1) / (kBQWaves * kBQCentres), b);
    if (contract == HCM_CONTRACT_FMA)
      ball_query_wave_kernel<true><<<gridw, kBQWaves * 64, cloud_lds, (hipStream_t)stream>>>(n, m, radius * radius, nsample,
                                                                                         new_xyz, xyz, idx);
    else
      ball_query_wave_kernel<false><<<gridw, kBQWaves * 64, cloud_lds, (hipStream_t)stream>>>(n, m, radius * radius, nsample,
                                                                                          new_xyz, xyz, idx);
    HCM_CHECK_LAUNCH();
    return 0;
  }
  // one thread per centre: shrink the workgroup when there are too few centres to fill 256 CUs
  int threads = kT;
  while (threads > 64 && (long long)b * ((m + threads - 1) / threads) < 1024) threads >>= 1;
  while (threads > 64 && (size_t)threads * nsample * sizeof(int) > 96 * 1024) threads >>= 1;
  const size_t lds = (size_t)threads * nsample * sizeof(int);
  if (lds > 128 * 1024) return (int)hipErrorInvalidValue;   // nsample > 512
  const void* fn = contract == HCM_CONTRACT_FMA ? reinterpret_cast<const void*>(ball_query_kernel<true>)
                                                : reinterpret_cast<const void*>(ball_query_kernel<false>);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  dim3 grid((m + threads - 1) / threads, b);
  if (contract == HCM_CONTRACT_FMA)
    ball_query_kernel<true><<<grid, threads, lds, (hipStream_t)stream>>>(n, m, radius * radius, nsample, new_xyz, xyz, idx);
  else
    ball_query_kernel<false><<<grid, threads, lds, (hipStream_t)stream>>>(n, m, radius * radius, nsample, new_xyz, xyz, idx);
  HCM_CHECK_LAUNCH();
  return 0;
}
int hcm_three_nn(int b, int n, int m, const float* unknown, const float* known, float* dist2,
                 int* idx, hcm_stream_t stream) {
  return hcm_three_nn_contract(b, n, m, unknown, known, dist2, idx, HCM_CONTRACT_FMA, stream);
}
int hcm_three_nn_contract(int b, int n, int m, const float* unknown, const float* known, float* dist2,
                          int* idx, int contract, hcm_stream_t stream) {
  if (contract != HCM_CONTRACT_FMA && contract != HCM_CONTRACT_IEEE) return (int)hipErrorInvalidValue;
  if (b <= 0 || n <= 0) return b < 0 || n < 0 ? (int)hipErrorInvalidValue : 0;
  hcm::ProfSpan span(HCM_PROF_THREE_NN, (hipStream_t)stream, (double)b * n * (double)m);
  // few unknowns (the feature-propagation levels: b x n <= 2^17 ... 2^19): split each scan across 16 lanes; many
  // (pts2depth: 2 M unknowns) -> a thread per two unknowns keeps every SIMD busy.
  if ((long long)b * n <= (1ll << 19) && m >= 16) {
    dim3 gs((n + (kT / 16) * kNU - 1) / ((kT / 16) * kNU), b);
    if (contract == HCM_CONTRACT_FMA)
      three_nn_split_kernel<true><<<gs, kT, 0, (hipStream_t)stream>>>(n, m, unknown, known, dist2, idx);
    else
      three_nn_split_kernel<false><<<gs, kT, 0, (hipStream_t)stream>>>(n, m, unknown, known, dist2, idx);
    HCM_CHECK_LAUNCH();
    return 0;
  }
  dim3 grid((n + 2 * kT - 1) / (2 * kT), b);
  if (contract == HCM_CONTRACT_FMA)
    three_nn_kernel<true><<<grid, kT, 0, (hipStream_t)stream>>>(n, m, unknown, known, dist2, idx);
  else
    three_nn_kernel<false><<<grid, kT, 0, (hipStream_t)stream>>>(n, m, unknown, known, dist2, idx);
  HCM_CHECK_LAUNCH();
  return 0;
}

int hcm_scatter_add_lds(const float* grad_out, const float* coef, const int* idx, int B, int C,
                        int Qsrc, int Q, int m, int div, float* grad_points, hcm_stream_t stream) {
  if (B <= 0 || C <= 0 || Q <= 0 || m <= 0 || div <= 0 || Qsrc <= 0) return (int)hipErrorInvalidValue;
  int cbl = kLdsScatterBytes / (int)(m * sizeof(float));
  if (cbl < 1) return (int)hipErrorInvalidConfiguration;  // target row does not fit LDS: use the atomic kernels
  if (cbl > 32) cbl = 32;
  if (cbl > C) cbl = C;
  // keep at least ~2 workgroups per CU when there are enough channels to split
  while (cbl > 1 && (long long)B * ((C + cbl - 1) / cbl) < 512) cbl = (cbl + 1) / 2;
  const size_t lds = (size_t)cbl * m * sizeof(float);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lds_scatter_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  dim3 grid((C + cbl - 1) / cbl, B);
  lds_scatter_kernel<<<grid, kLdsScatterThreads, lds, (hipStream_t)stream>>>(grad_out, coef, idx, C, Qsrc,
                                                                            Q, m, div, cbl, grad_points);
  HCM_CHECK_LAUNCH();
  return 0;
}

int hcm_furthest_point_sampling(int b, int n, int m, const float* dataset, float* temp, int* idxs,
                                hcm_stream_t stream) {
  return hcm_furthest_point_sampling_contract(b, n, m, dataset, temp, idxs, HCM_CONTRACT_FMA, stream);
}
int hcm_furthest_point_sampling_contract(int b, int n, int m, const float* dataset, float* temp, int* idxs,
                                         int contract, hcm_stream_t stream) {
  if (contract != HCM_CONTRACT_FMA && contract != HCM_CONTRACT_IEEE) return (int)hipErrorInvalidValue;
  if (b <= 0 || n <= 0 || m <= 0) return b < 0 || n < 0 || m < 0 ? (int)hipErrorInvalidValue : 0;
  const int bs = opt_n_threads(n);   // the reference's block size: defines the tie-break order only
  int threads = bs < 64 ? 64 : bs;
  if (threads > 512) threads = 512;       // 512 threads x 8 points at n = 4096: the round's serial chain is shortest there (r04)
  const int per = (n + threads - 1) / threads;
  int log2bs = 0;
  while ((1 << log2bs) < bs) ++log2bs;
  const size_t lds = (size_t)n * 3 * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  const bool fma = contract == HCM_CONTRACT_FMA;
  hcm::ProfSpan span(HCM_PROF_FPS, st, (double)b * m * (double)n);
#define HCM_FPS1(P, F)                                                                      \
  do {                                                                                      \
    hipFuncSetAttribute(reinterpret_cast<const void*>(fps_kernel<P, F>),                     \
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
    fps_kernel<P, F><<<b, threads, lds, st>>>(n, m, bs, log2bs, dataset, temp, idxs);        \
  } while (0)
#define HCM_FPS(P) do { if (fma) HCM_FPS1(P, true); else HCM_FPS1(P, false); } while (0)
  if (per <= 1 && lds <= 150 * 1024) HCM_FPS(1);
  else if (per <= 2 && lds <= 150 * 1024) HCM_FPS(2);
  else if (per <= 4 && lds <= 150 * 1024) HCM_FPS(4);
  else if (per <= 8 && lds <= 150 * 1024) HCM_FPS(8);
  else if (per <= 16 && lds <= 150 * 1024) HCM_FPS(16);
  else if (fma) fps_generic_kernel<true><<<b, threads, 0, st>>>(n, m, bs, log2bs, dataset, temp, idxs);
  else fps_generic_kernel<false><<<b, threads, 0, st>>>(n, m, bs, log2bs, dataset, temp, idxs);
#undef HCM_FPS
#undef HCM_FPS1
  HCM_CHECK_LAUNCH();
  return 0;
}


int hcm_rowmax_forward(const float* x, long long rows, int ns, float* y, int* arg, hcm_stream_t stream) {
  if (rows < 0 || ns <= 0 || !x || !y || !arg) return (int)hipErrorInvalidValue;
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if ((ns == 16 || ns == 32) && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    const long long threads = rows * (ns / 4);
    const unsigned grid = (unsigned)((threads + 255) / 256);
    if (ns == 16) rowmax_fwd_vec_kernel<16><<<grid, 256, 0, st>>>(reinterpret_cast<const float4*>(x), rows, y, arg);
    else rowmax_fwd_vec_kernel<32><<<grid, 256, 0, st>>>(reinterpret_cast<const float4*>(x), rows, y, arg);
    HCM_CHECK_LAUNCH();
    return 0;
  }
  int lpr = 1;
  while (lpr < 64 && lpr * 2 <= ns) lpr *= 2;                    // largest power of two <= min(ns, 64)
  const long long threads = rows * lpr;
  const unsigned grid = (unsigned)((threads + 255) / 256);
  switch (lpr) {
    case 1: rowmax_fwd_kernel<1><<<grid, 256, 0, st>>>(x, rows, ns, y, arg); break;
    case 2: rowmax_fwd_kernel<2><<<grid, 256, 0, st>>>(x, rows, ns, y, arg); break;
    case 4: rowmax_fwd_kernel<4><<<grid, 256, 0, st>>>(x, rows, ns, y, arg); break;
    case 8: rowmax_fwd_kernel<8><<<grid, 256, 0, st>>>(x, rows, ns, y, arg); break;
    case 16: rowmax_fwd_kernel<16><<<grid, 256, 0, st>>>(x, rows, ns, y, arg); break;
    case 32: rowmax_fwd_kernel<32><<<grid, 256, 0, st>>>(x, rows, ns, y, arg); break;
    default: rowmax_fwd_kernel<64><<<grid, 256, 0, st>>>(x, rows, ns, y, arg); break;
  }
  HCM_CHECK_LAUNCH();
  return 0;
}

int hcm_rowmax_backward(const float* dy, const int* arg, long long rows, int ns, float* dx, hcm_stream_t stream) {
  if (rows < 0 || ns <= 0 || !dy || !arg || !dx) return (int)hipErrorInvalidValue;
  if (rows == 0) return 0;
  const long long total = rows * ns;
  if (ns % 4 == 0 && (reinterpret_cast<uintptr_t>(dx) & 15) == 0) {
    rowmax_bwd_vec_kernel<<<(unsigned)((total / 4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        dy, arg, total / 4, ns, reinterpret_cast<float4*>(dx));
    HCM_CHECK_LAUNCH();
    return 0;
  }
  rowmax_bwd_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(dy, arg, total, ns, dx);
  HCM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
