// Deterministic backward of the three gather-type PointNet++ ops:
//     grad_points[b, c, j] = sum_{q : idx[b, q] == j} coef[b, q] * grad_out[b, c, q / div]
// (group_points_grad: coef = 1, div = 1, q over npoint * nsample; gather_points_grad: coef = 1, div = 1;
//  three_interpolate_grad: coef = weight, div = 3, q = 3 n + t).   Reference: one global float atomicAdd per (channel,
// contribution) in arbitrary order (src/group_points_gpu.cu:8-25, src/sampling_gpu.cu:46-63, src/interpolate_gpu.cu:120-142).
//
// Two steps, no float atomics, every sum in a fixed order:
//   hcm_scatter_sort        ONCE per index tensor (forward's idx is reused by every backward that needs it): the
//                           contributions are sorted by (b, target, q) -- rocPRIM radix sort on the key b * m + idx, which
//                           is stable, so q ascends inside a target -- and `seg` receives the first sorted position of
//                           every (b, target).
//   hcm_scatter_add_sorted  workgroup = (b, tile of JT targets, block of CBL channels); the tile's accumulators
//                           acc[CBL][JT] live in LDS; its contributions are ONE contiguous piece of the sorted list, dealt
//                           to the 16 waves in equal chunks whatever the targets are (a hub target -- an empty-mask image
//                           sends all 196 608 contributions of pts2depth to 3 targets -- is shared by all waves).  A wave
//                           takes 64 consecutive sorted contributions per step: coalesced loads of (q, target), one
//                           gather of grad_out per channel, a segmented inclusive scan across the lanes (lanes are sorted
//                           by target, so the six compare masks are computed once per step and reused by every channel),
//                           and the tail lane of each segment adds the segment's sum to the accumulator.  Targets that can
//                           straddle two waves' chunks (the first and the last of a chunk) go to a per-wave side slot and
//                           are merged after a barrier in wave order = sorted order.  The tree shape depends only on the
//                           sorted positions: bit-reproducible.
// The r02 kernel this replaces (hcm_scatter_add_lds: LDS float atomics in arrival order, every channel block re-reading
// the whole index stream) ran at 1.0-1.2 TB/s of algorithmic bytes and was not reproducible.
#include <hipcub/hipcub.hpp>

#include "hcm_common.h"
#include "../../include/hcmoco_hip.h"

namespace {

using namespace hcm;

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;
constexpr int kMaxCBL = 64;

__global__ void make_keys_kernel(const int* __restrict__ idx, int64_t total, int Q, int m,
                                 unsigned* __restrict__ keys, int* __restrict__ vals) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(e / Q);
    int j = idx[e];
    j = j < 0 ? 0 : (j >= m ? m - 1 : j);                  // like a raw scatter the caller guarantees 0 <= idx < m
    keys[e] = (unsigned)(b * m + j);
    vals[e] = (int)(e - (int64_t)b * Q);
  }
}

// seg[k] = first sorted position whose key is >= k, k = 0 .. B*m (lower bound)
__global__ void seg_kernel(const unsigned* __restrict__ skeys, int64_t total, int nkeys, int* __restrict__ seg) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > nkeys) return;
  int64_t lo = 0, hi = total;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (skeys[mid] < (unsigned)k) lo = mid + 1; else hi = mid;
  }
  seg[k] = (int)lo;
}

struct SortWs {
  unsigned *keys_in, *keys_out;
  int* vals_in;
  void* temp;
  size_t temp_bytes, bytes;
};
SortWs carve_sort(void* ws, int64_t total, int end_bit) {
  SortWs o;
  size_t tb = 0;
  hipcub::DeviceRadixSort::SortPairs((void*)nullptr, tb, (const unsigned*)nullptr, (unsigned*)nullptr,
                                     (const int*)nullptr, (int*)nullptr, (int)total, 0, end_bit, (hipStream_t)0);
  char* base = reinterpret_cast<char*>(ws);
  size_t off = 0;
  auto take = [&](size_t n) { const size_t at = off; off += (n + 255) & ~(size_t)255; return base ? base + at : nullptr; };
  o.keys_in = reinterpret_cast<unsigned*>(take((size_t)total * 4));
  o.keys_out = reinterpret_cast<unsigned*>(take((size_t)total * 4));
  o.vals_in = reinterpret_cast<int*>(take((size_t)total * 4));
  o.temp = take(tb);
  o.temp_bytes = tb;
  o.bytes = off;
  return o;
}
inline int key_bits(int64_t nkeys) {
  int b = 1;
  while (((int64_t)1 << b) < nkeys) ++b;
  return b;
}

// ------------------------------------------------------------------------------------------
// grid (target tiles, channel blocks, B), 1024 threads.  Dynamic LDS: acc [CBL][JT] floats, side [16][2][CBL] floats.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void scatter_sorted_kernel(
    const float* __restrict__ grad_out, const float* __restrict__ coef, const int* __restrict__ order,
    const unsigned* __restrict__ skey, const int* __restrict__ seg, int C, int Qsrc, int Q, int m, int div, int JT,
    int CBL, float* __restrict__ grad_points) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ int wfirst[kWaves], wlast[kWaves];
  float* acc = lds;                                   // [CBL][JT]
  float* side = lds + (size_t)CBL * JT;               // [kWaves][2][CBL]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, j0 = blockIdx.x * JT, c0 = blockIdx.y * CBL;
  const int nj = min(JT, m - j0), nc = min(CBL, C - c0);
  for (int e = tid; e < CBL * JT; e += kThreads) acc[e] = 0.f;
  for (int e = tid; e < kWaves * 2 * CBL; e += kThreads) side[e] = 0.f;
  const int E0 = seg[b * m + j0], E1 = seg[b * m + j0 + nj];        // this tile's piece of the sorted list
  const int total = E1 - E0;
  // equal chunks of whole 64-contribution steps per wave
  const int steps = (total + 63) / 64;
  const int per = (steps + kWaves - 1) / kWaves;
  const int wbeg = E0 + min(wave * per, steps) * 64;
  const int wend = min(E1, E0 + min((wave + 1) * per, steps) * 64);
  const unsigned kbase = (unsigned)(b * m + j0);
  // first / last target of the chunk: may be shared with the neighbouring waves -> side slots
  int jf = -1, jl = -1;
  if (wbeg < wend) {
    jf = (int)(skey[wbeg] - kbase);
    jl = (int)(skey[wend - 1] - kbase);
  }
  if (lane == 0) { wfirst[wave] = jf; wlast[wave] = jl; }
  __syncthreads();
  const float* g = grad_out + ((int64_t)b * C + c0) * Qsrc;
  float* sA = side + (size_t)(wave * 2) * CBL;
  float* sB = sA + CBL;
  for (int e0 = wbeg; e0 < wend; e0 += 64) {
    const int e = e0 + lane;
    const bool valid = e < wend;
    int q = 0, j = -2 - lane;                          // invalid lanes: singleton segments nobody writes
    float w = 0.f;
    if (valid) {
      q = order[e];
      j = (int)(skey[e] - kbase);
      w = coef != nullptr ? coef[(int64_t)b * Q + q] : 1.f;
    }
    const int src = q / div;
    // lanes are sorted by target: lane l continues the segment of lane l - d iff their targets are equal
    bool cont[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      const int d = 1 << s;
      const int jo = __shfl_up(j, d, 64);
      cont[s] = lane >= d && jo == j;
    }
    const int jn = __shfl_down(j, 1, 64);
    const bool tail = valid && (lane == 63 || jn != j);
    const bool toA = tail && j == jf, toB = tail && !toA && j == jl;
    for (int ch = 0; ch < nc; ++ch) {
      float v = valid ? w * g[(int64_t)ch * Qsrc + src] : 0.f;
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        const float o = __shfl_up(v, 1 << s, 64);
        v += cont[s] ? o : 0.f;
      }
      if (tail) {
        float* dst = toA ? sA + ch : (toB ? sB + ch : acc + (size_t)ch * JT + j);
        *dst += v;                                     // one writer per address: this wave, this instruction
      }
    }
  }
  __syncthreads();
  // merge the boundary partials in wave order (= sorted order): one thread per channel
  if (tid < nc) {
    for (int w = 0; w < kWaves; ++w) {
      const int a = wfirst[w], z = wlast[w];
      if (a >= 0) acc[(size_t)tid * JT + a] += side[(size_t)(w * 2) * CBL + tid];
      if (z >= 0 && z != a) acc[(size_t)tid * JT + z] += side[(size_t)(w * 2 + 1) * CBL + tid];
    }
  }
  __syncthreads();
  float* out = grad_points + ((int64_t)b * C + c0) * m + j0;
  for (int e = tid; e < nc * nj; e += kThreads) {
    const int ch = e / nj, j = e - ch * nj;
    out[(int64_t)ch * m + j] = acc[(size_t)ch * JT + j];
  }
}

}  // namespace

extern "C" {

size_t hcm_scatter_sort_workspace_bytes(int B, int Q, int m) {
  if (B <= 0 || Q <= 0 || m <= 0 || (int64_t)B * Q >= ((int64_t)1 << 31) || (int64_t)B * m >= ((int64_t)1 << 31)) return 0;
  return carve_sort(nullptr, (int64_t)B * Q, key_bits((int64_t)B * m)).bytes;
}

int hcm_scatter_sort(const int* idx, int B, int Q, int m, int* order, int* sorted_key, int* seg, void* workspace,
                     size_t workspace_bytes, hcm_stream_t stream) {
  if (B <= 0 || Q <= 0 || m <= 0 || (int64_t)B * Q >= ((int64_t)1 << 31) || (int64_t)B * m >= ((int64_t)1 << 31))
    return (int)hipErrorInvalidValue;
  const int64_t total = (int64_t)B * Q;
  const int bits = key_bits((int64_t)B * m);
  SortWs ws = carve_sort(workspace, total, bits);
  if (workspace == nullptr || workspace_bytes < ws.bytes) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  make_keys_kernel<<<blocks, 256, 0, st>>>(idx, total, Q, m, ws.keys_in, ws.vals_in);
  HCM_CHECK_LAUNCH();
  size_t tb = ws.temp_bytes;
  hipError_t e = hipcub::DeviceRadixSort::SortPairs(ws.temp, tb, (const unsigned*)ws.keys_in,
                                                    reinterpret_cast<unsigned*>(sorted_key), (const int*)ws.vals_in, order,
                                                    (int)total, 0, bits, st);
  if (e != hipSuccess) return (int)e;
  const int nkeys = B * m;
  seg_kernel<<<(nkeys + 1 + 255) / 256, 256, 0, st>>>(reinterpret_cast<const unsigned*>(sorted_key), total, nkeys, seg);
  HCM_CHECK_LAUNCH();
  return 0;
}

int hcm_scatter_add_sorted(const float* grad_out, const float* coef, const int* order, const int* sorted_key,
                           const int* seg, int B, int C, int Qsrc, int Q, int m, int div, float* grad_points,
                           hcm_stream_t stream) {
  if (B <= 0 || C <= 0 || Q <= 0 || m <= 0 || div <= 0 || Qsrc <= 0) return (int)hipErrorInvalidValue;
  // tile of targets x block of channels: at most 28 K accumulators (112 KB of LDS)
  int JT = m < 512 ? m : 512;
  int CBL = 28672 / JT;
  if (CBL > kMaxCBL) CBL = kMaxCBL;
  if (CBL > C) CBL = C;
  // keep the device busy: at least ~512 workgroups when the channels allow it
  while (CBL > 8 && (long long)B * ((m + JT - 1) / JT) * ((C + CBL - 1) / CBL) < 512) CBL = (CBL + 1) / 2;
  const size_t ldsb = ((size_t)CBL * JT + (size_t)kWaves * 2 * CBL) * sizeof(float);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(scatter_sorted_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  if (e != hipSuccess) return (int)e;
  dim3 grid((m + JT - 1) / JT, (C + CBL - 1) / CBL, B);
  scatter_sorted_kernel<<<grid, kThreads, ldsb, (hipStream_t)stream>>>(grad_out, coef, order,
                                                                      reinterpret_cast<const unsigned*>(sorted_key), seg, C,
                                                                      Qsrc, Q, m, div, JT, CBL, grad_points);
  HCM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
