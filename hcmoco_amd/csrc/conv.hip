// Direct 3x3 convolution (stride 1, pad 1, fp32 NCHW) for the two high-resolution HRNet branches, gfx950.
//
//   forward        Y[n][k][y][x]  = sum over c, r, s of  Wt[k][c][r][s] * X[n][c][y+r-1][x+s-1]
//   data gradient  dX[n][c][y][x] = sum over k, r, s of  Wt[k][c][2-r][2-s] * dY[n][k][y+r-1][x+s-1]
//
// (official_hrnet.py:40-70 BasicBlock: 18 channels on 64x64 and 36 on 32x32 at B=32 -- 256 + 256 of the 852 3x3
// stride-1 convolution launches of a training step.)  Each is 0.76 GFLOP over a 9.4 / 4.7 MB map: small enough
// that a general library kernel spends its time on its own set-up (MIOpen's Winograd assembly: 25 / 21 us).
//
// One workgroup = one image x a band of RB rows, 4 waves.  The band of X with its halo, every input channel,
// sits in LDS ([4*CG planes][RB+2][WT+8], interior float4-aligned, zero planes for the padded channels); so do
// the weights, already in MFMA operand order.  As a GEMM: M = output channels (MT tiles of 16), N = 16 pixels
// of a row, reduction over (tap, channel) in steps of 4 channels of one tap -- so the B operand of a step is ONE
// ds_read_b32 at a compile-time offset from a per-lane base (lane = channel-in-group x pixel), no address math.
// A wave owns one row of the band: WT/16 pixel tiles x MT channel tiles of accumulators, and walks the 9*CG steps
// once: (MT + WT/16) LDS reads per MT*WT/16 v_mfma_f32_16x16x4_f32.  The data gradient is the same kernel with the weights
// read transposed and flipped when they are put into operand order.
#include <cstdlib>

#include "hcm_common.h"
#include "../../include/hcmoco_hip.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));


// CG: groups of 4 input channels (padded), MT: tiles of 16 output channels on the matrix cores, WT: map width,
// RB: rows per workgroup = waves per workgroup (one row per wave), LO: output channels beyond 16*MT computed on the vector ALU
// (0, or the capacity 2 * 64 / WT: two per lane), FLIP: data gradient.
//
// Why LO: 18 = 16 + 2 and 36 = 32 + 4.  A second / third MFMA tile for 2 / 4 channels is 44 % / 25 % of the
// matrix work spent on zeros.  A wave owns one map row; its 64 lanes are (pixel, channel pair) for the leftover
// channels and do them as plain FMAs -- x from the same LDS tile (lane = pixel: conflict-free), the two weights
// of the pair from a small LDS table (broadcast read) -- in the shadow of the MFMAs, which run 32 cycles each.
// STATS (forward only): the epilogue also leaves this workgroup's per-channel sums of (y - k) and (y - k)^2 in
// stat_part[2*blockIdx.x + {0,1}][K] -- the statistics pass of the BatchNorm that follows every one of these
// convolutions (hcm_bn_act_forward_pre), taken from the accumulators instead of re-reading y.  k[c] = stat_shift[c]
// (the BatchNorm's running mean: close to the batch mean, so the variance is not a difference of two large sums) or 0;
// workgroup 0 leaves the k it used in row 2*gridDim.x, where the BatchNorm kernel reads it back (it updates the
// running mean itself, so it must not read k from there).
template <int CG, int MT, int WT, int RB, int LO, bool FLIP, bool STATS>
__global__ __launch_bounds__(64 * RB) void conv3x3_mfma_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                                float* __restrict__ y, int C, int K, int H,
                                                                float* __restrict__ stat_part,
                                                                const float* __restrict__ stat_shift) {
  constexpr int LW = WT + 8;                       // [3 unused][left halo][WT][right halo][3 unused]
  constexpr int LH = RB + 2;
  constexpr int PLANE = LH * LW;
  constexpr int STEPS = 9 * CG;
  constexpr int P = WT / 16;                       // pixel tiles of a row
  constexpr int kThreads = 64 * RB;               // one row per wave
  static_assert(LO == 0 || LO == 2 * (64 / WT), "leftover capacity: two channels per lane");
  extern __shared__ float lds[];
  float* Xs = lds;                                 // [4*CG][LH][LW]
  float* As = lds + 4 * CG * PLANE;                // [STEPS][MT][64]
  float* Ls = As + STEPS * MT * 64;                // [4*CG][9][LO]
  const int bands = H / RB;
  const int img = blockIdx.x / bands, y0 = (blockIdx.x % bands) * RB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int np = lane & 15, kq = lane >> 4;

  // ---- weights -> operand order: As[step = tap*CG + g][mt][lane] = W(out = 16mt + np, in = 4g + kq, tap) ----
  // forward: W(out, in, tap) = wt[out][in][tap]; data gradient: out runs over C, in over K, wt[in][out][8 - tap]
  // Per thread and trip one (out, in) pair of the PADDED operand: its nine taps are contiguous in memory, padded
  // pairs are written as zeros (no separate clear).
  // Every global load of the prologue (weights and the band of the input) is issued before the first LDS store:
  // one memory round trip instead of one per loop trip.
  const int n_out = FLIP ? C : K, n_in = FLIP ? K : C;
  // weights: a wave's 64 lanes are the 64 entries (4 input channels x 16 output channels) of ONE operand row, so
  // each of its nine LDS stores fills a row (conflict-free) and its global reads fall into 16 short runs
  constexpr int WROWS = CG * MT * 64, WTRIPS = (WROWS + kThreads - 1) / kThreads;
  constexpr int LPAIRS = 4 * CG * LO, LTRIPS = (LPAIRS + kThreads - 1) / kThreads;
  constexpr int Q = WT / 4;                        // float4 per row
  constexpr int XQ = 4 * CG * LH * Q, XTRIPS = (XQ + kThreads - 1) / kThreads;
  float wv[WTRIPS][9], lv[LTRIPS > 0 ? LTRIPS : 1][9];
  float4 xv4[XTRIPS];
  auto wsrc = [&](int o, int i) { return wt + (FLIP ? ((size_t)i * C + o) : ((size_t)o * C + i)) * 9; };
#pragma unroll
  for (int t = 0; t < WTRIPS; ++t) {
    const int e = tid + t * kThreads;
    const int i = 4 * ((e >> 6) / MT) + (e & 3), o = 16 * ((e >> 6) % MT) + ((e >> 2) & 15);
    const bool real = e < WROWS && o < n_out && i < n_in;
    const float* src = wsrc(o, i);
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[t][k] = real ? src[k] : 0.f;
  }
#pragma unroll
  for (int t = 0; t < LTRIPS; ++t) {
    const int e = tid + t * kThreads;
    const int o = 16 * MT + e % (LO ? LO : 1), i = e / (LO ? LO : 1);
    const bool real = e < LPAIRS && o < n_out && i < n_in;
    const float* src = wsrc(o, i);
#pragma unroll
    for (int k = 0; k < 9; ++k) lv[t][k] = real ? src[k] : 0.f;
  }
  const float* xin = x + (size_t)img * n_in * H * WT;
#pragma unroll
  for (int t = 0; t < XTRIPS; ++t) {
    const int e = tid + t * kThreads;
    const int q = e % Q, r = (e / Q) % LH, c = e / (Q * LH);
    const int yy = y0 - 1 + r;
    xv4[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < XQ && c < n_in && yy >= 0 && yy < H) xv4[t] = *reinterpret_cast<const float4*>(xin + ((size_t)c * H + yy) * WT + 4 * q);
  }
#pragma unroll
  for (int t = 0; t < WTRIPS; ++t) {
    const int e = tid + t * kThreads;              // As[tap*CG + g][mt][kq*16 + np] with (g, mt) = e >> 6
    if (e < WROWS) {
      float* dst = As + (e >> 6) * 64 + (e & 3) * 16 + ((e >> 2) & 15);
#pragma unroll
      for (int k = 0; k < 9; ++k) dst[(FLIP ? 8 - k : k) * CG * MT * 64] = wv[t][k];
    }
  }
#pragma unroll
  for (int t = 0; t < LTRIPS; ++t) {
    const int e = tid + t * kThreads;
    if (e < LPAIRS) {
      float* dst = Ls + (e / (LO ? LO : 1)) * 9 * LO + e % (LO ? LO : 1);
#pragma unroll
      for (int k = 0; k < 9; ++k) dst[(FLIP ? 8 - k : k) * LO] = lv[t][k];
    }
  }
#pragma unroll
  for (int t = 0; t < XTRIPS; ++t) {
    const int e = tid + t * kThreads;
    const int q = e % Q, r = (e / Q) % LH, c = e / (Q * LH);
    if (e < XQ) *reinterpret_cast<float4*>(Xs + (c * LH + r) * LW + 4 + 4 * q) = xv4[t];
  }
  for (int e = tid; e < 4 * CG * LH * 2; e += kThreads) {            // halo columns
    const int side = e & 1, r = (e >> 1) % LH, c = (e >> 1) / LH;
    Xs[(c * LH + r) * LW + (side ? 4 + WT : 3)] = 0.f;
  }
  __syncthreads();

  // ---- main loop: row `wave` of the band ----
  v4f acc[MT][P];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int p = 0; p < P; ++p) acc[m][p] = v4f{0.f, 0.f, 0.f, 0.f};
  float lo0 = 0.f, lo1 = 0.f;                      // leftover channels 16*MT + 2*pair + {0, 1} at pixel lpx
  const int lpx = lane % WT, pair = lane / WT;
  const float* Bx = Xs + kq * PLANE + wave * LW + 3 + np;      // MFMA B operand: channel kq of a group, pixel np of tile 0
  const float* Lx = Xs + wave * LW + 3 + lpx;                  // leftover: channel 0, this lane's pixel
  const float* Lw = Ls + 2 * pair;
  const float* Aw = As + lane;
  // Operands of step s+1 are read while the MFMAs of step s run; the scheduling barrier keeps the compiler from
  // hoisting every LDS read of the unrolled loop to the top (it did: 512 VGPRs and scratch spills).
  float a[2][MT], b[2][P], xl[2][4];
  float2 wl[2][4];
  auto fetch = [&](int step, float (&av)[MT], float (&bv)[P], float (&xv)[4], float2 (&wv)[4]) {
    const int tap = step / CG, g = step % CG;
    const int off = 4 * g * PLANE + (tap / 3) * LW + (tap % 3);
#pragma unroll
    for (int m = 0; m < MT; ++m) av[m] = Aw[(step * MT + m) * 64];
#pragma unroll
    for (int p = 0; p < P; ++p) bv[p] = Bx[off + 16 * p];
    if (LO) {
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        xv[c4] = Lx[off + c4 * PLANE];
        wv[c4] = *reinterpret_cast<const float2*>(Lw + ((4 * g + c4) * 9 + tap) * LO);
      }
    }
  };
  fetch(0, a[0], b[0], xl[0], wl[0]);
#pragma unroll
  for (int step = 0; step < STEPS; ++step) {
    const int cur = step & 1, nxt = cur ^ 1;
    if (step + 1 < STEPS) fetch(step + 1, a[nxt], b[nxt], xl[nxt], wl[nxt]);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int p = 0; p < P; ++p) acc[m][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][m], b[cur][p], acc[m][p], 0, 0, 0);
    if (LO) {
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        lo0 = fmaf(wl[cur][c4].x, xl[cur][c4], lo0);
        lo1 = fmaf(wl[cur][c4].y, xl[cur][c4], lo1);
      }
      asm volatile("" : "+v"(lo0), "+v"(lo1));      // pins the FMAs to this step (pure arithmetic is not ordered by the barrier)
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- D[4*kq + r][np] of tile (m, p) -> Y[img][16m + 4kq + r][y0 + wave][16 p + np] ----
  float* yout = y + (size_t)img * n_out * H * WT + (size_t)(y0 + wave) * WT;
#pragma unroll
  for (int p = 0; p < P; ++p)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = 16 * m + 4 * kq + r;
        if (o < n_out) yout[(size_t)o * H * WT + p * 16 + np] = acc[m][p][r];
      }
  if (LO) {
    const int o = 16 * MT + 2 * pair;
    if (o < n_out) yout[(size_t)o * H * WT + lpx] = lo0;
    if (o + 1 < n_out) yout[(size_t)(o + 1) * H * WT + lpx] = lo1;
  }
  if (STATS) {
    // this wave's row: sums over its WT pixels per channel; then the RB rows through LDS in wave order
    float* Ss = Ls + 4 * CG * 9 * LO;                 // [RB][2][16*MT + LO]
    constexpr int NC = 16 * MT + (LO ? LO : 0);
    float* mine = Ss + wave * 2 * NC;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s1 = 0.f, s2 = 0.f;
        const int oc = 16 * m + 4 * kq + r;
        const float k = (stat_shift && oc < n_out) ? stat_shift[oc] : 0.f;
#pragma unroll
        for (int p = 0; p < P; ++p) { const float v = acc[m][p][r] - k; s1 += v; s2 = fmaf(v, v, s2); }
        s1 = hcm::row16_sum(s1);                      // the 16 pixels of a tile sit in one DPP row (same kq)
        s2 = hcm::row16_sum(s2);
        if (np == 0) { mine[16 * m + 4 * kq + r] = s1; mine[NC + 16 * m + 4 * kq + r] = s2; }
      }
    if (LO) {
      const int oc = 16 * MT + 2 * pair;
      float a0 = lo0 - ((stat_shift && oc < n_out) ? stat_shift[oc] : 0.f);
      float a1 = lo1 - ((stat_shift && oc + 1 < n_out) ? stat_shift[oc + 1] : 0.f);
      float b0 = a0 * a0, b1 = a1 * a1;
      // lanes of one channel pair: all 64 (WT = 64) or one 32-lane half (WT = 32)
      a0 = hcm::row16_sum(a0); a1 = hcm::row16_sum(a1); b0 = hcm::row16_sum(b0); b1 = hcm::row16_sum(b1);
      a0 += __shfl_xor(a0, 16, 64); a1 += __shfl_xor(a1, 16, 64); b0 += __shfl_xor(b0, 16, 64); b1 += __shfl_xor(b1, 16, 64);
      if (WT == 64) {
        a0 += __shfl_xor(a0, 32, 64); a1 += __shfl_xor(a1, 32, 64); b0 += __shfl_xor(b0, 32, 64); b1 += __shfl_xor(b1, 32, 64);
      }
      if (lpx == 0) {
        mine[16 * MT + 2 * pair] = a0; mine[16 * MT + 2 * pair + 1] = a1;
        mine[NC + 16 * MT + 2 * pair] = b0; mine[NC + 16 * MT + 2 * pair + 1] = b1;
      }
    }
    __syncthreads();
    for (int e = tid; e < 2 * NC; e += kThreads) {
      const int c = e % NC, which = e / NC;
      if (c < n_out) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < RB; ++w) v += Ss[w * 2 * NC + e];
        stat_part[(size_t)(2 * blockIdx.x + which) * n_out + c] = v;
      }
    }
    if (blockIdx.x == 0)
      for (int c = tid; c < n_out; c += kThreads) stat_part[(size_t)2 * gridDim.x * n_out + c] = stat_shift ? stat_shift[c] : 0.f;
  }
}

template <int CG, int MT, int WT, int RB, int LO>
int launch(const float* x, const float* w, float* y, int N, int C, int K, int H, bool flip, hipStream_t st, float* stat_part = nullptr,
           const float* stat_shift = nullptr) {
  constexpr size_t lds = ((size_t)4 * CG * (RB + 2) * (WT + 8) + (size_t)9 * CG * MT * 64 + (size_t)4 * CG * 9 * LO +
                          (size_t)RB * 2 * (16 * MT + LO)) * sizeof(float);
  const dim3 grid(N * (H / RB));
  if (stat_part != nullptr && !flip) {
    static const hipError_t attr = hipFuncSetAttribute(
        reinterpret_cast<const void*>(&conv3x3_mfma_kernel<CG, MT, WT, RB, LO, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) return (int)attr;
    conv3x3_mfma_kernel<CG, MT, WT, RB, LO, false, true><<<grid, 64 * RB, lds, st>>>(x, w, y, C, K, H, stat_part, stat_shift);
    HCM_CHECK_LAUNCH();
    return 0;
  }
  if (flip) {
    static const hipError_t attr = hipFuncSetAttribute(
        reinterpret_cast<const void*>(&conv3x3_mfma_kernel<CG, MT, WT, RB, LO, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) return (int)attr;
    conv3x3_mfma_kernel<CG, MT, WT, RB, LO, true, false><<<grid, 64 * RB, lds, st>>>(x, w, y, C, K, H, nullptr, nullptr);
  } else {
    static const hipError_t attr = hipFuncSetAttribute(
        reinterpret_cast<const void*>(&conv3x3_mfma_kernel<CG, MT, WT, RB, LO, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) return (int)attr;
    conv3x3_mfma_kernel<CG, MT, WT, RB, LO, false, false><<<grid, 64 * RB, lds, st>>>(x, w, y, C, K, H, nullptr, nullptr);
  }
  HCM_CHECK_LAUNCH();
  return 0;
}

// shapes with a kernel instance: C == K (BasicBlock), 64- or 32-wide maps
int dispatch(const float* x, const float* w, float* y, int N, int C, int K, int H, int W, bool flip, hipStream_t st,
             float* stat_part = nullptr, const float* stat_shift = nullptr) {
  if (N <= 0 || C != K || !x || !w || !y || H % 4 != 0) return (int)hipErrorInvalidValue;
  // 4-row bands (4 waves): 8-row bands with 8 waves halve the workgroup count and the weight re-reads, but the
  // kernel gets slower (36ch: 14.6 -> 19.5 us) and so does the step (654 -> 642 samples/s)
  if (W == 64 && C > 16 && C <= 18) return launch<5, 1, 64, 4, 2>(x, w, y, N, C, K, H, flip, st, stat_part, stat_shift);
  if (W == 64 && C > 16 && C <= 20) return launch<5, 2, 64, 4, 0>(x, w, y, N, C, K, H, flip, st, stat_part, stat_shift);
  // (a padded 32-channel instance for HRNet-w32's first branch was measured: 436 vs 441 samples/s with MIOpen -- not kept)
  if (W == 32 && C > 32 && C <= 36) return launch<9, 2, 32, 4, 4>(x, w, y, N, C, K, H, flip, st, stat_part, stat_shift);
  if (W == 32 && C > 32 && C <= 36) return launch<9, 3, 32, 4, 0>(x, w, y, N, C, K, H, flip, st, stat_part, stat_shift);
  return (int)hipErrorInvalidValue;
}

}  // namespace

extern "C" {

int hcm_conv3x3_supported(int C, int K, int H, int W) {
  if (C != K) return 0;
  if (W == 64 && C > 16 && C <= 20 && H % 4 == 0) return 1;
  if (W == 32 && C > 32 && C <= 36 && H % 4 == 0) return 1;
  return 0;
}

int hcm_conv3x3_forward(const float* x, const float* w, float* y, int N, int C, int K, int H, int W, hcm_stream_t stream) {
  return dispatch(x, w, y, N, C, K, H, W, false, (hipStream_t)stream);
}

int hcm_conv3x3_stats_slots(int N, int H) { return (N > 0 && H > 0 && H % 4 == 0) ? N * (H / 4) : 0; }

int hcm_conv3x3_forward_stats(const float* x, const float* w, float* y, int N, int C, int K, int H, int W, const float* shift,
                              float* partial_sums, hcm_stream_t stream) {
  if (!partial_sums) return (int)hipErrorInvalidValue;
  return dispatch(x, w, y, N, C, K, H, W, false, (hipStream_t)stream, partial_sums, shift);
}

int hcm_conv3x3_backward_data(const float* dy, const float* w, float* dx, int N, int C, int K, int H, int W,
                              hcm_stream_t stream) {
  return dispatch(dy, w, dx, N, C, K, H, W, true, (hipStream_t)stream);
}

}  // extern "C"
