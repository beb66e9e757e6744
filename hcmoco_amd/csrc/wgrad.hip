// Weight gradients of the encoders' convolutions (3x3 at stride 1 or 2 with pad 1, and 1x1), gfx950.
//
//   dW[k][c][r][s] = sum over n, y, x of  dY[n][k][y][x] * X[n][c][y+r-1][x+s-1]        (fp32, NCHW)
//
// 416 of the 620 convolutions of an HRNet pair have this shape (official_hrnet.py:40-70 BasicBlock),
// with C = K in {18,36,72,144} on 64^2..8^2 maps.  MIOpen's best solution for them is an NHWC
// implicit GEMM behind three layout transposes and a clear: five launches, 26-47 us, 24 ms of GPU time
// per step and the largest block of launches on each encoder's stream.
//
// As a GEMM the problem is M = K output channels, N = C*9 (input channel, tap) pairs, reduction over
// the N*H*W pixels: a tiny output with a long reduction.  Here:
//   wgrad3x3_mfma_kernel: a workgroup owns a run of (image, row-block) units and a range of N tiles;
//       it stages the unit's dY rows [K][rb][W] and X rows with halo [c][rb+2][W+2] in LDS (coalesced
//       NCHW reads, no layout transposes) and runs v_mfma_f32_16x16x4_f32 over 4 pixels at a time:
//       A[k][p] = dY, B[p][(c,tap)] = X shifted by the tap -- the shift is just an LDS address offset.
//       Waves split the N tiles (WN) and the rows of the unit (WP); the WP partial tiles are summed
//       through LDS, one partial [K][C*9] per workgroup goes to the workspace;
//   wgrad3x3_reduce_kernel: sums the per-workgroup partials in fixed order (deterministic, no atomics).
// Two launches, no layout transposes, deterministic.  Measured (tools/bench_wgrad.py, N=32, C=K):
// 18ch@64^2 25 us (MIOpen 49), 36ch@32^2 22 (38), 32ch@64^2 39 (58), 72ch@16^2 28 (31), 144ch@8^2 42 (28):
// the encoder runtime uses it for layers of at most 48 channels (+3.7 % on the step), MIOpen for the rest.
// What it took (PMC, 18ch@64^2, per launch): 0.79 M MFMA instructions against 12.7 M VALU in the first
// version -- integer divisions by run-time map sizes in the staging loops.  Map width and rows per unit
// as template constants and float4 staging of an aligned X tile brought VALU to 3.8 M; keeping only the
// K real dY rows (+ one shared zero row for the padded channels) in LDS lets four workgroups share a CU.
// A first attempt fed dY through the scalar cache into packed VALU FMAs (one lane per (c,tap), K
// accumulators): correct, but every 4 pixels waited on ~9 scalar-load round trips -- 111 us at 18ch@64^2.
#include <cstdio>
#include <cstdlib>

#include "hcm_common.h"
#include "../../include/hcmoco_hip.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

struct WgradGeo {
  int N, C, K, H, W;
  int mt, nt;             // 16-row tiles over K, 16-column tiles over C*9
  int ntw, wn, wp;        // N tiles per wave, waves across N, waves across the rows of a unit
  int ngroups;            // workgroups across N (each covers wn*ntw tiles)
  int rb, rblocks;        // rows per unit, units per image
  int units, chunks, per; // total units, workgroups along the reduction, units per workgroup
  int cmax;               // input channels staged per workgroup
  int dstride;            // floats between consecutive k rows of the dY tile (padded against bank conflicts)
  int threads;
  int taps;               // 9 (3x3, pad 1) or 1 (1x1, pad 0: only the centre tap of the same tile layout)
  int st;                 // convolution stride (1 or 2): x is [N,C,st*H,st*W], dy [N,K,H,W]
  size_t lds_bytes;
};

// WT, RBT: map width and rows per unit as compile-time constants (0 = take them from g).  With them
// every index computation of the staging loops is a division by a constant (multiply-shift); with
// runtime values the integer divisions were two thirds of the kernel's VALU instructions (PMC:
// 12.7 M VALU vs 0.79 M MFMA per launch at 18ch@64x64).
template <int MT, int NTW, int WT, int RBT, int ST>
__global__ __launch_bounds__(512) void wgrad3x3_mfma_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ partial, WgradGeo gin) {
  WgradGeo g = gin;
  if (WT) { g.W = WT; g.rb = RBT; g.dstride = RBT * WT + 4; g.st = ST; }
  const int Wx = g.st * g.W, Hx = g.st * g.H;          // input map
  extern __shared__ float lds[];
  const int chunk = blockIdx.x, ng = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int np = lane & 15, kq = lane >> 4;
  const int wn = wave % g.wn, wp = wave / g.wn;
  // X tile row: [3 unused][left halo][W interior][right halo][3 unused] -> the interior is float4-aligned
  const int LW = Wx + 8, LH = g.st * g.rb + 2, plane = LH * LW;
  const int n9 = g.C * g.taps;
  const int tile0 = (ng * g.wn + wn) * NTW;            // first N tile of this wave
  const int c_lo = (ng * g.wn * NTW * 16) / g.taps;    // first input channel this workgroup touches
  float* Xs = lds;                                     // [cmax][rb+2][W+2]
  float* Ds = lds + g.cmax * plane;                    // [K + 1][dstride]
  // per-lane LDS offset of the (c, tap) column of every N tile
  int boff[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    int j = (tile0 + t) * 16 + np;
    if (j >= n9) j = n9 - 1;                            // padded columns: any valid address, never stored
    const int c = j / g.taps, tap = g.taps == 9 ? j - c * 9 : 4, r = tap / 3, s = tap - r * 3;
    boff[t] = (c - c_lo) * plane + r * LW + s + 3 + g.st * kq;
  }
  int aoff[MT];                                        // dY tile row of output channel 16m+np; padded channels read the zero row K
#pragma unroll
  for (int m = 0; m < MT; ++m) aoff[m] = min(m * 16 + np, g.K) * g.dstride;
  v4f acc[MT][NTW];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[m][t] = (v4f){0.f, 0.f, 0.f, 0.f};

  const size_t HW = (size_t)g.H * g.W, HWx = (size_t)Hx * Wx;
  const int u_beg = chunk * g.per, u_end = min(g.units, u_beg + g.per);
  for (int i = threadIdx.x; i < g.cmax * LH * 2; i += blockDim.x)      // the halo columns stay zero for every unit
    Xs[(i >> 1) * LW + ((i & 1) ? Wx + 4 : 3)] = 0.f;
  for (int i = threadIdx.x; i < g.dstride; i += blockDim.x) Ds[g.K * g.dstride + i] = 0.f;   // and so does dY row K
  for (int u = u_beg; u < u_end; ++u) {
    const int n = u / g.rblocks, y0 = (u - n * g.rblocks) * g.rb;
    const int rows = min(g.rb, g.H - y0);
    __syncthreads();
    {                                                                    // X rows y0-1 .. y0+rb, float4 per thread
      const int w4 = Wx >> 2, per_c = LH * w4;
      for (int i = threadIdx.x; i < g.cmax * per_c; i += blockDim.x) {
        const int cc = i / per_c, rem = i - cc * per_c, ly = rem / w4, q4 = rem - ly * w4;
        const int gy = g.st * y0 + ly - 1, c = c_lo + cc;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gy >= 0 && gy < Hx && c < g.C)
          v = *reinterpret_cast<const float4*>(x + ((size_t)n * g.C + c) * HWx + (size_t)gy * Wx + q4 * 4);
        *reinterpret_cast<float4*>(Xs + cc * plane + ly * LW + 4 + q4 * 4) = v;
      }
    }
    const int rowf = g.rb * g.W;                                         // dY rows [k < K][rb][W]; row K stays zero
    for (int i = threadIdx.x * 4; i < g.K * rowf; i += blockDim.x * 4) {
      const int k = i / rowf, rem = i - k * rowf, yy = rem / g.W;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (yy < rows)
        v = *reinterpret_cast<const float4*>(dy + ((size_t)n * g.K + k) * HW + (size_t)y0 * g.W + rem);
      *reinterpret_cast<float4*>(Ds + k * g.dstride + rem) = v;
    }
    __syncthreads();
    for (int yy = wp; yy < rows; yy += g.wp) {
      const float* arow = Ds + yy * g.W + kq;
      const float* brow = Xs + g.st * yy * LW;
      for (int xx = 0; xx < g.W; xx += 4) {
        float a[MT], b[NTW];
#pragma unroll
        for (int m = 0; m < MT; ++m) a[m] = arow[aoff[m] + xx];
#pragma unroll
        for (int t = 0; t < NTW; ++t) b[t] = brow[boff[t] + g.st * xx];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int t = 0; t < NTW; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[t], acc[m][t], 0, 0, 0);
      }
    }
  }
  // sum the row-split waves through LDS: slot [wp-1][wn][m][t][lane] of float4
  if (g.wp > 1) {
    __syncthreads();
    v4f* red = reinterpret_cast<v4f*>(lds);
    if (wp > 0) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NTW; ++t) red[((((wp - 1) * g.wn + wn) * MT + m) * NTW + t) * 64 + lane] = acc[m][t];
    }
    __syncthreads();
    if (wp == 0) {
      for (int p = 0; p < g.wp - 1; ++p) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int t = 0; t < NTW; ++t) acc[m][t] += red[(((p * g.wn + wn) * MT + m) * NTW + t) * 64 + lane];
      }
    }
  }
  if (wp == 0) {                                       // acc[m][t][q] = dW[k = 16m + 4kq + q][j = 16(tile0+t) + np]
    float* out = partial + (size_t)chunk * g.K * n9;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        const int j = (tile0 + t) * 16 + np;
        if (j < n9 && tile0 + t < g.nt) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int k = m * 16 + kq * 4 + q;
            if (k < g.K) out[(size_t)k * n9 + j] = acc[m][t][q];
          }
        }
      }
  }
}

// Fixed-order sum of the per-workgroup partials.  64 consecutive outputs x 16 chunk lanes per
// workgroup: lane q sums chunks q, q+16, ... (eight loads in flight), the 16 lane sums are added in
// lane order through LDS.
__global__ __launch_bounds__(1024) void wgrad3x3_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                               int total, int chunks) {
  __shared__ float sh[16][64];
  const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + o;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (i < total) {
    int j = q;
    for (; j + 7 * 16 < chunks; j += 8 * 16) {
#pragma unroll
      for (int u = 0; u < 8; ++u) s[u] += partial[(size_t)(j + u * 16) * total + i];
    }
    for (int u = 0; j < chunks; j += 16, ++u) s[u & 7] += partial[(size_t)j * total + i];
  }
  sh[q][o] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (q == 0 && i < total) {
    float r = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) r += sh[u][o];
    dw[i] = r;
  }
}

// The same fixed-order sum for MANY layers in one launch: the encoder runtime defers the reductions of a stretch of
// the backward walk (up to kReduceBatch layers, their partials parked in one arena) and runs them together -- 398
// launches of 5.6 us per step were a serial 1.5 ms on each encoder stream for 3 MB of work apiece.
constexpr int kReduceBatch = 64;
struct ReduceBatch {
  const float* partial[kReduceBatch];
  float* dw[kReduceBatch];
  int total[kReduceBatch];
  int chunks[kReduceBatch];
  int block0[kReduceBatch + 1];     // first workgroup of each layer; block0[n] = grid size
  int n;
};

__global__ __launch_bounds__(1024) void wgrad_reduce_batch_kernel(ReduceBatch b) {
  __shared__ float sh[16][64];
  int d = 0;
  while (d + 1 < b.n && b.block0[d + 1] <= (int)blockIdx.x) ++d;
  const float* __restrict__ partial = b.partial[d];
  const int total = b.total[d], chunks = b.chunks[d];
  const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int i = ((int)blockIdx.x - b.block0[d]) * 64 + o;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (i < total) {
    int j = q;
    for (; j + 7 * 16 < chunks; j += 8 * 16) {
#pragma unroll
      for (int u = 0; u < 8; ++u) s[u] += partial[(size_t)(j + u * 16) * total + i];
    }
    for (int u = 0; j < chunks; j += 16, ++u) s[u & 7] += partial[(size_t)j * total + i];
  }
  sh[q][o] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (q == 0 && i < total) {
    float r = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) r += sh[u][o];
    b.dw[d][i] = r;
  }
}

// (MT, NTW) instantiations: K <= 16*MT, at most ~24 tiles (96 accumulator registers) per wave
// (MT, NTW, waves across N) per K: few N tiles per workgroup where K is small (more workgroups across
// N, fewer and smaller partials), at most ~24 tiles (96 accumulator registers) per wave
bool pick_tiles_default(int mt, int& ntw, int& wn_max);
bool pick_tiles(int mt, int& ntw, int& wn_max) { return pick_tiles_default(mt, ntw, wn_max); }
bool pick_tiles_default(int mt, int& ntw, int& wn_max) {
  wn_max = 4;
  switch (mt) {
    case 1: ntw = 12; return true;
    case 2: ntw = 3; wn_max = 1; return true;
    case 3: ntw = 3; wn_max = 1; return true;
    case 4: ntw = 6; return true;
    case 5: ntw = 2; wn_max = 1; return true;
    case 8: ntw = 3; return true;
    case 9: ntw = 2; return true;
    case 16: ntw = 1; return true;
    default: return false;
  }
}

bool make_wgeo(int N, int C, int K, int H, int W, int taps, int st, WgradGeo& g) {
  g.taps = taps; g.st = st;
  if (st != 1 && st != 2) return false;
  if (N <= 0 || C <= 0 || K <= 0 || H <= 0 || W <= 0 || (W & 3) != 0) return false;
  g.N = N; g.C = C; g.K = K; g.H = H; g.W = W;
  g.mt = (K + 15) / 16;
  g.nt = (C * taps + 15) / 16;
  int wn_max;
  if (!pick_tiles(g.mt, g.ntw, wn_max)) return false;
  int wn = (g.nt + g.ntw - 1) / g.ntw;
  if (wn > wn_max) wn = wn_max;
  // fewer waves across N (= fewer input channels per workgroup) until one unit row of X and dY fits LDS: the 256 -> 64
  // 1x1 convolutions of layer1 on 64-wide maps staged 194 channels per workgroup (167 KB) and fell back to MIOpen's
  // atomic kernels -- the last six layers of an HRNet pair that were not run-to-run reproducible (r03)
  for (;; --wn) {
    g.cmax = (wn * g.ntw * 16 + taps - 1) / taps + 2;
    if (g.cmax > C) g.cmax = C;
    const size_t min_bytes = ((size_t)g.cmax * (st + 2) * (st * W + 8) + (size_t)(K + 1) * (W + 4)) * 4;
    if (min_bytes <= 150 * 1024 || wn == 1) break;
  }
  g.wn = wn;
  g.ngroups = (g.nt + wn * g.ntw - 1) / (wn * g.ntw);
  // rows per unit: X tile + dY tile within ~48 KB of LDS
  int rb = H;
  for (;;) {
    g.dstride = rb * W + 4;
    size_t bytes = ((size_t)g.cmax * (st * rb + 2) * (st * W + 8) + (size_t)(K + 1) * g.dstride) * 4;
    constexpr size_t cap = 48 * 1024;
    if (bytes <= cap || rb == 1) break;
    rb = (rb + 1) / 2;
  }
  g.rb = rb;
  g.rblocks = (H + rb - 1) / rb;
  g.units = N * g.rblocks;
  int wp = 8 / wn;
  if (wp > rb) wp = rb;
  if (wp < 1) wp = 1;
  g.wp = wp;
  g.threads = 64 * wn * wp;
  size_t tile = ((size_t)g.cmax * (st * rb + 2) * (st * W + 8) + (size_t)(K + 1) * g.dstride) * 4;
  size_t red = (size_t)(wp - 1) * wn * g.mt * g.ntw * 64 * 16;
  g.lds_bytes = tile > red ? tile : red;
  if (g.lds_bytes > 150 * 1024) return false;
  // workgroups along the reduction.  r02 sweep (tools/sweep_wgrad.sh): for the wide layers (K >= 65) 256 instead of
  // 1024 and two waves across N make the ISOLATED kernel faster than MIOpen's five launches (72ch@16x16: 28 -> 21.7 us
  // vs 30.8; 144ch@8x8 27.7 vs 28.7), but routing those layers here is SLOWER in the step (625 vs 632 samples/s:
  // 512-thread workgroups with 30-50 KB of LDS crowd out the other encoder's stream) -- the defaults stay as in r01
  // and the wide layers stay on MIOpen (the sweep's switches were removed in r05; DESIGN 4.6 keeps the record)
  int want = 1024 / g.ngroups;
  if (want < 1) want = 1;
  if (want > g.units) want = g.units;
  g.per = (g.units + want - 1) / want;
  g.chunks = (g.units + g.per - 1) / g.per;
  return true;
}

template <int MT, int NTW, int WT, int RBT, int ST>
void launch_wgrad_t(const float* x, const float* dy, float* partial, const WgradGeo& g, hipStream_t st) {
  if (g.lds_bytes > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3_mfma_kernel<MT, NTW, WT, RBT, ST>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes);
  wgrad3x3_mfma_kernel<MT, NTW, WT, RBT, ST><<<dim3(g.chunks, g.ngroups), g.threads, g.lds_bytes, st>>>(x, dy, partial, g);
}

template <int MT, int NTW>
void launch_wgrad(const float* x, const float* dy, float* partial, const WgradGeo& g, hipStream_t st) {
  // the (W, rows-per-unit) pairs the HRNet branches produce get constant-folded instantiations
  if (g.st == 1) {
    if (g.W == 64 && g.rb == 4)      launch_wgrad_t<MT, NTW, 64, 4, 1>(x, dy, partial, g, st);
    else if (g.W == 32 && g.rb == 4) launch_wgrad_t<MT, NTW, 32, 4, 1>(x, dy, partial, g, st);
    else if (g.W == 32 && g.rb == 8) launch_wgrad_t<MT, NTW, 32, 8, 1>(x, dy, partial, g, st);
    else if (g.W == 16 && g.rb == 8) launch_wgrad_t<MT, NTW, 16, 8, 1>(x, dy, partial, g, st);
    else if (g.W == 16 && g.rb == 16) launch_wgrad_t<MT, NTW, 16, 16, 1>(x, dy, partial, g, st);
    else if (g.W == 8 && g.rb == 8)  launch_wgrad_t<MT, NTW, 8, 8, 1>(x, dy, partial, g, st);
    else                             launch_wgrad_t<MT, NTW, 0, 0, 0>(x, dy, partial, g, st);
  } else {
    if (g.W == 32 && g.rb == 2)      launch_wgrad_t<MT, NTW, 32, 2, 2>(x, dy, partial, g, st);
    else if (g.W == 32 && g.rb == 4) launch_wgrad_t<MT, NTW, 32, 4, 2>(x, dy, partial, g, st);
    else if (g.W == 16 && g.rb == 4) launch_wgrad_t<MT, NTW, 16, 4, 2>(x, dy, partial, g, st);
    else if (g.W == 16 && g.rb == 8) launch_wgrad_t<MT, NTW, 16, 8, 2>(x, dy, partial, g, st);
    else if (g.W == 8 && g.rb == 8)  launch_wgrad_t<MT, NTW, 8, 8, 2>(x, dy, partial, g, st);
    else                             launch_wgrad_t<MT, NTW, 0, 0, 0>(x, dy, partial, g, st);
  }
}

}  // namespace

extern "C" {

static size_t wgrad_ws_bytes(int N, int C, int K, int H, int W, int taps, int st) {
  WgradGeo g;
  if (!make_wgeo(N, C, K, H, W, taps, st, g)) return 0;
  return (size_t)g.chunks * K * C * taps * sizeof(float);
}

static int wgrad_run(const float* x, const float* dy, int N, int C, int K, int H, int W, int taps, int cst, float* dw,
                     void* workspace, size_t workspace_bytes, hcm_stream_t stream, int* chunks_out = nullptr) {
  WgradGeo g;
  if (!x || !dy || (!dw && !chunks_out) || !workspace || !make_wgeo(N, C, K, H, W, taps, cst, g)) return (int)hipErrorInvalidValue;
  if (workspace_bytes < (size_t)g.chunks * K * C * taps * sizeof(float)) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)workspace;
  switch (g.mt) {
    case 1: launch_wgrad<1, 12>(x, dy, partial, g, st); break;
    case 2: launch_wgrad<2, 3>(x, dy, partial, g, st); break;
    case 3: launch_wgrad<3, 3>(x, dy, partial, g, st); break;
    case 4: launch_wgrad<4, 6>(x, dy, partial, g, st); break;
    case 5: launch_wgrad<5, 2>(x, dy, partial, g, st); break;
    case 8: launch_wgrad<8, 3>(x, dy, partial, g, st); break;
    case 9: launch_wgrad<9, 2>(x, dy, partial, g, st); break;
    case 16: launch_wgrad<16, 1>(x, dy, partial, g, st); break;
    default: return (int)hipErrorInvalidValue;
  }
  HCM_CHECK_LAUNCH();
  if (chunks_out != nullptr) {            // partial sums only: the caller reduces later (hcm_wgrad_reduce_batch)
    *chunks_out = g.chunks;
    return 0;
  }
  const int total = K * C * taps;
  wgrad3x3_reduce_kernel<<<(total + 63) / 64, 1024, 0, st>>>(partial, dw, total, g.chunks);
  HCM_CHECK_LAUNCH();
  return 0;
}

size_t hcm_conv3x3_wgrad_workspace_bytes(int N, int C, int K, int H, int W) { return wgrad_ws_bytes(N, C, K, H, W, 9, 1); }
int hcm_conv3x3_wgrad(const float* x, const float* dy, int N, int C, int K, int H, int W, float* dw, void* workspace,
                      size_t workspace_bytes, hcm_stream_t stream) {
  return wgrad_run(x, dy, N, C, K, H, W, 9, 1, dw, workspace, workspace_bytes, stream);
}

size_t hcm_conv1x1_wgrad_workspace_bytes(int N, int C, int K, int H, int W) { return wgrad_ws_bytes(N, C, K, H, W, 1, 1); }
int hcm_conv1x1_wgrad(const float* x, const float* dy, int N, int C, int K, int H, int W, float* dw, void* workspace,
                      size_t workspace_bytes, hcm_stream_t stream) {
  return wgrad_run(x, dy, N, C, K, H, W, 1, 1, dw, workspace, workspace_bytes, stream);
}

// 3x3 / stride 2 / pad 1: x [N,C,2Ho,2Wo], dy [N,K,Ho,Wo] (Ho, Wo are the arguments)
size_t hcm_conv3x3s2_wgrad_workspace_bytes(int N, int C, int K, int Ho, int Wo) { return wgrad_ws_bytes(N, C, K, Ho, Wo, 9, 2); }
int hcm_conv3x3s2_wgrad(const float* x, const float* dy, int N, int C, int K, int Ho, int Wo, float* dw, void* workspace,
                        size_t workspace_bytes, hcm_stream_t stream) {
  return wgrad_run(x, dy, N, C, K, Ho, Wo, 9, 2, dw, workspace, workspace_bytes, stream);
}

// kind: 3 = 3x3 stride 1, 1 = 1x1, 2 = 3x3 stride 2 (H, W = the OUTPUT map, as in the entry points above)
int hcm_conv_wgrad_partial(int kind, const float* x, const float* dy, int N, int C, int K, int H, int W, void* workspace,
                           size_t workspace_bytes, int* chunks, hcm_stream_t stream) {
  if (!chunks || (kind != 1 && kind != 2 && kind != 3)) return (int)hipErrorInvalidValue;
  return wgrad_run(x, dy, N, C, K, H, W, kind == 1 ? 1 : 9, kind == 2 ? 2 : 1, nullptr, workspace, workspace_bytes, stream, chunks);
}

int hcm_wgrad_reduce_batch(const hcm_wgrad_reduce_desc* descs, int n, hcm_stream_t stream) {
  if (n < 0 || (n > 0 && !descs)) return (int)hipErrorInvalidValue;
  for (int lo = 0; lo < n; lo += kReduceBatch) {
    ReduceBatch b;
    b.n = n - lo < kReduceBatch ? n - lo : kReduceBatch;
    int blocks = 0;
    for (int d = 0; d < b.n; ++d) {
      const hcm_wgrad_reduce_desc& e = descs[lo + d];
      if (!e.partial || !e.dw || e.total <= 0 || e.chunks <= 0) return (int)hipErrorInvalidValue;
      b.partial[d] = e.partial; b.dw[d] = e.dw; b.total[d] = e.total; b.chunks[d] = e.chunks;
      b.block0[d] = blocks;
      blocks += (e.total + 63) / 64;
    }
    b.block0[b.n] = blocks;
    wgrad_reduce_batch_kernel<<<blocks, 1024, 0, (hipStream_t)stream>>>(b);
    HCM_CHECK_LAUNCH();
  }
  return 0;
}

}  // extern "C"
