// 1x1 convolutions of the PointNet++ shared MLPs on ball tensors [B, C, npoint, nsample] (reference:
// networks/pointnet2/pytorch_utils.py:5-33 -> nn.Conv2d(kernel_size=1, bias=False) inside SharedMLP), gfx950, r05.
//
// MIOpen answers these shapes -- 16..512 channels on maps of 1 K .. 131 K positions -- with Winograd-class and generic GEMM
// kernels (profiles/r05_hrnetpn_timeline.txt: 3.5 ms forward, 4.3 ms data gradient per HRNetPN step for 4 GB of tensors
// that HBM moves in ~1 ms).  With positions contiguous (NCHW) a 1x1 convolution is, per image,
//     Z[k][p] = sum_c W[k][c] X[c][p]          (forward;  data gradient: the same with W^T: dX[c][p] = sum_k W[k][c] dZ[k][p])
// and v_mfma_f32_16x16x4_f32 takes both operands as they lie: lane (np, g) feeds X[c0 + g][p0 + np] (16 lanes = 64 contiguous
// bytes of one channel row) and W[m0 + np][c0 + g] (the weights: a few KB, cache resident), and owns
// Z[m0 + np][p0 + 4 g .. + 3] (one 16-byte store).  No transposes, LDS only for the weights; every activation byte is read once
// per 64-channel output block and written once.  A wave owns 64 positions x 16 MT output channels, a workgroup 4 waves = 256
// positions; grid = (position blocks, output-channel blocks, images).  Exact fp32 (an fmaf chain per output element).
#include "hcm_common.h"
#include "../../include/hcmoco_hip.h"

namespace {

using namespace hcm;
typedef float v4f __attribute__((ext_vector_type(4)));

// TRANS: the data gradient (W^T).  W is [Kw][Cw] row-major as nn.Conv2d stores it; M = output rows of this product
// (forward: Kw, data gradient: Cw), R = its reduction length (forward: Cw, data gradient: Kw).
// The MFMA runs transposed -- A = a 16-position tile of X^T, B = 16 channels of W^T -- so that a lane's four accumulator
// registers are four CONSECUTIVE positions of one output channel: one 16-byte store per (channel tile, position tile)
// instead of four 4-byte ones (the forward pass writes twice what it reads).
// r06, measured and NOT adopted (profiles/r06_conv1x1_layers.txt): 16-byte activation loads with a permuted-position operand
// (one load per lane and k-step instead of four; the weight gradient's trick).  (a) np permuted so that the store pattern below is
// kept, prefetch 3-6 k-steps deep: forward 416 -> 446 us at 32 -> 64 / P = 131072, 311 -> 337 at 64 -> 128 / P = 32768, data
// gradient 337 -> 430; (b) identity permutation (256 contiguous bytes per row and instruction), the k-steps of all of a
// workgroup's position blocks as ONE software pipeline: 596 / 379 / 283 us forward.  Both slower on all five layers: the layer
// is bound by moving 1.6 GB (one third read, two thirds written) through rows that lie 64 - 512 KB apart, not by the number of
// load instructions.  (c) the r05 access pattern with three k-steps in flight instead of one: 442 / 306 / 242 us, inside the
// run-to-run spread of r05's 416 / 311 / 249.  The r05 form stays.
template <int MT, bool TRANS>
__global__ __launch_bounds__(256) void conv1x1_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                      float* __restrict__ Z, int M, int R, int Cw, int P, int PB) {
  constexpr int RC = 128;                              // reduction rows of W staged per round: 128 x 16 MT floats (<= 32 KB)
  __shared__ float Ws[RC * 16 * MT];                   // Ws[r][np][mt] = W^T[rc + r][m0 + 16 mt + np]: one MT-wide read per lane
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int np = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.y * (16 * MT);
  // a workgroup walks PB consecutive blocks of 256 positions: the channel rows of a ball tensor lie 32 .. 512 KB apart, so a
  // block touches R + M distant pieces of memory, and staying on them for PB KB each (and staging W once) is what pays
  for (int i = 0; i < PB; ++i) {
    const int p0 = ((blockIdx.x * PB + i) * 4 + wave) * 64;
    const bool active = p0 < P;                        // P is a multiple of 64
    const float* x = X + (size_t)blockIdx.z * R * P + (active ? p0 : 0) + np;
    v4f acc[4][MT];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[t][mt] = (v4f){0.f, 0.f, 0.f, 0.f};
    for (int rc = 0; rc < R; rc += RC) {
      const int rows = min(RC, R - rc);
      if (R > RC || i == 0) {
        if (rc || i) __syncthreads();
        // the weights go through LDS: read from global memory by the MFMA lane pattern they are 16-byte pieces of 16
        // different rows per instruction, four times the requests of the activations for the same bytes
        for (int e = threadIdx.x; e < rows * 16 * MT; e += 256) {
          int r, m;
          if (TRANS) { r = e / (16 * MT); m = e - r * (16 * MT); }            // W[r][m]: m contiguous
          else { m = e / rows; r = e - m * rows; }                            // W[m][r]: r contiguous
          const float v = m0 + m < M ? (TRANS ? W[(size_t)(rc + r) * Cw + m0 + m] : W[(size_t)(m0 + m) * Cw + rc + r]) : 0.f;
          Ws[r * (16 * MT) + (m & 15) * MT + (m >> 4)] = v;
        }
        __syncthreads();
      }
      if (active) {
        const float* xr = x + (size_t)(rc + g) * P;
        float xv[4], xn[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) xv[t] = xr[16 * t];
        for (int r0 = 0; r0 < rows; r0 += 4) {
          const bool more = r0 + 4 < rows;
#pragma unroll
          for (int t = 0; t < 4; ++t) xn[t] = more ? xr[(size_t)(r0 + 4) * P + 16 * t] : 0.f;
          float wv[MT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) wv[mt] = Ws[(r0 + g) * (16 * MT) + np * MT + mt];
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
              acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[t], wv[mt], acc[t][mt], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < 4; ++t) xv[t] = xn[t];
        }
      }
    }
    if (active) {
      // acc[t][mt][q] = Z[m0 + 16 mt + np][p0 + 16 t + 4 g + q]
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = m0 + 16 * mt + np;
        if (m < M) {
          float* z = Z + ((size_t)blockIdx.z * M + m) * P + p0 + 4 * g;
#pragma unroll
          for (int t = 0; t < 4; ++t) *reinterpret_cast<v4f*>(z + 16 * t) = acc[t][mt];
        }
      }
    }
  }
}

template <bool TRANS>
int launch(const float* X, const float* W, float* Z, int N, int M, int R, int Cw, int P, hipStream_t st) {
  const int mblocks = M <= 32 ? 1 : (M + 63) / 64;
  // PB blocks of 256 positions per workgroup: as many as leave >= 2048 workgroups (8 per CU); one when W takes several rounds,
  // and one for the narrow outputs (M <= 32: little W to stage, and the short workgroups balance better -- 32 -> 64 data
  // gradient at 131 K positions 327 us against 397 us, tools/bench_conv1x1.py)
  int PB = 1;
  if (R <= 128 && M > 32)
    while (PB < 8 && (long long)((P + 512 * PB - 1) / (512 * PB)) * mblocks * N >= 2048) PB *= 2;
  const int pb = (P + 256 * PB - 1) / (256 * PB);
  if (M <= 16) conv1x1_kernel<1, TRANS><<<dim3(pb, 1, N), 256, 0, st>>>(X, W, Z, M, R, Cw, P, PB);
  else if (M <= 32) conv1x1_kernel<2, TRANS><<<dim3(pb, 1, N), 256, 0, st>>>(X, W, Z, M, R, Cw, P, PB);
  else conv1x1_kernel<4, TRANS><<<dim3(pb, mblocks, N), 256, 0, st>>>(X, W, Z, M, R, Cw, P, PB);
  HCM_CHECK_LAUNCH();
  return 0;
}

// ---- weight gradient ------------------------------------------------------------------------------------------------
//     dW[k][c] = sum over images n and positions p of dZ[n][k][p] X[n][c][p]
// A [K x C] output of at most 256 x 128 with a reduction over 0.5 M .. 4 M positions.  MIOpen runs it as an NHWC implicit GEMM
// behind two layout transposes of the operands (profiles/r05_hrnetpn_timeline.txt: batched_transpose 2.5 ms + igemm_wrw
// 2.3 ms per HRNetPN step).  v_mfma_f32_16x16x4_f32 contracts over its 4 "k" slots, and a SUM over positions does not care
// which position sits in which slot as long as A and B agree: lane (np, g) loads the float4 dZ[k0 + np][p + 4 g ..] and
// X[c0 + np][p + 4 g ..] (16 lanes x 4 g = 16 rows x 64 contiguous bytes, twice = one 128-byte line per row) and register j
// of both goes to MFMA number j.  No transposes, no LDS on the way in.  A wave owns a [16 KT x 16 CT] block of dW and a run
// of L positions of one image, the four waves of a workgroup four consecutive runs; their accumulators meet in LDS and ONE
// partial block per workgroup goes to the workspace [chunk][K][C]; wgrad1x1_reduce_kernel sums the chunks in fixed order
// (deterministic, no atomics).  Loads of step s + 1 are issued before the MFMAs of step s.
template <int KT, int CT>
__global__ __launch_bounds__(256) void wgrad1x1_ball_kernel(const float* __restrict__ X, const float* __restrict__ DZ,
                                                            float* __restrict__ partial, int C, int K, int P, int L,
                                                            int cblocks) {
  extern __shared__ float red[];                       // [4 waves][KT*CT*4][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int np = lane & 15, g = lane >> 4;
  // Workgroup id -> (chunk of positions, block of dW).  The blocks of one chunk read the same rows of X / dZ again: they get
  // ids 8 apart -- dispatched back to back onto the SAME XCD (ids go round-robin over the 8 XCDs), so the second reader
  // finds the rows in that XCD's L2 (r05_conv1x1_pmc.json: 1.33x the algorithmic bytes fetched with the blocks a whole
  // grid dimension apart, 64 -> 128 channels).
  const int nblocks = gridDim.y, chunks = gridDim.x;
  const int id = blockIdx.y * chunks + blockIdx.x;                 // linear dispatch order
  int chunk, blk;
  if ((chunks & 7) == 0) {
    const int group = id / (8 * nblocks), r = id - group * (8 * nblocks);
    blk = r >> 3;
    chunk = group * 8 + (r & 7);
  } else {
    blk = blockIdx.y;
    chunk = blockIdx.x;
  }
  const int per_image = P / (4 * L);
  const int n = chunk / per_image, run = chunk - n * per_image;
  const int k0 = (blk / cblocks) * (16 * KT), c0 = (blk % cblocks) * (16 * CT);
  const size_t pw = (size_t)run * 4 * L + (size_t)wave * L + 4 * g;
  const float* a_ptr = DZ + ((size_t)n * K + k0 + np) * P + pw;
  const float* b_ptr = X + ((size_t)n * C + c0 + np) * P + pw;
  const size_t tile = (size_t)16 * P;
  v4f acc[KT][CT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[kt][ct] = (v4f){0.f, 0.f, 0.f, 0.f};
  v4f a[2][KT][2], b[2][CT][2];
  auto load = [&](int buf, int s) {
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int h = 0; h < 2; ++h) a[buf][kt][h] = *reinterpret_cast<const v4f*>(a_ptr + kt * tile + 32 * s + 16 * h);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int h = 0; h < 2; ++h) b[buf][ct][h] = *reinterpret_cast<const v4f*>(b_ptr + ct * tile + 32 * s + 16 * h);
  };
  auto mul = [&](int buf) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
            acc[kt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[buf][kt][h][j], b[buf][ct][h][j], acc[kt][ct], 0, 0, 0);
  };
  const int steps = L / 32;                            // even (L is a multiple of 64)
  load(0, 0);
  for (int s = 0; s < steps; s += 2) {
    load(1, s + 1);
    mul(0);
    if (s + 2 < steps) load(0, s + 2);
    mul(1);
  }
  // acc[kt][ct][q] = dW[k0 + 16 kt + 4 g + q][c0 + 16 ct + np] of this wave's run
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int q = 0; q < 4; ++q) red[(wave * (KT * CT * 4) + (kt * CT + ct) * 4 + q) * 64 + lane] = acc[kt][ct][q];
  __syncthreads();
  float* out = partial + (size_t)chunk * K * C;
  for (int e = threadIdx.x; e < KT * CT * 4 * 64; e += 256) {
    const float v = (red[e] + red[KT * CT * 256 + e]) + (red[2 * KT * CT * 256 + e] + red[3 * KT * CT * 256 + e]);
    const int l = e & 63, t = e >> 6, q = t & 3, ct = (t >> 2) % CT, kt = (t >> 2) / CT;
    out[(size_t)(k0 + 16 * kt + 4 * (l >> 4) + q) * C + c0 + 16 * ct + (l & 15)] = v;
  }
}

// dw[o] = sum over chunks of partial[chunk][o], chunks taken in four interleaved slices and the slices in fixed order.
__global__ __launch_bounds__(256) void wgrad1x1_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int KC,
                                                              int chunks) {
  __shared__ float part[4][64];
  const int o = blockIdx.x * 64 + (threadIdx.x & 63), slice = threadIdx.x >> 6;
  float s0 = 0.f, s1 = 0.f;
  if (o < KC) {
    int ch = slice;
    for (; ch + 4 < chunks; ch += 8) { s0 += partial[(size_t)ch * KC + o]; s1 += partial[(size_t)(ch + 4) * KC + o]; }
    if (ch < chunks) s0 += partial[(size_t)ch * KC + o];
  }
  part[slice][threadIdx.x & 63] = s0 + s1;
  __syncthreads();
  if (slice == 0 && o < KC) dw[o] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

struct BallWgradGeo { int kt, ct, kblocks, cblocks, L, chunks; };

bool ball_wgrad_geo(int N, int C, int K, int P, BallWgradGeo& g) {
  if (N <= 0 || C <= 0 || K <= 0 || P <= 0 || (K & 15) || (C & 15) || (P & 255)) return false;
  g.kt = (K & 63) == 0 ? 4 : (K & 31) == 0 ? 2 : 1;
  g.ct = (C & 63) == 0 ? 4 : (C & 31) == 0 ? 2 : 1;
  g.kblocks = K / (16 * g.kt);
  g.cblocks = C / (16 * g.ct);
  // per-wave run L (a multiple of 64 that divides P / 4): the longest that still leaves ~512 workgroups for the 256 CUs
  g.L = 64;
  for (int L = 2048; L >= 64; L >>= 1)
    if (P % (4 * L) == 0 && (long long)N * (P / (4 * L)) * g.kblocks * g.cblocks >= 512) { g.L = L; break; }
  g.chunks = N * (P / (4 * g.L));
  return true;
}

template <int KT, int CT>
int ball_wgrad_launch(const float* x, const float* dz, float* partial, int C, int K, int P, const BallWgradGeo& g, hipStream_t st) {
  const size_t lds = (size_t)4 * KT * CT * 256 * sizeof(float);
  // the attribute is per DEVICE: set on every launch (as rowproj.hip / pointnet2.hip do), not once per process
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad1x1_ball_kernel<KT, CT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  wgrad1x1_ball_kernel<KT, CT><<<dim3(g.chunks, g.kblocks * g.cblocks), 256, lds, st>>>(x, dz, partial, C, K, P, g.L, g.cblocks);
  return 0;
}

}  // namespace

extern "C" {

size_t hcm_conv1x1_ball_wgrad_workspace_bytes(int N, int C, int K, int H, int W) {
  BallWgradGeo g;
  if (H <= 0 || W <= 0 || !ball_wgrad_geo(N, C, K, H * W, g)) return 0;
  return (size_t)g.chunks * K * C * sizeof(float);
}

int hcm_conv1x1_ball_wgrad(const float* x, const float* dy, int N, int C, int K, int H, int W, float* dw, void* workspace,
                           size_t workspace_bytes, hcm_stream_t stream) {
  BallWgradGeo g;
  if (!x || !dy || !dw || !workspace || H <= 0 || W <= 0 || !ball_wgrad_geo(N, C, K, H * W, g)) return (int)hipErrorInvalidValue;
  if (workspace_bytes < (size_t)g.chunks * K * C * sizeof(float)) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  float* partial = static_cast<float*>(workspace);
  const int P = H * W;
  hcm::ProfSpan span(HCM_PROF_CONV1X1_DW, st, 2.0 * N * (double)C * K * P);
  int rc;
  if (g.kt == 4 && g.ct == 4) rc = ball_wgrad_launch<4, 4>(x, dy, partial, C, K, P, g, st);
  else if (g.kt == 4 && g.ct == 2) rc = ball_wgrad_launch<4, 2>(x, dy, partial, C, K, P, g, st);
  else if (g.kt == 4) rc = ball_wgrad_launch<4, 1>(x, dy, partial, C, K, P, g, st);
  else if (g.kt == 2 && g.ct == 4) rc = ball_wgrad_launch<2, 4>(x, dy, partial, C, K, P, g, st);
  else if (g.kt == 2 && g.ct == 2) rc = ball_wgrad_launch<2, 2>(x, dy, partial, C, K, P, g, st);
  else if (g.kt == 2) rc = ball_wgrad_launch<2, 1>(x, dy, partial, C, K, P, g, st);
  else if (g.ct == 4) rc = ball_wgrad_launch<1, 4>(x, dy, partial, C, K, P, g, st);
  else if (g.ct == 2) rc = ball_wgrad_launch<1, 2>(x, dy, partial, C, K, P, g, st);
  else rc = ball_wgrad_launch<1, 1>(x, dy, partial, C, K, P, g, st);
  if (rc != 0) return rc;
  HCM_CHECK_LAUNCH();
  wgrad1x1_reduce_kernel<<<(K * C + 63) / 64, 256, 0, st>>>(partial, dw, K * C, g.chunks);
  HCM_CHECK_LAUNCH();
  return 0;
}

int hcm_conv1x1_supported(int C, int K, int P) {
  return C > 0 && K > 0 && P > 0 && (C & 3) == 0 && (K & 3) == 0 && (P & 63) == 0 ? 1 : 0;
}

int hcm_conv1x1_forward(const float* x, const float* w, float* z, int N, int C, int K, int P, hcm_stream_t stream) {
  if (!x || !w || !z || N <= 0 || !hcm_conv1x1_supported(C, K, P)) return (int)hipErrorInvalidValue;
  hcm::ProfSpan span(HCM_PROF_CONV1X1_FWD, (hipStream_t)stream, 2.0 * N * (double)C * K * P);
  return launch<false>(x, w, z, N, K, C, C, P, (hipStream_t)stream);
}

int hcm_conv1x1_backward_data(const float* dz, const float* w, float* dx, int N, int C, int K, int P, hcm_stream_t stream) {
  if (!dz || !w || !dx || N <= 0 || !hcm_conv1x1_supported(C, K, P)) return (int)hipErrorInvalidValue;
  hcm::ProfSpan span(HCM_PROF_CONV1X1_DX, (hipStream_t)stream, 2.0 * N * (double)C * K * P);
  return launch<true>(dz, w, dx, N, C, K, C, P, (hipStream_t)stream);
}

}  // extern "C"
