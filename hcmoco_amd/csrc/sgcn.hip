// Fused SemGCN layer for gfx950.
//
// One layer of the 2D-keypoint encoder (pycontrast/networks/SGCN/sem_graph_conv.py:34-48 +
// sem_gcn.py:8-28) is SemGraphConv -> BatchNorm1d -> ReLU on a [B, J<=32, C<=128] tensor.  In eager
// PyTorch the whole 10-layer encoder is ~1500 kernel launches per training step for 2.7 ms of GPU
// work, i.e. it is paid in host launch time (8.7 ms).  Here a layer is one library GEMM
// H = X [W0 | W1] plus
//     A      = row-softmax of the learned edge weights e over the skeleton adjacency
//     Y      = A_diag (.) H0 + A_off H1 + bias                (graph mixing over the J joints)
//     out    = ReLU(BatchNorm(Y))                              (batch statistics over B*J rows, running
//                                                               statistics updated in place)
// on a grid of ONE WORKGROUP PER BATCH ELEMENT (512 threads = C channels x 512/C joint slices; the
// sample's H tile, J x 2C floats = 17 KB, is staged through LDS with float4 loads).  BatchNorm1d needs
// the statistics of all B*J rows, so a direction is split at that dependency, like bn_stats / bn_apply
// in bnact.hip:
//     forward : sgc_mix_kernel   (Y, per-sample partial sums)  ->  sgc_norm_kernel (merge, normalise)
//     backward: sgc_bwd_stats_kernel (per-sample sums of dz, dz*xhat)
//               -> sgc_bwd_kernel (merge, dY, transposed mixing -> dH, per-sample dbias / dA partials)
//               -> sgc_bwd_finish_kernel (one workgroup: merge, softmax backward -> de)
// plus two library GEMMs for dX and dW.  Every merge walks the per-sample partials in sample order:
// deterministic, no atomics.  (Round 1 ran each direction in ONE 1024-thread workgroup -- 106 / 278 us
// per layer on one of 256 CUs.)
#include "hcm_common.h"
#include "../../include/hcmoco_hip.h"

namespace {

using namespace hcm;

constexpr int kThreads = 512;
constexpr int kMaxJ = 32;
constexpr int kMaxE = 256;  // edges incl. self loops
constexpr int kMaxC = 128;

struct Graph {   // CSR by row (i: receiving joint) and CSC by column (j: sending joint), edge ids in
  const int* row_ptr;   // [J+1]   row-major order = the order of the reference's `adj[self.m]`
  const int* col_idx;   // [E]     j of edge e
  const int* csc_ptr;   // [J+1]
  const int* csc_edge;  // [E]     edge ids sorted by column
  const int* edge_row;  // [E]     i of edge e
};

__device__ __forceinline__ void edge_softmax(const float* __restrict__ e, const Graph& g, int J, float* sA) {
  // thread i < J normalises its row
  const int i = threadIdx.x;
  if (i < J) {
    const int lo = g.row_ptr[i], hi = g.row_ptr[i + 1];
    float mx = -3.0e38f;
    for (int k = lo; k < hi; ++k) mx = fmaxf(mx, e[k]);
    float s = 0.f;
    for (int k = lo; k < hi; ++k) s += __expf(e[k] - mx);
    for (int k = lo; k < hi; ++k) sA[k] = __expf(e[k] - mx) / s;
  }
}

// sample b's H tile [J][2C] -> LDS, float4 loads (2C is a multiple of 4)
__device__ __forceinline__ void stage_tile(const float* __restrict__ Hb, float* sH, int n) {
  const float4* src = reinterpret_cast<const float4*>(Hb);
  float4* dst = reinterpret_cast<float4*>(sH);
  for (int q = threadIdx.x; q < n / 4; q += kThreads) dst[q] = src[q];
}

// Sum over the joint slices r = 0..RS-1 of a per-thread value, in slice order; valid in threads < C.
__device__ __forceinline__ float slice_sum(float* sRed, float v, int C, int RS) {
  __syncthreads();
  sRed[threadIdx.x] = v;
  __syncthreads();
  float a = 0.f;
  if ((int)threadIdx.x < C)
    for (int r = 0; r < RS; ++r) a += sRed[r * C + threadIdx.x];
  return a;
}

// ---------------------------------------------------------------------------------------------
// forward 1/2: graph mixing for sample b = blockIdx.x; per-sample partial sums for the statistics.
// Without BatchNorm (output layer), or with BatchNorm in eval mode (running statistics), the whole
// layer finishes here.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void sgc_mix_kernel(
    const float* __restrict__ H, const float* __restrict__ e, Graph g, const float* __restrict__ bias,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ running_mean,
    const float* __restrict__ running_var, int J, int C, int E, int has_bn, int relu, int training, float eps,
    float* __restrict__ out, float* __restrict__ xhat, float* __restrict__ invstd_out, float* __restrict__ A_out,
    float* __restrict__ part) {
  __shared__ float sA[kMaxE];
  __shared__ float sRed[kThreads];
  extern __shared__ __attribute__((aligned(16))) float sH[];   // [J][2C]
  const int tid = threadIdx.x, b = blockIdx.x;
  const int c = tid % C, r = tid / C, RS = kThreads / C, ld = 2 * C;
  edge_softmax(e, g, J, sA);
  stage_tile(H + (int64_t)b * J * ld, sH, J * ld);
  __syncthreads();
  if (b == 0 && tid < E) A_out[tid] = sA[tid];
  const bool inline_bn = has_bn && !training;
  float mu = 0.f, is = 1.f, ga = 1.f, be = 0.f;
  if (inline_bn) {
    mu = running_mean[c];
    is = 1.f / sqrtf(running_var[c] + eps);
    ga = gamma[c];
    be = beta[c];
    if (b == 0 && r == 0) invstd_out[c] = is;
  }
  const float bc = bias ? bias[c] : 0.f;
  float sum = 0.f, sq = 0.f;
  for (int i = r; i < J; i += RS) {
    float y = bc;
    for (int k = g.row_ptr[i]; k < g.row_ptr[i + 1]; ++k) {
      const int j = g.col_idx[k];
      y = fmaf(sA[k], (j == i) ? sH[i * ld + c] : sH[j * ld + C + c], y);
    }
    const int64_t o = ((int64_t)b * J + i) * C + c;
    if (inline_bn) {
      const float xh = (y - mu) * is;
      xhat[o] = xh;
      y = fmaf(ga, xh, be);
    }
    if (relu && (!has_bn || inline_bn)) y = fmaxf(y, 0.f);
    out[o] = y;
    sum += y;
    sq = fmaf(y, y, sq);
  }
  if (has_bn && training) {
    const float a = slice_sum(sRed, sum, C, RS);
    const float q = slice_sum(sRed, sq, C, RS);
    if (tid < C) {
      part[((int64_t)b * 2) * C + tid] = a;
      part[((int64_t)b * 2 + 1) * C + tid] = q;
    }
  }
}

// Merge the per-sample pairs part[b][0..1][c] over b in sample order (threads (c, r) take b = r, r+RS, ...,
// the slices are then added in slice order): valid in threads < C.
__device__ __forceinline__ void merge_pairs(const float* __restrict__ part, float* sRed, int B, int C, int RS,
                                            float& a, float& q) {
  const int c = threadIdx.x % C, r = threadIdx.x / C;
  float pa = 0.f, pq = 0.f;
  for (int b = r; b < B; b += RS) {
    pa += part[((int64_t)b * 2) * C + c];
    pq += part[((int64_t)b * 2 + 1) * C + c];
  }
  a = slice_sum(sRed, pa, C, RS);
  q = slice_sum(sRed, pq, C, RS);
}

// forward 2/2 (training-mode BatchNorm only): statistics, running statistics, normalise + ReLU.
__global__ __launch_bounds__(kThreads) void sgc_norm_kernel(
    const float* __restrict__ part, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ running_mean, float* __restrict__ running_var, int B, int J, int C, int relu,
    float momentum, float eps, float* __restrict__ out, float* __restrict__ xhat, float* __restrict__ invstd_out) {
  __shared__ float sRed[kThreads];
  __shared__ float sMean[kMaxC], sInv[kMaxC];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int c = tid % C, r = tid / C, RS = kThreads / C;
  float a, q;
  merge_pairs(part, sRed, B, C, RS, a, q);
  if (tid < C) {
    const float n = (float)(B * J);
    const float mean = a / n;
    const float var = fmaxf(q / n - mean * mean, 0.f);
    const float is = 1.f / sqrtf(var + eps);
    sMean[tid] = mean;
    sInv[tid] = is;
    if (b == 0) {
      invstd_out[tid] = is;
      if (running_mean) {
        running_mean[tid] = (1.f - momentum) * running_mean[tid] + momentum * mean;
        running_var[tid] = (1.f - momentum) * running_var[tid] + momentum * var * (n / fmaxf(n - 1.f, 1.f));
      }
    }
  }
  __syncthreads();
  const float mu = sMean[c], is = sInv[c], ga = gamma[c], be = beta[c];
  for (int i = r; i < J; i += RS) {
    const int64_t o = ((int64_t)b * J + i) * C + c;
    const float xh = (out[o] - mu) * is;
    xhat[o] = xh;
    const float z = fmaf(ga, xh, be);
    out[o] = relu ? fmaxf(z, 0.f) : z;
  }
}

// ---------------------------------------------------------------------------------------------
// backward 1/3 (BatchNorm only): per-sample sums of dz and dz * xhat.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void sgc_bwd_stats_kernel(
    const float* __restrict__ dOut, const float* __restrict__ out, const float* __restrict__ xhat, int J, int C,
    int relu, float* __restrict__ part) {
  __shared__ float sRed[kThreads];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int c = tid % C, r = tid / C, RS = kThreads / C;
  float s1 = 0.f, s2 = 0.f;
  for (int i = r; i < J; i += RS) {
    const int64_t o = ((int64_t)b * J + i) * C + c;
    const float dz = (relu && !(out[o] > 0.f)) ? 0.f : dOut[o];
    s1 += dz;
    s2 = fmaf(dz, xhat[o], s2);
  }
  const float a = slice_sum(sRed, s1, C, RS);
  const float q = slice_sum(sRed, s2, C, RS);
  if (tid < C) {
    part[((int64_t)b * 2) * C + tid] = a;
    part[((int64_t)b * 2 + 1) * C + tid] = q;
  }
}

// backward 2/3: dY of sample b, transposed graph mixing -> dH, per-sample partials of dbias and dA.
__global__ __launch_bounds__(kThreads) void sgc_bwd_kernel(
    const float* __restrict__ dOut, const float* __restrict__ out, const float* __restrict__ xhat,
    const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ A, Graph g,
    const float* __restrict__ H, const float* __restrict__ part, int B, int J, int C, int E, int has_bn, int relu,
    int training, float* __restrict__ dH, float* __restrict__ pdb, float* __restrict__ pda) {
  __shared__ float sA[kMaxE];
  __shared__ float sRed[kThreads];
  __shared__ float sW[kMaxE][2];                 // per-wave partial of an edge gradient (C = 128: two waves per slice)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int c = tid % C, r = tid / C, RS = kThreads / C, ld = 2 * C;
  float* sH = smem;                              // [J][2C]
  float* sDy = smem + J * ld;                    // [J][C]
  if (tid < E) sA[tid] = A[tid];
  stage_tile(H + (int64_t)b * J * ld, sH, J * ld);
  float m1 = 0.f, m2 = 0.f;
  if (has_bn) {
    float a, q;
    merge_pairs(part, sRed, B, C, RS, a, q);     // valid in threads < C ...
    __syncthreads();
    if (tid < C) { sRed[tid] = a; sRed[C + tid] = q; }
    __syncthreads();
    const float n = (float)(B * J);
    m1 = sRed[c] / n;                            // ... broadcast to every slice
    m2 = sRed[C + c] / n;
  }
  __syncthreads();
  const float ga_is = has_bn ? gamma[c] * invstd[c] : 1.f;
  float db = 0.f;
  for (int i = r; i < J; i += RS) {
    const int64_t o = ((int64_t)b * J + i) * C + c;
    const float dz = (relu && !(out[o] > 0.f)) ? 0.f : dOut[o];
    float dy = dz;
    if (has_bn) dy = training ? ga_is * (dz - m1 - xhat[o] * m2) : ga_is * dz;
    sDy[i * C + c] = dy;
    db += dy;
  }
  const float dbs = slice_sum(sRed, db, C, RS);  // also the barrier that publishes sDy
  if (tid < C) pdb[(int64_t)b * C + tid] = dbs;
  // dA[e] of this sample = sum_c dY[i,c] * (i==j ? H0[i,c] : H1[j,c]): slice r takes edges r, r+RS, ...;
  // the C channels of a slice are C/64 whole waves: DPP/shuffle wave sums, then the waves in order
  const int wps = C / 64, wv = (tid % C) / 64;
  for (int k = r; k < E; k += RS) {
    const int i = g.edge_row[k], j = g.col_idx[k];
    float v = sDy[i * C + c] * ((j == i) ? sH[i * ld + c] : sH[j * ld + C + c]);
    v = wave_sum(v);
    if ((tid & 63) == 0) sW[k][wv] = v;
  }
  __syncthreads();
  if (tid < E) pda[(int64_t)b * E + tid] = wps == 2 ? sW[tid][0] + sW[tid][1] : sW[tid][0];
  // dH0[b,j,c] = A_jj dY[b,j,c] ;  dH1[b,j,c] = sum_{i != j} A_ij dY[b,i,c]
  float* dHb = dH + (int64_t)b * J * ld;
  for (int j = r; j < J; j += RS) {
    float h0 = 0.f, h1 = 0.f;
    for (int q = g.csc_ptr[j]; q < g.csc_ptr[j + 1]; ++q) {
      const int k = g.csc_edge[q], i = g.edge_row[k];
      const float t = sA[k] * sDy[i * C + c];
      if (i == j) h0 += t; else h1 += t;
    }
    dHb[j * ld + c] = h0;
    dHb[j * ld + C + c] = h1;
  }
}

// backward 3/3 (one workgroup): merge the per-sample partials in sample order; softmax backward.
__global__ __launch_bounds__(kThreads) void sgc_bwd_finish_kernel(
    const float* __restrict__ part, const float* __restrict__ pdb, const float* __restrict__ pda,
    const float* __restrict__ A, Graph g, int B, int J, int C, int E, int has_bn, float* __restrict__ dgamma,
    float* __restrict__ dbeta, float* __restrict__ dbias, float* __restrict__ de) {
  __shared__ float sA[kMaxE], sDA[kMaxE];
  __shared__ float sRed[kThreads];
  const int tid = threadIdx.x;
  const int c = tid % C, r = tid / C, RS = kThreads / C;
  // channel sums: threads (c, r) take samples r, r+RS, ...; the slices are then added in slice order
  float db = 0.f;
  for (int b = r; b < B; b += RS) db += pdb[(int64_t)b * C + c];
  db = slice_sum(sRed, db, C, RS);
  if (tid < C && dbias) dbias[tid] = db;
  if (has_bn) {
    float a, q;
    merge_pairs(part, sRed, B, C, RS, a, q);
    if (tid < C) { dbeta[tid] = a; dgamma[tid] = q; }
  }
  // edge sums: threads (k, r2) with r2 < RE = kThreads / E sample slices
  const int RE = kThreads / E;
  const int k = tid % E, r2 = tid / E;
  float v = 0.f;
  if (r2 < RE)
    for (int b = r2; b < B; b += RE) v += pda[(int64_t)b * E + k];
  __syncthreads();
  sRed[tid] = v;
  __syncthreads();
  if (tid < E) {
    float t = 0.f;
    for (int q = 0; q < RE; ++q) t += sRed[q * E + tid];
    sDA[tid] = t;
    sA[tid] = A[tid];
  }
  __syncthreads();
  // de = A (.) (dA - sum_k A dA) per adjacency row
  if (tid < J) {
    const int lo = g.row_ptr[tid], hi = g.row_ptr[tid + 1];
    float dot = 0.f;
    for (int q = lo; q < hi; ++q) dot = fmaf(sA[q], sDA[q], dot);
    for (int q = lo; q < hi; ++q) de[q] = sA[q] * (sDA[q] - dot);
  }
}

inline bool ok(int B, int J, int C, int E) {
  return B > 0 && J > 0 && J <= kMaxJ && (C == 64 || C == 128) && E > 0 && E <= kMaxE;
}

}  // namespace

extern "C" {

size_t hcm_sgc_workspace_floats(int B, int J, int C, int E) {
  if (!ok(B, J, C, E)) return 0;
  return (size_t)B * (size_t)(3 * C + E);     // part [B][2][C] | pdb [B][C] | pda [B][E]
}

int hcm_sgc_forward(const float* H, const float* e, const int* row_ptr, const int* col_idx,
                    const int* csc_ptr, const int* csc_edge, const int* edge_row, const float* bias,
                    const float* gamma, const float* beta, float* running_mean, float* running_var,
                    int B, int J, int C, int E, int has_bn, int relu, int training, float momentum,
                    float eps, float* out, float* xhat, float* invstd, float* A_out, float* workspace,
                    hcm_stream_t stream) {
  if (!ok(B, J, C, E) || !H || !e || !out || !A_out || !workspace) return (int)hipErrorInvalidValue;
  if (has_bn && (!gamma || !beta || !xhat || !invstd)) return (int)hipErrorInvalidValue;
  if (has_bn && !training && (!running_mean || !running_var)) return (int)hipErrorInvalidValue;
  Graph g{row_ptr, col_idx, csc_ptr, csc_edge, edge_row};
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)J * 2 * C * sizeof(float);
  ProfSpan span(HCM_PROF_SGC_FWD, st);
  sgc_mix_kernel<<<B, kThreads, lds, st>>>(H, e, g, bias, gamma, beta, running_mean, running_var, J, C, E, has_bn,
                                           relu, training, eps, out, xhat, invstd, A_out, workspace);
  HCM_CHECK_LAUNCH();
  if (has_bn && training) {
    sgc_norm_kernel<<<B, kThreads, 0, st>>>(workspace, gamma, beta, running_mean, running_var, B, J, C, relu,
                                            momentum, eps, out, xhat, invstd);
    HCM_CHECK_LAUNCH();
  }
  span.stop();
  return 0;
}

int hcm_sgc_backward(const float* dOut, const float* out, const float* xhat, const float* invstd,
                     const float* gamma, const float* A, const int* row_ptr, const int* col_idx,
                     const int* csc_ptr, const int* csc_edge, const int* edge_row, const float* H, int B,
                     int J, int C, int E, int has_bn, int relu, int training, float* dH, float* dgamma,
                     float* dbeta, float* dbias, float* de, float* workspace, hcm_stream_t stream) {
  if (!ok(B, J, C, E) || !dOut || !A || !H || !dH || !de || !workspace) return (int)hipErrorInvalidValue;
  if ((has_bn || relu) && !out) return (int)hipErrorInvalidValue;
  if (has_bn && (!xhat || !invstd || !gamma || !dgamma || !dbeta)) return (int)hipErrorInvalidValue;
  Graph g{row_ptr, col_idx, csc_ptr, csc_edge, edge_row};
  hipStream_t st = (hipStream_t)stream;
  float* part = workspace;
  float* pdb = part + (size_t)B * 2 * C;
  float* pda = pdb + (size_t)B * C;
  ProfSpan span(HCM_PROF_SGC_BWD, st);
  if (has_bn) {
    sgc_bwd_stats_kernel<<<B, kThreads, 0, st>>>(dOut, out, xhat, J, C, relu, part);
    HCM_CHECK_LAUNCH();
  }
  const size_t lds = (size_t)J * 3 * C * sizeof(float);
  sgc_bwd_kernel<<<B, kThreads, lds, st>>>(dOut, out, xhat, invstd, gamma, A, g, H, part, B, J, C, E, has_bn, relu,
                                           training, dH, pdb, pda);
  HCM_CHECK_LAUNCH();
  sgc_bwd_finish_kernel<<<1, kThreads, 0, st>>>(part, pdb, pda, A, g, B, J, C, E, has_bn, dgamma, dbeta, dbias, de);
  HCM_CHECK_LAUNCH();
  span.stop();
  return 0;
}

}  // extern "C"
