// Fused SemGCN layer for gfx950.
//
// One layer of the 2D-keypoint encoder (pycontrast/networks/SGCN/sem_graph_conv.py:34-48 +
// sem_gcn.py:8-28) is SemGraphConv -> BatchNorm1d -> ReLU on a [B, J<=32, C<=128] tensor.  In eager
// PyTorch the whole 10-layer encoder is ~1500 kernel launches per training step for 2.7 ms of GPU
// work, i.e. it is paid in host launch time (8.7 ms).  Here a layer is one library GEMM
// H = X [W0 | W1] plus ONE single-workgroup kernel (1024 threads: thread = (channel, row slice)):
//     A      = row-softmax of the learned edge weights e over the skeleton adjacency
//     Y      = A_diag (.) H0 + A_off H1 + bias                (graph mixing over the J joints)
//     out    = ReLU(BatchNorm(Y))                              (batch statistics over B*J rows, running
//                                                               statistics updated in place)
// and the backward is one kernel (ReLU', BatchNorm', transposed graph mixing, edge-weight and
// bias gradients) plus two library GEMMs for dX and dW.
#include "hcm_common.h"
#include "../../include/hcmoco_hip.h"

namespace {

using namespace hcm;

constexpr int kThreads = 1024;
constexpr int kMaxJ = 32;
constexpr int kMaxE = 256;  // edges incl. self loops

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

__global__ __launch_bounds__(kThreads) void sgc_fwd_kernel(
    const float* __restrict__ H, const float* __restrict__ e, Graph g, const float* __restrict__ bias,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ running_mean,
    float* __restrict__ running_var, int B, int J, int C, int E, int has_bn, int relu, int training,
    float momentum, float eps, float* __restrict__ out, float* __restrict__ xhat,
    float* __restrict__ invstd_out, float* __restrict__ A_out) {
  __shared__ float sA[kMaxE];
  __shared__ float sRed[2][kThreads];
  __shared__ float sMean[128], sInv[128];
  const int tid = threadIdx.x;
  const int c = tid % C, s = tid / C, nsl = kThreads / C;
  edge_softmax(e, g, J, sA);
  __syncthreads();
  if (tid < E) A_out[tid] = sA[tid];
  const bool worker = s < nsl;
  const int ld = 2 * C;
  float sum = 0.f, sq = 0.f;
  if (worker) {
    const float bc = bias ? bias[c] : 0.f;
    for (int b = s; b < B; b += nsl) {
      const float* Hb = H + (int64_t)b * J * ld;
      for (int i = 0; i < J; ++i) {
        float y = bc;
        for (int k = g.row_ptr[i]; k < g.row_ptr[i + 1]; ++k) {
          const int j = g.col_idx[k];
          y = fmaf(sA[k], (j == i) ? Hb[i * ld + c] : Hb[j * ld + C + c], y);
        }
        out[((int64_t)b * J + i) * C + c] = y;
        sum += y;
        sq = fmaf(y, y, sq);
      }
    }
  }
  if (!has_bn) {
    if (relu && worker)
      for (int b = s; b < B; b += nsl)
        for (int i = 0; i < J; ++i) {
          float* o = out + ((int64_t)b * J + i) * C + c;
          *o = fmaxf(*o, 0.f);
        }
    return;
  }
  // batch statistics: fixed-order reduction over the row slices
  sRed[0][tid] = worker ? sum : 0.f;
  sRed[1][tid] = worker ? sq : 0.f;
  __syncthreads();
  if (tid < C) {
    float mean, var;
    const float n = (float)(B * J);
    if (training) {
      float a = 0.f, q = 0.f;
      for (int k = 0; k < nsl; ++k) { a += sRed[0][k * C + tid]; q += sRed[1][k * C + tid]; }
      mean = a / n;
      var = fmaxf(q / n - mean * mean, 0.f);
      running_mean[tid] = (1.f - momentum) * running_mean[tid] + momentum * mean;
      running_var[tid] = (1.f - momentum) * running_var[tid] + momentum * var * (n / fmaxf(n - 1.f, 1.f));
    } else {
      mean = running_mean[tid];
      var = running_var[tid];
    }
    sMean[tid] = mean;
    sInv[tid] = 1.f / sqrtf(var + eps);
    invstd_out[tid] = sInv[tid];
  }
  __syncthreads();
  if (worker) {
    const float mu = sMean[c], is = sInv[c], ga = gamma[c], be = beta[c];
    for (int b = s; b < B; b += nsl)
      for (int i = 0; i < J; ++i) {
        const int64_t o = ((int64_t)b * J + i) * C + c;
        const float xh = (out[o] - mu) * is;
        xhat[o] = xh;
        const float z = fmaf(ga, xh, be);
        out[o] = relu ? fmaxf(z, 0.f) : z;
      }
  }
}

__global__ __launch_bounds__(kThreads) void sgc_bwd_kernel(
    const float* __restrict__ dOut, const float* __restrict__ out, const float* __restrict__ xhat,
    const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ A,
    Graph g, const float* __restrict__ H, int B, int J, int C, int E, int has_bn, int relu,
    int training, float* __restrict__ dH, float* __restrict__ dgamma, float* __restrict__ dbeta,
    float* __restrict__ dbias, float* __restrict__ de) {
  __shared__ float sA[kMaxE], sDA[kMaxE];
  __shared__ float sRed[2][kThreads];
  __shared__ float sS1[128], sS2[128];
  extern __shared__ __attribute__((aligned(16))) float sDy[];  // [J][kThreads]
  const int tid = threadIdx.x;
  const int c = tid % C, s = tid / C, nsl = kThreads / C;
  const bool worker = s < nsl;
  const int ld = 2 * C;
  if (tid < E) { sA[tid] = A[tid]; sDA[tid] = 0.f; }
  // pass 1: dz sums for the BatchNorm backward
  float s1 = 0.f, s2 = 0.f;
  if (has_bn && worker) {
    for (int b = s; b < B; b += nsl)
      for (int i = 0; i < J; ++i) {
        const int64_t o = ((int64_t)b * J + i) * C + c;
        const float dz = (relu && !(out[o] > 0.f)) ? 0.f : dOut[o];
        s1 += dz;
        s2 = fmaf(dz, xhat[o], s2);
      }
  }
  sRed[0][tid] = worker ? s1 : 0.f;
  sRed[1][tid] = worker ? s2 : 0.f;
  __syncthreads();
  if (tid < C) {
    float a = 0.f, q = 0.f;
    for (int k = 0; k < nsl; ++k) { a += sRed[0][k * C + tid]; q += sRed[1][k * C + tid]; }
    sS1[tid] = a;
    sS2[tid] = q;
    if (has_bn) { dbeta[tid] = a; dgamma[tid] = q; }
  }
  __syncthreads();
  // pass 2: dY, then the transposed graph mixing, the edge-weight and bias gradients
  float db = 0.f;
  if (worker) {
    const float n = (float)(B * J);
    const float ga_is = has_bn ? gamma[c] * invstd[c] : 1.f;
    const float m1 = sS1[c] / n, m2 = sS2[c] / n;
    for (int b = s; b < B; b += nsl) {
      const float* Hb = H + (int64_t)b * J * ld;
      float* dHb = dH + (int64_t)b * J * ld;
      for (int i = 0; i < J; ++i) {
        const int64_t o = ((int64_t)b * J + i) * C + c;
        const float dz = (relu && !(out[o] > 0.f)) ? 0.f : dOut[o];
        float dy = dz;
        if (has_bn) dy = training ? ga_is * (dz - m1 - xhat[o] * m2) : ga_is * dz;
        sDy[i * kThreads + tid] = dy;
        db += dy;
      }
      // d A[e] += dY[b,i,c] * (i==j ? H0[b,i,c] : H1[b,j,c]) : reduce over the 64 channels of the wave
      // first (DPP/bpermute), then one LDS atomic per wave and edge
      for (int k = 0; k < E; ++k) {
        const int i = g.edge_row[k], j = g.col_idx[k];
        float v = sDy[i * kThreads + tid] * ((j == i) ? Hb[i * ld + c] : Hb[j * ld + C + c]);
        v = wave_sum(v);
        if ((tid & 63) == 0) atomicAdd(&sDA[k], v);
      }
      // dH0[b,j,c] = A_jj dY[b,j,c] ;  dH1[b,j,c] = sum_{i != j} A_ij dY[b,i,c]
      for (int j = 0; j < J; ++j) {
        float h0 = 0.f, h1 = 0.f;
        for (int q = g.csc_ptr[j]; q < g.csc_ptr[j + 1]; ++q) {
          const int k = g.csc_edge[q], i = g.edge_row[k];
          const float t = sA[k] * sDy[i * kThreads + tid];
          if (i == j) h0 += t; else h1 += t;
        }
        dHb[j * ld + c] = h0;
        dHb[j * ld + C + c] = h1;
      }
    }
  }
  sRed[0][tid] = worker ? db : 0.f;
  __syncthreads();
  if (tid < C && dbias != nullptr) {
    float a = 0.f;
    for (int k = 0; k < nsl; ++k) a += sRed[0][k * C + tid];
    dbias[tid] = a;
  }
  // softmax backward per adjacency row: de = A (.) (dA - sum_k A dA)
  if (tid < J) {
    const int lo = g.row_ptr[tid], hi = g.row_ptr[tid + 1];
    float dot = 0.f;
    for (int k = lo; k < hi; ++k) dot = fmaf(sA[k], sDA[k], dot);
    for (int k = lo; k < hi; ++k) de[k] = sA[k] * (sDA[k] - dot);
  }
}

inline bool ok(int B, int J, int C, int E) {
  return B > 0 && J > 0 && J <= kMaxJ && C > 0 && C <= 128 && (kThreads % C) == 0 && (C % 64 == 0 || C == 128) &&
         E > 0 && E <= kMaxE;
}

}  // namespace

extern "C" {

int hcm_sgc_forward(const float* H, const float* e, const int* row_ptr, const int* col_idx,
                    const int* csc_ptr, const int* csc_edge, const int* edge_row, const float* bias,
                    const float* gamma, const float* beta, float* running_mean, float* running_var,
                    int B, int J, int C, int E, int has_bn, int relu, int training, float momentum,
                    float eps, float* out, float* xhat, float* invstd, float* A_out,
                    hcm_stream_t stream) {
  if (!ok(B, J, C, E)) return (int)hipErrorInvalidValue;
  Graph g{row_ptr, col_idx, csc_ptr, csc_edge, edge_row};
  sgc_fwd_kernel<<<1, kThreads, 0, (hipStream_t)stream>>>(H, e, g, bias, gamma, beta, running_mean,
                                                          running_var, B, J, C, E, has_bn, relu, training,
                                                          momentum, eps, out, xhat, invstd, A_out);
  HCM_CHECK_LAUNCH();
  return 0;
}

int hcm_sgc_backward(const float* dOut, const float* out, const float* xhat, const float* invstd,
                     const float* gamma, const float* A, const int* row_ptr, const int* col_idx,
                     const int* csc_ptr, const int* csc_edge, const int* edge_row, const float* H, int B,
                     int J, int C, int E, int has_bn, int relu, int training, float* dH, float* dgamma,
                     float* dbeta, float* dbias, float* de, hcm_stream_t stream) {
  if (!ok(B, J, C, E)) return (int)hipErrorInvalidValue;
  Graph g{row_ptr, col_idx, csc_ptr, csc_edge, edge_row};
  const size_t lds = (size_t)J * kThreads * sizeof(float);
  hipError_t er = hipFuncSetAttribute(reinterpret_cast<const void*>(sgc_bwd_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (er != hipSuccess) return (int)er;
  sgc_bwd_kernel<<<1, kThreads, lds, (hipStream_t)stream>>>(dOut, out, xhat, invstd, gamma, A, g, H, B, J, C, E,
                                                            has_bn, relu, training, dH, dgamma, dbeta, dbias, de);
  HCM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
