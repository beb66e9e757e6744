// Row 8 of SURVEY.md 8a -- merge_all_res + the 1x1 projection (networks/build_backbone.py:243-254, :290-300) -- at the
// sampled pixels, on the fp32 matrix cores (r04; r03 staged a [2, B R, 272] matrix for three rocBLAS GEMMs):
//
//   project_rows_kernel   forward.  A workgroup owns a run of rows of one (modality, image).  It stages the two
//                         coarsest branch maps of that image in LDS (coalesced 16-byte loads; at 256 x 256 they are 84 %
//                         of the 1026 stencil taps of a row), gathers / bilinearly samples 32 rows at a time into an
//                         LDS tile [32][Ctot | 1 | 0], and multiplies the tile by the [W | b] panel with
//                         v_mfma_f32_16x16x4_f32: each of the 8 waves owns 16 of the 128 output channels and holds its
//                         B fragments in registers for the whole launch.  Writes rows [2, B R, 128], zero-fills grows,
//                         and stores the sampled tile xs [2, B R, ld] once (the weight gradient reads it back).
//   proj_dw_partial_kernel, proj_dw_reduce_kernel
//                         d[W | b] = grows^T xs, split over row chunks (MFMA), then summed chunk by chunk in a fixed order
//                         (deterministic; no atomics), scaled, unpacked into dWp [F, Ctot] and dbp [F].
//
//   stencil_plan_kernel   backward, once per step: for every (image, branch) the (row, tap) stencil entries sorted by
//                         target pixel (bitonic sort of 32-bit keys pixel * E + entry in LDS) + per-pixel offsets: a CSR
//                         transpose of the sampling operator, shared by both modalities and all channels.
//   branch_grad_t_kernel  backward of sampling + projection + average pooling, owner computes, in the TRANSPOSED order:
//                         dX_i = W_i^T (S_i^T grows) instead of S_i^T (grows W): a workgroup owns 64 pixels of one
//                         branch map; its waves walk the pixels' entry lists and accumulate T[pixel][128] = sum of
//                         weight * grows[row] (512-byte coalesced row reads) into LDS, then multiply by the branch's
//                         slice of W on the matrix cores ([C_i x 128] x [128 x 64]) and store the map tile once, pooling
//                         gradient included.  No dxs matrix, no per-workgroup list building (r03: 70 % of the wave
//                         cycles of branch_grad_kernel were barrier / LDS waits of its list construction), linear in the
//                         length of a pixel's list (r03 was quadratic for a pixel sampled hundreds of times).
//                         r06: workgroups dispatched most expensive kind first; the walk of the entry lists without control
//                         flow (DPP row broadcasts of a 16-entry chunk, two named row buffers: 8-16 row loads in flight);
//                         the pooling gradient and the first A fragments requested with the offsets.
//   finest_rows_kernel, finest_tiles (inside branch_grad_t_kernel)
//                         r06: the finest branch in the reference's order, S_0^T (grows W_0) -- see the comment above them.
// Everything is fp32; an fp32 MFMA is an exact fmaf chain (MI355X_MICROARCH.md), sums run in a fixed order.
#include "hcm_common.h"
#include "../../include/hcmoco_hip.h"
#include "section_common.h"

namespace {

using namespace hcm;

typedef float v4f __attribute__((ext_vector_type(4)));

// n / d by one multiply: magic(d) = floor((2^32 - 1) / d) + 1 once (block-uniform), then the high word of n * magic.  Exact
// while n d < 2^32: n (magic d - 2^32) <= n d.  d = 1 has no 32-bit magic (0 stands for it).  (A run-time unsigned division is
// ~30 VALU instructions; the gather loops of the forward kernel did one per element.)
__device__ __forceinline__ unsigned div_magic(int d) { return (unsigned)(0xffffffffu / (unsigned)d) + 1u; }
__device__ __forceinline__ int fast_div(int n, unsigned magic) { return magic != 0u ? (int)__umulhi((unsigned)n, magic) : n; }

// Diagnostic build only (-DHCM_ROW8_TIMING, tools/probes/row8_timing.sh): s_memtime stamps of the phases of a workgroup,
// 16 slots per workgroup in a buffer the probe hands over.  Compiled out of the product library.
#ifdef HCM_ROW8_TIMING
__device__ unsigned long long* g_row8_dbg = nullptr;
#define HCM_STAMP(k)                                                                                              \
  do {                                                                                                            \
    if (g_row8_dbg != nullptr && threadIdx.x == 0)                                                                \
      g_row8_dbg[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + (k)] = clock64(); \
  } while (0)
#else
#define HCM_STAMP(k) do {} while (0)
#endif

constexpr int kPW = 512;     // forward workgroup: 8 waves, wave w owns output channels [16 w, 16 w + 16)
constexpr int kSR = 32;      // rows per LDS tile: two 16-row MFMA tiles = two independent accumulator chains per wave
constexpr int kF = 128;      // projected channels (the loss kernels' C)

// ---- channels-last copies of the branch maps the forward kernel gathers from global memory (r06; north_star's "coalesced
// reads of the modality feature maps", SURVEY 8f-1: "channels-last so each gather is one line").  The HRNet writes NCHW: the C_i
// values of one sampled pixel lie H_i W_i * 4 bytes apart (16 KB for the finest branch at 256 x 256), so a tap of a row costs
// C_i 4-byte accesses to C_i different cache lines (r04 counters: 7.9 L1 accesses per line requested from L2).  This kernel
// writes [B, H_i W_i, C_i] next to the NCHW map -- 64-pixel tiles through LDS, 256-byte runs in, fully contiguous runs out --
// for the branches that are NOT staged in LDS (the two finest at 256 x 256: 14 MB per modality), one launch per map;
// project_rows_kernel then reads ONE contiguous run of C_i floats per tap.
// one map: [B, C, HW] -> [B, HW, C]; a workgroup owns one (image, 64-pixel tile)
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int HW) {
  extern __shared__ float tl[];                   // [C][65]
  const int tiles = (HW + 63) >> 6;
  const int b = blockIdx.x / tiles, q0 = (blockIdx.x - b * tiles) << 6, nq = min(64, HW - q0);
  for (int e = threadIdx.x; e < C * 64; e += 256) {
    const int c = e >> 6, q = e & 63;
    if (q < nq) tl[c * 65 + q] = src[((size_t)b * C + c) * HW + q0 + q];
  }
  __syncthreads();
  float* out = dst + ((size_t)b * HW + q0) * C;
  for (int e = threadIdx.x; e < nq * C; e += 256) {
    const int q = e / C, c = e - q * C;
    out[e] = tl[c * 65 + q];
  }
}

// one channel of one row from a branch: a plain gather (finest branch) or the 4-tap bilinear stencil
struct RowTaps {
  int o00, o01, o10, o11;
  float hy, ly, hx, lx;
};
__device__ __forceinline__ float tap4(const float* __restrict__ xc, const RowTaps& t) {
  return t.hy * (t.hx * xc[t.o00] + t.lx * xc[t.o01]) + t.ly * (t.hx * xc[t.o10] + t.lx * xc[t.o11]);
}

// KS = k-steps of 4 channels: 4 KS >= Ctot + 1 (the bias column).  LDS: [kSR][4 KS + 2] tile, then the staged maps.
// The tile's row stride 4 KS + 2 makes the A-fragment read (lane (i, g) -> row i, column 4 s + g) conflict-free:
// (stride * i + g) mod 32 takes 32 different values for i < 16, g < 2.
template <int KS>
__global__ __launch_bounds__(kPW) void project_rows_kernel(
    Maps8 e, Maps8T et, int B, const int64_t* __restrict__ pix, int R, int Ctot, const float* __restrict__ Wp1,
    const float* __restrict__ bp1, const float* __restrict__ Wp2, const float* __restrict__ bp2,
    float* __restrict__ xs, int ld, float* __restrict__ rows, float* __restrict__ grows, int per, int off2, int off3) {
  constexpr int XS = 4 * KS + 2;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ __attribute__((aligned(16))) int s_to[kSR][4][4];      // the tile's stencils: tap offsets ...
  __shared__ __attribute__((aligned(16))) float s_tw[kSR][4][4];    // ... and (hy, ly, hx, lx); the finest branch is the
                                                                    // stencil (p, p, p, p) / (1, 0, 1, 0): hy (hx a + 0 a) + 0 = a
  // the branches gathered from GLOBAL memory (not staged in LDS), one entry per group k < ng: where its map starts for this
  // (modality, image), the strides of a pixel and of a channel in floats, its first column in the gathered run and in the tile,
  // its branch index.  Read per element by a lane-varying k: LDS reads, no control flow (r06: the ternary chains over the
  // kernel-argument arrays they replace compiled to branch trees, and every element's four tap loads were closed by
  // s_waitcnt vmcnt(0) -- 4 loads in flight per thread where the loop asks for 16)
  __shared__ const float* s_gptr[4];
  __shared__ int s_gmeta[4][6];                                      // pixel stride, channel stride, first cg, tile column, branch, -
  float* lx = lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = blockIdx.z, b = blockIdx.y;
  const int r_begin = blockIdx.x * per, r_end = min(R, r_begin + per);
  if (r_begin >= r_end) return;
  HCM_STAMP(0);
  const int n = lane & 15, g = lane >> 4;
  // ---- stage the coarse maps of this image: plane stride odd, so that the channel lanes of a tap hit different banks
  const int h0 = e.H[0], w0 = e.W[0];
#pragma unroll
  for (int i = 2; i < 4; ++i) {
    const int off = i == 2 ? off2 : off3;
    if (off < 0) continue;
    const int C = e.C[i], hw = e.H[i] * e.W[i], P = hw | 1;
    const float* src = (m ? e.p[4 + i] : e.p[i]) + (int64_t)b * C * hw;
    float* dst = lds + off;
    if ((hw & 3) == 0) {
      const int n4 = (C * hw) >> 2;
      const unsigned mw = div_magic(hw);
      for (int base = tid; base < n4; base += 16 * kPW) {         // sixteen 16-byte loads in flight per thread
        float4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = reinterpret_cast<const float4*>(src)[min(base + u * kPW, n4 - 1)];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          // (only the loads are clamped: a clamped WRITE is the whole workgroup storing to one LDS address, a 64-way bank
          // conflict per instruction -- r06 stamps: 40 k cycles of staging, 38 k of them the 18 clamped slots x 4 words x 8 waves)
          if (base + u * kPW < n4) {
            const int e0 = (base + u * kPW) << 2, c = fast_div(e0, mw), q = e0 - c * hw;
            float* d = dst + c * P + q;
            d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
          }
        }
      }
    } else {
      for (int e0 = tid; e0 < C * hw; e0 += kPW) {
        const int c = e0 / hw, q = e0 - c * hw;
        dst[c * P + q] = src[e0];
      }
    }
  }
  HCM_STAMP(12);
  // ---- the wave's B fragments: [W | b | 0]^T, column 16 wave + n, rows 4 s + g
  float breg[KS];
  {
    const float* Wp = (m ? Wp2 : Wp1) + (int64_t)(16 * wave + n) * Ctot;
    const float bias = (m ? bp2 : bp1)[16 * wave + n];
    // unconditional loads (clamped index), selected afterwards: a guarded load is a branch plus a full s_waitcnt each
#pragma unroll
    for (int s = 0; s < KS; ++s) breg[s] = Wp[min(4 * s + g, Ctot - 1)];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = 4 * s + g;
      breg[s] = k < Ctot ? breg[s] : (k == Ctot ? bias : 0.f);
    }
  }
  HCM_STAMP(13);
  const int64_t row0 = ((int64_t)m * B + b) * R;
  int g_ng = 0, g_ctot = 0, g_end0 = 0, g_end1 = 0, g_end2 = 0;
  {
    int col = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool staged = (i == 2 && off2 >= 0) || (i == 3 && off3 >= 0);
      const int C = e.C[i], hw = e.H[i] * e.W[i];
      if (!staged) {
        if (tid == 0) {
          const float* xt = m ? et.t[4 + i] : et.t[i];
          const float* xp = m ? e.p[4 + i] : e.p[i];
          s_gptr[g_ng] = xt != nullptr ? xt + (int64_t)b * hw * C : xp + (int64_t)b * C * hw;
          s_gmeta[g_ng][0] = xt != nullptr ? C : 1;
          s_gmeta[g_ng][1] = xt != nullptr ? 1 : hw;
          s_gmeta[g_ng][2] = g_ctot;
          s_gmeta[g_ng][3] = col;
          s_gmeta[g_ng][4] = i;
        }
        g_ctot += C;
        if (g_ng == 0) g_end0 = g_ctot;
        if (g_ng <= 1) g_end1 = g_ctot;
        if (g_ng <= 2) g_end2 = g_ctot;
        ++g_ng;
      }
      col += C;
    }
  }
  for (int r0 = r_begin; r0 < r_end; r0 += kSR) {
    __syncthreads();        // the staged maps (and the group table) are in place / the previous tile has been multiplied
    if (r0 == r_begin) HCM_STAMP(1);
    // ---- the tile's stencils: one thread per (row, branch)
    if (tid < kSR * 4) {
      const int lr = tid >> 2, br = tid & 3, r = min(r0 + lr, r_end - 1);
      const int p = (int)pix[(int64_t)b * R + r];
      int* o = &s_to[lr][br][0];
      float* w = &s_tw[lr][br][0];
      if (br == 0) {
        o[0] = p; o[1] = p; o[2] = p; o[3] = p;
        w[0] = 1.f; w[1] = 0.f; w[2] = 1.f; w[3] = 0.f;
      } else {
        const int py = p / w0, px = p - py * w0;
        const int hi = sel4(e.H, br), wi = sel4(e.W, br);
        const Taps t = bilinear_taps(py, px, hi, wi, (float)hi / (float)h0, (float)wi / (float)w0);
        o[0] = t.y0 * wi + t.x0; o[1] = t.y0 * wi + t.x1; o[2] = t.y1 * wi + t.x0; o[3] = t.y1 * wi + t.x1;
        w[0] = t.hy; w[1] = t.ly; w[2] = t.hx; w[3] = t.lx;
      }
    }
    __syncthreads();
    // ---- gather.  Every (row, channel) of the tile is one independent element; consecutive threads take consecutive
    // channels of a row.  Rows past the end repeat the last row (computed, never stored).  Branches that are not in LDS
    // first: the 16 loads of four elements per thread are requested together, with NO store between them -- on this
    // part loads and stores retire through one in-order counter (vmcnt), so a store issued between two loads makes the
    // second load's wait include the store's round trip (r04 stamps: 23 k cycles per tile with an xs store after every
    // element).  xs is written once per tile from LDS, after the barrier, and retires under the MFMA phase.
    {
      const int ctot = g_ctot, total = kSR * ctot;
      const unsigned mg = div_magic(max(ctot, 1));
      for (int base = tid; base < total; base += 4 * kPW) {
        float rv[4][4];
        int dst[4];
        float4 wv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int ge = min(base + u * kPW, total - 1);
          const int lr = fast_div(ge, mg), cg = ge - lr * ctot;
          const int k = (cg >= g_end0) + (cg >= g_end1) + (cg >= g_end2);       // < ng: the ends past the last group equal ctot
          const int st = s_gmeta[k][0], cs = s_gmeta[k][1], c = cg - s_gmeta[k][2], i = s_gmeta[k][4];
          const int4 o = *reinterpret_cast<const int4*>(&s_to[lr][i][0]);
          wv[u] = *reinterpret_cast<const float4*>(&s_tw[lr][i][0]);
          // channels-last copy present (r06): consecutive threads = consecutive channels = consecutive addresses, one
          // contiguous run of C floats per tap; otherwise the NCHW plane walk (one word per line).
          // All four taps UNCONDITIONALLY: the finest branch's stencil is (p, p, p, p) / (1, 0, 1, 0), its three extra loads are
          // cache hits on the word just requested.
          // (a pointer that comes out of LDS has no address space for the compiler: say it is global, or the taps are flat loads)
          typedef const float __attribute__((address_space(1))) gfloat;
          const gfloat* x = (const gfloat*)(s_gptr[k] + (int64_t)c * cs);
          rv[u][0] = x[o.x * st];
          rv[u][1] = x[o.y * st];
          rv[u][2] = x[o.z * st];
          rv[u][3] = x[o.w * st];
          dst[u] = lr * XS + s_gmeta[k][3] + c;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          lx[dst[u]] = wv[u].x * (wv[u].z * rv[u][0] + wv[u].w * rv[u][1]) + wv[u].y * (wv[u].z * rv[u][2] + wv[u].w * rv[u][3]);
      }
      int coff = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int C = e.C[i], hw = e.H[i] * e.W[i];
        const int off = i == 2 ? off2 : (i == 3 ? off3 : -1);
        if (off >= 0) {                                     // block-uniform
          const float* xl = lds + off;
          const int P = hw | 1, tot = kSR * C;
          const unsigned mc = div_magic(C);
#pragma unroll 4
          for (int e0 = tid; e0 < tot + kPW - 1 - (tot + kPW - 1) % kPW; e0 += kPW) {
            const int ec = min(e0, tot - 1);
            const int lr = fast_div(ec, mc), c = ec - lr * C;
            const int4 o = *reinterpret_cast<const int4*>(&s_to[lr][i][0]);
            const float4 w = *reinterpret_cast<const float4*>(&s_tw[lr][i][0]);
            RowTaps q;
            q.o00 = o.x; q.o01 = o.y; q.o10 = o.z; q.o11 = o.w;
            q.hy = w.x; q.ly = w.y; q.hx = w.z; q.lx = w.w;
            lx[lr * XS + coff + c] = tap4(xl + c * P, q);
          }
        }
        coff += C;
      }
      const int npad = 4 * KS - Ctot;                          // the bias column, then padding
      const unsigned mp = div_magic(npad);
      for (int e0 = tid; e0 < kSR * npad; e0 += kPW) {
        const int lr = fast_div(e0, mp), k = Ctot + e0 - lr * npad;
        lx[lr * XS + k] = k == Ctot ? 1.f : 0.f;
      }
    }
    __syncthreads();
    HCM_STAMP(2 + 2 * ((r0 - r_begin) / kSR));
    if (xs != nullptr) {                                       // the sampled rows, coalesced 8-byte stores out of the tile
      const int h2 = ld >> 1, nrow = min(kSR, r_end - r0);
      const unsigned mh = div_magic(h2);
      for (int e0 = tid; e0 < nrow * h2; e0 += kPW) {
        const int lr = fast_div(e0, mh), k2 = e0 - lr * h2;
        *reinterpret_cast<float2*>(xs + (row0 + r0 + lr) * ld + 2 * k2) = *reinterpret_cast<const float2*>(lx + lr * XS + 2 * k2);
      }
    }
    // ---- multiply: two 16-row tiles, one accumulator chain each (an fp32 MFMA depends on its predecessor for 40 cycles
    // and issues every 32: two chains keep the pipe full with the second wave of the SIMD as further cover)
    v4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const float* a0 = lx + n * XS + g;
    const float* a1 = a0 + 16 * XS;
    const bool two = r0 + 16 < r_end;                           // block-uniform (the second tile is multiplied regardless)
    // The A fragments of the next kCH k-steps are requested from LDS before the MFMAs of the current kCH issue (left to
    // itself the compiler reads each fragment right in front of its MFMA: LDS latency + the 40-cycle accumulator
    // dependency per k-step instead of 32 cycles of issue)
    constexpr int kCH = KS > 121 ? 4 : 16, kNC = (KS + kCH - 1) / kCH;      // (w48: 181 B fragments leave room for 8)
    float fa[2][kCH], fb[2][kCH];
#pragma unroll
    for (int u = 0; u < kCH; ++u) { fa[0][u] = a0[4 * u]; fb[0][u] = a1[4 * u]; }
#pragma unroll
    for (int c = 0; c < kNC; ++c) {
      if (c + 1 < kNC) {
#pragma unroll
        for (int u = 0; u < kCH; ++u)
          if ((c + 1) * kCH + u < KS) { fa[(c + 1) & 1][u] = a0[4 * ((c + 1) * kCH + u)]; fb[(c + 1) & 1][u] = a1[4 * ((c + 1) * kCH + u)]; }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < kCH; ++u)
        if (c * kCH + u < KS) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[c & 1][u], breg[c * kCH + u], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[c & 1][u], breg[c * kCH + u], acc1, 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    // D: lane (g, n) holds rows 4 g + j, column n
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ra = r0 + 4 * g + j, rb = ra + 16;
      if (ra < r_end) {
        const int64_t o = (row0 + ra) * kF + 16 * wave + n;
        rows[o] = acc0[j];
        if (grows != nullptr) grows[o] = 0.f;
      }
      if (two && rb < r_end) {
        const int64_t o = (row0 + rb) * kF + 16 * wave + n;
        rows[o] = acc1[j];
        if (grows != nullptr) grows[o] = 0.f;
      }
    }
    HCM_STAMP(3 + 2 * ((r0 - r_begin) / kSR));
  }
}

// ------------------------------------------------------------------------------------------
// d[W | b] partials: grid (chunks, column groups, 2).  A workgroup (8 waves) reduces `rpc` rows of one modality into a
// [128][<= 16 kNT] block: wave w owns output channels [16 w, 16 w + 16) (A = grows^T, straight from global memory: lane
// (i, g) reads grows[row 4 s + g][16 w + i], 64 contiguous bytes per g) times up to kNT 16-column tiles of xs.
// ------------------------------------------------------------------------------------------
constexpr int kNT = 17;      // column tiles per workgroup: 272 = the HRNet-w18 row (270 channels + bias + pad)
constexpr int kDW = 256;     // 4 waves: wave w owns output channels [32 w, 32 w + 32) = two 16-row tiles x kNT column tiles
constexpr int kRing = 3;     // k-steps of fragments in flight per wave (one wave per SIMD: the ring is the latency cover)
__global__ __launch_bounds__(kDW, 1) void proj_dw_partial_kernel(const float* __restrict__ grows,
                                                                 const float* __restrict__ xs, int M, int ld, int rpc,
                                                                 float* __restrict__ part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int chunk = blockIdx.x, cg = blockIdx.y, m = blockIdx.z, nchunk = gridDim.x;
  const int col0 = cg * kNT * 16;
  const int nt = min(kNT, (ld - col0 + 15) / 16);
  const int r_begin = chunk * rpc, r_end = min(M, r_begin + rpc);
  const float* A = grows + (int64_t)m * M * kF + 32 * wave + n;
  const float* Bm = xs + (int64_t)m * M * ld + col0 + n;
  v4f acc[2][kNT];
#pragma unroll
  for (int t = 0; t < kNT; ++t) { acc[0][t] = (v4f){0.f, 0.f, 0.f, 0.f}; acc[1][t] = (v4f){0.f, 0.f, 0.f, 0.f}; }
  // Every load is unconditional (a row past the end is clamped and its A values zeroed, a column past ld is clamped and
  // its product never stored): a guarded load is a branch and a full s_waitcnt in the middle of the stream.
  int cofs[kNT];
#pragma unroll
  for (int t = 0; t < kNT; ++t) cofs[t] = min(col0 + 16 * t + n, ld - 1) - col0 - n;
  float ra[kRing][2], rb[kRing][kNT];
  auto load = [&](int r, float (&a)[2], float (&bv)[kNT]) {
    const int rg = min(r + g, M - 1);
    const bool ok = r + g < r_end;
    const float a0 = A[(int64_t)rg * kF], a1 = A[(int64_t)rg * kF + 16];
    a[0] = ok ? a0 : 0.f;
    a[1] = ok ? a1 : 0.f;
    const float* brow = Bm + (int64_t)rg * ld;
#pragma unroll
    for (int t = 0; t < kNT; ++t) bv[t] = brow[cofs[t]];
  };
#pragma unroll
  for (int j = 0; j < kRing; ++j) load(r_begin + 4 * j, ra[j], rb[j]);
  for (int r = r_begin; r < r_end; r += 4 * kRing) {
#pragma unroll
    for (int j = 0; j < kRing; ++j) {
      if (r + 4 * j < r_end) {                        // block-uniform
        float a0 = ra[j][0], a1 = ra[j][1], bv[kNT];
#pragma unroll
        for (int t = 0; t < kNT; ++t) bv[t] = rb[j][t];
        load(r + 4 * (j + kRing), ra[j], rb[j]);      // the slot is free again: request k-step r + 4 (j + kRing)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < kNT; ++t) {
          acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv[t], acc[0][t], 0, 0, 0);
          acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv[t], acc[1][t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  // part [2][nchunk][128][ld]; D: lane (g, n) holds output channels 32 wave + 16 h + 4 g + j, column 16 t + n
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float* out = part + (((int64_t)m * nchunk + chunk) * kF + 32 * wave + 16 * h + 4 * g) * ld + col0 + n;
#pragma unroll
    for (int t = 0; t < kNT; ++t) {
      if (t < nt && col0 + 16 * t + n < ld) {
#pragma unroll
        for (int j = 0; j < 4; ++j) out[(int64_t)j * ld + 16 * t] = acc[h][t][j];
      }
    }
  }
}

// one thread per (modality, output channel, column): chunks summed in ascending order
__global__ __launch_bounds__(256) void proj_dw_reduce_kernel(const float* __restrict__ part, int nchunk, int ld, int Ctot,
                                                             const float* __restrict__ scale, float* __restrict__ dWp1,
                                                             float* __restrict__ dbp1, float* __restrict__ dWp2,
                                                             float* __restrict__ dbp2) {
  const int e = blockIdx.x * 256 + threadIdx.x, m = blockIdx.y;
  if (e >= kF * ld) return;
  const int o = e / ld, k = e - o * ld;
  if (k > Ctot) return;
  const float* p = part + (int64_t)m * nchunk * kF * ld + e;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int c = 0;
  // sixteen loads in flight (r06: four per trip were 32 dependent round trips for 128 chunks, 12 us for 35 MB); the adds in the
  // same order as the four-at-a-time loop below: the association is fixed by the chunk count alone, the sums bit-identical
  for (; c + 15 < nchunk; c += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = p[(int64_t)(c + u) * kF * ld];
#pragma unroll
    for (int u = 0; u < 16; u += 4) { s0 += v[u]; s1 += v[u + 1]; s2 += v[u + 2]; s3 += v[u + 3]; }
  }
  for (; c + 3 < nchunk; c += 4) {
    s0 += p[(int64_t)c * kF * ld];
    s1 += p[(int64_t)(c + 1) * kF * ld];
    s2 += p[(int64_t)(c + 2) * kF * ld];
    s3 += p[(int64_t)(c + 3) * kF * ld];
  }
  for (; c < nchunk; ++c) s0 += p[(int64_t)c * kF * ld];
  const float v = (scale != nullptr ? scale[0] : 1.f) * ((s0 + s1) + (s2 + s3));
  float* dW = m ? dWp2 : dWp1;
  float* db = m ? dbp2 : dbp1;
  if (k < Ctot) dW[(int64_t)o * Ctot + k] = v;
  else db[o] = v;
}

// ------------------------------------------------------------------------------------------
// Stencil plan.  grid (4 branches, B), 256 threads, dynamic LDS = N2 keys.  Entry e of image b, branch i: branch 0 has one
// entry per row (e = r: the gather), branches 1-3 four (e = 4 r + tap, tap order y0x0, y0x1, y1x0, y1x1 = the order the
// forward adds them).  key = pixel * E + e: ascending keys = pixels in raster order, inside a pixel ascending (row, tap) --
// the summation order of the gradient.  The first S rows of a dropped image (keep[b] == 0) are left out: their gradient is
// exactly zero and they all sit on pixel 0.
// ------------------------------------------------------------------------------------------
constexpr int kRowBits = 14;  // plan entry .x = row | pixel << 14: R < 16384, H W < 2^17
struct PlanGeom {
  int H[4], W[4];
  int ent_off[4], off_off[4];      // per-image offsets of branch i inside ent (entries) / off (ints)
  int ent_per_image, off_per_image;
};

constexpr int kPT = 1024;     // plan workgroup
__global__ __launch_bounds__(kPT) void stencil_plan_kernel(const int64_t* __restrict__ pix, int R, PlanGeom gm,
                                                           const int32_t* __restrict__ keep, int S, int N2max,
                                                           int2* __restrict__ ent, int* __restrict__ off) {
  extern __shared__ uint32_t keys[];
  const int tid = threadIdx.x, i = blockIdx.x, b = blockIdx.y;
  const int hi = sel4(gm.H, i), wi = sel4(gm.W, i), hw = hi * wi;
  const int h0 = gm.H[0], w0 = gm.W[0];
  const int taps = i == 0 ? 1 : 4, E = taps * R;
  int N2 = 128;                                     // a wave sorts runs of 128 keys on its own
  while (N2 < E) N2 <<= 1;
  (void)N2max;
  const bool dropped = keep != nullptr && keep[b] == 0;
  const float sy = (float)hi / (float)h0, sx = (float)wi / (float)w0;
  for (int e = tid; e < N2; e += kPT) {
    uint32_t key = 0xffffffffu;
    if (e < E) {
      const int r = e / taps, t = e - r * taps;
      if (!(dropped && r < S)) {
        const int p = (int)pix[(int64_t)b * R + r];
        int q = p;
        if (i > 0) {
          const int py = p / w0, px = p - py * w0;
          const Taps tp = bilinear_taps(py, px, hi, wi, sy, sx);
          q = ((t & 2) ? tp.y1 : tp.y0) * wi + ((t & 1) ? tp.x1 : tp.x0);
        }
        key = (uint32_t)q * (uint32_t)E + (uint32_t)e;
      }
    }
    keys[e] = key;
  }
  __syncthreads();
  // Bitonic network.  Thread t handles pair t of each 2 kPT-key slab.  A compare-exchange distance j < 64 keeps both
  // keys inside the 128-key run of the thread's own wave (pair index t -> keys with the same t / 64), and LDS requests of
  // one wave complete in order: those stages need no workgroup barrier, only the stages with j >= 64 do.
  for (int k = 2; k <= N2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (N2 >> 1); t += kPT) {
        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi2 = lo | j;
        const uint32_t a = keys[lo], c = keys[hi2];
        const bool up = (lo & k) == 0;
        if ((a > c) == up) { keys[lo] = c; keys[hi2] = a; }
      }
      if (j >= 64) __syncthreads();
      else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
    if (k >= 128) __syncthreads();                  // the next phase starts at distance k: other waves' runs
  }
  __syncthreads();
  int2* eo = ent + (int64_t)b * gm.ent_per_image + sel4(gm.ent_off, i);
  int* oo = off + (int64_t)b * gm.off_per_image + sel4(gm.off_off, i);
  for (int j = tid; j < E; j += kPT) {
    const uint32_t key = keys[j];
    if (key == 0xffffffffu) continue;
    const int e = (int)(key % (uint32_t)E), r = e / taps, t = e - r * taps;
    float w = 1.f;
    if (i > 0) {
      const int p = (int)pix[(int64_t)b * R + r];
      const int py = p / w0, px = p - py * w0;
      const Taps tp = bilinear_taps(py, px, hi, wi, sy, sx);
      w = ((t & 2) ? tp.ly : tp.hy) * ((t & 1) ? tp.lx : tp.hx);
    }
    eo[j] = make_int2(r | ((int)(key / (uint32_t)E) << kRowBits), __builtin_bit_cast(int, w));
  }
  // off[q] = number of entries with pixel < q (lower bound of q * E), q = 0 .. hw
  for (int q = tid; q <= hw; q += kPT) {
    const uint64_t want = (uint64_t)q * (uint64_t)E;
    int lo = 0, hi2 = E;                         // first j with keys[j] >= want (padding keys are the largest)
    while (lo < hi2) {
      const int mid = (lo + hi2) >> 1;
      if ((uint64_t)keys[mid] < want) lo = mid + 1; else hi2 = mid;
    }
    oo[q] = lo;
  }
}

// ------------------------------------------------------------------------------------------
// Branch gradients, transposed order.  grid (B, modalities, tiles of all four branches: the most expensive kind first), 256 threads.
// ------------------------------------------------------------------------------------------
constexpr int kTP = 64;       // pixels per sub-tile (a workgroup walks `span` of them)
constexpr int kTS = 130;      // LDS row stride of T: the B-fragment read (lane (g, n) -> T[n][4 s + g]) is conflict-free
struct TilePlan {
  int first[5];               // first workgroup (blockIdx.z) of the k-th branch in dispatch order; first[4] = total
  int ord[4];                 // the branch dispatched k-th: most expensive workgroups first (the kernel's tail is its last workgroups)
  int span[4];                // sub-tiles per workgroup of branch i
  int tpix[4];                // pixels per sub-tile of branch i: 64, or 16 on the small maps, where every row of the image lands in
                              // every tile (a 64-pixel tile of an 8 x 8 map collects all 4 R stencil entries: r04 timing stamps
                              // showed that workgroup living 141 k cycles against a mean of 25 k)
};

constexpr int kSpanMax = 4;   // sub-tiles per workgroup on the large (mostly empty) maps
constexpr int kGW = 256;      // 4 waves; <= 256 VGPRs -> two workgroups per SIMD set (the row ring of the T phase is the cover)
// Where feature f of a pixel sits inside its row of T (r06 counters: SQ_LDS_BANK_CONFLICT 5.0 M quad-cycles against 2.5 M of LDS
// activity).  A lane of the walk owns features 8 l .. 8 l + 7 (its 32 bytes of a grows row) and stores them as four 8-byte pairs;
// in feature order lanes l and l + 8 of a 16-lane store hit the same banks (8 l mod 64).  Pair k of lane l goes to words
// 2 l + 32 k (+ 0 / 1): sixteen lanes = 32 consecutive words.  The multiply's B-fragment read (lane (g, n): feature 4 s + g of pixel
// n) then sees banks 2 n + {0, 1 | 32, 33}: conflict-free in each half wave like the plain order was.
__host__ __device__ constexpr int t_col(int f) { return 2 * (f >> 3) + 32 * ((f & 7) >> 1) + (f & 1); }
constexpr int kSE = 4;        // stencil entries per stage of the T phase (two stages in flight per 16-lane group)
template <int K>
__device__ __forceinline__ int row_bcast(int v) {       // lane K of every DPP row (16 lanes) to all lanes of that row
  return __builtin_amdgcn_update_dpp(0, v, 0x150 + K, 0xf, 0xf, false);
}
// the 32 bytes this lane owns of the grows rows of entries K0 .. K0 + 3 of the group's chunk (entry l is held by lane l)
template <int K0>
__device__ __forceinline__ void fetch_rows(int ex, const float* __restrict__ gl, float4 (&r)[kSE][2]) {
  const int rows[kSE] = {row_bcast<K0>(ex), row_bcast<K0 + 1>(ex), row_bcast<K0 + 2>(ex), row_bcast<K0 + 3>(ex)};
#pragma unroll
  for (int u = 0; u < kSE; ++u) {
    const float* rp = gl + (int64_t)(rows[u] & ((1 << kRowBits) - 1)) * kF;
    r[u][0] = *reinterpret_cast<const float4*>(rp);
    r[u][1] = *reinterpret_cast<const float4*>(rp + 4);
  }
}
template <int K0>
__device__ __forceinline__ void add_rows(int2 ech, int j, int jend, int q0, int spare, float* __restrict__ tl,
                                         const float4 (&r)[kSE][2], float (&acc)[8], int& cur) {
  const int ex[kSE] = {row_bcast<K0>(ech.x), row_bcast<K0 + 1>(ech.x), row_bcast<K0 + 2>(ech.x), row_bcast<K0 + 3>(ech.x)};
  const int ew[kSE] = {row_bcast<K0>(ech.y), row_bcast<K0 + 1>(ech.y), row_bcast<K0 + 2>(ech.y), row_bcast<K0 + 3>(ech.y)};
#pragma unroll
  for (int u = 0; u < kSE; ++u) {
    const int px = j + K0 + u < jend ? (ex[u] >> kRowBits) - q0 : spare;
    const float w = __builtin_bit_cast(float, ew[u]);
    const bool fresh = px != cur;
    cur = px;
    const float v[8] = {r[u][0].x, r[u][0].y, r[u][0].z, r[u][0].w, r[u][1].x, r[u][1].y, r[u][1].z, r[u][1].w};
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = fmaf(w, v[k], fresh ? 0.f : acc[k]);
    float* z = tl + px * kTS;
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<float2*>(z + 32 * k) = make_float2(acc[2 * k], acc[2 * k + 1]);    // t_col(8 l + 2 k)
  }
}

template <int TPX>
__device__ __forceinline__ void branch_grad_tiles(
    float* __restrict__ T, int* soff, float* __restrict__ pool_l, const float* __restrict__ grows, const float* __restrict__ Wp1,
    const float* __restrict__ Wp2, const float* __restrict__ dpooled, const float* __restrict__ scale,
    const int2* __restrict__ ent, const int* __restrict__ off, int R, int B, int Ctot, const Maps8Out& g,
    const TilePlan& tp, const PlanGeom& gm) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x, m = blockIdx.y;
  int k0 = 0;
  while (k0 < 3 && (int)blockIdx.z >= tp.first[k0 + 1]) ++k0;
  const int i = sel4(tp.ord, k0);
  const int wg = blockIdx.z - (k0 == 0 ? tp.first[0] : k0 == 1 ? tp.first[1] : k0 == 2 ? tp.first[2] : tp.first[3]);
  const int span = sel4(tp.span, i);
  constexpr int tpx = TPX, ppg = TPX >> 4, ntl = TPX >> 4;     // pixels per 16-lane group, 16-pixel MFMA tiles per sub-tile
  const int C = sel4(g.C, i), hw = sel4(g.H, i) * sel4(g.W, i);
  int coff = 0;
  for (int k = 0; k < i; ++k) coff += sel4(g.C, k);
  const int* offb = off + (int64_t)b * gm.off_per_image + sel4(gm.off_off, i);
  const int2* entb = ent + (int64_t)b * gm.ent_per_image + sel4(gm.ent_off, i);
  const int n = lane & 15, gq = lane >> 4;
  const float sc = scale != nullptr ? scale[0] : 1.f;
  const float inv = 1.f / (float)hw;
  const float* Wp = (m ? Wp2 : Wp1) + coff;
  const int nmt = (C + 15) >> 4;
  float* out = sel8(g.p, m * 4 + i) + (int64_t)b * C * hw;
  const float* dp = dpooled != nullptr ? dpooled + ((int64_t)m * B + b) * Ctot + coff : nullptr;
  HCM_STAMP(0);
  // the per-pixel offsets of ALL the workgroup's sub-tiles in one round trip (kSpanMax * 64 + 1 <= 257 words): a sub-tile
  // without entries then costs its stores and nothing else (r04 stamps: an empty sub-tile alone, as its own workgroup, lived
  // 5.4 k cycles, most of it this load; the finest maps are mostly empty tiles)
  int* const soff_all = soff;
  for (int e0 = tid; e0 <= span * tpx; e0 += kGW) soff_all[e0] = offb[min(wg * span * tpx + e0, hw)];
  // (and the pooling gradient of the branch's channels: read after the multiply it was a dependent global load per round)
  for (int c = tid; c < C; c += kGW) pool_l[c] = dp != nullptr ? dp[c] * inv : 0.f;
  HCM_STAMP(6);
  __syncthreads();
  bool used_t = false;
  for (int sub = 0; sub < span; ++sub) {
    const int q0 = (wg * span + sub) * tpx;
    if (q0 >= hw) break;
    soff = soff_all + sub * tpx;
    const bool any = soff[tpx] > soff[0];           // block-uniform
    if (any && used_t) __syncthreads();             // the previous sub-tile's products have been formed
    used_t = used_t || any;
    HCM_STAMP(1);
    if (!any) {
      // nothing sampled here: the pooling gradient alone, 16-byte stores
      if ((hw & 3) == 0) {
        for (int e0 = tid; e0 < C * (tpx >> 2); e0 += kGW) {
          const int c = e0 / (tpx >> 2), q = q0 + 4 * (e0 - c * (tpx >> 2));
          if (q < hw) {
            const float pool = pool_l[c];
            *reinterpret_cast<float4*>(out + (int64_t)c * hw + q) = make_float4(pool, pool, pool, pool);
          }
        }
      } else {
        for (int e0 = tid; e0 < C * tpx; e0 += kGW) {
          const int c = e0 / tpx, q = q0 + e0 - c * tpx;
          if (q < hw) out[(int64_t)c * hw + q] = pool_l[c];
        }
      }
      HCM_STAMP(4);
      continue;
    }
    // (lane coordinates the optimiser cannot see through: it otherwise lifts the ~40 64-bit addresses of the multiply out of
    // the sub-tile loop and they live, spilled to scratch, across the T phase)
    int nn = n, gg = gq;
    asm volatile("" : "+v"(nn), "+v"(gg));
    auto load_a = [&](int mt, float (&a)[32]) {
      const unsigned lo = (unsigned)(gg * Ctot + min(16 * mt + nn, C - 1));   // uniform base + 32-bit lane offset: one
#pragma unroll                                                                  // VGPR of address for all 32 loads
      for (int s = 0; s < 32; ++s) a[s] = (Wp + (size_t)(4 * s) * Ctot)[lo];
    };
    // the A fragments of the wave's first channel tile are requested HERE: their round trip runs under the T walk (r06
    // stamps: ~2 k cycles of every multiply round were this latency)
    float a0[32];
    load_a(min(wave, nmt - 1), a0);
    // ---- T[pixel][f] = sum over the pixel's entries of weight * grows[row][f], entries in (row, tap) order.  Sixteen
    // lanes (one DPP row, 32 bytes of the 512-byte row each) own `ppg` consecutive pixels = ONE run of the sorted entry list;
    // the sixteen groups of the workgroup walk their runs side by side.  The walk has NO control flow inside a trip (r06;
    // the ISA of the r04 loop showed a `vmcnt(0)` on its back edge -- register copies of a ring with loads in flight -- and a
    // scratch reload + `vmcnt(0)` inside the "next pixel begins" branch: every pixel change drained the ring):
    //   * lane l of a group loads entry j + l of the run -- ONE 8-byte load per sixteen entries, the next sixteen a trip ahead --
    //     and an entry reaches the group's lanes through a DPP row broadcast (no memory, no LDS);
    //   * the rows of four entries are requested while the previous four are added (two named buffers, trip unrolled four
    //     stages deep: no copies);
    //   * "a new pixel begins" is a select on the accumulator (restart from 0) and the running sum is stored to T after
    //     EVERY entry -- the last store of a pixel is its sum, in the same (row, tap) order with the same fmaf chain as before:
    //     bit-identical -- and an entry beyond the run adds into the group's own spare row of T.
    {
      const int grp = tid >> 4, l16 = tid & 15;
#pragma unroll
      for (int k = 0; k < ppg; ++k) {
        float* z = &T[(grp * ppg + k) * kTS + 2 * l16];
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<float2*>(z + 32 * u) = make_float2(0.f, 0.f);
      }
      int j = soff[grp * ppg];
      const int jend = soff[grp * ppg + ppg], jlast = soff[tpx] - 1;      // jlast >= 0: the sub-tile has entries
      const float* gl = grows + ((int64_t)m * B + b) * R * kF + 8 * l16;
      const int spare = kTP + grp;                                        // this group's spare row of T
      float* const tl = T + 2 * l16;                                      // t_col(8 l16)
      float4 ra[kSE][2], rb[kSE][2];
      float acc[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = 0.f;
      int cur = spare;
      int2 ech = entb[min(j + l16, jlast)];
      fetch_rows<0>(ech.x, gl, ra);
      while (j < jend) {
        const int2 enx = entb[min(j + 16 + l16, jlast)];
        fetch_rows<4>(ech.x, gl, rb);
        asm volatile("" : "+v"(ech.x), "+v"(ech.y));      // (left alone all sixteen broadcasts are lifted to the top of the trip)
        add_rows<0>(ech, j, jend, q0, spare, tl, ra, acc, cur);
        asm volatile("" : "+v"(ech.x), "+v"(ech.y));
        fetch_rows<8>(ech.x, gl, ra);
        asm volatile("" : "+v"(ech.x), "+v"(ech.y));
        add_rows<4>(ech, j, jend, q0, spare, tl, rb, acc, cur);
        asm volatile("" : "+v"(ech.x), "+v"(ech.y));
        fetch_rows<12>(ech.x, gl, rb);
        asm volatile("" : "+v"(ech.x), "+v"(ech.y));
        add_rows<8>(ech, j, jend, q0, spare, tl, ra, acc, cur);
        asm volatile("" : "+v"(ech.x), "+v"(ech.y));
        fetch_rows<0>(enx.x, gl, ra);
        asm volatile("" : "+v"(ech.x), "+v"(ech.y));
        add_rows<12>(ech, j, jend, q0, spare, tl, rb, acc, cur);
        ech = enx;
        j += 16;
      }
    }
    __syncthreads();
    HCM_STAMP(2);
    // ---- dX[c][q] = sum_f W[f][coff + c] T[q][f]: a task = one 16-channel tile x the four 16-pixel tiles of the sub-tile
    // (four accumulator chains sharing the A fragments).  A = W^T straight from global memory (64-byte runs per lane
    // group); B = T from LDS, four k-steps ahead.
    {
      for (int mt = wave; mt < nmt; mt += kGW / 64) {
        float a[32];
        if (mt == wave) {
#pragma unroll
          for (int s = 0; s < 32; ++s) a[s] = a0[s];
        } else {
          load_a(mt, a);
        }
        if (16 * mt + nn >= C) {
#pragma unroll
          for (int s = 0; s < 32; ++s) a[s] = 0.f;
        }
        v4f acc4[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc4[t] = (v4f){0.f, 0.f, 0.f, 0.f};
        const float* tq = T + nn * kTS + t_col(gg);                     // feature 4 s + gg: t_col(4 s) + t_col(gg)
        float tb[2][4][4];                               // [buffer][k-step][pixel tile]
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int t = 0; t < 4; ++t) tb[0][u][t] = tq[(t < ntl ? 16 * t : 0) * kTS + t_col(4 * u)];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          if (c < 7) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
              for (int t = 0; t < 4; ++t) tb[(c + 1) & 1][u][t] = tq[(t < ntl ? 16 * t : 0) * kTS + t_col(4 * (4 * c + 4 + u))];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t)
              if (t < ntl) acc4[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * c + u], tb[c & 1][u][t], acc4[t], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        // D: lane (gg, nn) holds channels 16 mt + 4 gg + j, pixel 16 t + nn
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = 16 * mt + 4 * gg + j;
          if (c >= C) continue;
          const float pool = pool_l[c];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int q = q0 + 16 * t + nn;
            if (t < ntl && q < hw) out[(int64_t)c * hw + q] = fmaf(sc, acc4[t][j], pool);
          }
        }
      }
    }
    HCM_STAMP(3);
  }
}

// ------------------------------------------------------------------------------------------
// The finest branch in the reference's own order, dX_0 = S_0^T (grows W_0) (r06).  Branch 0 is a gather: one entry of
// weight 1 per sampled row, R entries on a map of H_0 W_0 >> R pixels.  The transposed order above multiplies 64-pixel
// tiles of T of which ~8 columns are not zero and pays a T pass + a multiply per 64 pixels (r06 stamps: 58 k cycles per
// 256-pixel workgroup, HALF of the kernel's wave time for 7 % of its channels).  Instead:
//   finest_rows_kernel   dxs0[m][b][r][0 .. C_0) = grows[m][b][r][:] W_m[:, 0 .. C_0): [R x 128] x [128 x C_0] per image and
//                        modality on the matrix cores (64 rows per workgroup, staged in LDS with 16-byte loads), 1.9 MB;
//   finest_tiles         a workgroup of branch_grad_t_kernel owns 256 pixels: offsets + pooling gradient, the entries of its
//                        range, their dxs0 rows (three round trips, everything of a trip in flight together), then ONE pass over
//                        (channel, four pixels): pixel sums in entry order out of LDS, 16-byte stores.  No T, no multiply.
// More entries in a workgroup's range than its LDS holds (a pixel sampled hundreds of times) -> the generic path, any input.
// ------------------------------------------------------------------------------------------
constexpr int kPoolMax = 8 * 48;   // channels of the widest branch (HRNet-w48)
constexpr int kFR = 64;       // rows per workgroup of finest_rows_kernel
__global__ __launch_bounds__(256) void finest_rows_kernel(const float* __restrict__ grows, const float* __restrict__ Wp1,
                                                          const float* __restrict__ Wp2, int R, int B, int Ctot, int C0,
                                                          int C0p, float* __restrict__ dxs0) {
  __shared__ __attribute__((aligned(16))) float Tl[kFR * kTS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = blockIdx.x * kFR, b = blockIdx.y, m = blockIdx.z;
  const int n = lane & 15, gq = lane >> 4;
  const float* Wp = m ? Wp2 : Wp1;
  // A fragments of TWO channel tiles (W^T, columns 16 mt + n; uniform base + 32-bit lane offset), requested before the rows are
  // staged: their round trip runs under the staging, and the two tiles are two independent accumulator chains (the first form --
  // A after the barrier, one chain -- took 15 us for 0.12 GFLOP)
  float a[2][32];
  auto load_a = [&](int mt) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const unsigned lo = (unsigned)(gq * Ctot + min(16 * (mt + h) + n, C0 - 1));
#pragma unroll
      for (int s = 0; s < 32; ++s) a[h][s] = (Wp + (size_t)(4 * s) * Ctot)[lo];
    }
  };
  load_a(0);
  const float* gsrc = grows + (((int64_t)m * B + b) * R + r0) * kF;
  const int nr = min(kFR, R - r0);
  {
    float4 v[kFR * (kF / 4) / 256];
#pragma unroll
    for (int k = 0; k < kFR * (kF / 4) / 256; ++k) {                 // eight 16-byte loads in flight (clamped row, zeroed below)
      const int e = tid + 256 * k, r = e >> 5, c4 = e & 31;
      v[k] = *reinterpret_cast<const float4*>(gsrc + (int64_t)min(r, nr - 1) * kF + 4 * c4);
    }
#pragma unroll
    for (int k = 0; k < kFR * (kF / 4) / 256; ++k) {
      const int e = tid + 256 * k, r = e >> 5, c4 = e & 31;
      const float4 w = r < nr ? v[k] : make_float4(0.f, 0.f, 0.f, 0.f);
      float* z = &Tl[r * kTS + 4 * c4];
      *reinterpret_cast<float2*>(z) = make_float2(w.x, w.y);
      *reinterpret_cast<float2*>(z + 2) = make_float2(w.z, w.w);
    }
  }
  __syncthreads();
  const float* tq = Tl + (16 * wave + n) * kTS + gq;
  const int r = r0 + 16 * wave + n;
  float* drow = dxs0 + (((int64_t)m * B + b) * R + r) * C0p;
  for (int mt = 0; 16 * mt < C0p; mt += 2) {
    if (mt != 0) load_a(mt);
    const bool live0 = 16 * mt + n < C0, live1 = 16 * (mt + 1) + n < C0;
    v4f acc0 = (v4f){0.f, 0.f, 0.f, 0.f}, acc1 = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      const float bv = tq[4 * s];
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(live0 ? a[0][s] : 0.f, bv, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(live1 ? a[1][s] : 0.f, bv, acc1, 0, 0, 0);
    }
    // D: lane (gq, n) holds channels 16 mt + 4 gq + j of row 16 wave + n (channels >= C0: A was zero)
    if (r < R && 16 * mt + 4 * gq < C0p) *reinterpret_cast<float4*>(drow + 16 * mt + 4 * gq) = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]);
    if (r < R && 16 * (mt + 1) + 4 * gq < C0p)
      *reinterpret_cast<float4*>(drow + 16 * (mt + 1) + 4 * gq) = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);
  }
}

constexpr int kTFloats = (kTP + 256 / 16) * kTS;       // floats of the tile kernel's T region (64 pixel rows + 16 spare rows)
// entries a workgroup of the finest path can stage: each costs an int2 + a row of C0p floats of the T region
__host__ __device__ inline int finest_cap(int C0p) { return (kTFloats - 4) / (C0p + 2); }

// -> true: the workgroup's tiles are written; false: too many entries in its range, the caller runs the generic path
__device__ __forceinline__ bool finest_tiles(float* __restrict__ T, int* __restrict__ soff, float* __restrict__ pool_l,
                                             const float* __restrict__ dxs0, int C0p, const float* __restrict__ dpooled,
                                             const float* __restrict__ scale, const int2* __restrict__ ent,
                                             const int* __restrict__ off, int R, int B, int Ctot, const Maps8Out& g,
                                             const PlanGeom& gm, int wg, int wpx) {
  const int tid = threadIdx.x;
  const int b = blockIdx.x, m = blockIdx.y;
  const int C = g.C[0], hw = g.H[0] * g.W[0];
  const int P0 = wg * wpx, npx = min(wpx, hw - P0);            // wpx = pixels per workgroup (sub-tiles x pixels of a sub-tile)
  const int* offb = off + (int64_t)b * gm.off_per_image + gm.off_off[0];
  const int2* entb = ent + (int64_t)b * gm.ent_per_image + gm.ent_off[0];
  const float* dp = dpooled != nullptr ? dpooled + ((int64_t)m * B + b) * Ctot : nullptr;
  const float sc = scale != nullptr ? scale[0] : 1.f;
  const float inv = 1.f / (float)hw;
  HCM_STAMP(0);
  // ---- trip 1: the per-pixel offsets of the range and the pooling gradient of the branch's channels
  for (int e = tid; e <= npx; e += kGW) soff[e] = offb[P0 + e];
  for (int c = tid; c < C; c += kGW) pool_l[c] = dp != nullptr ? dp[c] * inv : 0.f;
  __syncthreads();
  const int j0 = soff[0], ne = soff[npx] - j0;
  if (ne > finest_cap(C0p)) return false;                // block-uniform
  HCM_STAMP(1);
  // ---- trip 2: the entries; trip 3: their rows of dxs0, 16 bytes per thread and load
  int2* el = reinterpret_cast<int2*>(T);
  float* dl = T + ((2 * ne + 3) & ~3);
  for (int e = tid; e < ne; e += kGW) el[e] = entb[j0 + e];
  __syncthreads();
  const int C4 = C0p >> 2;
  const float* dsrc = dxs0 + ((int64_t)m * B + b) * R * C0p;
  for (int k = tid; k < ne * C4; k += kGW) {
    const int e = k / C4, c4 = k - e * C4;
    const int r = el[e].x & ((1 << kRowBits) - 1);
    *reinterpret_cast<float4*>(dl + e * C0p + 4 * c4) = *reinterpret_cast<const float4*>(dsrc + (int64_t)r * C0p + 4 * c4);
  }
  __syncthreads();
  HCM_STAMP(2);
  // ---- one pass over (channel, four pixels): consecutive threads = consecutive quads of one channel's row of the map
  float* out = sel8(g.p, m * 4) + (int64_t)b * C * hw + P0;
  const int nq = npx >> 2;                                // hw % 4 == 0 on this path, P0 is a multiple of 16
  for (int it = tid; it < C * nq; it += kGW) {
    const int c = it / nq, quad = it - c * nq;
    const float pool = pool_l[c];
    int o[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) o[k] = soff[4 * quad + k] - j0;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float acc = 0.f;
      for (int e = o[k]; e < o[k + 1]; ++e) acc = fmaf(__builtin_bit_cast(float, el[e].y), dl[e * C0p + c], acc);
      v[k] = fmaf(sc, acc, pool);
    }
    *reinterpret_cast<float4*>(out + (int64_t)c * hw + 4 * quad) = make_float4(v[0], v[1], v[2], v[3]);
  }
  HCM_STAMP(3);
  return true;
}


__global__ __launch_bounds__(kGW, 3) void branch_grad_t_kernel(
    const float* __restrict__ grows, const float* __restrict__ Wp1, const float* __restrict__ Wp2,
    const float* __restrict__ dpooled, const float* __restrict__ scale, const int2* __restrict__ ent,
    const int* __restrict__ off, int R, int B, int Ctot, Maps8Out g, TilePlan tp, PlanGeom gm,
    const float* __restrict__ dxs0, int C0p) {
  __shared__ __attribute__((aligned(16))) float T[kTFloats];                   // 64 pixel rows + one spare row per 16-lane group
  __shared__ int soff[kSpanMax * kTP + 1];
  __shared__ float pool_l[kPoolMax];            // the pooling gradient of the workgroup's branch, / (H W)
  HCM_STAMP(5);
  int k0 = 0;
  while (k0 < 3 && (int)blockIdx.z >= tp.first[k0 + 1]) ++k0;
  const int i = sel4(tp.ord, k0);
  if (i == 0 && dxs0 != nullptr) {
    const int wg = blockIdx.z - (k0 == 0 ? tp.first[0] : k0 == 1 ? tp.first[1] : k0 == 2 ? tp.first[2] : tp.first[3]);
    if (finest_tiles(T, soff, pool_l, dxs0, C0p, dpooled, scale, ent, off, R, B, Ctot, g, gm, wg, tp.span[0] * tp.tpix[0])) return;
    __syncthreads();                                     // (soff is loaded again below)
  }
  if (sel4(tp.tpix, i) == 16)
    branch_grad_tiles<16>(T, soff, pool_l, grows, Wp1, Wp2, dpooled, scale, ent, off, R, B, Ctot, g, tp, gm);
  else
    branch_grad_tiles<kTP>(T, soff, pool_l, grows, Wp1, Wp2, dpooled, scale, ent, off, R, B, Ctot, g, tp, gm);
}

int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

int dw_chunks(int B) {
  (void)B;
  return 128;                     // x 2 modalities = one workgroup per CU of an MI355X
}

// LDS budget of the forward kernel: the tile, then branch 3, then branch 2 if they fit in 160 KiB.  -> floats used, or -1
template <int KS>
int lds_plan(const int (&C)[4], const int (&H)[4], const int (&W)[4], int* off2, int* off3) {
  constexpr int XS = 4 * KS + 2;
  const int cap = (160 * 1024 - 4096) / 4;            // 3.2 KB of static LDS: the stencil table
  int used = kSR * XS;
  *off2 = -1; *off3 = -1;
  if (kSR * XS > cap) return -1;
  const int n3 = C[3] * ((H[3] * W[3]) | 1), n2 = C[2] * ((H[2] * W[2]) | 1);
  if (used + n3 <= cap) { *off3 = used; used += n3; }
  if (used + n2 <= cap) { *off2 = used; used += n2; }
  return used;
}

// floats of channels-last workspace the forward wants for ONE modality (the branches it gathers from global memory)
size_t nhwc_floats_one(const int (&C)[4], const int (&H)[4], const int (&W)[4], int B, int Ctot) {
  int off2, off3, used;
  if (Ctot + 1 <= 4 * 68) used = lds_plan<68>(C, H, W, &off2, &off3);
  else if (Ctot + 1 <= 4 * 121) used = lds_plan<121>(C, H, W, &off2, &off3);
  else used = lds_plan<181>(C, H, W, &off2, &off3);
  if (used < 0) return 0;
  size_t n = 0;
  for (int i = 0; i < 4; ++i)
    if (!((i == 2 && off2 >= 0) || (i == 3 && off3 >= 0))) n += (size_t)B * C[i] * H[i] * W[i];
  return n;
}

template <int KS>
int launch_project(const Maps8& e, int nmod, int B, const int64_t* pix, int R, int Ctot, const float* Wp1, const float* bp1,
                   const float* Wp2, const float* bp2, float* xs, int ld, float* rows, float* grows, float* nhwc_ws,
                   hipStream_t s) {
  int off2, off3;
  int used = lds_plan<KS>(e.C, e.H, e.W, &off2, &off3);
  if (used < 0) return (int)hipErrorInvalidValue;
  // ---- channels-last copies of the branches that stay in global memory (workspace given = the caller wants them)
  Maps8T et;
  for (int k = 0; k < 8; ++k) et.t[k] = nullptr;
  if (nhwc_ws != nullptr) {
    float* wp = nhwc_ws;
    hcm::ProfSpan tspan(HCM_PROF_ROW8_NHWC, s);
    double moved = 0.0;
    for (int m = 0; m < nmod; ++m)
      for (int i = 0; i < 4; ++i) {
        if ((i == 2 && off2 >= 0) || (i == 3 && off3 >= 0)) continue;
        const int hw = e.H[i] * e.W[i];
        const size_t tb = (size_t)e.C[i] * 65 * sizeof(float);
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(nchw_to_nhwc_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)tb);
        if (e2 != hipSuccess) return (int)e2;
        nchw_to_nhwc_kernel<<<B * ((hw + 63) / 64), 256, tb, s>>>(e.p[m * 4 + i], wp, e.C[i], hw);
        HCM_CHECK_LAUNCH();
        et.t[m * 4 + i] = wp;
        wp += (size_t)B * e.C[i] * hw;
        moved += 8.0 * (double)B * e.C[i] * hw;          // every float read once, written once
      }
    tspan.add_work(moved);
    tspan.stop();
  }
  // rows per workgroup: fewest (rounds over the CUs) x (tiles per workgroup + the staging prologue)
  const int cus = num_cus();
  int best = 1;
  double best_cost = 1e30;
  for (int ns = 1; ns <= 32 && ns <= R; ++ns) {
    const int per = (R + ns - 1) / ns, tiles = (per + kSR - 1) / kSR;
    const int rounds = (nmod * B * ns + cus - 1) / cus;
    const double cost = rounds * (tiles + 1.5);
    if (cost < best_cost - 1e-9) { best_cost = cost; best = ns; }
  }
  const int per = (R + best - 1) / best;
  const size_t bytes = (size_t)used * sizeof(float);
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(project_rows_kernel<KS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (err != hipSuccess) return (int)err;
  hcm::ProfSpan span(HCM_PROF_ROW8_FWD, s);
  project_rows_kernel<KS><<<dim3(best, B, nmod), kPW, bytes, s>>>(e, et, B, pix, R, Ctot, Wp1, bp1, Wp2, bp2, xs, ld, rows, grows,
                                                             per, off2, off3);
  span.stop();
  HCM_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" {

#ifdef HCM_ROW8_TIMING
int hcm_debug_row8_timing(unsigned long long* buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_row8_dbg), &buf, sizeof(buf));
}
// resident workgroups per CU the runtime computes for the backward tile kernel (registers, LDS, wave slots)
int hcm_debug_row8_occupancy() {
  int nb = -1;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(branch_grad_t_kernel), kGW, 0) != hipSuccess) return -1;
  return nb;
}
#endif

size_t hcm_project_rows_nhwc_floats(hcm_branches enc1, hcm_branches enc2, int B, int Ctot) {
  if (B <= 0 || !branches_ok(enc1, Ctot)) return 0;
  return (absent(enc2) ? 1 : 2) * nhwc_floats_one(enc1.C, enc1.H, enc1.W, B, Ctot);
}

int hcm_project_rows(hcm_branches enc1, hcm_branches enc2, int B, const int64_t* pix, int R, int Ctot, int F,
                     const float* Wp1, const float* bp1, const float* Wp2, const float* bp2, float* xs, float* rows,
                     float* grows, hcm_stream_t stream) {
  return hcm_project_rows_cl(enc1, enc2, B, pix, R, Ctot, F, Wp1, bp1, Wp2, bp2, xs, rows, grows, nullptr, 0, stream);
}

int hcm_project_rows_cl(hcm_branches enc1, hcm_branches enc2, int B, const int64_t* pix, int R, int Ctot, int F,
                        const float* Wp1, const float* bp1, const float* Wp2, const float* bp2, float* xs, float* rows,
                        float* grows, float* nhwc_ws, size_t nhwc_floats, hcm_stream_t stream) {
  if (nhwc_ws != nullptr && nhwc_floats < hcm_project_rows_nhwc_floats(enc1, enc2, B, Ctot)) return (int)hipErrorInvalidValue;
  const int nmod = absent(enc2) ? 1 : 2;          // absent: rows[1] / xs[1] / grows[1] are left to the caller
  if (B <= 0 || R <= 0 || F != kF || pix == nullptr || rows == nullptr || !branches_ok(enc1, Ctot) ||
      (nmod == 2 && !branches_ok(enc2, Ctot)))
    return (int)hipErrorInvalidValue;
  for (int i = 0; i < 4 && nmod == 2; ++i)
    if (enc1.C[i] != enc2.C[i] || enc1.H[i] != enc2.H[i] || enc1.W[i] != enc2.W[i]) return (int)hipErrorInvalidValue;
  const int ld = hcm_sample_branches_ld(Ctot);
  const Maps8 e = pack8(enc1, nmod == 2 ? enc2 : enc1);
  hipStream_t s = (hipStream_t)stream;
  // HRNet-w18 / w32 / w48: 270 / 480 / 720 channels + the bias column
  if (Ctot + 1 <= 4 * 68) return launch_project<68>(e, nmod, B, pix, R, Ctot, Wp1, bp1, Wp2, bp2, xs, ld, rows, grows, nhwc_ws, s);
  if (Ctot + 1 <= 4 * 121) return launch_project<121>(e, nmod, B, pix, R, Ctot, Wp1, bp1, Wp2, bp2, xs, ld, rows, grows, nhwc_ws, s);
  if (Ctot + 1 <= 4 * 181) return launch_project<181>(e, nmod, B, pix, R, Ctot, Wp1, bp1, Wp2, bp2, xs, ld, rows, grows, nhwc_ws, s);
  return (int)hipErrorInvalidValue;
}

size_t hcm_project_rows_dw_workspace_bytes(int B, int Ctot) {
  return (size_t)2 * dw_chunks(B) * kF * hcm_sample_branches_ld(Ctot) * sizeof(float);
}

int hcm_project_rows_dw(const float* grows, const float* xs, const float* scale, int B, int R, int Ctot, int F,
                        float* dWp1, float* dbp1, float* dWp2, float* dbp2, void* workspace, size_t workspace_bytes,
                        hcm_stream_t stream) {
  if (B <= 0 || R <= 0 || Ctot <= 0 || F != kF || !grows || !xs || !dWp1 || !dbp1 || (dWp2 == nullptr) != (dbp2 == nullptr) ||
      !workspace || workspace_bytes < hcm_project_rows_dw_workspace_bytes(B, Ctot))
    return (int)hipErrorInvalidValue;
  const int nmod = dWp2 != nullptr ? 2 : 1;        // dWp2 == NULL: modality 0 only (the second encoder is absent)
  hipStream_t s = (hipStream_t)stream;
  const int ld = hcm_sample_branches_ld(Ctot), M = B * R;
  const int nchunk = dw_chunks(B);
  int rpc = (M + nchunk - 1) / nchunk;
  rpc = (rpc + 3) & ~3;
  const int ncg = (ld + kNT * 16 - 1) / (kNT * 16);
  float* part = static_cast<float*>(workspace);
  hcm::ProfSpan span(HCM_PROF_ROW8_DW, s);
  proj_dw_partial_kernel<<<dim3(nchunk, ncg, nmod), kDW, 0, s>>>(grows, xs, M, ld, rpc, part);
  HCM_CHECK_LAUNCH();
  proj_dw_reduce_kernel<<<dim3((kF * ld + 255) / 256, nmod), 256, 0, s>>>(part, nchunk, ld, Ctot, scale, dWp1, dbp1, dWp2, dbp2);
  span.stop();
  HCM_CHECK_LAUNCH();
  return 0;
}


// the finest branch takes the rows-first path (finest_rows_kernel + finest_tiles) when its map rows split into 16-byte quads
// and a row of its channels fits the tile kernel's LDS: floats of dxs0 per modality, 0 = generic path
static size_t finest_rows_floats(int B, int R, const hcm_branches_out& g1) {
  const int C0p = (g1.C[0] + 3) & ~3;
  if (((g1.H[0] * g1.W[0]) & 3) != 0 || C0p > 64 || finest_cap(C0p) < 64) return 0;
  return (size_t)B * R * C0p;
}

static int plan_geom(const hcm_branches_out& g1, int R, PlanGeom* gm) {
  int eo = 0, oo = 0;
  for (int i = 0; i < 4; ++i) {
    gm->H[i] = g1.H[i]; gm->W[i] = g1.W[i];
    gm->ent_off[i] = eo; gm->off_off[i] = oo;
    eo += (i == 0 ? 1 : 4) * R;
    oo += g1.H[i] * g1.W[i] + 1;
  }
  gm->ent_per_image = eo;
  gm->off_per_image = (oo + 3) & ~3;
  return 0;
}

size_t hcm_project_rows_backward_workspace_bytes(int B, int R, int Ctot, hcm_branches_out g1) {
  PlanGeom gm;
  plan_geom(g1, R, &gm);
  const size_t dw = hcm_project_rows_dw_workspace_bytes(B, Ctot);
  return dw + (size_t)B * gm.ent_per_image * sizeof(int2) + (size_t)B * gm.off_per_image * sizeof(int) + 256 +
         finest_rows_floats(B, R, g1) * 2 * sizeof(float);
}

int hcm_project_rows_backward(const float* grows, const float* xs, const float* Wp1, const float* Wp2,
                              const float* dpooled, const float* scale, const int64_t* pix, int R, int B, int Ctot,
                              int F, hcm_branches_out g1, hcm_branches_out g2, const int32_t* keep, int S, float* dWp1,
                              float* dbp1, float* dWp2, float* dbp2, void* workspace, size_t workspace_bytes,
                              hcm_stream_t stream) {
  const int nmod = absent(g2) ? 1 : 2;             // absent: modality 0's maps and weight gradients only
  if (B <= 0 || R <= 0 || F != kF || !grows || !xs || !Wp1 || (nmod == 2 && !Wp2) || !pix || !workspace ||
      !branches_ok(g1, Ctot) || (nmod == 2 && !branches_ok(g2, Ctot)) || (nmod == 1 && (dWp2 != nullptr || dbp2 != nullptr)) ||
      workspace_bytes < hcm_project_rows_backward_workspace_bytes(B, R, Ctot, g1))
    return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  PlanGeom gm;
  plan_geom(g1, R, &gm);
  TilePlan tp;
  int v = 0, n2 = 2;
  int nwg[4];
  double cost[4];
  for (int i = 0; i < 4; ++i) {
    if (nmod == 2 && (g1.C[i] != g2.C[i] || g1.H[i] != g2.H[i] || g1.W[i] != g2.W[i])) return (int)hipErrorInvalidValue;
    if (g1.C[i] > kPoolMax) return (int)hipErrorInvalidValue;      // the tile kernel keeps a branch's pooling gradient in LDS
    // keys are pixel * E + entry in 32 bits
    if ((uint64_t)g1.H[i] * g1.W[i] * 4ull * (uint64_t)R >= 0xffffffffull) return (int)hipErrorInvalidValue;
    if (g1.H[i] * g1.W[i] >= (1 << 17) || R >= (1 << kRowBits)) return (int)hipErrorInvalidValue;
    tp.tpix[i] = g1.H[i] * g1.W[i] <= 256 ? 16 : kTP;
    const int nt = (g1.H[i] * g1.W[i] + tp.tpix[i] - 1) / tp.tpix[i];
    tp.span[i] = nt >= 32 ? kSpanMax : 1;            // maps of >= 2048 pixels: kSpanMax 64-pixel sub-tiles per workgroup
    nwg[i] = (nt + tp.span[i] - 1) / tp.span[i];
    // a workgroup's expected life, in "sub-tile passes": its sub-tiles x (walk of its share of the entries + multiply rounds)
    const double ent_per_wg = (double)(i == 0 ? 1 : 4) * R / nwg[i];
    cost[i] = tp.span[i] * (1.0 + ((g1.C[i] + 15) / 16 + 3) / 4) + ent_per_wg / (16.0 * tp.tpix[i] / 4);
    if (i == 0 && finest_rows_floats(B, R, g1) != 0) cost[i] = 1.0;          // rows-first path: three round trips and stores
  }
  // dispatch order: expensive workgroups first.  grid (B, nmod, tiles): x runs fastest, so ALL images' workgroups of the most
  // expensive kind start before the first cheap one (r06 stamps: with the tiles in x every image's heaviest workgroups started
  // behind its light ones, the last image's at the very end of the kernel)
  for (int k = 0; k < 4; ++k) tp.ord[k] = k;
  for (int a = 0; a < 4; ++a)
    for (int c = a + 1; c < 4; ++c)
      if (cost[tp.ord[c]] > cost[tp.ord[a]]) { const int t = tp.ord[a]; tp.ord[a] = tp.ord[c]; tp.ord[c] = t; }
  for (int k = 0; k < 4; ++k) { tp.first[k] = v; v += nwg[tp.ord[k]]; }
  tp.first[4] = v;
  n2 = 128;
  while (n2 < 4 * R) n2 <<= 1;
  if ((size_t)n2 * 4 > 150 * 1024) return (int)hipErrorInvalidValue;            // R <= 9600 rows per image
  char* base = static_cast<char*>(workspace);
  const size_t dwb = hcm_project_rows_dw_workspace_bytes(B, Ctot);
  int2* ent = reinterpret_cast<int2*>(base + ((dwb + 15) & ~(size_t)15));
  int* off = reinterpret_cast<int*>(ent + (size_t)B * gm.ent_per_image);
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(stencil_plan_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, n2 * 4);
  if (err != hipSuccess) return (int)err;
  stencil_plan_kernel<<<dim3(4, B), kPT, (size_t)n2 * 4, s>>>(pix, R, gm, keep, S, n2, ent, off);
  HCM_CHECK_LAUNCH();
  if (dWp1 != nullptr) {
    const int rc = hcm_project_rows_dw(grows, xs, scale, B, R, Ctot, F, dWp1, dbp1, dWp2, dbp2, workspace, dwb, stream);
    if (rc != 0) return rc;
  }
  hcm::ProfSpan span(HCM_PROF_ROW8_BWD, s);
  const size_t fr = finest_rows_floats(B, R, g1);
  const int C0p = (g1.C[0] + 3) & ~3;
  float* dxs0 = nullptr;
  if (fr != 0) {
    dxs0 = reinterpret_cast<float*>(base + ((dwb + 15) & ~(size_t)15) + (((size_t)B * gm.ent_per_image * sizeof(int2) +
                                    (size_t)B * gm.off_per_image * sizeof(int) + 15) & ~(size_t)15));
    finest_rows_kernel<<<dim3((R + kFR - 1) / kFR, B, nmod), 256, 0, s>>>(grows, Wp1, nmod == 2 ? Wp2 : Wp1, R, B, Ctot, g1.C[0],
                                                                         C0p, dxs0);
    HCM_CHECK_LAUNCH();
  }
  branch_grad_t_kernel<<<dim3(B, nmod, v), kGW, 0, s>>>(grows, Wp1, nmod == 2 ? Wp2 : Wp1, dpooled, scale, ent, off, R, B, Ctot,
                                                        pack8(g1, nmod == 2 ? g2 : g1), tp, gm, dxs0, C0p);
  span.stop();
  HCM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
