// The serial section of a second-stage step between the encoders' forward and backward, as a handful of gfx950
// kernels:
//
//   head_pool_kernel        4-branch global average pool of both HRNets -> pooled [2, B, Ctot]
//                           (networks/build_backbone.py:265-276: AdaptiveAvgPool2d + cat)
//   heads_fwd_kernel        Linear(Ctot|D3 -> F) + L2 normalise for the three heads (:277-288, networks/util.py:74-80),
//                           head 3 fed by the mean over the joints of the SemGCN output (:273)
//   heads_bwd_kernel,       their backward: normalize-backward, dX (-> pooled gradient / SemGCN gradient), then
//   heads_dw_kernel         dW / db summed over the batch in a fixed order
//   pixel_sample_kernel     nearest-resize of the depth mask, keep flags, S pixels per image drawn with replacement
//                           from the valid ones (learning/contrast_trainer.py:671-685) by Philox inverse-CDF on the
//                           integer prefix counts, and the joints' pixels (:757-761)
//   sample_branches_kernel  merge_all_res restricted to the sampled pixels (build_backbone.py:247-254): all four
//                           branches of both modalities gathered (branch 0) / bilinearly sampled (branches 1-3) into
//                           the padded row matrix xs [2, B*R, ld] that the 1x1 projection (:243-245) multiplies
//   branch_grad_kernel      the backward of both: every pixel of every branch gradient map is written once as
//                           (pooled gradient / HW) + sum over the rows whose stencil touches it, rows in ascending
//                           order -- owner computes, deterministic, no atomics, no zero-fill
//
// Everything is fp32; sums run in a fixed order.
#include "hcm_common.h"
#include "../../include/hcmoco_hip.h"
#include "section_common.h"

namespace {

using namespace hcm;

constexpr int kWG = 256;
constexpr int kMaxIn = 1024;   // widest head input (HRNet-w48: 720 pooled channels)

// ------------------------------------------------------------------------------------------
// one wave per (modality, image, channel) plane
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWG) void head_pool_kernel(Maps8 e, int B, int Ctot, float* __restrict__ pooled, int nmod) {
  const int lane = threadIdx.x & 63;
  const int plane = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (plane >= nmod * B * Ctot) return;
  const int m = plane / (B * Ctot), rem = plane - m * B * Ctot;
  const int b = rem / Ctot;
  int c = rem - b * Ctot;
  int i = 0;
  while (i < 3 && c >= sel4(e.C, i)) { c -= sel4(e.C, i); ++i; }
  const int hw = sel4(e.H, i) * sel4(e.W, i);
  const float* src = sel8(e.p, m * 4 + i) + ((int64_t)b * sel4(e.C, i) + c) * hw;
  float s = 0.f;
  if ((hw & 3) == 0) {
    // four independent 16-byte loads per lane and trip: the plane (<= 16 KB at 256 x 256) is in flight at once
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    int k = lane * 4;
    for (; k + 768 < hw; k += 1024) {
      const float4 v0 = *reinterpret_cast<const float4*>(src + k);
      const float4 v1 = *reinterpret_cast<const float4*>(src + k + 256);
      const float4 v2 = *reinterpret_cast<const float4*>(src + k + 512);
      const float4 v3 = *reinterpret_cast<const float4*>(src + k + 768);
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
      a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
      a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
      a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
    }
    for (; k < hw; k += 256) {
      const float4 v = *reinterpret_cast<const float4*>(src + k);
      a0.x += v.x; a0.y += v.y; a0.z += v.z; a0.w += v.w;
    }
    s = (((a0.x + a0.y) + (a0.z + a0.w)) + ((a1.x + a1.y) + (a1.z + a1.w))) +
        (((a2.x + a2.y) + (a2.z + a2.w)) + ((a3.x + a3.y) + (a3.z + a3.w)));
  } else {
    for (int k = lane; k < hw; k += 64) s += src[k];
  }
  s = wave_sum(s);
  if (lane == 0) pooled[plane] = s / (float)hw;
}

// ------------------------------------------------------------------------------------------
// grid (B, 3 heads), 256 threads.  x -> LDS, every output = a coalesced wave dot product.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWG) void heads_fwd_kernel(
    const float* __restrict__ pooled, const float* __restrict__ feat3, int B, int J, int Ctot, int D3, int F,
    const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2,
    const float* __restrict__ b2, const float* __restrict__ W3, const float* __restrict__ b3,
    float* __restrict__ mean3, float* __restrict__ ypre, float* __restrict__ f, int ldf,
    float* __restrict__ fT, const int64_t* __restrict__ index) {
  __shared__ float x[kMaxIn];
  __shared__ float y[kWG];
  __shared__ float red[4];
  const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Cin = h < 2 ? Ctot : D3;
  const float* W = h == 0 ? W1 : (h == 1 ? W2 : W3);
  const float* bias = h == 0 ? b1 : (h == 1 ? b2 : b3);
  if (h < 2) {
    for (int i = tid; i < Cin; i += kWG) x[i] = pooled[((int64_t)h * B + b) * Ctot + i];
  } else {
    for (int i = tid; i < Cin; i += kWG) {
      float s = 0.f;
      for (int j = 0; j < J; ++j) s += feat3[((int64_t)b * J + j) * D3 + i];
      s /= (float)J;
      x[i] = s;
      mean3[(int64_t)b * D3 + i] = s;
    }
  }
  __syncthreads();
  // a wave owns outputs wave, wave + 4, ...: 32 of them per trip (all of them at F = 128) x two 64-wide steps of the input = 64
  // weight loads in flight (r06; one step of eight outputs per trip was 20 dependent round trips per workgroup: 19 us for a
  // 128 x 270 matrix-vector product; now 3).  Unconditional clamped loads; the same fmaf chain per lane and output as before.
  for (int o0 = wave; o0 < F; o0 += 128) {
    float acc[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) acc[j] = 0.f;
    for (int i0 = lane; i0 < Cin; i0 += 128) {
      float wv[2][32], xv[2];
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int ic = min(i0 + 64 * it, Cin - 1);
        xv[it] = x[ic];
#pragma unroll
        for (int j = 0; j < 32; ++j) wv[it][j] = W[(int64_t)min(o0 + 4 * j, F - 1) * Cin + ic];
      }
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const bool in = i0 + 64 * it < Cin;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
          const float t = fmaf(o0 + 4 * j < F ? wv[it][j] : 0.f, xv[it], acc[j]);
          acc[j] = in ? t : acc[j];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const int o = o0 + 4 * j;
      const float a = wave_sum(acc[j]);
      if (lane == 0 && o < F) y[o] = a + bias[o];
    }
  }
  __syncthreads();
  float v = tid < F ? y[tid] : 0.f;
  float ss = wave_sum(v * v);
  if (lane == 0) red[wave] = ss;
  __syncthreads();
  ss = (red[0] + red[1]) + (red[2] + red[3]);
  const float den = fmaxf(sqrtf(ss), 1e-12f);
  if (tid < F) {
    ypre[((int64_t)h * B + b) * F + tid] = v;
    const float o = v / den;
    f[(int64_t)b * ldf + h * F + tid] = o;
    fT[((int64_t)h * B + b) * F + tid] = o;
  }
  // the packed all-gather row carries the int64 bank index bit-cast into its last two floats
  if (index != nullptr && h == 0 && tid < 2) {
    const int64_t v64 = index[b];
    const uint32_t half = tid == 0 ? (uint32_t)v64 : (uint32_t)((uint64_t)v64 >> 32);
    f[(int64_t)b * ldf + 3 * F + tid] = __builtin_bit_cast(float, half);
  }
}

// grid (B, 3).  dy (normalize backward) -> dyws; dX -> dpooled (heads 1, 2) or the SemGCN gradient (head 3).
__global__ __launch_bounds__(kWG) void heads_bwd_kernel(
    const float* __restrict__ gf, const float* __restrict__ scale, const float* __restrict__ ypre, int B, int J,
    int Ctot, int D3, int F, const float* __restrict__ W1, const float* __restrict__ W2,
    const float* __restrict__ W3, const float* __restrict__ gjoint, float* __restrict__ dyws,
    float* __restrict__ dpooled, float* __restrict__ gfeat3) {
  __shared__ float dy[kWG];
  __shared__ float red[2][4];
  const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float sc = scale != nullptr ? scale[0] : 1.f;
  const int Cin = h < 2 ? Ctot : D3;
  const float* W = h == 0 ? W1 : (h == 1 ? W2 : W3);
  const float yv = tid < F ? ypre[((int64_t)h * B + b) * F + tid] : 0.f;
  const float g = tid < F ? gf[((int64_t)h * B + b) * F + tid] * sc : 0.f;
  float ss = wave_sum(yv * yv), gy = wave_sum(g * yv);
  if (lane == 0) { red[0][wave] = ss; red[1][wave] = gy; }
  __syncthreads();
  ss = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  gy = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  const float nrm = sqrtf(ss), den = fmaxf(nrm, 1e-12f);
  // F.normalize backward; with the clamp active the map is y / eps (no projection term)
  const float d = nrm < 1e-12f ? g / den : (g - (gy / (den * den)) * yv) / den;
  if (tid < F) {
    dy[tid] = d;
    dyws[((int64_t)h * B + b) * F + tid] = d;
  }
  __syncthreads();
  // two inputs per thread and pass (i, i + 256: at 270 inputs the second pass of a one-input loop was 14 threads paying the
  // whole chain again), 32 weight rows each in flight; o ascending in every sum as before
  for (int i0 = tid; i0 < Cin; i0 += 2 * kWG) {
    const int i1 = i0 + kWG, ic1 = min(i1, Cin - 1);
    float acc0 = 0.f, acc1 = 0.f;
    for (int o = 0; o < F; o += 32) {
      float w0[32], w1[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        const int64_t ro = (int64_t)min(o + u, F - 1) * Cin;
        w0[u] = W[ro + i0];
        w1[u] = W[ro + ic1];
      }
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        const float dv = dy[min(o + u, F - 1)];
        const float t0 = fmaf(w0[u], dv, acc0), t1 = fmaf(w1[u], dv, acc1);
        acc0 = o + u < F ? t0 : acc0;
        acc1 = o + u < F ? t1 : acc1;
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = k == 0 ? i0 : i1;
      const float acc = k == 0 ? acc0 : acc1;
      if (i >= Cin) continue;
      if (h < 2) {
        dpooled[((int64_t)h * B + b) * Ctot + i] = acc;
      } else {
        const float share = acc / (float)J;
        // sixteen joints' gradients in flight, then their stores (one load, one store per trip was J dependent round trips: loads
        // and stores retire through one in-order counter)
        for (int j0 = 0; j0 < J; j0 += 16) {
          float gj[16];
#pragma unroll
          for (int u = 0; u < 16; ++u)
            gj[u] = gjoint != nullptr ? gjoint[((int64_t)b * J + min(j0 + u, J - 1)) * D3 + i] : 0.f;
#pragma unroll
          for (int u = 0; u < 16; ++u)
            if (j0 + u < J) gfeat3[((int64_t)b * J + j0 + u) * D3 + i] = (gjoint != nullptr ? sc * gj[u] : 0.f) + share;
        }
      }
    }
  }
}

// grid (ceil(F*Cin/256), 3).  dW[o][i] = sum_b dy[b][o] x[b][i], db[o] = sum_b dy[b][o]; b ascending.
__global__ __launch_bounds__(kWG) void heads_dw_kernel(const float* __restrict__ dyws,
                                                       const float* __restrict__ pooled,
                                                       const float* __restrict__ mean3, int B, int Ctot, int D3,
                                                       int F, float* __restrict__ dW1, float* __restrict__ db1,
                                                       float* __restrict__ dW2, float* __restrict__ db2,
                                                       float* __restrict__ dW3, float* __restrict__ db3) {
  const int h = blockIdx.y;
  const int Cin = h < 2 ? Ctot : D3;
  const int e = blockIdx.x * kWG + threadIdx.x;
  if (e >= F * Cin) return;
  const int o = e / Cin, i = e - o * Cin;
  const float* x = h < 2 ? pooled + (int64_t)h * B * Ctot : mean3;
  const float* dy = dyws + (int64_t)h * B * F;
  float acc = 0.f, accb = 0.f;
  for (int b0 = 0; b0 < B; b0 += 16) {            // sixteen images in flight; b ascending in the sums as before
    float dv[16], xv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int b = min(b0 + u, B - 1);
      dv[u] = dy[(int64_t)b * F + o];
      xv[u] = x[(int64_t)b * Cin + i];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const float t = fmaf(dv[u], xv[u], acc), tb = accb + dv[u];
      acc = b0 + u < B ? t : acc;
      accb = b0 + u < B ? tb : accb;
    }
  }
  float* dW = h == 0 ? dW1 : (h == 1 ? dW2 : dW3);
  float* db = h == 0 ? db1 : (h == 1 ? db2 : db3);
  dW[e] = acc;
  if (i == 0) db[o] = accb;
}

// ------------------------------------------------------------------------------------------
// one workgroup per image.  cnt[q] = number of valid pixels with index <= q (dynamic LDS, h*w ints).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWG) void pixel_sample_kernel(
    const float* __restrict__ mask, int B, int H, int W, int h, int w, int S, const int32_t* __restrict__ use_depth,
    const float* __restrict__ j2d, int J, uint64_t seed, uint64_t offset, int64_t* __restrict__ pix,
    int64_t* __restrict__ coord, int32_t* __restrict__ keep) {
  extern __shared__ int cnt[];
  __shared__ int part[kWG];
  __shared__ int any_depth;
  const int b = blockIdx.x, tid = threadIdx.x, hw = h * w, R = S + J;
  const int per = (hw + kWG - 1) / kWG, q0 = tid * per, q1 = min(hw, q0 + per);
  // torch's nearest up/down-sampling: src = min(int(floorf(dst * (in / out))), in - 1), scale in fp32
  const float sy = (float)H / (float)h, sx = (float)W / (float)w;
  int local = 0;
  for (int q = q0; q < q1; ++q) {
    const int y = q / w, x = q - y * w;
    const int yy = min((int)floorf((float)y * sy), H - 1), xx = min((int)floorf((float)x * sx), W - 1);
    local += mask[((int64_t)b * H + yy) * W + xx] > 0.f ? 1 : 0;
    cnt[q] = local;
  }
  part[tid] = local;
  if (tid == 0) {
    int any = use_depth == nullptr ? 1 : 0;
    if (use_depth != nullptr)
      for (int i = 0; i < B; ++i) any |= use_depth[i] != 0 ? 1 : 0;     // reference early return (:663-665)
    any_depth = any;
  }
  __syncthreads();
  // exclusive scan of the 256 chunk counts (Hillis-Steele in LDS: 8 steps)
  int v = part[tid];
  for (int d = 1; d < kWG; d <<= 1) {
    const int add = tid >= d ? part[tid - d] : 0;
    __syncthreads();
    v += add;
    part[tid] = v;
    __syncthreads();
  }
  const int base = v - local, total = part[kWG - 1];
  for (int q = q0; q < q1; ++q) cnt[q] += base;
  __syncthreads();
  const bool kept = total > 0 && any_depth != 0;
  if (tid == 0) keep[b] = kept ? 1 : 0;
  for (int s = tid; s < S; s += kWG) {
    int p = 0;
    if (kept) {
      const uint64_t e = (uint64_t)b * (uint64_t)S + (uint64_t)s;
      uint32_t c[4] = {(uint32_t)e, (uint32_t)(e >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)};
      philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
      const int k = (int)(((uint64_t)c[0] * (uint64_t)(uint32_t)total) >> 32);   // uniform in [0, total)
      int lo = 0, hi = hw - 1;                                                     // smallest q with cnt[q] > k
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cnt[mid] > k) hi = mid; else lo = mid + 1;
      }
      p = lo;
    }
    pix[(int64_t)b * R + s] = p;
    coord[(int64_t)b * S + s] = p;
  }
  for (int j = tid; j < J; j += kWG) {
    // (original_joints2d // 4).long() clamped to [0, h-1]  (contrast_trainer.py:757-761)
    const int e = b * J + j;
    int r = (int)floorf(j2d[2 * e] / 4.f), c = (int)floorf(j2d[2 * e + 1] / 4.f);
    r = min(max(r, 0), h - 1);
    c = min(max(c, 0), h - 1);
    pix[(int64_t)b * R + S + j] = (int64_t)r * h + c;
  }
}

// blocks [0, nrowblocks): one wave per (modality, row); the rest pack [W | bias | 0] into Wpad [2, F, ld].
__global__ __launch_bounds__(kWG) void sample_branches_kernel(
    Maps8 e, int B, const int64_t* __restrict__ pix, int R, int Ctot, int F, int ld,
    const float* __restrict__ Wp1, const float* __restrict__ bp1, const float* __restrict__ Wp2,
    const float* __restrict__ bp2, float* __restrict__ xs, float* __restrict__ Wpad, float* __restrict__ grows,
    int nrowblocks) {
  const int lane = threadIdx.x & 63;
  const int M = B * R;
  if ((int)blockIdx.x >= nrowblocks) {
    const int e0 = ((int)blockIdx.x - nrowblocks) * kWG + threadIdx.x;
    for (int e = e0; e < 2 * F * ld; e += ((int)gridDim.x - nrowblocks) * kWG) {
      const int m = e / (F * ld), rem = e - m * F * ld, o = rem / ld, k = rem - o * ld;
      const float* Wp = m ? Wp2 : Wp1;
      const float* bp = m ? bp2 : bp1;
      Wpad[e] = k < Ctot ? Wp[(int64_t)o * Ctot + k] : (k == Ctot ? bp[o] : 0.f);
    }
    return;
  }
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= 2 * M) return;
  const int m = row / M, r = row - m * M, b = r / R;
  const int h0 = e.H[0], w0 = e.W[0];
  const int p = (int)pix[r];
  const int py = p / w0, px = p - py * w0;
  float* out = xs + (int64_t)row * ld;
  int coff = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int C = e.C[i], hi = e.H[i], wi = e.W[i];
    const float* x = (m ? e.p[4 + i] : e.p[i]) + (int64_t)b * C * hi * wi;
    if (i == 0) {
      for (int c = lane; c < C; c += 64) out[coff + c] = x[(int64_t)c * hi * wi + p];
    } else {
      const Taps t = bilinear_taps(py, px, hi, wi, (float)hi / (float)h0, (float)wi / (float)w0);
      const int o00 = t.y0 * wi + t.x0, o01 = t.y0 * wi + t.x1, o10 = t.y1 * wi + t.x0, o11 = t.y1 * wi + t.x1;
      for (int c = lane; c < C; c += 64) {
        const float* xc = x + (int64_t)c * hi * wi;
        out[coff + c] = t.hy * (t.hx * xc[o00] + t.lx * xc[o01]) + t.ly * (t.hx * xc[o10] + t.lx * xc[o11]);
      }
    }
    coff += C;
  }
  for (int k = Ctot + lane; k < ld; k += 64) out[k] = k == Ctot ? 1.f : 0.f;      // the bias column, then padding
  if (grows != nullptr)
    for (int k = lane; k < F; k += 64) grows[(int64_t)row * F + k] = 0.f;       // the loss kernels accumulate into it
}

// ------------------------------------------------------------------------------------------
// Backward of sampling + pooling.  A workgroup = (branch i, tile of PT <= 256 pixels, block of <= kCB channels) of one
// (modality, image).  All in LDS:
//   (1) the R stencils; ordered compaction of the rows that reach the tile (rows of dropped images' dense samples are
//       skipped: their gradient is exactly zero, and they all sit on pixel 0 -- a 400-entry hub);
//   (2) one coalesced staging pass of those rows' gradients for the channel block;
//   (3) the in-tile (row, tap) contributions, compacted IN ORDER, as a byte array of local target pixels;
//       per-pixel counts (integer LDS atomics: order-independent) and their prefix sum;
//   (4) every pixel's owner thread scans the byte array, 16 targets per ds_read_b128 with a branch-free byte match,
//       and appends its hits in scan order -> per-pixel lists in ascending (row, tap) order without a sort;
//   (5) thread (pixel, channel subgroup) adds exactly its own list: owner computes, fixed order, no float atomics,
//       every output element written once (the maps need no zero-fill), coalesced over the pixels.
// History (r03, B = 32, R = 417, profiles/r03_loss_section.txt): every pixel scanning all R rows in global memory
// 1.59 ms; the same with the rows in LDS 0.98 ms; per-pixel lists filled by atomics + insertion sort 3.2 ms (the
// pixel-0 hub of the dropped images is quadratic there).
// blocks with blockIdx.y == B unpack the projection's weight gradient from the padded GEMM output.
// ------------------------------------------------------------------------------------------
constexpr int kCB = 36;          // channels per workgroup
constexpr int kRB = 128;         // gradient rows staged in LDS at a time (128 x 36 floats = 18 KB)
struct GradPlan {
  int first[5];                  // first workgroup (blockIdx.x) of branch i; first[4] = total
  int ntile[4];                  // pixel tiles of branch i (workgroups per channel block)
};

// 0x80 in every byte of the result where the byte of `x` is zero (exact, no carries between bytes)
__device__ __forceinline__ uint32_t zero_bytes(uint32_t x) {
  const uint32_t t = (x & 0x7f7f7f7fu) + 0x7f7f7f7fu;
  return ~(t | x | 0x7f7f7f7fu);
}

__global__ __launch_bounds__(kWG) void branch_grad_kernel(
    const float* __restrict__ dxs, int ld, const float* __restrict__ dpooled, const float* __restrict__ scale,
    const int64_t* __restrict__ pix, int R, int B, int Ctot, Maps8Out g, GradPlan plan,
    const int32_t* __restrict__ keep, int S, const float* __restrict__ dWpad, int F, float* __restrict__ dWp1,
    float* __restrict__ dbp1, float* __restrict__ dWp2, float* __restrict__ dbp2) {
  extern __shared__ __attribute__((aligned(16))) int ldsraw[];
  __shared__ int wcount[4];
  __shared__ int nlist, ncon;
  __shared__ int cnt[kWG];        // contributions per pixel of the tile -> exclusive offsets
  __shared__ int scan[kWG];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float sc = scale != nullptr ? scale[0] : 1.f;
  if ((int)blockIdx.y == B) {             // unpack d[W | bias] of the two projections (scaled)
    const int m = blockIdx.z;
    float* dW = m ? dWp2 : dWp1;
    float* db = m ? dbp2 : dbp1;
    if (dWpad == nullptr || dW == nullptr) return;
    for (int e = blockIdx.x * kWG + tid; e < F * ld; e += gridDim.x * kWG) {
      const int o = e / ld, k = e - o * ld;
      const float v = sc * dWpad[(int64_t)m * F * ld + e];
      if (k < Ctot) dW[(int64_t)o * Ctot + k] = v;
      else if (k == Ctot) db[o] = v;
    }
    return;
  }
  const int b = blockIdx.y, m = blockIdx.z;
  int i = 0;
  while (i < 3 && (int)blockIdx.x >= plan.first[i + 1]) ++i;
  const int wg = blockIdx.x - plan.first[i];
  const int C = sel4(g.C, i), hi = sel4(g.H, i), wi = sel4(g.W, i), hw = hi * wi;
  const int h0 = g.H[0], w0 = g.W[0];
  int coff = 0;
  for (int k = 0; k < i; ++k) coff += sel4(g.C, k);
  const int ntile = sel4(plan.ntile, i);
  const int cblk = wg / ntile, tile = wg - cblk * ntile;
  const int c0 = cblk * kCB, nc = min(kCB, C - c0);
  const int PT = hw < kWG ? hw : kWG;               // pixels per tile
  const int nsub = kWG / PT;                        // channel subgroups sharing a pixel; when PT does not divide the
                                                    // workgroup (10 x 10 maps of a 320 x 320 crop) the last
                                                    // kWG - nsub * PT threads have no (pixel, subgroup) and idle in (5)
  const int q0 = tile * PT;
  // LDS carve (ints): stencil pixels [4R] (later: the per-pixel entry lists), stencil weights [4R], row list [R],
  // contribution keys [4R], contribution targets [4R bytes, 16-byte aligned], rows [R][kCB]
  int* tq = ldsraw;
  int* ent = ldsraw;                                 // aliases tq: the stencil pixels are dead once (3) is done
  float* tw = reinterpret_cast<float*>(ldsraw + 4 * R);
  int* list = ldsraw + 8 * R;
  int* ckey = ldsraw + 9 * R + 4;
  unsigned char* ctgt = reinterpret_cast<unsigned char*>(ldsraw + ((13 * R + 8 + 3) & ~3));
  float* rows = reinterpret_cast<float*>(ldsraw + ((13 * R + 8 + 3) & ~3) + ((R + 3) & ~3) + 4);   // [kRB][kCB]
  // ---- (1) stencils + ordered compaction of the rows that reach this tile
  if (tid == 0) { nlist = 0; ncon = 0; }
  cnt[tid] = 0;
  __syncthreads();
  const bool dropped = keep != nullptr && keep[b] == 0;
  for (int rb = 0; rb < R; rb += kWG) {
    const int r = rb + tid;
    bool rel = false;
    if (r < R) {
      const int p = (int)pix[(int64_t)b * R + r];
      const int py = p / w0, px = p - py * w0;
      int4 q4;
      float4 w4;
      if (i == 0) {
        q4 = make_int4(p, -1, -1, -1);
        w4 = make_float4(1.f, 0.f, 0.f, 0.f);
      } else {
        const Taps t = bilinear_taps(py, px, hi, wi, (float)hi / (float)h0, (float)wi / (float)w0);
        q4 = make_int4(t.y0 * wi + t.x0, t.y0 * wi + t.x1, t.y1 * wi + t.x0, t.y1 * wi + t.x1);
        w4 = make_float4(t.hy * t.hx, t.hy * t.lx, t.ly * t.hx, t.ly * t.lx);
      }
      *reinterpret_cast<int4*>(tq + 4 * r) = q4;
      *reinterpret_cast<float4*>(tw + 4 * r) = w4;
      const int lo = q0, up = q0 + PT;
      rel = ((q4.x >= lo && q4.x < up) || (q4.y >= lo && q4.y < up) || (q4.z >= lo && q4.z < up) ||
             (q4.w >= lo && q4.w < up)) && !(dropped && r < S);
    }
    const unsigned long long mask = __ballot(rel);
    if (lane == 0) wcount[wave] = __popcll(mask);
    __syncthreads();
    int base = nlist;
    for (int w = 0; w < wave; ++w) base += wcount[w];
    if (rel) list[base + __popcll(mask & ((1ull << lane) - 1ull))] = r;
    __syncthreads();
    if (tid == 0) nlist += wcount[0] + wcount[1] + wcount[2] + wcount[3];
    __syncthreads();
  }
  const int n = nlist;
  // ---- (3) in-tile contributions (row li, tap t) compacted in (li, t) order; per-pixel counts
  for (int lb = 0; lb < n; lb += kWG) {
    const int li = lb + tid;
    int ql[4], mine = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) ql[t] = -1;
    if (li < n) {
      const int4 q4 = *reinterpret_cast<const int4*>(tq + 4 * list[li]);
      const int qq[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int v = qq[t] - q0;
        if (v >= 0 && v < PT) { ql[t] = v; ++mine; }
      }
    }
    scan[tid] = mine;
    __syncthreads();
    int v = mine;                                    // inclusive block scan of the per-thread counts
    for (int d = 1; d < kWG; d <<= 1) {
      const int add = tid >= d ? scan[tid - d] : 0;
      __syncthreads();
      v += add;
      scan[tid] = v;
      __syncthreads();
    }
    int pos = ncon + v - mine;
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (ql[t] >= 0) {
        ckey[pos] = li * 4 + t;
        ctgt[pos] = (unsigned char)ql[t];
        atomicAdd(&cnt[ql[t]], 1);                   // integer: order-independent
        ++pos;
      }
    __syncthreads();
    if (tid == kWG - 1) ncon += v;
    __syncthreads();
  }
  const int nco = ncon;
  // exclusive prefix of the per-pixel counts
  const int mycnt = cnt[tid];
  {
    int v = mycnt;
    scan[tid] = v;
    __syncthreads();
    for (int d = 1; d < kWG; d <<= 1) {
      const int add = tid >= d ? scan[tid - d] : 0;
      __syncthreads();
      v += add;
      scan[tid] = v;
      __syncthreads();
    }
    cnt[tid] = v - mycnt;                            // offset of pixel `tid`
  }
  __syncthreads();                                   // tq is dead from here: `ent` may overwrite it
  // ---- (4) owner scan: 16 targets per 128-bit read, hits appended in scan order
  if (tid < PT && mycnt > 0) {
    const uint32_t pat = (uint32_t)tid * 0x01010101u;
    int pos = cnt[tid];
    const int full = nco & ~15;
    for (int e0 = 0; e0 < full; e0 += 16) {
      const uint4 v = *reinterpret_cast<const uint4*>(ctgt + e0);
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint32_t hit = zero_bytes(w[j] ^ pat);
        while (hit) {
          const int byte = (__ffs((int)hit) - 1) >> 3;
          hit &= hit - 1;
          ent[pos++] = ckey[e0 + 4 * j + byte];
        }
      }
    }
    for (int e = full; e < nco; ++e)
      if (ctgt[e] == (unsigned char)tid) ent[pos++] = ckey[e];
  }
  __syncthreads();
  // ---- (5) owner-computes accumulation, the listed rows staged kRB at a time (a pixel's list is sorted by row, so a
  // batch is a contiguous piece of it): 18 KB of LDS for the rows instead of R x 36 floats, four workgroups per CU
  const int px_ = tid % PT, sub = tid / PT;
  const int q = q0 + px_;
  const bool live = q < hw && sub < nsub;
  float acc[kCB];
#pragma unroll
  for (int k = 0; k < kCB; ++k) acc[k] = 0.f;
  int cur_ = live ? cnt[px_] : 0;
  const int up = live ? (px_ + 1 < kWG ? cnt[px_ + 1] : nco) : 0;
  const float* src = dxs + ((int64_t)m * B * R + (int64_t)b * R) * ld + coff + c0;
  for (int lb = 0; lb < n; lb += kRB) {
    const int nb = min(kRB, n - lb);
    if (lb > 0) __syncthreads();                     // the previous batch has been consumed
    // coalesced over the channels; independent loads, several in flight per thread
#pragma unroll 6
    for (int e = tid; e < nb * kCB; e += kWG) {
      const int li = e / kCB, k = e - li * kCB;
      rows[e] = k < nc ? src[(int64_t)list[lb + li] * ld + k] : 0.f;
    }
    __syncthreads();
    const int lim = (lb + nb) * 4;                   // keys of this batch: li * 4 + t < lim
    while (cur_ < up) {
      const int key = ent[cur_];
      if (key >= lim) break;
      const int li = key >> 2;
      const float wq = tw[4 * list[li] + (key & 3)];
      const float* row = rows + (li - lb) * kCB + sub;
#pragma unroll
      for (int k = 0; k < kCB; ++k)
        if (sub + k * nsub < kCB) acc[k] = fmaf(wq, row[k * nsub], acc[k]);
      ++cur_;
    }
  }
  if (!live) return;
  const float inv = 1.f / (float)hw;
  float* out = sel8(g.p, m * 4 + i) + ((int64_t)b * C + c0) * hw + q;
  const float* dp = dpooled != nullptr ? dpooled + ((int64_t)m * B + b) * Ctot + coff + c0 : nullptr;
#pragma unroll
  for (int k = 0; k < kCB; ++k) {
    const int c = sub + k * nsub;
    if (c < nc) out[(int64_t)c * hw] = fmaf(sc, acc[k], dp != nullptr ? dp[c] * inv : 0.f);
  }
}

// total = sum(losses6) + the five differentiable feature-map terms (learning/contrast_trainer.py:980)
__global__ void section_total_kernel(const float* __restrict__ losses, const float* __restrict__ meters,
                                     float* __restrict__ total) {
  if (threadIdx.x == 0) {
    float t = ((losses[0] + losses[1]) + (losses[2] + losses[3])) + (losses[4] + losses[5]);
    if (meters != nullptr) t += ((meters[0] + meters[1]) + (meters[4] + meters[5])) + meters[8];
    total[0] = t;
  }
}

}  // namespace

extern "C" {

int hcm_heads_forward(hcm_branches enc1, hcm_branches enc2, const float* feat3, int B, int J, int Ctot, int D3,
                      int F, const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                      const float* b3, const int64_t* index, float* pooled, float* mean3, float* ypre, float* f,
                      int ldf, float* fT, hcm_stream_t stream) {
  if (B <= 0 || J <= 0 || F <= 0 || F > kWG || Ctot <= 0 || Ctot > kMaxIn || D3 <= 0 || D3 > kMaxIn ||
      ldf < 3 * F + (index != nullptr ? 2 : 0) || !branches_ok(enc1, Ctot) || (!absent(enc2) && !branches_ok(enc2, Ctot)))
    return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  const int nmod = absent(enc2) ? 1 : 2;          // absent: pooled[1] is the caller's (already filled, same stream)
  const int planes = nmod * B * Ctot;
  for (int i = 0; i < 4 && nmod == 2; ++i)
    if (enc1.C[i] != enc2.C[i] || enc1.H[i] != enc2.H[i] || enc1.W[i] != enc2.W[i]) return (int)hipErrorInvalidValue;
  head_pool_kernel<<<(planes + 3) / 4, kWG, 0, s>>>(pack8(enc1, nmod == 2 ? enc2 : enc1), B, Ctot, pooled, nmod);
  HCM_CHECK_LAUNCH();
  heads_fwd_kernel<<<dim3(B, 3), kWG, 0, s>>>(pooled, feat3, B, J, Ctot, D3, F, W1, b1, W2, b2, W3, b3, mean3, ypre,
                                              f, ldf, fT, index);
  HCM_CHECK_LAUNCH();
  return 0;
}

int hcm_heads_backward(const float* gfT, const float* scale, const float* pooled, const float* mean3,
                       const float* ypre, int B, int J, int Ctot, int D3, int F, const float* W1, const float* W2,
                       const float* W3, const float* gfeat3_joint, float* dyws, float* dW1, float* db1, float* dW2,
                       float* db2, float* dW3, float* db3, float* dpooled, float* gfeat3, hcm_stream_t stream) {
  if (B <= 0 || J <= 0 || F <= 0 || F > kWG || Ctot <= 0 || Ctot > kMaxIn || D3 <= 0 || D3 > kMaxIn)
    return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  heads_bwd_kernel<<<dim3(B, 3), kWG, 0, s>>>(gfT, scale, ypre, B, J, Ctot, D3, F, W1, W2, W3, gfeat3_joint, dyws,
                                              dpooled, gfeat3);
  HCM_CHECK_LAUNCH();
  const int cmax = Ctot > D3 ? Ctot : D3;
  heads_dw_kernel<<<dim3((F * cmax + kWG - 1) / kWG, 3), kWG, 0, s>>>(dyws, pooled, mean3, B, Ctot, D3, F, dW1, db1,
                                                                     dW2, db2, dW3, db3);
  HCM_CHECK_LAUNCH();
  return 0;
}

int hcm_pixel_sample(const float* depth_mask, int B, int H, int W, int h, int w, int S, const int32_t* use_depth,
                     const float* joints2d, int J, uint64_t seed, uint64_t offset, int64_t* pix, int64_t* coord,
                     int32_t* keep, hcm_stream_t stream) {
  if (B <= 0 || H <= 0 || W <= 0 || h <= 0 || w <= 0 || S <= 0 || J < 0 || h != w || (size_t)h * w * 4 > 150 * 1024)
    return (int)hipErrorInvalidValue;
  const size_t lds = (size_t)h * w * sizeof(int);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pixel_sample_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  pixel_sample_kernel<<<B, kWG, lds, (hipStream_t)stream>>>(depth_mask, B, H, W, h, w, S, use_depth, joints2d, J, seed,
                                                            offset, pix, coord, keep);
  HCM_CHECK_LAUNCH();
  return 0;
}

int hcm_sample_branches_ld(int Ctot) { return (Ctot + 1 + 3) & ~3; }

int hcm_sample_branches(hcm_branches enc1, hcm_branches enc2, int B, const int64_t* pix, int R, int Ctot, int F,
                        const float* Wp1, const float* bp1, const float* Wp2, const float* bp2, float* xs, float* Wpad,
                        float* grows, hcm_stream_t stream) {
  if (B <= 0 || R <= 0 || F <= 0 || !branches_ok(enc1, Ctot) || !branches_ok(enc2, Ctot))
    return (int)hipErrorInvalidValue;
  for (int i = 0; i < 4; ++i)
    if (enc1.C[i] != enc2.C[i] || enc1.H[i] != enc2.H[i] || enc1.W[i] != enc2.W[i]) return (int)hipErrorInvalidValue;
  const int ld = hcm_sample_branches_ld(Ctot);
  const int nrowblocks = (2 * B * R + 3) / 4;
  int npack = (2 * F * ld + kWG - 1) / kWG;
  if (npack > 64) npack = 64;
  sample_branches_kernel<<<nrowblocks + npack, kWG, 0, (hipStream_t)stream>>>(pack8(enc1, enc2), B, pix, R, Ctot, F, ld,
                                                                             Wp1, bp1, Wp2, bp2, xs, Wpad, grows,
                                                                             nrowblocks);
  HCM_CHECK_LAUNCH();
  return 0;
}

int hcm_branch_grad(const float* dxs, const float* dpooled, const float* scale, const int64_t* pix, int R, int B,
                    int Ctot, hcm_branches_out g1, hcm_branches_out g2, const int32_t* keep, int S, const float* dWpad,
                    int F, float* dWp1, float* dbp1, float* dWp2, float* dbp2, hcm_stream_t stream) {
  if (B <= 0 || R < 0 || F <= 0 || !branches_ok(g1, Ctot) || !branches_ok(g2, Ctot) || (R > 0 && (dxs == nullptr || pix == nullptr)))
    return (int)hipErrorInvalidValue;
  const int ld = hcm_sample_branches_ld(Ctot);
  GradPlan plan;
  int v = 0;
  for (int i = 0; i < 4; ++i) {
    if (g1.C[i] != g2.C[i] || g1.H[i] != g2.H[i] || g1.W[i] != g2.W[i]) return (int)hipErrorInvalidValue;
    const int hw = g1.H[i] * g1.W[i];
    plan.ntile[i] = (hw + kWG - 1) / kWG;
    plan.first[i] = v;
    v += plan.ntile[i] * ((g1.C[i] + kCB - 1) / kCB);
  }
  plan.first[4] = v;
  const size_t lds = ((size_t)R * 14 + (size_t)kRB * kCB + 32) * sizeof(int);
  if (lds > 150 * 1024) return (int)hipErrorInvalidValue;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(branch_grad_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  branch_grad_kernel<<<dim3(v, B + 1, 2), kWG, lds, (hipStream_t)stream>>>(dxs, ld, dpooled, scale, pix, R, B, Ctot,
                                                                         pack8(g1, g2), plan, keep, S, dWpad, F, dWp1, dbp1,
                                                                         dWp2, dbp2);
  HCM_CHECK_LAUNCH();
  return 0;
}

int hcm_section_total(const float* losses6, const float* meters9, float* total, hcm_stream_t stream) {
  if (losses6 == nullptr || total == nullptr) return (int)hipErrorInvalidValue;
  section_total_kernel<<<1, 64, 0, (hipStream_t)stream>>>(losses6, meters9, total);
  HCM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
