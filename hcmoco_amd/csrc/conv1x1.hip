// 1x1 convolutions of the PointNet++ shared MLPs on ball tensors [B, C, npoint, nsample] (reference:
// networks/pointnet2/pytorch_utils.py:5-33 -> nn.Conv2d(kernel_size=1, bias=False) inside SharedMLP), gfx950, r05 / r06.
//
// MIOpen answers these shapes -- 16..512 channels on maps of 1 K .. 131 K positions -- with Winograd-class and generic GEMM
// kernels (profiles/r05_hrnetpn_timeline.txt: 3.5 ms forward, 4.3 ms data gradient per HRNetPN step for 4 GB of tensors
// that HBM moves in ~1 ms).  With positions contiguous (NCHW) a 1x1 convolution is, per image,
//     Z[k][p] = sum_c W[k][c] X[c][p]          (forward;  data gradient: the same with W^T: dX[c][p] = sum_k W[k][c] dZ[k][p])
// No transposes, no layout change; LDS only for the weights; every activation byte is read once per output-channel block and
// written once.  Three kernels (r06; numbers: profiles/r06_conv1x1_layers.txt):
//   conv1x1_split_kernel  the layers whose fp32 MFMA floor reaches their HBM floor (R * M >= 4096, R % 32 == 0): bf16 matrix cores
//                         on split operands, three terms, fp32 accumulate (4.4e-6 of float64 as a vector, the fmap.hip scheme);
//   conv1x1_rows_kernel   exact fp32 (v_mfma_f32_16x16x4_f32 = an fmaf chain per output element): the narrow layers, and every
//                         layer under hcm_conv1x1_set_arith(1);
//   conv1x1_kernel        r05's form, the fallback for channel counts the two above do not take (R % 16 != 0 or R > 512).
// The first two share the r06 findings: W' as the A operand and X as the B operand, so that one 16-byte load / store per lane
// covers 256 contiguous bytes of a channel row per 16 lanes; loads kept in flight across the whole stream of a wave's blocks, in
// straight-line code and refilled in place (what the compiler does to any other form is recorded at the kernels).
#include <atomic>
#include "hcm_common.h"
#include "../../include/hcmoco_hip.h"

namespace {

using namespace hcm;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));

// x = hi + mid + lo EXACTLY (three bf16 pieces, 3 x 8 significand bits): hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid),
// both residuals exact in fp32.  A product of two such operands on the bf16 matrix cores, fp32 accumulate, keeps six of the nine
// piece products: hi hi, hi mid, mid hi, mid mid, hi lo, lo hi -- every one exact in fp32 -- and drops mid lo, lo mid, lo lo
// (<= 2^-23 of the product together): fp32-class accuracy (measured against float64 beside the fp32 MFMA form in
// tests/test_pointnet2_gpu.py) for 6 x 16 matrix-pipe cycles per 32-deep contraction where eight v_mfma_f32_16x16x4_f32 take 256.
// (Two pieces / three terms -- fmap.hip's scheme, 4.4e-6 of float64 -- was built first and measured 10 % faster still, but it
// failed the reference gate of row a18 in train mode: level-4 features 1.23 x the 1e-4 / 1e-5 bound, BatchNorm over two clouds
// amplifies it; profiles/r06_conv1x1_layers.txt.)
__device__ __forceinline__ void split3(const v2f x2, int& h, int& m, int& l) {
  const v2bf h2 = __builtin_convertvector(x2, v2bf);
  const v2f r1 = x2 - __builtin_convertvector(h2, v2f);
  const v2bf m2 = __builtin_convertvector(r1, v2bf);
  const v2bf l2 = __builtin_convertvector(r1 - __builtin_convertvector(m2, v2f), v2bf);
  h = __builtin_bit_cast(int, h2);
  m = __builtin_bit_cast(int, m2);
  l = __builtin_bit_cast(int, l2);
}
// eight consecutive floats (two float4) -> the three fragments of a v_mfma_f32_16x16x32_bf16 operand
__device__ __forceinline__ void split8(const v4f& lo4, const v4f& hi4, v8bf& fh, v8bf& fm, v8bf& fl) {
  v4i ph, pm, pl;
#pragma unroll
  for (int j = 0; j < 4; ++j)
  {
    int h, m, l;
    split3(j < 2 ? (v2f){lo4[2 * j], lo4[2 * j + 1]} : (v2f){hi4[2 * j - 4], hi4[2 * j - 3]}, h, m, l);
    ph[j] = h; pm[j] = m; pl[j] = l;
  }
  fh = __builtin_bit_cast(v8bf, ph);
  fm = __builtin_bit_cast(v8bf, pm);
  fl = __builtin_bit_cast(v8bf, pl);
}
// Ordering fences for the in-place load rings below.  sched_barrier only binds the machine scheduler; IR passes are free to
// hoist a (side-effect-free) refill load above the arithmetic that reads the slot it refills -- the two then need two register
// sets, and the copies land on the loop's back edge behind s_waitcnt vmcnt(0) (seen three times in this file's ISA).  An empty
// asm with a "memory" clobber that also CONSUMES the results of that arithmetic cannot be crossed by either: the loads stay
// behind it, the arithmetic in front of it.
#define HCM_FENCE() asm volatile("" ::: "memory")
template <typename T>
__device__ __forceinline__ void pin(T& v) { asm volatile("" : "+v"(v) :: "memory"); }

// TRANS: the data gradient (W^T).  W is [Kw][Cw] row-major as nn.Conv2d stores it; M = output rows of this product
// (forward: Kw, data gradient: Cw), R = its reduction length (forward: Cw, data gradient: Kw).
// The MFMA runs transposed -- A = a 16-position tile of X^T, B = 16 channels of W^T -- so that a lane's four accumulator
// registers are four CONSECUTIVE positions of one output channel: one 16-byte store per (channel tile, position tile)
// instead of four 4-byte ones (the forward pass writes twice what it reads).
// r05's kernel; since r06 only the fallback for channel counts conv1x1_rows_kernel / conv1x1_split_kernel do not take.
// Three r06 variants of THIS kernel with 16-byte loads and deeper prefetch all measured slower (profiles/r06_conv1x1_layers.txt,
// first part) and were read as "the layer is bound by its 96 row streams, not by load instructions".  The ISA says otherwise:
// their conditional prefetch loads (if (rn < rows) ... else if (next) ...) made the compiler drain vmcnt(0) behind every load,
// and their ring registers were copied on the loop's back edge behind another full drain -- no load was ever in flight across
// a k-step.  The kernels below keep the loads unconditional, in program order and in place.
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


// r06, exact fp32: W' is the A operand and X the B operand -- D = W' X, lane (np, g) register q = Z[m0 + 16 mt + 4 g + q][position
// of column np].  A sum over channels does not care which position an MFMA column stands for: the lane loads ONE float4
// X[c0 + g][p0 + 4 np .. + 3] per k-step (16 lanes = 256 contiguous bytes of a channel row) and feeds element t to MFMA number t,
// whose column np is then position p0 + 4 np + t; register q of the four MFMAs is four CONSECUTIVE positions of output row
// m0 + 16 mt + 4 g + q: one 16-byte store, and the 16 lanes of a row write 256 contiguous bytes (r05's form: 64-byte pieces of 16
// rows per instruction and four 4-byte loads per k-step).  The k-steps of all PB blocks of a wave are one stream, PF loads in flight.
// An fp32 MFMA is an fmaf chain in k order: results are those of r05's kernel bit for bit.
template <int MT, int PF, bool TRANS, bool FULL>
__global__ __launch_bounds__(256) void conv1x1_rows_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                           float* __restrict__ Z, int M, int R, int Cw, int P, int PB,
                                                           int mblocks) {
  extern __shared__ float Ws[];                        // Ws[r][np][mt] = W'[m0 + 16 mt + np][r], all R rows, staged once
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: loop bounds in SGPRs
  const int np = lane & 15, g = lane >> 4;
  // (position block, output-channel block): the blocks of one position block read the same rows of X -- ids 8 apart land on the
  // same XCD back to back (wgrad1x1_ball_kernel's mapping)
  const int pbn = gridDim.x / mblocks, id = blockIdx.x;
  int pblk, mblk;
  if ((pbn & 7) == 0) {
    const int group = id / (8 * mblocks), r = id - group * (8 * mblocks);
    mblk = r >> 3;
    pblk = group * 8 + (r & 7);
  } else {
    mblk = id / pbn;
    pblk = id - mblk * pbn;
  }
  const int m0 = mblk * (16 * MT);
  for (int e = threadIdx.x; e < R * 16 * MT; e += 256) {
    int r, m;
    if (TRANS) { r = e / (16 * MT); m = e - r * (16 * MT); }            // W[r][m]: m contiguous
    else { m = e / R; r = e - m * R; }                                  // W[m][r]: r contiguous
    const float v = m0 + m < M ? (TRANS ? W[(size_t)r * Cw + m0 + m] : W[(size_t)(m0 + m) * Cw + r]) : 0.f;
    Ws[r * (16 * MT) + (m & 15) * MT + (m >> 4)] = v;
  }
  __syncthreads();
  // this wave's blocks of 64 positions: pos0(i) = first + i * 256, the first na of the PB lie inside the row (P % 64 == 0)
  const int first = (pblk * PB * 4 + wave) * 64;
  const int na = first < P ? min(PB, (P - first + 255) >> 8) : 0;
  if (na == 0) return;
  const float* xb = X + (size_t)blockIdx.z * R * P + first + 4 * np + (size_t)g * P;
  float* zb = Z + (size_t)blockIdx.z * M * P + first + 4 * np;
  // The k-steps (4 channels each) of all na blocks are ONE stream; the load PF steps ahead is issued before the MFMAs of a
  // step, UNCONDITIONALLY and in straight-line code (a load behind a branch makes the compiler drain vmcnt at the join -- what
  // slowed the three r06 pipeline variants above); behind the last block the stream re-reads its last rows (PF cache hits).
  // The ring is refilled IN PLACE: slot u is loaded again right after the MFMAs that read it have been issued (a load issued
  // ahead of them needs a second register set, and the compiler then copies ring registers on the loop's back edge behind a
  // full vmcnt drain -- seen in the ISA of the first attempt), so a load is PF - 1 k-steps ahead of its use.
  const float* pf_ptr = xb;                            // row pf_r of block pf_i: PF steps ahead of the stream
  int pf_r = 0, pf_i = 0;
  const size_t P4 = 4 * (size_t)P;
  auto group_done = [&]() {                            // PF k-steps further; R % (4 PF) == 0, so a block ends between groups
    pf_r += 4 * PF;
    pf_ptr += PF * P4;
    if (pf_r == R) {                                   // behind the last block the stream stays on its last rows (cache-hot)
      const bool end = pf_i + 1 >= na;
      pf_r = end ? R - 4 * PF : 0;
      pf_i = end ? pf_i : pf_i + 1;
      pf_ptr = xb + pf_i * 256 + (size_t)pf_r * P;
    }
  };
  v4f xq[PF];
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    xq[u] = *reinterpret_cast<const v4f*>(pf_ptr + u * P4);
    __builtin_amdgcn_sched_barrier(0);                 // in program order (the loop's vmcnt waits count on it)
  }
  group_done();
  for (int i = 0; i < na; ++i) {
    v4f acc[4][MT];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[t][mt] = (v4f){0.f, 0.f, 0.f, 0.f};
    for (int r0 = 0; r0 < R; r0 += 4 * PF) {           // R % (4 PF) == 0
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        float wv[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) wv[mt] = Ws[(r0 + 4 * u + g) * (16 * MT) + np * MT + mt];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int t = 0; t < 4; ++t)
            acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[mt], xq[u][t], acc[t][mt], 0, 0, 0);
        HCM_FENCE();
        __builtin_amdgcn_sched_barrier(0);
        xq[u] = *reinterpret_cast<const v4f*>(pf_ptr + u * P4);
        HCM_FENCE();
        __builtin_amdgcn_sched_barrier(0);
      }
      group_done();
    }
    // acc[t][mt][q] = Z[m0 + 16 mt + 4 g + q][first + 256 i + 4 np + t]
    float* z = zb + i * 256;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = m0 + 16 * mt + 4 * g + q;
        if (FULL || m < M)
          *reinterpret_cast<v4f*>(z + (size_t)m * P) = (v4f){acc[0][mt][q], acc[1][mt][q], acc[2][mt][q], acc[3][mt][q]};
      }
  }
}

// r06: the same product on the bf16 matrix cores with operands split in three pieces, six terms (split3 above): fp32-class results
// for 96 matrix-pipe cycles per k-step of 32 channels and 16 x 16 outputs instead of 256, which takes the layers with 64+ channels
// (fp32 MFMA floor >= HBM floor) off the matrix pipes.  A k-step is 32 channels: lane (np, g) loads the float4
// X[r0 + 8 g + e][p0 + 4 np .. + 3], e < 8 (eight rows x 256 contiguous bytes per instruction), converts them to the B fragments of
// the four position sub-tiles t (element e of fragment t = row 8 g + e at position 4 np + t) and refills the slot at once; W' lies
// pre-split in LDS as [piece][r / 8][m][8] bf16, one 16-byte read per fragment.  Two k-steps in flight per wave (16 KB), stream of
// all the wave's blocks as in conv1x1_rows_kernel.
template <int MT, bool TRANS, bool FULL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv1x1_split_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                            float* __restrict__ Z, int M, int R, int Cw, int P, int PB,
                                                            int mblocks) {
  extern __shared__ __attribute__((aligned(16))) __bf16 Wl[];     // [hi | mid | lo][R / 8][16 MT][8]
  constexpr int MW = 16 * MT;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: loop bounds in SGPRs
  const int np = lane & 15, g = lane >> 4;
  const int pbn = gridDim.x / mblocks, id = blockIdx.x;
  int pblk, mblk;
  if ((pbn & 7) == 0) {
    const int group = id / (8 * mblocks), r = id - group * (8 * mblocks);
    mblk = r >> 3;
    pblk = group * 8 + (r & 7);
  } else {
    mblk = id / pbn;
    pblk = id - mblk * pbn;
  }
  const int m0 = mblk * MW;
  __bf16* Wh = Wl;
  const int plane = R * MW;
  for (int e = threadIdx.x; e < R * MW; e += 256) {
    int r, m;
    if (TRANS) { r = e / MW; m = e - r * MW; }                          // W[r][m]: m contiguous
    else { m = e / R; r = e - m * R; }                                  // W[m][r]: r contiguous
    const float v = m0 + m < M ? (TRANS ? W[(size_t)r * Cw + m0 + m] : W[(size_t)(m0 + m) * Cw + r]) : 0.f;
    const __bf16 h = (__bf16)v;
    const float r1 = v - (float)h;
    const __bf16 md = (__bf16)r1;
    const int o = ((r >> 3) * MW + m) * 8 + (r & 7);
    Wh[o] = h;
    Wh[plane + o] = md;
    Wh[2 * plane + o] = (__bf16)(r1 - (float)md);
  }
  __syncthreads();
  const int first = (pblk * PB * 4 + wave) * 64;
  const int na = first < P ? min(PB, (P - first + 255) >> 8) : 0;
  if (na == 0) return;
  // addresses = a wave-uniform base (SGPRs) + ONE 32-bit lane offset: the saddr form of global_load / global_store, no
  // per-row address registers
  const float* xb = X + (size_t)blockIdx.z * R * P + first;
  float* zb = Z + ((size_t)blockIdx.z * M + m0) * P + first;
  const unsigned xlane = (unsigned)(4 * np + 8 * g * P) * 4u, zlane = (unsigned)(4 * np + 4 * g * P) * 4u;     // bytes
  const int KS = R >> 5;                               // k-steps per block
  // rows pf_r + 8 g + e of block pf_i: two k-steps ahead of the stream.  No branches here: selects keep the priming loads in one
  // basic block, where the sched_barriers hold them in program order -- the loop's vmcnt waits are the minimum over both ways
  // into it, and a priming the compiler had reordered (the slot read first loaded last) made every one of them vmcnt(0).
  int pf_r = 0, pf_i = 0;
  auto advance = [&]() {
    const int r = pf_r + 32;
    const bool wrap = r == R, end = wrap && pf_i + 1 >= na;      // behind the last block: stay on its last rows (cache-hot)
    pf_i = wrap && !end ? pf_i + 1 : pf_i;
    pf_r = end ? pf_r : wrap ? 0 : r;
  };
  auto pf_ptr = [&]() { return xb + pf_i * 256 + (size_t)pf_r * P; };
  v4f xq[2][8];
  const float* const prime0 = pf_ptr();
  advance();
  const float* const prime1 = pf_ptr();
  advance();
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      xq[h][e] = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>((h ? prime1 : prime0) + (size_t)e * P) + xlane);
      __builtin_amdgcn_sched_barrier(0);
    }
  v4f acc[4][MT];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[t][mt] = (v4f){0.f, 0.f, 0.f, 0.f};
  int ks = 0, blk = 0;
  const int total = (na * KS + 1) & ~1;                // an odd stream runs one k-step more on re-read rows; it is never stored
  int s = 0;
  do {                                                 // total >= 2: no guard branch for the priming loads to sink behind
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      v8bf bh[4], bm[4], bl[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        v4i ph, pm, pl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int vh, vm, vl;
          split3((v2f){xq[h][2 * j][t], xq[h][2 * j + 1][t]}, vh, vm, vl);
          ph[j] = vh; pm[j] = vm; pl[j] = vl;
        }
        pin(ph);                                                 // the conversions that read the slot are in front ...
        pin(pm);
        pin(pl);
        bh[t] = __builtin_bit_cast(v8bf, ph);
        bm[t] = __builtin_bit_cast(v8bf, pm);
        bl[t] = __builtin_bit_cast(v8bf, pl);
      }
      __builtin_amdgcn_sched_barrier(0);
      {
        const float* const src = pf_ptr();
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xq[h][e] = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(src + (size_t)e * P) + xlane);
          __builtin_amdgcn_sched_barrier(0);
        }
        HCM_FENCE();                                               // ... and its refill stays in front of the MFMAs
      }
      advance();
      const __bf16* wrow = Wh + ((size_t)(4 * ks + g) * MW + np) * 8;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const v8bf ah = *reinterpret_cast<const v8bf*>(wrow + mt * 128);
        const v8bf am = *reinterpret_cast<const v8bf*>(wrow + plane + mt * 128);
        const v8bf al = *reinterpret_cast<const v8bf*>(wrow + 2 * plane + mt * 128);
        // term by term over the four sub-tiles: four independent accumulators between two MFMAs of one chain
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[t], acc[t][mt], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[t], acc[t][mt], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm[t], acc[t][mt], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh[t], acc[t][mt], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm[t], acc[t][mt], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[t], acc[t][mt], 0, 0, 0);
      }
      if (++ks == KS) {
        if (blk < na) {
          // acc[t][mt][q] = Z[m0 + 16 mt + 4 g + q][first + 256 blk + 4 np + t]
          float* z = zb + blk * 256;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              if (FULL || m0 + 16 * mt + 4 * g + q < M)
                *reinterpret_cast<v4f*>(reinterpret_cast<char*>(z + (size_t)(16 * mt + q) * P) + zlane) =
                    (v4f){acc[0][mt][q], acc[1][mt][q], acc[2][mt][q], acc[3][mt][q]};
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[t][mt] = (v4f){0.f, 0.f, 0.f, 0.f};
        ks = 0;
        ++blk;
      }
    }
    s += 2;
  } while (s < total);
}

// 0: split-bf16 where the fp32 MFMA floor reaches the HBM floor (default); 1: exact fp32 everywhere (hcm_conv1x1_set_arith)
std::atomic<int> g_arith{0};

template <bool TRANS>
int launch(const float* X, const float* W, float* Z, int N, int M, int R, int Cw, int P, hipStream_t st, bool exact) {
  if (R % 16 == 0 && R <= 512) {
    const bool split = !exact && g_arith.load(std::memory_order_relaxed) == 0 && R % 32 == 0 && (long long)R * M >= 4096;
    // all of W' of an output-channel block in LDS, <= 64 KB (two workgroups per CU and more): R x 16 MT floats, or the same
    // elements as three bf16 planes
    const size_t per_mt = (size_t)R * 16 * (split ? 3 * sizeof(__bf16) : sizeof(float));
    int mt = M <= 16 ? 1 : M <= 32 ? 2 : 4;
    while (mt > 1 && per_mt * mt > 65536) mt >>= 1;
    const int mb = (M + 16 * mt - 1) / (16 * mt);
    const bool full = M % (16 * mt) == 0;
    const size_t lds = per_mt * mt;
    // PB blocks of 256 positions per workgroup: as many as leave >= 2048 workgroups (8 per CU)
    int pbv = 1;
    while (pbv < 8 && (long long)((P + 512 * pbv - 1) / (512 * pbv)) * mb * N >= 2048) pbv *= 2;
    const int pbn = (P + 256 * pbv - 1) / (256 * pbv);
    const int pf = R % 32 == 0 ? 8 : 4;
    hipError_t e = hipSuccess;
    // (the LDS attribute is per DEVICE: set on every launch)
#define HCM_GO(KERNEL)                                                                                        \
  do {                                                                                                        \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e == hipSuccess) KERNEL<<<dim3(pbn * mb, 1, N), 256, lds, st>>>(X, W, Z, M, R, Cw, P, pbv, mb);           \
  } while (0)
#define HCM_PICK(MT_)                                                                       \
  do {                                                                                      \
    if (split) {                                                                            \
      if (full) HCM_GO((conv1x1_split_kernel<MT_, TRANS, true>));                           \
      else HCM_GO((conv1x1_split_kernel<MT_, TRANS, false>));                               \
    } else if (pf == 8) {                                                                   \
      if (full) HCM_GO((conv1x1_rows_kernel<MT_, 8, TRANS, true>));                         \
      else HCM_GO((conv1x1_rows_kernel<MT_, 8, TRANS, false>));                             \
    } else {                                                                                \
      if (full) HCM_GO((conv1x1_rows_kernel<MT_, 4, TRANS, true>));                         \
      else HCM_GO((conv1x1_rows_kernel<MT_, 4, TRANS, false>));                             \
    }                                                                                       \
  } while (0)
    if (mt == 1) HCM_PICK(1);
    else if (mt == 2) HCM_PICK(2);
    else HCM_PICK(4);
#undef HCM_PICK
#undef HCM_GO
    if (e != hipSuccess) return (int)e;
    HCM_CHECK_LAUNCH();
    return 0;
  }
  const int mblocks = M <= 32 ? 1 : (M + 63) / 64;
  // PB blocks of 256 positions per workgroup: as many as leave >= 2048 workgroups (8 per CU); one when W takes several rounds,
  // and one for the narrow outputs
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
// SPLIT (r06, the default arithmetic): the same sum on the bf16 matrix cores with three-piece operands, six terms (split3 above:
// fp32-class).  v_mfma_f32_16x16x32_bf16 contracts 32 positions; its lane (np, g) holds 8 CONSECUTIVE k of its row -- positions
// p + 8 g .. + 7, two float4 of the row as it lies (16 lanes x 4 g = 16 rows x 128 contiguous bytes per pair of loads, as before)
// -- converted to fragments (split8) on arrival: 6 x 16 cycles per 32 positions and 16 x 16 block where the fp32 form takes
// 8 x 32.  One step's operands in flight per wave; the X rows are refilled as soon as they are converted, the dZ rows tile by
// tile inside the MFMA phase (their fragments live for one tile only), all unconditionally: the last refill re-reads the last step.
template <int KT, int CT, bool SPLIT>
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
  const size_t pw = (size_t)run * 4 * L + (size_t)wave * L + (SPLIT ? 8 : 4) * g;
  const float* a_ptr = DZ + ((size_t)n * K + k0 + np) * P + pw;
  const float* b_ptr = X + ((size_t)n * C + c0 + np) * P + pw;
  const size_t tile = (size_t)16 * P;
  v4f acc[KT][CT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[kt][ct] = (v4f){0.f, 0.f, 0.f, 0.f};
  const int steps = L / 32;                            // even (L is a multiple of 64)
  if constexpr (SPLIT) {
    v4f a[KT][2], b[CT][2];
    auto load_a = [&](int kt, int s) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        a[kt][h] = *reinterpret_cast<const v4f*>(a_ptr + kt * tile + 32 * s + 4 * h);
        __builtin_amdgcn_sched_barrier(0);             // program order: the loop's vmcnt waits count on it
      }
    };
    auto load_b = [&](int s) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          b[ct][h] = *reinterpret_cast<const v4f*>(b_ptr + ct * tile + 32 * s + 4 * h);
          __builtin_amdgcn_sched_barrier(0);
        }
    };
    load_b(0);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) load_a(kt, 0);
    int s = 0;
    do {
      const int sn = s + 1 < steps ? s + 1 : s;
      v8bf bh[CT], bm[CT], bl[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) split8(b[ct][0], b[ct][1], bh[ct], bm[ct], bl[ct]);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) { pin(bh[ct]); pin(bm[ct]); pin(bl[ct]); }
      __builtin_amdgcn_sched_barrier(0);
      load_b(sn);
      HCM_FENCE();
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        v8bf ah, am, al;
        split8(a[kt][0], a[kt][1], ah, am, al);
        pin(ah); pin(am); pin(al);
        __builtin_amdgcn_sched_barrier(0);
        load_a(kt, sn);
        HCM_FENCE();
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[kt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[ct], acc[kt][ct], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[kt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[ct], acc[kt][ct], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[kt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm[ct], acc[kt][ct], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[kt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh[ct], acc[kt][ct], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[kt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm[ct], acc[kt][ct], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[kt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[ct], acc[kt][ct], 0, 0, 0);
      }
    } while (++s < steps);
  } else {
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
  load(0, 0);
  for (int s = 0; s < steps; s += 2) {
    load(1, s + 1);
    mul(0);
    if (s + 2 < steps) load(0, s + 2);
    mul(1);
  }
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

template <int KT, int CT, bool SPLIT>
int ball_wgrad_launch_arith(const float* x, const float* dz, float* partial, int C, int K, int P, const BallWgradGeo& g, hipStream_t st) {
  const size_t lds = (size_t)4 * KT * CT * 256 * sizeof(float);
  // the attribute is per DEVICE: set on every launch (as rowproj.hip / pointnet2.hip do), not once per process
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad1x1_ball_kernel<KT, CT, SPLIT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  wgrad1x1_ball_kernel<KT, CT, SPLIT><<<dim3(g.chunks, g.kblocks * g.cblocks), 256, lds, st>>>(x, dz, partial, C, K, P, g.L, g.cblocks);
  return 0;
}

template <int KT, int CT>
int ball_wgrad_launch(const float* x, const float* dz, float* partial, int C, int K, int P, const BallWgradGeo& g, hipStream_t st,
                      bool exact) {
  return !exact && g_arith.load(std::memory_order_relaxed) == 0
             ? ball_wgrad_launch_arith<KT, CT, true>(x, dz, partial, C, K, P, g, st)
             : ball_wgrad_launch_arith<KT, CT, false>(x, dz, partial, C, K, P, g, st);
}

int ball_wgrad(const float* x, const float* dy, int N, int C, int K, int H, int W, float* dw, void* workspace,
               size_t workspace_bytes, hipStream_t st, bool exact) {
  BallWgradGeo g;
  if (!x || !dy || !dw || !workspace || H <= 0 || W <= 0 || !ball_wgrad_geo(N, C, K, H * W, g)) return (int)hipErrorInvalidValue;
  if (workspace_bytes < (size_t)g.chunks * K * C * sizeof(float)) return (int)hipErrorInvalidValue;
  float* partial = static_cast<float*>(workspace);
  const int P = H * W;
  hcm::ProfSpan span(HCM_PROF_CONV1X1_DW, st, 4.0 * N * (double)(C + K) * P);
  int rc;
  if (g.kt == 4 && g.ct == 4) rc = ball_wgrad_launch<4, 4>(x, dy, partial, C, K, P, g, st, exact);
  else if (g.kt == 4 && g.ct == 2) rc = ball_wgrad_launch<4, 2>(x, dy, partial, C, K, P, g, st, exact);
  else if (g.kt == 4) rc = ball_wgrad_launch<4, 1>(x, dy, partial, C, K, P, g, st, exact);
  else if (g.kt == 2 && g.ct == 4) rc = ball_wgrad_launch<2, 4>(x, dy, partial, C, K, P, g, st, exact);
  else if (g.kt == 2 && g.ct == 2) rc = ball_wgrad_launch<2, 2>(x, dy, partial, C, K, P, g, st, exact);
  else if (g.kt == 2) rc = ball_wgrad_launch<2, 1>(x, dy, partial, C, K, P, g, st, exact);
  else if (g.ct == 4) rc = ball_wgrad_launch<1, 4>(x, dy, partial, C, K, P, g, st, exact);
  else if (g.ct == 2) rc = ball_wgrad_launch<1, 2>(x, dy, partial, C, K, P, g, st, exact);
  else rc = ball_wgrad_launch<1, 1>(x, dy, partial, C, K, P, g, st, exact);
  if (rc != 0) return rc;
  HCM_CHECK_LAUNCH();
  wgrad1x1_reduce_kernel<<<(K * C + 63) / 64, 256, 0, st>>>(partial, dw, K * C, g.chunks);
  HCM_CHECK_LAUNCH();
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
  return ball_wgrad(x, dy, N, C, K, H, W, dw, workspace, workspace_bytes, (hipStream_t)stream, false);
}

int hcm_conv1x1_ball_wgrad_exact(const float* x, const float* dy, int N, int C, int K, int H, int W, float* dw, void* workspace,
                                 size_t workspace_bytes, hcm_stream_t stream) {
  return ball_wgrad(x, dy, N, C, K, H, W, dw, workspace, workspace_bytes, (hipStream_t)stream, true);
}

int hcm_conv1x1_set_arith(int mode) {
  if (mode != 0 && mode != 1) return -1;
  return g_arith.exchange(mode, std::memory_order_relaxed);
}

int hcm_conv1x1_supported(int C, int K, int P) {
  return C > 0 && K > 0 && P > 0 && (C & 3) == 0 && (K & 3) == 0 && (P & 63) == 0 ? 1 : 0;
}

int hcm_conv1x1_forward(const float* x, const float* w, float* z, int N, int C, int K, int P, hcm_stream_t stream) {
  if (!x || !w || !z || N <= 0 || !hcm_conv1x1_supported(C, K, P)) return (int)hipErrorInvalidValue;
  hcm::ProfSpan span(HCM_PROF_CONV1X1_FWD, (hipStream_t)stream, 4.0 * N * (double)(C + K) * P);
  return launch<false>(x, w, z, N, K, C, C, P, (hipStream_t)stream, false);
}

int hcm_conv1x1_backward_data(const float* dz, const float* w, float* dx, int N, int C, int K, int P, hcm_stream_t stream) {
  if (!dz || !w || !dx || N <= 0 || !hcm_conv1x1_supported(C, K, P)) return (int)hipErrorInvalidValue;
  hcm::ProfSpan span(HCM_PROF_CONV1X1_DX, (hipStream_t)stream, 4.0 * N * (double)(C + K) * P);
  return launch<true>(dz, w, dx, N, C, K, C, P, (hipStream_t)stream, false);
}

int hcm_conv1x1_forward_exact(const float* x, const float* w, float* z, int N, int C, int K, int P, hcm_stream_t stream) {
  if (!x || !w || !z || N <= 0 || !hcm_conv1x1_supported(C, K, P)) return (int)hipErrorInvalidValue;
  hcm::ProfSpan span(HCM_PROF_CONV1X1_FWD, (hipStream_t)stream, 4.0 * N * (double)(C + K) * P);
  return launch<false>(x, w, z, N, K, C, C, P, (hipStream_t)stream, true);
}

int hcm_conv1x1_backward_data_exact(const float* dz, const float* w, float* dx, int N, int C, int K, int P, hcm_stream_t stream) {
  if (!dz || !w || !dx || N <= 0 || !hcm_conv1x1_supported(C, K, P)) return (int)hipErrorInvalidValue;
  hcm::ProfSpan span(HCM_PROF_CONV1X1_DX, (hipStream_t)stream, 4.0 * N * (double)(C + K) * P);
  return launch<true>(dz, w, dx, N, C, K, C, P, (hipStream_t)stream, true);
}

}  // extern "C"
