// Feature-map contrastive losses for gfx950 (MI355X): dense intra-sample soft InfoNCE,
// joint<->graph-node InfoNCE and the cross-subject SCL, forward + backward
// (pycontrast/learning/contrast_trainer.py:642-892).
//
// Structure
//   gather_norm_kernel   one wave per sampled pixel: strided read of the 128 channels straight out
//                        of the NCHW / channels-last map, L2-normalise, store the unit row.
//   strip_kernel<Policy> S x S (dense, per image) or N x N (SCL) similarity on the fp32 MFMA
//                        (v_mfma_f32_16x16x4_f32): a 256-thread workgroup owns 64 query rows (one
//                        16-row strip per wave) and walks the key set in 16-row tiles staged once
//                        per workgroup in LDS.  STATS pass: online row softmax + soft-target sums.
//                        GRAD pass: re-forms the tile TRANSPOSED (operands of the first MFMA chain swapped),
//                        so the logit gradient built in registers already has the A-operand layout of the
//                        second MFMA chain (G x K) -- no LDS transpose; the S x S matrices never exist in memory.
//   joint_nce_kernel     one workgroup per image (J<=32 joints, launch/latency bound): VALU.
//   scatter_rows_kernel  deterministic owner-computes scatter-add of the sampled-pixel gradients
//                        into the map gradient (duplicates summed in index order, no atomics).
#include "hcm_common.h"
#include "../../include/hcmoco_hip.h"

namespace {

using namespace hcm;

constexpr int kC = 128;  // linear_merge channels (build_backbone.py:243-245)
constexpr int kWG = 256;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v8f __attribute__((ext_vector_type(8)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 v4bf __attribute__((ext_vector_type(4)));
typedef short v4s __attribute__((ext_vector_type(4)));

struct MapView {
  int64_t sN, sC, sH, sW;
  int w;
  __device__ __forceinline__ int64_t at(int b, int ch, int pix) const {
    return b * sN + ch * sC + (int64_t)(pix / w) * sH + (int64_t)(pix % w) * sW;
  }
};

// ------------------------------------------------------------------------------------------
// gather + L2 normalise (F.normalize, eps 1e-12).  rows r = b*R + s; grid.y selects the map.
// F [2][nrows][128]; invn [2][nrows] = 1/max(|x|,eps), stored NEGATIVE when the clamp was active
// (then d xhat/dx = I/eps, without the projection term).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWG) void gather_norm_kernel(const float* __restrict__ map1,
                                                          const float* __restrict__ map2,
                                                          MapView mv, const int64_t* __restrict__ pix,
                                                          int R, int nrows,
                                                          const int32_t* __restrict__ keep,
                                                          float* __restrict__ F,
                                                          float* __restrict__ invn) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int mod = blockIdx.y;
  if (r >= nrows) return;
  const int b = r / R;
  float* out = F + ((int64_t)mod * nrows + r) * kC;
  if (keep != nullptr && keep[b] == 0) {
    out[lane] = 0.f;
    out[lane + 64] = 0.f;
    if (lane == 0) invn[(int64_t)mod * nrows + r] = 0.f;
    return;
  }
  const float* map = mod == 0 ? map1 : map2;
  const int p = (int)pix[r];
  const float x0 = map[mv.at(b, lane, p)];
  const float x1 = map[mv.at(b, lane + 64, p)];
  const float nrm = sqrtf(wave_sum(fmaf(x0, x0, x1 * x1)));
  const float den = fmaxf(nrm, 1e-12f);
  out[lane] = x0 / den;
  out[lane + 64] = x1 / den;
  if (lane == 0) invn[(int64_t)mod * nrows + r] = (nrm < 1e-12f) ? -1.f / den : 1.f / den;
}

// ------------------------------------------------------------------------------------------
// Target / weight policies of the strip kernel.  meta is one int per row.
// ------------------------------------------------------------------------------------------
struct DensePolicy {
  // split-bf16 contraction (see strip_kernel): hi.hi + hi.mid + mid.hi.  Q and K are different modalities' rows, the dropped
  // terms (~2^-16 |q||k|, random sign) leave 3e-9..8e-8 on the loss and 4e-6..6e-6 on the gradients against float64
  // (tools/studies/split_bf16_error.py, profiles/r05_split_bf16_error_study.txt; gate 1e-5 / 1e-4)
  static constexpr int kTerms = 3;
  // soft target exp(-|q_r - q_c|_2) from integer pixel coordinates (contrast_trainer.py:702-706)
  // meta arrives as the flat pixel index; pack() turns it into (row << 16 | column) ONCE per query row and once
  // per staged key (the divisions by the run-time map width used to sit in weight(): 16 integer divisions per
  // lane and key tile, more VALU time than the tile's 32 MFMAs)
  int w;
  __device__ __forceinline__ int pack(int m) const { return ((m / w) << 16) | (m % w); }
  __device__ __forceinline__ float weight(int mr, int mc, int r, int c) const {
    const float dy = (float)((mr >> 16) - (mc >> 16)), dx = (float)((mr & 0xffff) - (mc & 0xffff));
    // v_sqrt_f32 (1 ulp) instead of the correctly rounded sqrtf (a dozen instructions): the argument is a small exact
    // integer and the result only feeds exp(-d)
    return __expf(-__builtin_amdgcn_sqrtf(dy * dy + dx * dx));
  }
  // row statistics -> (loss term, alpha, beta):  G = alpha*softmax - beta*weight
  __device__ __forceinline__ void finish(float lse, float tdot, float z, float& loss, float& alpha,
                                         float& beta) const {
    loss = lse - tdot / z;
    alpha = 1.f;
    beta = 1.f / z;
  }
};
struct SclPolicy {
  // + mid.mid: Q = K here, and on the diagonal x.x the dropped mid.mid term is a SUM OF SQUARES (systematic, 3.5e-6 on the
  // loss with three terms, 9e-8 with four)
  static constexpr int kTerms = 4;
  // positives: same joint id, different row, both rows' modality present (:873-885)
  // meta = joint id | (valid << 16)
  __device__ __forceinline__ int pack(int m) const { return m; }
  __device__ __forceinline__ float weight(int mr, int mc, int r, int c) const {
    return ((mr & 0xffff) == (mc & 0xffff) && r != c && (mr >> 16) && (mc >> 16)) ? 1.f : 0.f;
  }
  __device__ __forceinline__ void finish(float lse, float tdot, float z, float& loss, float& alpha,
                                         float& beta) const {
    const float c = fmaxf(z, 1.f);
    loss = (z * lse - tdot) / c;
    alpha = z / c;
    beta = 1.f / c;
  }
};

struct StripArgs {
  const float* F;       // [2][nbatch*S][128] unit rows (dense: modality 0 = rgb, 1 = depth)
  const float* invn;    // [2][nbatch*S]
  const int* meta;      // [nbatch*S] (dense) or [2*nbatch*S] (scl)
  const int32_t* keep;  // [nbatch] or null
  int S;                // rows per problem
  int nbatch;
  int symmetric;        // 1: SCL (Q = K = all 2*nbatch*S rows, one problem); 0: dense
  float inv_tau;
  int bounded;          // 1: rows are unit vectors and exp(-2/tau) is representable: softmax around the bound 1/tau
  const float* gscale;  // device scalar: 1/(B'S) or 1/N (0 disables the loss)
  float* stat;          // [norient][rows][4] = lse, alpha, beta, unused
  float* rowloss;       // [norient][rows]
  float* rowcorrect;    // [norient][rows]
  float* dX;            // [2][nbatch*S][128]  gradient wrt the gathered (un-normalised) rows
  // Key-axis split (symmetric problems only; blockIdx.y = key chunk).  The SCL problem is ONE 1088 x 1088
  // similarity: 17 row blocks cannot fill 256 CUs, so the key tiles are dealt to nkc chunks and every
  // (row block, chunk) workgroup leaves a partial -- online-softmax state per row (stats pass) or an
  // un-normalised d/dq-hat row (grad pass) -- that strip_merge_* combine in chunk order (deterministic).
  int nkc;              // 1: no split, results are final
  float* pstat;         // [nkc][N][8] = running max, sum, target dot, target mass, best logit, its column
  float* pdq;           // [nkc][N][128]
};

// LDS image of a key tile (16 rows x 32 float4 slots): row stride 128 floats, NO padding, slot s of row r stored at
// slot s ^ r.  Found by exhaustive search over strides and row-dependent rotations / XORs against the part's real lane
// groups (MI355X_MICROARCH.md, LDS): with this image all three accesses are conflict-free --
//   GEMM 1   ds_read_b128, lane (np, g) reads slot 4j + g of row np (four non-contiguous 16-lane groups, bank =
//            float4 slot mod 16).  A padded linear image cannot do that: the 8 + 8 lanes of a group need slots
//            {np st} and {np st + 1} disjoint, impossible for a shift of a proper subset of Z_16 -- r02's stride 148
//            left 96 conflict cycles per tile here (SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS = 1.5 in the stats pass);
//   GEMM 2   ds_read_b32, lane (np, g) reads float 16nt + np of row 4g + ks (32-lane halves, bank = float mod 32);
//   commit   ds_write_b128, thread e writes slot e & 31 of row e >> 5 (8 contiguous lanes, bank = slot mod 8).
constexpr int kKS = 128;
__device__ __forceinline__ int ksw(int row, int slot) { return (row * 32 + (slot ^ row)) * 4; }   // float index

// FP32-ACCURATE CONTRACTIONS ON THE BF16 MATRIX CORES (r05; the `BF16 == false` instantiations).  v_mfma_f32_16x16x4_f32 runs
// at the fp32 VECTOR rate: the 32 of them a key tile's similarity needs are 1024 cycles per SIMD, 47 % of the critical
// workgroups' time in r04 (profiles/r04_strip_phase_stamps.txt), and the gradient contraction takes 32 more.  Every fp32
// operand is split once into bf16 pieces, x = hi + mid (+ 2^-16 |x|): hi = bf16(x), mid = bf16(x - hi); a product of two bf16
// values is exact in the fp32 accumulator, so  x y = hi hi + hi mid + mid hi (+ mid mid)  carries ~2^-16 relative error per
// PRODUCT but 2^-24-class error per SUM of 128 random-sign products (Policy::kTerms, error study above).  Per tile that is
// 12 (16) v_mfma_f32_16x16x32_bf16 for the similarity (~17 cycles each) and 24 (32) v_mfma_f32_16x16x16_bf16 for G K.
// The key tile is split ONCE per workgroup on its way into LDS (commit): two row-major bf16 planes [16 rows][128] for the
// similarity (16-byte slot s of row r at slot s ^ r: conflict-free ds_read_b128); the second contraction's B operand -- key
// index = reduction index -- is read out of the SAME planes with the LDS transpose read (ds_read_b64_tr_b16).  The query
// fragments are split once per wave, the logit gradient G (four values per lane and tile) in registers.
//
// BF16 (BASELINE config 5, "bf16 feature-map GEMMs"): both contractions run on the bf16 matrix cores with
// fp32 accumulation -- the similarity P = Q K^T as 4 x v_mfma_f32_16x16x32_bf16 per tile (32 fp32 MFMAs
// otherwise) and the gradient contraction G K as 8 x v_mfma_f32_16x16x16_bf16 (32 otherwise).  Operands
// are rounded to bf16 (v_cvt_pk_bf16_f32, round-to-nearest-even) in registers on their way from the
// fp32 unit rows / the fp32 LDS tile into the MFMA; everything else (softmax statistics, targets,
// normalisation backward) stays fp32.  Parity is stated against the fp32 oracle at 1e-2.
// Diagnostic build only (-DHCM_STRIP_TIMING, tools/probes/strip_timing.sh): s_memtime stamps of wave 0 of every workgroup,
// summed per phase of the key-tile loop.
#ifdef HCM_STRIP_TIMING
__device__ unsigned long long* g_strip_dbg = nullptr;
#define HCM_TS(k)                                                                    \
  do {                                                                               \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();                    \
    tacc_[k] += now_ - tlast_;                                                        \
    tlast_ = now_;                                                                   \
  } while (0)
#else
#define HCM_TS(k) do {} while (0)
#endif
// EXACT (r06, `--fmap_dtype fp32_exact`; ADVICE r05: a true-fp32 contraction must stay selectable): every operand is split into
// THREE bf16 pieces, x = hi + mid + lo EXACTLY (3 x 8 significand bits), and all nine piece products -- each exact in the fp32
// accumulator -- are issued, smallest first: the contraction is an fp32 dot product whose only rounding is the accumulator's, the
// class of v_mfma_f32_16x16x4_f32.  3 x the matrix work of the default (9 terms against 3), not a speed mode.
template <class Policy, bool GRAD, bool BF16, bool BND, bool EXACT = false>
__global__ __launch_bounds__(kWG) void strip_kernel(StripArgs a, Policy pol) {
  static_assert(!(BF16 && EXACT), "EXACT is a mode of the fp32 instantiation");
#ifdef HCM_STRIP_TIMING
  unsigned long long tacc_[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long tlast_ = __builtin_amdgcn_s_memtime();
#endif
  // Software pipeline over the key tiles, three LDS buffers, ONE barrier per tile: while tile t is in its element-wise
  // phase (softmax / targets / gradient entries) the similarity of tile t+1 has already been issued and tile t+2 is
  // travelling global -> registers, so no phase waits for a load or for a matrix result.
  // What this does NOT buy (r02, PMC of the dense stats pass: per wave 25.6 k cycles of fp32-MFMA busy time + 26.6 k
  // of VALU busy time in 113 k cycles, two waves per SIMD): on this part the fp32 MFMAs and the VALU instructions of a
  // SIMD's waves do not overlap -- their times ADD (removing either phase removes its full time; interleaving them
  // in the instruction stream, independent accumulators, conflict-free LDS strides each changed nothing).  The pass is
  // bounded by (MFMA + VALU) instruction time: fewer VALU instructions per element is the only lever left in fp32.
  // BF16: the fp32 tile (operands rounded on their way into the MFMA).  Split: hi / mid planes, row-major [2][16][128] bf16 = 8 KB.
  constexpr int kBufFloats = BF16 ? 16 * kKS : (EXACT ? 12288 / 4 : 8192 / 4);      // EXACT: a third (lo) plane
  constexpr int kT = EXACT ? 9 : Policy::kTerms;
  __shared__ __attribute__((aligned(16))) float sKb[3][kBufFloats];
  __shared__ int sMetaCb[3][16];
  __shared__ float sStatCb[3][16][3];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int np = lane & 15, g = lane >> 4;
  const int o = blockIdx.z;  // orientation (dense): 0 -> Q = depth rows, K = rgb rows; 1 -> swapped
  const int b = a.symmetric ? 0 : blockIdx.y;
  const int kc = a.symmetric ? blockIdx.y : 0;   // key chunk (StripArgs::nkc)
  const int rows_total = a.nbatch * a.S;           // rows per modality
  const int S = a.symmetric ? 2 * rows_total : a.S;  // problem size
  if (!a.symmetric && a.keep != nullptr && a.keep[b] == 0) return;

  const int qmod = a.symmetric ? 0 : (o == 0 ? 1 : 0);
  const int kmod = a.symmetric ? 0 : 1 - qmod;
  const int64_t base = a.symmetric ? 0 : (int64_t)b * a.S;
  const float* Q = a.F + ((int64_t)qmod * rows_total + base) * kC;
  const float* K = a.F + ((int64_t)kmod * rows_total + base) * kC;
  const int* metaQ = a.meta + base;
  const int* metaK = a.meta + base;
  const int64_t statQ = ((int64_t)o * (a.symmetric ? S : rows_total) + base);  // own stats
  const int64_t statK = a.symmetric ? 0 : ((int64_t)(1 - o) * rows_total + base);  // other orientation

  const int row0 = blockIdx.x * 64 + wave * 16;  // first row of this wave's strip
  // A operand: lane (m = np, kslot = g) holds Q[row0+np][16j + 4g + e], j<8, e<4
  // (BF16: Q[row0+np][32j + 8g + e], j<4, e<8, rounded to bf16)
  // lane (m = np, kslot = g) holds Q[row0+np][32j + 8g + e], j<4, e<8, as bf16: rounded (BF16) or split into hi + mid
  v8bf qh[4], qm[BF16 ? 1 : 4], ql[EXACT ? 4 : 1];
  {
    const int r = row0 + np;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
      if (r < S) {
        lo = *reinterpret_cast<const float4*>(Q + (int64_t)r * kC + 32 * j + 8 * g);
        hi = *reinterpret_cast<const float4*>(Q + (int64_t)r * kC + 32 * j + 8 * g + 4);
      }
      const v8f v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      qh[j] = __builtin_convertvector(v, v8bf);
      if constexpr (!BF16) qm[j] = __builtin_convertvector(v - __builtin_convertvector(qh[j], v8f), v8bf);
      if constexpr (EXACT)
        ql[j] = __builtin_convertvector((v - __builtin_convertvector(qh[j], v8f)) - __builtin_convertvector(qm[j], v8f), v8bf);
    }
  }
  // rows owned in the C layout: row0 + 4g + reg (stats pass: P = Q K^T) or row0 + np for every reg (grad pass: the
  // TRANSPOSED similarity K Q^T is formed there, see GEMM 1)
  int mrow[4];
  float lse_r[4], al_r[4], be_r[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = GRAD ? row0 + np : row0 + 4 * g + q;
    mrow[q] = (r < S) ? pol.pack(metaQ[r]) : 0;
    if (GRAD) {
      lse_r[q] = (r < S) ? a.stat[(statQ + r) * 4 + 0] : 0.f;
      al_r[q] = (r < S) ? a.stat[(statQ + r) * 4 + 1] : 0.f;
      be_r[q] = (r < S) ? a.stat[(statQ + r) * 4 + 2] : 0.f;
    }
  }
  const float gs = GRAD ? *a.gscale : 0.f;

  float m[4], ssum[4], td[4], z[4], best[4];
  int bestc[4];
  v4f dq[8];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    m[q] = BND ? a.inv_tau : -1.0e30f; ssum[q] = 0.f; td[q] = 0.f; z[q] = 0.f; best[q] = -3.0e38f; bestc[q] = 0;
  }
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) dq[nt] = (v4f){0.f, 0.f, 0.f, 0.f};

  const int ntiles = (S + 15) / 16;
  const int tiles_per = (ntiles + a.nkc - 1) / a.nkc;
  const int tile_lo = kc * tiles_per, tile_hi = min(ntiles, tile_lo + tiles_per);
  // fetch: this thread's share of a key tile (16 rows x 32 float4 = 2 float4 per thread; threads < 16 also the
  // tile's meta / stats of the other orientation)
  float4 kreg[2];
  int mreg = 0;
  float sreg[3] = {0.f, 0.f, 0.f};
  auto fetch = [&](int tile) {
    const int c0 = tile * 16;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int e = threadIdx.x + h * kWG;
      const int rr = e >> 5, cc = (e & 31) * 4;
      kreg[h] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c0 + rr < S) kreg[h] = *reinterpret_cast<const float4*>(K + (int64_t)(c0 + rr) * kC + cc);
    }
    if (threadIdx.x < 16) {
      const int c = c0 + threadIdx.x;
      mreg = (c < S) ? pol.pack(metaK[c]) : 0;
      if (GRAD) {
#pragma unroll
        for (int i = 0; i < 3; ++i) sreg[i] = (c < S) ? a.stat[(statK + c) * 4 + i] : 0.f;
      }
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int e = threadIdx.x + h * kWG;
      if constexpr (BF16) {
        *reinterpret_cast<float4*>(&sKb[buf][ksw(e >> 5, e & 31)]) = kreg[h];
      } else {
        // split the four values once for the whole workgroup: hi = bf16(x), mid = bf16(x - hi)
        const int rr = e >> 5, c4 = e & 31;
        const v4f v = {kreg[h].x, kreg[h].y, kreg[h].z, kreg[h].w};
        const v4bf hi = __builtin_convertvector(v, v4bf);
        const v4bf mid = __builtin_convertvector(v - __builtin_convertvector(hi, v4f), v4bf);
        char* base = reinterpret_cast<char*>(sKb[buf]);
        const int o = rr * 256 + (((c4 >> 1) ^ rr) << 4) + ((c4 & 1) << 3);
        *reinterpret_cast<v4bf*>(base + o) = hi;
        *reinterpret_cast<v4bf*>(base + 4096 + o) = mid;
        if constexpr (EXACT)
          *reinterpret_cast<v4bf*>(base + 8192 + o) =
              __builtin_convertvector((v - __builtin_convertvector(hi, v4f)) - __builtin_convertvector(mid, v4f), v4bf);
      }
    }
    if (threadIdx.x < 16) {
      sMetaCb[buf][threadIdx.x] = mreg;
      if (GRAD) {
#pragma unroll
        for (int i = 0; i < 3; ++i) sStatCb[buf][threadIdx.x][i] = sreg[i];
      }
    }
  };
  // GEMM 1: P[16 x 16] = Qstrip . Ktile^T   (32 x v_mfma_f32_16x16x4_f32) into four independent accumulators
  auto gemm1 = [&](const float* sK, v4f (&ap)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) ap[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    if constexpr (BF16) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 lo = *reinterpret_cast<const float4*>(&sK[ksw(np, 8 * j + 2 * g)]);
        const float4 hi = *reinterpret_cast<const float4*>(&sK[ksw(np, 8 * j + 2 * g + 1)]);
        const v8f kv = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        if (GRAD) ap[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_convertvector(kv, v8bf), qh[j], ap[j], 0, 0, 0);
        else ap[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qh[j], __builtin_convertvector(kv, v8bf), ap[j], 0, 0, 0);
      }
    } else {
      // split operands: the smallest terms go in first.  GRAD: A <-> B, the accumulator holds P^T, i.e. lane (np, g) reg q =
      // P[row0 + np][c0 + 4g + q]
      const char* base = reinterpret_cast<const char*>(sK);
      v8bf kh[4], km[4], kl[EXACT ? 4 : 1];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int o = np * 256 + (((4 * j + g) ^ np) << 4);
        kh[j] = *reinterpret_cast<const v8bf*>(base + o);
        km[j] = *reinterpret_cast<const v8bf*>(base + 4096 + o);
        if constexpr (EXACT) kl[j] = *reinterpret_cast<const v8bf*>(base + 8192 + o);
      }
      // term-major issue order: the four accumulators take turns, so no MFMA waits for the one before it (a dependent
      // v_mfma_f32_16x16x32_bf16 issues ~40 cycles after its producer, an independent one after ~17)
#define HCM_TERM(QA, KB)                                                                                          \
  _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                    \
      ap[j] = GRAD ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(KB[j], QA[j], ap[j], 0, 0, 0)                           \
                   : __builtin_amdgcn_mfma_f32_16x16x32_bf16(QA[j], KB[j], ap[j], 0, 0, 0)
      if constexpr (EXACT) {                        // 2^-32, 2^-24, 2^-24, 2^-16, 2^-16 class terms first
        HCM_TERM(ql, kl);
        HCM_TERM(qm, kl);
        HCM_TERM(ql, km);
        HCM_TERM(qh, kl);
        HCM_TERM(ql, kh);
      }
      if constexpr (kT >= 4) { HCM_TERM(qm, km); }
      HCM_TERM(qh, km);
      HCM_TERM(qm, kh);
      HCM_TERM(qh, kh);
#undef HCM_TERM
    }
  };
  v4f apn[4];                                       // similarity of the NEXT tile (in flight)
  if (tile_lo < tile_hi) {
    fetch(tile_lo);
    commit(0);
  }
  __syncthreads();
  gemm1(sKb[0], apn);
  if (tile_lo + 1 < tile_hi) fetch(tile_lo + 1);
  HCM_TS(0);                                          // prologue: query fragments, first tile staged, first GEMM issued
  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    const int c0 = tile * 16;
    const int ib = (tile - tile_lo) % 3, nb = (ib + 1) % 3;
    const float* sK = sKb[ib];
    const int* sMetaC = sMetaCb[ib];
    const float (*sStatC)[3] = sStatCb[ib];
    const v4f acc = (apn[0] + apn[1]) + (apn[2] + apn[3]);
    if (tile + 1 < tile_hi) commit(nb);             // buffer nb was last read two barriers ago
    HCM_TS(1);                                      // accumulator read (the previous GEMM must have finished) + commit
    __syncthreads();                                // tile t+1 is in LDS
    HCM_TS(2);
    if (tile + 2 < tile_hi) fetch(tile + 2);
    gemm1(sKb[nb], apn);                            // unconditional (a stale buffer after the last tile): keeps the
                                                    // MFMAs and the element-wise code below in one basic block
    HCM_TS(3);                                      // operand reads + MFMA issue
    // C layout: acc[q] = P[row0 + 4g + q][c0 + np]
    const int c = c0 + np;
    const bool cvalid = c < S;
    const int mc = sMetaC[np];

    // Element-wise part, branch-free (selects): with divergent branches and a two-exp online softmax it cost as many
    // VALU cycles per tile as the tile's 32 MFMAs.  Unit rows bound every logit by 1/tau, so when exp(-2/tau) is
    // far from underflow (a.bounded; tau = 0.07: 4e-13) the row maximum is replaced by that bound -- one exp per
    // element, no rescaling; otherwise the running-maximum form.
    if (!GRAD) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = row0 + 4 * g + q;
        const bool ok = cvalid && r < S;
        const float okf = ok ? 1.f : 0.f;            // multiplied in, not selected: a select on an expensive value is
        const float P = acc[q] * a.inv_tau;          // turned back into a divergent branch, and a branch ends the block
        const float wgt = pol.weight(mrow[q], mc, r, c) * okf;
        if (BND) {
          ssum[q] = fmaf(__expf(P - a.inv_tau), okf, ssum[q]);
        } else {
          const float mn = ok ? fmaxf(m[q], P) : m[q];
          ssum[q] = fmaf(__expf(P - mn), okf, ssum[q] * __expf(m[q] - mn));
          m[q] = mn;
        }
        td[q] = fmaf(wgt, P, td[q]);
        z[q] += wgt;
        const bool better = ok && P > best[q];
        best[q] = better ? P : best[q];
        bestc[q] = better ? c : bestc[q];
      }
    } else {
      // Grad pass: GEMM 1 ran with its operands swapped, so this lane owns QUERY ROW row0 + np and its four
      // accumulator registers are the keys c0 + 4g + q -- exactly the A-operand layout of GEMM 2 when the key index of
      // its reduction is ordered (4 x k-slot + step).  The logit gradient goes from the element-wise code straight into
      // the second MFMA chain; r02 wrote it to LDS in C layout and read it back transposed (4-way bank conflicts on the
      // writes: SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS = 2.4 in this pass).
      const int r = row0 + np;
      float ga[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int kq = 4 * g + q, cq = c0 + kq;
        const float okf = (cq < S && r < S) ? gs * a.inv_tau : 0.f;
        const float P = fminf(acc[q] * a.inv_tau, 2.f * a.inv_tau);      // (clamp: padded rows must not make inf * 0)
        const float wgt = pol.weight(mrow[q], sMetaC[kq], r, cq);
        const float lse_c = sStatC[kq][0], al_c = sStatC[kq][1], be_c = sStatC[kq][2];
        const float G = al_r[q] * __expf(P - lse_r[q]) - be_r[q] * wgt + al_c * __expf(P - lse_c) - be_c * wgt;
        ga[q] = G * okf;
      }
      // GEMM 2: dQ[16 x 128] += G[16 x 16] . Ktile[16 x 128]   (4 k-steps x 8 channel tiles); key of (step, slot g) = 4g + step
      if constexpr (BF16) {
        // A = G[np][4g + i], B = Ktile[4g + i][16nt + np], i < 4: one 16-key contraction per channel tile
        const v4f gv = {ga[0], ga[1], ga[2], ga[3]};
        const v4s gb = __builtin_bit_cast(v4s, __builtin_convertvector(gv, v4bf));
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
          const int sl = 4 * nt + (np >> 2), el = np & 3;
          const v4f kv = {sK[ksw(4 * g, sl) + el], sK[ksw(4 * g + 1, sl) + el], sK[ksw(4 * g + 2, sl) + el],
                          sK[ksw(4 * g + 3, sl) + el]};
          dq[nt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(gb, __builtin_bit_cast(v4s, __builtin_convertvector(kv, v4bf)),
                                                             dq[nt], 0, 0, 0);
        }
      } else {
        // split: G = ghi + gmid in registers.  B[key 4g + i][channel 16nt + np], i < 4, comes out of the ROW-MAJOR planes through
        // the LDS transpose read (ds_read_b64_tr_b16; tools/probes/tr16_probe.hip pins its gather: lane l receives, for i < 4,
        // element (l & 3) of the 8-byte chunk that lane 4 i + ((l & 15) >> 2) of its 16-lane group points at): lane (np, g)
        // points at key row 4g + (np >> 2), channels 16nt + 4 (np & 3) .. + 3.  (A first version kept transposed copies of
        // the planes: their sixteen ds_write_b16 per thread and tile, 4-8 way conflicted, cost 700 cycles per tile.)
        const v4f gv = {ga[0], ga[1], ga[2], ga[3]};
        const v4bf gh4 = __builtin_convertvector(gv, v4bf);
        const v4bf gm4 = __builtin_convertvector(gv - __builtin_convertvector(gh4, v4f), v4bf);
        const v4s gh = __builtin_bit_cast(v4s, gh4), gm = __builtin_bit_cast(v4s, gm4);
        v4s gl = gh;
        if constexpr (EXACT)
          gl = __builtin_bit_cast(v4s, __builtin_convertvector((gv - __builtin_convertvector(gh4, v4f)) -
                                                               __builtin_convertvector(gm4, v4f), v4bf));
        typedef v4s __attribute__((address_space(3))) * lds_v4s;
        const char* tb = reinterpret_cast<const char*>(sK);
        const int trow = 4 * g + (np >> 2), tq = np & 3;
        v4s th[8], tm[8], tl[EXACT ? 8 : 1];
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
          const int c4 = 4 * nt + tq;
          const int o = trow * 256 + (((c4 >> 1) ^ trow) << 4) + ((c4 & 1) << 3);
          th[nt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(tb + o));
          tm[nt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(tb + 4096 + o));
          if constexpr (EXACT) tl[nt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(tb + 8192 + o));
        }
#define HCM_TERM2(GA, TB) \
  _Pragma("unroll") for (int nt = 0; nt < 8; ++nt) dq[nt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(GA, TB[nt], dq[nt], 0, 0, 0)
        if constexpr (EXACT) {
          HCM_TERM2(gl, tl);
          HCM_TERM2(gm, tl);
          HCM_TERM2(gl, tm);
          HCM_TERM2(gh, tl);
          HCM_TERM2(gl, th);
        }
        if constexpr (kT >= 4) { HCM_TERM2(gm, tm); }
        HCM_TERM2(gh, tm);
        HCM_TERM2(gm, th);
        HCM_TERM2(gh, th);
#undef HCM_TERM2
      }
    }
    HCM_TS(4);                                      // element-wise code (+ GEMM 2 issue in the grad pass)
  }
#ifdef HCM_STRIP_TIMING
  if (g_strip_dbg != nullptr && threadIdx.x == 0) {
    unsigned long long* o_ = g_strip_dbg + ((((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 2 + (GRAD ? 1 : 0)) * 8;
    for (int k = 0; k < 5; ++k) o_[k] = tacc_[k];
    o_[5] = (unsigned long long)(tile_hi - tile_lo);
  }
#endif

  if (!GRAD) {
    // merge the 16 lanes (columns) that share each row: they sit in one DPP row
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float M = row16_max(m[q]);
      const float sAll = row16_sum(ssum[q] * __expf(m[q] - M));
      const float tdAll = row16_sum(td[q]);
      const float zAll = row16_sum(z[q]);
      const float bAll = row16_max(best[q]);
      // lowest column among the maxima (torch.argmax returns the first)
      const float cand = (best[q] == bAll) ? (float)bestc[q] : 3.0e38f;
      const float cmin = -row16_max(-cand);
      const int r = row0 + 4 * g + q;
      if (a.nkc > 1) {
        if (np == 0 && r < S) {
          float* ps = a.pstat + ((int64_t)kc * S + r) * 8;
          ps[0] = M; ps[1] = sAll; ps[2] = tdAll; ps[3] = zAll; ps[4] = bAll; ps[5] = cmin;
        }
        continue;
      }
      if (np == 0 && r < S) {
        const float lse = M + __logf(sAll);
        float loss, alpha, beta;
        pol.finish(lse, tdAll, zAll, loss, alpha, beta);
        a.stat[(statQ + r) * 4 + 0] = lse;
        a.stat[(statQ + r) * 4 + 1] = alpha;
        a.stat[(statQ + r) * 4 + 2] = beta;
        a.rowloss[statQ + r] = loss;
        a.rowcorrect[statQ + r] = ((int)cmin == r) ? 1.f : 0.f;
      }
    }
  } else {
    // dq[nt][q] = d loss / d qhat[row0+4g+q][16nt+np]; push it through F.normalize
    const int64_t qrow_base = (int64_t)qmod * rows_total + base;
    if (a.nkc > 1) {       // partial over this chunk's keys: strip_merge_grad_kernel sums and normalises
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = row0 + 4 * g + q;
        if (r < S) {
          float* dst = a.pdq + ((int64_t)kc * S + r) * kC;
#pragma unroll
          for (int nt = 0; nt < 8; ++nt) dst[16 * nt + np] = dq[nt][q];
        }
      }
      return;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = row0 + 4 * g + q;
      const bool rv = r < S;
      float fh[8];
      float dot = 0.f;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        fh[nt] = rv ? Q[(int64_t)r * kC + 16 * nt + np] : 0.f;
        dot = fmaf(dq[nt][q], fh[nt], dot);
      }
      dot = row16_sum(dot);
      if (rv) {
        const float inv = a.invn[qrow_base + r];
        float* dst = a.dX + (qrow_base + r) * kC;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
          dst[16 * nt + np] = (inv < 0.f) ? dq[nt][q] * (-inv) : (dq[nt][q] - dot * fh[nt]) * inv;
      }
    }
  }
}

// Chunk merges of a key-split symmetric problem (N rows, chunks in ascending key order).
template <class Policy>
__global__ __launch_bounds__(kWG) void strip_merge_stats_kernel(StripArgs a, Policy pol, int N) {
  const int r = blockIdx.x * kWG + threadIdx.x;
  if (r >= N) return;
  float M = -1.0e30f, sum = 0.f, td = 0.f, z = 0.f, best = -3.0e38f, col = 3.0e38f;
  for (int kc = 0; kc < a.nkc; ++kc) {
    const float* ps = a.pstat + ((int64_t)kc * N + r) * 8;
    const float Mn = fmaxf(M, ps[0]);
    sum = sum * __expf(M - Mn) + ps[1] * __expf(ps[0] - Mn);
    M = Mn;
    td += ps[2];
    z += ps[3];
    if (ps[4] > best) { best = ps[4]; col = ps[5]; }   // strict: the first chunk keeps a tie (lowest column)
  }
  const float lse = M + __logf(sum);
  float loss, alpha, beta;
  pol.finish(lse, td, z, loss, alpha, beta);
  a.stat[(int64_t)r * 4 + 0] = lse;
  a.stat[(int64_t)r * 4 + 1] = alpha;
  a.stat[(int64_t)r * 4 + 2] = beta;
  a.rowloss[r] = loss;
  a.rowcorrect[r] = ((int)col == r) ? 1.f : 0.f;
}

// one wave per row: sum the chunk partials in order, then F.normalize's backward (as in strip_kernel)
__global__ __launch_bounds__(kWG) void strip_merge_grad_kernel(StripArgs a, int N) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= N) return;
  float d0 = 0.f, d1 = 0.f;
  for (int kc = 0; kc < a.nkc; ++kc) {
    const float* src = a.pdq + ((int64_t)kc * N + r) * kC;
    d0 += src[lane];
    d1 += src[lane + 64];
  }
  const float f0 = a.F[(int64_t)r * kC + lane], f1 = a.F[(int64_t)r * kC + lane + 64];
  const float dot = wave_sum(fmaf(d0, f0, d1 * f1));
  const float inv = a.invn[r];
  float* dst = a.dX + (int64_t)r * kC;
  dst[lane] = (inv < 0.f) ? d0 * (-inv) : (d0 - dot * f0) * inv;
  dst[lane + 64] = (inv < 0.f) ? d1 * (-inv) : (d1 - dot * f1) * inv;
}

// ------------------------------------------------------------------------------------------
// small helper kernels
// ------------------------------------------------------------------------------------------
// dense: gscale = 1/(B' * S) with B' = #kept images (0 when none); meta = pixel index
__global__ void dense_prep_kernel(const int32_t* __restrict__ keep, int B, int S,
                                  const int64_t* __restrict__ sample_ind, int* __restrict__ meta,
                                  float* __restrict__ gscale) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int cnt = 0;
    for (int i = 0; i < B; ++i) cnt += keep[i] != 0;
    *gscale = cnt > 0 ? 1.f / ((float)cnt * (float)S) : 0.f;
  }
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < B * S; e += gridDim.x * blockDim.x)
    meta[e] = (int)sample_ind[e];
}

// scl: rows u = (mod, b, j); meta = j | valid<<16 ; gscale = 1/N, or 0 when use_depth.sum()==0
__global__ void scl_prep_kernel(const int32_t* __restrict__ use_depth,
                                const int32_t* __restrict__ use_rgb, int B, int J,
                                int* __restrict__ meta, float* __restrict__ gscale) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int any = 0;
    for (int i = 0; i < B; ++i) any += use_depth[i] != 0;
    *gscale = any > 0 ? 1.f / (float)(2 * B * J) : 0.f;
  }
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < 2 * B * J; e += gridDim.x * blockDim.x) {
    const int mod = e / (B * J), bb = (e % (B * J)) / J, j = e % J;
    const int valid = mod == 0 ? (use_rgb == nullptr ? 1 : (use_rgb[bb] != 0)) : (use_depth[bb] != 0);
    meta[e] = j | (valid << 16);
  }
}

// fixed-order block reduction of n floats (deterministic): each thread a strided partial, then a tree
__device__ float block_sum(const float* __restrict__ v, int n, float* sh) {
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) acc += v[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  const float r = sh[0];
  __syncthreads();
  return r;
}

// out4 = {loss_r2d, loss_d2r, acc_r2d, acc_d2r}; orientation 1 rows are the r2d terms.
// Rows of dropped images were never written by the stats pass -> masked here through keep.
__global__ __launch_bounds__(1024) void dense_finish_kernel(const float* __restrict__ rowloss,
                                                            const float* __restrict__ rowcorrect,
                                                            const int32_t* __restrict__ keep, int B,
                                                            int S, const float* __restrict__ gscale,
                                                            float* __restrict__ out4) {
  // fixed-shape reduction (deterministic): strided per-thread partials, wave butterfly, 16-wave tree
  __shared__ float sh[4][16];
  const int n = B * S, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  // ONE workgroup walks B S rows: four strides of it in flight per trip, unconditional (clamped) loads and a select -- the guarded
  // form (keep, then the four values inside the branch) was two dependent round trips per stride, 10 us for 12800 rows (r06).  The
  // rows of a dropped image hold whatever the workspace held: they are loaded and never added.  Same adds in the same order.
  for (int i0 = tid; i0 < n; i0 += 4 * 1024) {
    float x[4][4];
    int kp[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = min(i0 + 1024 * u, n - 1);
      kp[u] = keep[i / S];
      x[u][0] = rowloss[n + i];      // loss_r2d: orientation 1
      x[u][1] = rowloss[i];          // loss_d2r: orientation 0
      x[u][2] = rowcorrect[n + i];
      x[u][3] = rowcorrect[i];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool ok = i0 + 1024 * u < n && kp[u] != 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = ok ? v[k] + x[u][k] : v[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[k] = wave_sum(v[k]);
    if (lane == 0) sh[k][wave] = v[k];
  }
  __syncthreads();
  if (tid < 4) {
    float s = 0.f;
    for (int w = 0; w < 16; ++w) s += sh[tid][w];
    out4[tid] = s * (*gscale);   // 1/(B'S); 0 when nothing is kept (reference early return -> zeros)
  }
}

__global__ __launch_bounds__(kWG) void scl_finish_kernel(const float* __restrict__ rowloss, int N,
                                                         const float* __restrict__ gscale,
                                                         float* __restrict__ out1) {
  __shared__ float sh[kWG];
  const float s = block_sum(rowloss, N, sh);
  if (threadIdx.x == 0) out1[0] = s * (*gscale);
}

// ------------------------------------------------------------------------------------------
// Owner-computes scatter-add of row gradients into the map gradient.
// dX [2][B*R][128]; pix [B*R]; one wave per (mod, row).  A row owns its pixel when no EARLIER row
// of the same image has the same pixel; the owner adds its own and all later duplicates' rows in
// index order -> deterministic, no atomics (torch.gather backward sums duplicates too).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWG) void scatter_rows_kernel(const float* __restrict__ dX,
                                                           const int64_t* __restrict__ pix, int R,
                                                           int nrows, const int32_t* __restrict__ keep,
                                                           MapView mv, float* __restrict__ gmap1,
                                                           float* __restrict__ gmap2) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int mod = blockIdx.y;
  if (r >= nrows) return;
  const int b = r / R, s = r - b * R;
  if (keep != nullptr && keep[b] == 0) return;
  const int64_t p = pix[r];
  const int64_t* pb = pix + (int64_t)b * R;
  bool earlier = false;
  for (int i = lane; i < s; i += 64) earlier |= (pb[i] == p);
  if (__any(earlier)) return;
  const float* src = dX + ((int64_t)mod * nrows + (int64_t)b * R) * kC;
  float a0 = src[(int64_t)s * kC + lane], a1 = src[(int64_t)s * kC + lane + 64];
  for (int i0 = s + 1; i0 < R; i0 += 64) {
    const int i = i0 + lane;
    const unsigned long long mask = __ballot(i < R && pb[i] == p);
    unsigned long long mm = mask;
    while (mm) {
      const int bit = __ffsll((long long)mm) - 1;
      mm &= mm - 1;
      a0 += src[(int64_t)(i0 + bit) * kC + lane];
      a1 += src[(int64_t)(i0 + bit) * kC + lane + 64];
    }
  }
  float* gm = mod == 0 ? gmap1 : gmap2;
  gm[mv.at(b, lane, (int)p)] += a0;
  gm[mv.at(b, lane + 64, (int)p)] += a1;
}

// ------------------------------------------------------------------------------------------
// Row 6: joint <-> graph-node InfoNCE (learning/contrast_trainer.py:744-828).
// r05: one workgroup per (image, MODALITY), 512 threads, and a finish kernel -- the r01 form ran one 256-thread workgroup per
// image: 32 of 256 CUs, 39 us for 0.8 MB and 4.7 MFLOP, most of it two serial stretches (a J-long column softmax on 2 J
// lanes, a 3 J x 128 gradient loop of 2 J fmas on four waves).  Now the two modalities of an image are independent until
// the graph rows' gradient (sum of the two partials, formed and pushed through F.normalize by the finish kernel), the
// softmax of column j is evaluated redundantly by the J threads (i, j) -- J LDS reads each, no serial lane --, and the
// gradient loops are J fmas on eight waves.
// ------------------------------------------------------------------------------------------
constexpr int kJMax = 32;
constexpr int kLS = kC + 4;  // LDS row stride: 16-byte aligned rows (ds_read_b128), consecutive rows 4 banks apart
constexpr int kJT = 512;     // threads of a joint workgroup

__global__ __launch_bounds__(kJT) void joint_nce_kernel(
    const float* __restrict__ map1, const float* __restrict__ map2, MapView mv,
    const float* __restrict__ feat3, const int64_t* __restrict__ pix,
    const int32_t* __restrict__ vis, const int32_t* __restrict__ use_depth, int B, int J,
    float inv_tau, float* __restrict__ part /*[B][6]*/, float* __restrict__ dX /*[2][B*J][128]*/,
    float* __restrict__ gpart /*[2][B*J][128]: d loss / d ghat, this modality's share*/) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* sV = lds;                     // [2][J][kLS] unit rows: graph nodes, this modality's joints
  float* sD = sV + 2 * J * kLS;        // [J][kLS] gradient wrt the modality's unit rows
  float* sA = sD + J * kLS;            // [J][J+1] logits
  float* sG = sA + J * (J + 1);        // [J][J+1] logit gradients
  float* sInv = sG + J * (J + 1);      // [J] of the modality's rows
  __shared__ int sCnt;
  __shared__ float sRed[3][kJMax];
  const int b = blockIdx.x, mm = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* map = mm == 0 ? map1 : map2;

  if (tid == 0) sCnt = 0;
  // 1. load + normalise 2 J rows, one wave per row.  A wave owns up to kRows rows; ALL their loads (pixel index, then the
  //    two halves of the row) are issued before the first is used: the rows are independent, and a dependent pair of
  //    global loads per round was 5 x ~2.5 us of this kernel's 20.
  constexpr int kRows = 2 * kJMax / (kJT / 64);
  int prow[kRows];
  float x0[kRows], x1[kRows];
#pragma unroll
  for (int k = 0; k < kRows; ++k) {
    const int rr = wave + k * (kJT / 64);
    prow[k] = (rr >= J && rr < 2 * J) ? (int)pix[(int64_t)b * J + (rr - J)] : 0;
  }
#pragma unroll
  for (int k = 0; k < kRows; ++k) {
    const int rr = wave + k * (kJT / 64);
    x0[k] = 0.f; x1[k] = 0.f;
    if (rr < J) {
      x0[k] = feat3[((int64_t)b * J + rr) * kC + lane];
      x1[k] = feat3[((int64_t)b * J + rr) * kC + lane + 64];
    } else if (rr < 2 * J) {
      x0[k] = map[mv.at(b, lane, prow[k])];
      x1[k] = map[mv.at(b, lane + 64, prow[k])];
    }
  }
  __syncthreads();                     // sCnt = 0 is visible
  {  // global count of valid targets (rgb: visible; depth: visible and use_depth), underneath the row loads
    int c0 = 0;
    for (int e = tid; e < B * J; e += kJT)
      c0 += (vis[e] != 0) && (mm == 0 || use_depth == nullptr || use_depth[e / J] != 0);
    if (c0) atomicAdd(&sCnt, c0);
  }
#pragma unroll
  for (int k = 0; k < kRows; ++k) {
    const int rr = wave + k * (kJT / 64);
    if (rr < 2 * J) {                  // wave-uniform
      const float nrm = sqrtf(wave_sum(fmaf(x0[k], x0[k], x1[k] * x1[k])));
      const float den = fmaxf(nrm, 1e-12f);
      sV[rr * kLS + lane] = x0[k] / den;
      sV[rr * kLS + lane + 64] = x1[k] / den;
      if (rr >= J && lane == 0) sInv[rr - J] = (nrm < 1e-12f) ? -1.f / den : 1.f / den;
    }
  }
  __syncthreads();
  const int cnt = sCnt;
  // 2. logits A[i][j] = ghat_i . fhat_j / tau
  for (int e = tid; e < J * J; e += kJT) {
    const int i = e / J, j = e - i * J;
    const float4* gi = reinterpret_cast<const float4*>(sV + i * kLS);
    const float4* fj = reinterpret_cast<const float4*>(sV + (J + j) * kLS);
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll 8
    for (int c = 0; c < kC / 4; ++c) {
      const float4 a = gi[c], f4 = fj[c];
      d0 = fmaf(a.x, f4.x, d0); d1 = fmaf(a.y, f4.y, d1); d2 = fmaf(a.z, f4.z, d2); d3 = fmaf(a.w, f4.w, d3);
    }
    sA[i * (J + 1) + j] = ((d0 + d1) + (d2 + d3)) * inv_tau;
  }
  __syncthreads();
  // 3. column softmax over the nodes i, CE target j, arg-max (first maximum); every thread (i, j) walks its column
  for (int e = tid; e < J * J; e += kJT) {
    const int i = e / J, j = e - i * J;
    const float* col = sA + j;
    float mx = -3.0e38f;
    int arg = 0;
    for (int k = 0; k < J; ++k) {
      const float v = col[k * (J + 1)];
      if (v > mx) { mx = v; arg = k; }
    }
    float se = 0.f;
    for (int k = 0; k < J; ++k) se += __expf(col[k * (J + 1)] - mx);
    const float lse = mx + __logf(se);
    const bool valid = vis[(int64_t)b * J + j] != 0 && (mm == 0 || use_depth == nullptr || use_depth[b] != 0);
    const float sc = (valid && cnt > 0) ? 1.f / (float)cnt : 0.f;
    const float pr = __expf(col[i * (J + 1)] - lse);
    sG[i * (J + 1) + j] = (pr - (i == j ? 1.f : 0.f)) * sc;
    if (i == 0) {
      sRed[0][j] = valid ? lse - col[j * (J + 1)] : 0.f;
      sRed[1][j] = (valid && arg == j) ? 1.f : 0.f;
      sRed[2][j] = valid ? 1.f : 0.f;
    }
  }
  __syncthreads();
  if (tid < 3) {      // per-image partial sums in column order
    float sum = 0.f;
    for (int j = 0; j < J; ++j) sum += sRed[tid][j];
    part[(int64_t)b * 6 + tid * 2 + mm] = sum;
  }
  // 4. gradients wrt the unit rows: d fhat[j] = sum_i dA[i][j] ghat[i]  (LDS, then F.normalize's backward);
  //    this modality's share of d ghat[i] = sum_j dA[i][j] fhat[j]       (straight to global memory)
  for (int e = tid; e < 2 * J * (kC / 4); e += kJT) {          // four channels per thread: one 16-byte LDS read per term
    const int which = e / (J * (kC / 4)), r = (e / (kC / 4)) % J, c = (e % (kC / 4)) * 4;
    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
    if (which == 0) {
      for (int jj = 0; jj < J; ++jj)
        fma4(d, sG[r * (J + 1) + jj], *reinterpret_cast<const float4*>(sV + (J + jj) * kLS + c));
      scale4(d, inv_tau);
      *reinterpret_cast<float4*>(gpart + ((int64_t)mm * B * J + (int64_t)b * J + r) * kC + c) = d;
    } else {
      for (int ii = 0; ii < J; ++ii)
        fma4(d, sG[ii * (J + 1) + r], *reinterpret_cast<const float4*>(sV + ii * kLS + c));
      scale4(d, inv_tau);
      *reinterpret_cast<float4*>(sD + r * kLS + c) = d;
    }
  }
  __syncthreads();
  // 5. normalize backward of the modality's rows, one wave per row
  for (int j = wave; j < J; j += kJT / 64) {
    const float v0 = sV[(J + j) * kLS + lane], v1 = sV[(J + j) * kLS + lane + 64];
    const float d0 = sD[j * kLS + lane], d1 = sD[j * kLS + lane + 64];
    const float dot = wave_sum(fmaf(d0, v0, d1 * v1));
    const float inv = sInv[j];
    float* dst = dX + ((int64_t)mm * B * J + (int64_t)b * J + j) * kC;
    dst[lane] = inv < 0.f ? d0 * (-inv) : (d0 - dot * v0) * inv;
    dst[lane + 64] = inv < 0.f ? d1 * (-inv) : (d1 - dot * v1) * inv;
  }
}

// One wave per graph row: d ghat = the two modalities' shares (rgb first), pushed through F.normalize; block 0 also
// reduces the per-image partial sums: out4 = {loss_rgb, loss_d, acc_rgb, acc_d}.  An empty target set gives 0/0 = NaN like
// nn.CrossEntropyLoss; accuracy averages ncorrect/nvalid over images with nvalid > 0 (:812-822).
__global__ __launch_bounds__(kWG) void joint_finish_kernel(const float* __restrict__ part, const float* __restrict__ gpart,
                                                           const float* __restrict__ feat3, int B, int J,
                                                           float* __restrict__ out4, float* __restrict__ gfeat3) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (blockIdx.x == 0 && threadIdx.x < 2) {
    const int mm = threadIdx.x;
    float lsum = 0.f, cnt = 0.f, accsum = 0.f, nimg = 0.f;
    // sixteen images' partials in flight (the third value sat behind a branch on the second: two dependent round trips per image
    // on two lanes, 9 us for B = 32 -- the whole kernel's duration, r06); b ascending in every sum as before
    for (int b0 = 0; b0 < B; b0 += 16) {
      float pl[16], pn[16], pa[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int b = min(b0 + u, B - 1);
        pl[u] = part[b * 6 + 0 + mm];
        pn[u] = part[b * 6 + 4 + mm];
        pa[u] = part[b * 6 + 2 + mm];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (b0 + u < B) {
          lsum += pl[u];
          cnt += pn[u];
          if (pn[u] > 0.f) { accsum += pa[u] / pn[u]; nimg += 1.f; }
        }
      }
    }
    out4[mm] = lsum / cnt;
    out4[2 + mm] = accsum / nimg;
  }
  if (row >= B * J) return;
  const float x0 = feat3[(int64_t)row * kC + lane], x1 = feat3[(int64_t)row * kC + lane + 64];
  const float nrm = sqrtf(wave_sum(fmaf(x0, x0, x1 * x1)));
  const float den = fmaxf(nrm, 1e-12f);
  const float v0 = x0 / den, v1 = x1 / den;
  const int64_t o = (int64_t)row * kC, o2 = ((int64_t)B * J + row) * kC;
  const float d0 = gpart[o + lane] + gpart[o2 + lane], d1 = gpart[o + lane + 64] + gpart[o2 + lane + 64];
  const float dot = wave_sum(fmaf(d0, v0, d1 * v1));
  gfeat3[o + lane] = nrm < 1e-12f ? d0 / den : (d0 - dot * v0) / den;
  gfeat3[o + lane + 64] = nrm < 1e-12f ? d1 / den : (d1 - dot * v1) / den;
}

__global__ void joint_pixels_kernel(const float* __restrict__ j2d, int n, int h,
                                    int64_t* __restrict__ pix) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  // (original_joints2d // 4).long() clamped to [0, h-1]  (contrast_trainer.py:757-761)
  int r = (int)floorf(j2d[2 * e] / 4.f), c = (int)floorf(j2d[2 * e + 1] / 4.f);
  r = min(max(r, 0), h - 1);
  c = min(max(c, 0), h - 1);
  pix[e] = (int64_t)r * h + c;
}


// ------------------------------------------------------------------------------------------
// Bilinear up-sampling, align_corners=False (F.interpolate(..., mode='bilinear'), used by HRNet's
// fuse layers, official_hrnet.py:231-236, and merge_all_res, build_backbone.py:247-254).
// PyTorch's NCHW forward kernel walks all N*C planes inside each thread (strided stores) and costs
// 22.5 ms of a 130 ms step on MI355X (profiles/r01_bench_one_step_summary.csv); this one gives each
// thread four consecutive output pixels of one plane -> 16-byte coalesced stores, inputs from L1/L2.
// Same arithmetic as at::native::upsample_bilinear2d (area_pixel_compute_source_index).
// ------------------------------------------------------------------------------------------
template <bool ACC, bool RELU>
__global__ __launch_bounds__(kWG) void upsample_bilinear_kernel(const float* __restrict__ in,
                                                                const float* __restrict__ acc,
                                                                float* __restrict__ out, int planes,
                                                                int Hi, int Wi, int Ho, int Wo,
                                                                float sy, float sx) {
  const int wq = (Wo + 3) >> 2;  // groups of four output columns
  const int64_t total = (int64_t)planes * Ho * wq;
  for (int64_t e = (int64_t)blockIdx.x * kWG + threadIdx.x; e < total; e += (int64_t)gridDim.x * kWG) {
    const int q = (int)(e % wq);
    const int oy = (int)((e / wq) % Ho);
    const int64_t pl = e / ((int64_t)wq * Ho);
    const float fy = fmaxf(sy * ((float)oy + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)fy;
    const int y1 = y0 + (y0 < Hi - 1 ? 1 : 0);
    const float ly = fy - (float)y0, hy = 1.f - ly;
    const float* r0 = in + (pl * Hi + y0) * Wi;
    const float* r1 = in + (pl * Hi + y1) * Wi;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int ox = 4 * q + k;
      const float fx = fmaxf(sx * ((float)ox + 0.5f) - 0.5f, 0.f);
      int x0 = (int)fx;
      x0 = min(x0, Wi - 1);
      const int x1 = x0 + (x0 < Wi - 1 ? 1 : 0);
      const float lx = fx - (float)x0, hx = 1.f - lx;
      v[k] = hy * (hx * r0[x0] + lx * r0[x1]) + ly * (hx * r1[x0] + lx * r1[x1]);
    }
    float* dst = out + (pl * Ho + oy) * Wo + 4 * q;
    const float* asrc = ACC ? acc + (pl * Ho + oy) * Wo + 4 * q : nullptr;
    if (4 * q + 3 < Wo && ((Wo & 3) == 0)) {
      if (ACC) {                                       // the running sum of the fuse layer + this term (+ ReLU): one launch
        const float4 a4 = *reinterpret_cast<const float4*>(asrc);
        v[0] = a4.x + v[0]; v[1] = a4.y + v[1]; v[2] = a4.z + v[2]; v[3] = a4.w + v[3];
      }
      if (RELU) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
      *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      for (int k = 0; k < 4 && 4 * q + k < Wo; ++k) {
        float r = ACC ? asrc[k] + v[k] : v[k];
        dst[k] = RELU ? fmaxf(r, 0.f) : r;
      }
    }
  }
}

// Backward of the same op in GATHER form: one thread per INPUT pixel sums, in a fixed order (output rows
// ascending, columns ascending), the contributions of every output pixel whose 2x2 stencil touches it.
// ATen's backward scatters with float atomics (upsample_bilinear2d_backward_out_frame: 62 launches x 30 us
// per step and the last run-to-run non-deterministic kernel of the loss section); this one is
// deterministic and reads grad_out ~(2 + 1/s)^2 / 4 times through L2.  The stencil of an output pixel is
// re-derived with exactly the forward's float expressions, so forward and backward agree on every tap.
constexpr int kBwdWin = 8;    // x-window whose weights are kept in registers (covers scale factors <= 4)

// source index of output position o along one axis, exactly as the forward computes it
__device__ __forceinline__ int src_floor(int o, float scale, int n_in) {
  const float f = fmaxf(scale * ((float)o + 0.5f) - 0.5f, 0.f);
  return min((int)f, n_in - 1);
}
// [lo, hi] = the outputs whose source floor is i-1 or i (the only ones that can touch input i); the floor is
// monotone in the output position, so the estimate from the inverse map is corrected by a few exact probes
__device__ __forceinline__ void touch_range(int i, float scale, float inv_scale, int n_in, int n_out, int& lo, int& hi) {
  lo = min(max((int)floorf(((float)i - 0.5f) * inv_scale - 0.5f), 0), n_out - 1);
  while (lo > 0 && src_floor(lo - 1, scale, n_in) >= i - 1) --lo;
  while (lo < n_out - 1 && src_floor(lo, scale, n_in) < i - 1) ++lo;
  hi = min(max((int)floorf(((float)i + 1.5f) * inv_scale - 0.5f), lo), n_out - 1);
  while (hi < n_out - 1 && src_floor(hi + 1, scale, n_in) <= i) ++hi;
  while (hi > lo && src_floor(hi, scale, n_in) > i) --hi;
}

__global__ __launch_bounds__(kWG) void upsample_bilinear_bwd_kernel(const float* __restrict__ g,
                                                                    float* __restrict__ dx, int planes,
                                                                    int Hi, int Wi, int Ho, int Wo,
                                                                    float sy, float sx) {
  const int64_t total = (int64_t)planes * Hi * Wi;
  const float isy = 1.f / sy, isx = 1.f / sx;
  for (int64_t e = (int64_t)blockIdx.x * kWG + threadIdx.x; e < total; e += (int64_t)gridDim.x * kWG) {
    const int xi = (int)(e % Wi);
    const int yi = (int)((e / Wi) % Hi);
    const int64_t pl = e / ((int64_t)Wi * Hi);
    int ylo, yhi, xlo, xhi;
    touch_range(yi, sy, isy, Hi, Ho, ylo, yhi);
    touch_range(xi, sx, isx, Wi, Wo, xlo, xhi);
    auto wx_of = [&](int ox) -> float {
      const float fx = fmaxf(sx * ((float)ox + 0.5f) - 0.5f, 0.f);
      const int x0 = min((int)fx, Wi - 1);
      const int x1 = x0 + (x0 < Wi - 1 ? 1 : 0);
      const float lx = fx - (float)x0;
      return (x0 == xi ? 1.f - lx : 0.f) + (x1 == xi ? lx : 0.f);
    };
    const int nwin = xhi - xlo + 1;
    float wxv[kBwdWin];
    if (nwin <= kBwdWin) {
#pragma unroll
      for (int k = 0; k < kBwdWin; ++k) wxv[k] = k < nwin ? wx_of(xlo + k) : 0.f;
    }
    float acc = 0.f;
    for (int oy = ylo; oy <= yhi; ++oy) {
      const float fy = fmaxf(sy * ((float)oy + 0.5f) - 0.5f, 0.f);
      const int y0 = min((int)fy, Hi - 1);
      const int y1 = y0 + (y0 < Hi - 1 ? 1 : 0);
      const float ly = fy - (float)y0;
      const float wy = (y0 == yi ? 1.f - ly : 0.f) + (y1 == yi ? ly : 0.f);
      const float* grow = g + (pl * Ho + oy) * Wo;
      float racc = 0.f;
      if (nwin <= kBwdWin) {
#pragma unroll
        for (int k = 0; k < kBwdWin; ++k)
          if (k < nwin) racc = fmaf(wxv[k], grow[xlo + k], racc);
      } else {
        for (int ox = xlo; ox <= xhi; ++ox) racc = fmaf(wx_of(ox), grow[ox], racc);
      }
      acc = fmaf(wy, racc, acc);
    }
    dx[e] = acc;
  }
}

// Separable form of the same backward, one workgroup per plane (used whenever the intermediate fits LDS --
// every HRNet fuse layer): dx = Wy^T (G Wx).  Stage 1 contracts the columns, T[yo][xi] = sum_xo wx G[yo][xo]
// (Ho*Wi outputs: 8x the parallelism of the per-input-pixel form at scale factor 8, and ~2S taps each instead of
// (2S)^2); stage 2 contracts the rows out of LDS.  Stencils come from per-workgroup tables built with the
// forward's float expressions; both sums run in ascending output order: deterministic.
// MASK: the op was fused with the ReLU that followed it (y = relu(acc + up(x))): the gradient is first masked by
// y > 0 -- staged in LDS for the two passes and written out as the gradient of the running sum `acc`.
template <bool MASK>
__global__ __launch_bounds__(kWG) void upsample_bilinear_bwd_plane_kernel(const float* __restrict__ g,
                                                                          const float* __restrict__ y,
                                                                          float* __restrict__ gmasked,
                                                                          float* __restrict__ dx, int Hi, int Wi,
                                                                          int Ho, int Wo, float sy, float sx) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Gs = smem;                                  // [Ho][Wo] masked gradient plane (MASK only)
  float* T = smem + (MASK ? Ho * Wo : 0);            // [Ho][Wi]
  float* lxt = T + Ho * Wi;                          // [Wo] lambda of output column
  float* lyt = lxt + Wo;                             // [Ho]
  int* x0t = reinterpret_cast<int*>(lyt + Ho);       // [Wo] source floor of output column
  int* y0t = x0t + Wo;                               // [Ho]
  int* xlo = y0t + Ho;                               // [Wi] first / last output column touching input column
  int* xhi = xlo + Wi;
  int* ylo = xhi + Wi;                               // [Hi]
  int* yhi = ylo + Hi;
  const int64_t pl = blockIdx.x;
  const int tid = threadIdx.x;
  const float isy = 1.f / sy, isx = 1.f / sx;
  for (int o = tid; o < Wo; o += kWG) {
    const float f = fmaxf(sx * ((float)o + 0.5f) - 0.5f, 0.f);
    const int x0 = min((int)f, Wi - 1);
    x0t[o] = x0;
    lxt[o] = f - (float)x0;
  }
  for (int o = tid; o < Ho; o += kWG) {
    const float f = fmaxf(sy * ((float)o + 0.5f) - 0.5f, 0.f);
    const int y0 = min((int)f, Hi - 1);
    y0t[o] = y0;
    lyt[o] = f - (float)y0;
  }
  for (int i = tid; i < Wi; i += kWG) touch_range(i, sx, isx, Wi, Wo, xlo[i], xhi[i]);
  for (int i = tid; i < Hi; i += kWG) touch_range(i, sy, isy, Hi, Ho, ylo[i], yhi[i]);
  const float* gp = g + pl * Ho * Wo;
  if (MASK) {
    const float* yp = y + pl * Ho * Wo;
    float* mp = gmasked + pl * Ho * Wo;
    for (int o = tid; o < Ho * Wo; o += kWG) {
      const float v = yp[o] > 0.f ? gp[o] : 0.f;
      Gs[o] = v;
      mp[o] = v;
    }
  }
  __syncthreads();
  for (int o = tid; o < Ho * Wi; o += kWG) {
    const int yo = o / Wi, xi = o - yo * Wi;
    const float* grow = (MASK ? Gs : gp) + (int64_t)yo * Wo;
    float acc = 0.f;
    for (int xo = xlo[xi]; xo <= xhi[xi]; ++xo) {
      const int x0 = x0t[xo], x1 = x0 + (x0 < Wi - 1 ? 1 : 0);
      const float lx = lxt[xo];
      const float w = (x0 == xi ? 1.f - lx : 0.f) + (x1 == xi ? lx : 0.f);
      acc = fmaf(w, grow[xo], acc);
    }
    T[o] = acc;
  }
  __syncthreads();
  float* dp = dx + pl * Hi * Wi;
  for (int o = tid; o < Hi * Wi; o += kWG) {
    const int yi = o / Wi, xi = o - yi * Wi;
    float acc = 0.f;
    for (int yo = ylo[yi]; yo <= yhi[yi]; ++yo) {
      const int y0 = y0t[yo], y1 = y0 + (y0 < Hi - 1 ? 1 : 0);
      const float ly = lyt[yo];
      const float w = (y0 == yi ? 1.f - ly : 0.f) + (y1 == yi ? ly : 0.f);
      acc = fmaf(w, T[yo * Wi + xi], acc);
    }
    dp[o] = acc;
  }
}

// channels-last variant: memory is [N][H][W][C]; one thread per (pixel, channel), channel fastest.
__global__ __launch_bounds__(kWG) void upsample_bilinear_nhwc_kernel(const float* __restrict__ in,
                                                                     float* __restrict__ out, int N,
                                                                     int Cc, int Hi, int Wi, int Ho,
                                                                     int Wo, float sy, float sx) {
  const int64_t total = (int64_t)N * Ho * Wo * Cc;
  for (int64_t e = (int64_t)blockIdx.x * kWG + threadIdx.x; e < total; e += (int64_t)gridDim.x * kWG) {
    const int c = (int)(e % Cc);
    const int ox = (int)((e / Cc) % Wo);
    const int oy = (int)((e / ((int64_t)Cc * Wo)) % Ho);
    const int64_t n = e / ((int64_t)Cc * Wo * Ho);
    const float fy = fmaxf(sy * ((float)oy + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)fy;
    const int y1 = y0 + (y0 < Hi - 1 ? 1 : 0);
    const float ly = fy - (float)y0, hy = 1.f - ly;
    const float fx = fmaxf(sx * ((float)ox + 0.5f) - 0.5f, 0.f);
    const int x0 = min((int)fx, Wi - 1);
    const int x1 = x0 + (x0 < Wi - 1 ? 1 : 0);
    const float lx = fx - (float)x0, hx = 1.f - lx;
    const float* base = in + n * Hi * Wi * Cc + c;
    const float v00 = base[((int64_t)y0 * Wi + x0) * Cc], v01 = base[((int64_t)y0 * Wi + x1) * Cc];
    const float v10 = base[((int64_t)y1 * Wi + x0) * Cc], v11 = base[((int64_t)y1 * Wi + x1) * Cc];
    out[e] = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
  }
}

// ------------------------------------------------------------------------------------------
// Sampled-pixel feature-map producer.  merge_all_res (bilinear up-sampling of the
// three coarser HRNet branches + concat, build_backbone.py:247-254) followed by the 1x1 projection
// (:243-245) is linear, and the three losses read only S+J pixels per image out of h*w: sample the
// branches AT those pixels (4 bilinear taps each) into a [rows, Ctot] matrix and project only that
// (one small library GEMM on the host side).  No 270-channel concat, no full-resolution projection.
//   rows r = b*R + s ; pixel pix[r] = py*w0 + px on the finest (h0 x w0) grid.
// ------------------------------------------------------------------------------------------
struct Taps {
  int y0, y1, x0, x1;
  float hy, ly, hx, lx;
};
__device__ __forceinline__ Taps bilinear_taps(int py, int px, int hi, int wi, float sy, float sx) {
  Taps t;
  const float fy = fmaxf(sy * ((float)py + 0.5f) - 0.5f, 0.f);
  const float fx = fmaxf(sx * ((float)px + 0.5f) - 0.5f, 0.f);
  t.y0 = min((int)fy, hi - 1);
  t.x0 = min((int)fx, wi - 1);
  t.y1 = t.y0 + (t.y0 < hi - 1 ? 1 : 0);
  t.x1 = t.x0 + (t.x0 < wi - 1 ? 1 : 0);
  t.ly = fy - (float)t.y0; t.hy = 1.f - t.ly;
  t.lx = fx - (float)t.x0; t.hx = 1.f - t.lx;
  return t;
}

// one wave per row; lanes walk the branch's channels
__global__ __launch_bounds__(kWG) void sample_rows_kernel(const float* __restrict__ x, hcm_strides4 st,
                                                          int C, int hi, int wi, int h0, int w0,
                                                          const int64_t* __restrict__ pix, int R,
                                                          int nrows, float* __restrict__ out, int ldo,
                                                          int col0) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= nrows) return;
  const int b = r / R;
  const int p = (int)pix[r];
  const Taps t = bilinear_taps(p / w0, p % w0, hi, wi, (float)hi / (float)h0, (float)wi / (float)w0);
  const int64_t base = b * st.sN;
  const int64_t o00 = base + t.y0 * st.sH + t.x0 * st.sW, o01 = base + t.y0 * st.sH + t.x1 * st.sW;
  const int64_t o10 = base + t.y1 * st.sH + t.x0 * st.sW, o11 = base + t.y1 * st.sH + t.x1 * st.sW;
  for (int c = lane; c < C; c += 64) {
    const int64_t oc = c * st.sC;
    out[(int64_t)r * ldo + col0 + c] = t.hy * (t.hx * x[o00 + oc] + t.lx * x[o01 + oc]) +
                                       t.ly * (t.hx * x[o10 + oc] + t.lx * x[o11 + oc]);
  }
}

// Backward when the branch IS the sampling grid (hi == h0, wi == w0: every row is a plain gather of pixel
// pix[r] -- the finest HRNet branch, the only one the trainer routes here): owner-computes scatter-add like
// scatter_rows_kernel.  A row owns its pixel when no EARLIER row of the same image has it; the owner adds
// its own and all later duplicates' rows in index order and STORES the sum (gx arrives zero-filled):
// deterministic, no atomics.
__global__ __launch_bounds__(kWG) void scatter_sample_rows_same_grid_kernel(
    const float* __restrict__ g, int ldo, int col0, hcm_strides4 st, int C, int w0,
    const int64_t* __restrict__ pix, int R, int nrows, float* __restrict__ gx) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= nrows) return;
  const int b = r / R, s = r - b * R;
  const int64_t p = pix[r];
  const int64_t* pb = pix + (int64_t)b * R;
  bool earlier = false;
  for (int i = lane; i < s; i += 64) earlier |= (pb[i] == p);
  if (__any(earlier)) return;
  const float* src = g + (int64_t)b * R * ldo + col0;
  const int64_t dst0 = b * st.sN + (p / w0) * st.sH + (p % w0) * st.sW;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + lane;
    float a = c < C ? src[(int64_t)s * ldo + c] : 0.f;
    for (int i0 = s + 1; i0 < R; i0 += 64) {
      const int i = i0 + lane;
      unsigned long long mm = __ballot(i < R && pb[i] == p);
      while (mm) {
        const int bit = __ffsll((long long)mm) - 1;
        mm &= mm - 1;
        if (c < C) a += src[(int64_t)(i0 + bit) * ldo + c];
      }
    }
    if (c < C) gx[dst0 + c * st.sC] = a;
  }
}

// general backward: gx[b, c, tap] += weight * g[r, col0 + c]   (atomic, like ATen's own bilinear backward)
__global__ __launch_bounds__(kWG) void scatter_sample_rows_kernel(const float* __restrict__ g, int ldo,
                                                                  int col0, hcm_strides4 st, int C,
                                                                  int hi, int wi, int h0, int w0,
                                                                  const int64_t* __restrict__ pix,
                                                                  int R, int nrows,
                                                                  float* __restrict__ gx) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= nrows) return;
  const int b = r / R;
  const int p = (int)pix[r];
  const Taps t = bilinear_taps(p / w0, p % w0, hi, wi, (float)hi / (float)h0, (float)wi / (float)w0);
  const int64_t base = b * st.sN;
  const int64_t o00 = base + t.y0 * st.sH + t.x0 * st.sW, o01 = base + t.y0 * st.sH + t.x1 * st.sW;
  const int64_t o10 = base + t.y1 * st.sH + t.x0 * st.sW, o11 = base + t.y1 * st.sH + t.x1 * st.sW;
  for (int c = lane; c < C; c += 64) {
    const float gv = g[(int64_t)r * ldo + col0 + c];
    const int64_t oc = c * st.sC;
    // zero-weight taps (always 3 of 4 on the finest branch) are skipped
    if (t.hy * t.hx != 0.f) atomicAdd(gx + o00 + oc, gv * t.hy * t.hx);
    if (t.hy * t.lx != 0.f) atomicAdd(gx + o01 + oc, gv * t.hy * t.lx);
    if (t.ly * t.hx != 0.f) atomicAdd(gx + o10 + oc, gv * t.ly * t.hx);
    if (t.ly * t.lx != 0.f) atomicAdd(gx + o11 + oc, gv * t.ly * t.lx);
  }
}

// Dense form of the same linear map for the COARSE branches: S[b, r, q] = bilinear weight of coarse
// pixel q for sampled pixel pix[b, r] (4 non-zeros per row, added in a fixed order so coinciding
// border taps sum).  With S in hand, sampling is bmm(S, x) and its backward bmm(S^T, g): two small
// library GEMMs, deterministic, no atomics.  S must be zero-filled by the caller.
__global__ void sampling_matrix_kernel(const int64_t* __restrict__ pix, int nrows, int hi, int wi,
                                       int h0, int w0, float* __restrict__ S) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  const int p = (int)pix[r];
  const Taps t = bilinear_taps(p / w0, p % w0, hi, wi, (float)hi / (float)h0, (float)wi / (float)w0);
  float* row = S + (int64_t)r * hi * wi;
  row[t.y0 * wi + t.x0] += t.hy * t.hx;
  row[t.y0 * wi + t.x1] += t.hy * t.lx;
  row[t.y1 * wi + t.x0] += t.ly * t.hx;
  row[t.y1 * wi + t.x1] += t.ly * t.lx;
}

// ---- workspace carving (all regions 16-byte aligned) --------------------------------------
struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* p) : base(reinterpret_cast<char*>(p)) {}
  template <class T>
  T* take(size_t n) {
    const size_t o = off;
    off += (n * sizeof(T) + 15) & ~(size_t)15;
    return base ? reinterpret_cast<T*>(base + o) : nullptr;
  }
};

struct DenseWs {
  float *F, *invn, *stat, *rowloss, *rowcorrect, *dX, *gscale;
  int* meta;
  size_t bytes;
};
DenseWs carve_dense(void* ws, int B, int S) {
  Carver c(ws);
  DenseWs o;
  const size_t rows = (size_t)B * S;
  o.F = c.take<float>(2 * rows * kC);
  o.dX = c.take<float>(2 * rows * kC);
  o.invn = c.take<float>(2 * rows);
  o.stat = c.take<float>(2 * rows * 4);
  o.rowloss = c.take<float>(2 * rows);
  o.rowcorrect = c.take<float>(2 * rows);
  o.meta = c.take<int>(rows);
  o.gscale = c.take<float>(4);
  o.bytes = c.off;
  return o;
}
struct SclWs {
  float *F, *invn, *stat, *rowloss, *rowcorrect, *dX, *gscale, *pstat, *pdq;
  int* meta;
  int nkc;
  size_t bytes;
};
// key chunks of the SCL problem: enough (row block, chunk) workgroups for ~3/4 of the 256 CUs, at most 16
inline int scl_key_chunks(int N) {
  const int row_blocks = (N + 63) / 64, ntiles = (N + 15) / 16;
  int nkc = (192 + row_blocks - 1) / row_blocks;
  if (nkc > 16) nkc = 16;
  if (nkc > ntiles) nkc = ntiles;
  if (nkc < 1) nkc = 1;
  const int tiles_per = (ntiles + nkc - 1) / nkc;
  return (ntiles + tiles_per - 1) / tiles_per;      // drop chunks that would be empty
}
SclWs carve_scl(void* ws, int B, int J) {
  Carver c(ws);
  SclWs o;
  const size_t rows = (size_t)B * J;
  o.nkc = scl_key_chunks((int)(2 * rows));
  o.pstat = c.take<float>(o.nkc > 1 ? (size_t)o.nkc * 2 * rows * 8 : 4);
  o.pdq = c.take<float>(o.nkc > 1 ? (size_t)o.nkc * 2 * rows * kC : 4);
  o.F = c.take<float>(2 * rows * kC);
  o.dX = c.take<float>(2 * rows * kC);
  o.invn = c.take<float>(2 * rows);
  o.stat = c.take<float>(2 * rows * 4);
  o.rowloss = c.take<float>(2 * rows);
  o.rowcorrect = c.take<float>(2 * rows);
  o.meta = c.take<int>(2 * rows);
  o.gscale = c.take<float>(4);
  o.bytes = c.off;
  return o;
}
struct JointWs {
  float *part, *dX, *gpart;
  size_t bytes;
};
JointWs carve_joint(void* ws, int B, int J) {
  Carver c(ws);
  JointWs o;
  o.part = c.take<float>((size_t)B * 6);
  o.dX = c.take<float>((size_t)2 * B * J * kC);
  o.gpart = c.take<float>((size_t)2 * B * J * kC);
  o.bytes = c.off;
  return o;
}

inline MapView view(hcm_strides4 st, int w) { return MapView{st.sN, st.sC, st.sH, st.sW, w}; }

}  // namespace

extern "C" {

size_t hcm_dense_soft_nce_workspace_bytes(int B, int S, int C) {
  (void)C;
  return carve_dense(nullptr, B, S).bytes;
}

int hcm_dense_soft_nce(const float* map1, const float* map2, hcm_strides4 st, int B, int C, int h,
                       int w, const int64_t* sample_ind, const int32_t* keep, int S,
                       float temperature, float* out4, float* gmap1, float* gmap2, void* workspace,
                       size_t workspace_bytes, hcm_stream_t stream) {
  return hcm_dense_soft_nce_coords(map1, map2, st, B, C, h, w, sample_ind, nullptr, w, keep, S,
                                   temperature, out4, gmap1, gmap2, workspace, workspace_bytes, stream);
}

static int dense_impl(const float* map1, const float* map2, hcm_strides4 st, int B, int C,
                      int h, int w, const int64_t* sample_ind, const int64_t* coord_ind,
                      int coord_w, const int32_t* keep, int S, float temperature, float* out4,
                      float* gmap1, float* gmap2, void* workspace, size_t workspace_bytes,
                      hcm_stream_t stream, int mode);

int hcm_dense_soft_nce_coords(const float* map1, const float* map2, hcm_strides4 st, int B, int C,
                              int h, int w, const int64_t* sample_ind, const int64_t* coord_ind,
                              int coord_w, const int32_t* keep, int S, float temperature, float* out4,
                              float* gmap1, float* gmap2, void* workspace, size_t workspace_bytes,
                              hcm_stream_t stream) {
  return dense_impl(map1, map2, st, B, C, h, w, sample_ind, coord_ind, coord_w, keep, S, temperature, out4, gmap1, gmap2,
                    workspace, workspace_bytes, stream, 0);
}
int hcm_dense_soft_nce_coords_bf16(const float* map1, const float* map2, hcm_strides4 st, int B, int C,
                                   int h, int w, const int64_t* sample_ind, const int64_t* coord_ind,
                                   int coord_w, const int32_t* keep, int S, float temperature, float* out4,
                                   float* gmap1, float* gmap2, void* workspace, size_t workspace_bytes,
                                   hcm_stream_t stream) {
  return dense_impl(map1, map2, st, B, C, h, w, sample_ind, coord_ind, coord_w, keep, S, temperature, out4, gmap1, gmap2,
                    workspace, workspace_bytes, stream, 1);
}
int hcm_dense_soft_nce_coords_exact(const float* map1, const float* map2, hcm_strides4 st, int B, int C,
                                    int h, int w, const int64_t* sample_ind, const int64_t* coord_ind,
                                    int coord_w, const int32_t* keep, int S, float temperature, float* out4,
                                    float* gmap1, float* gmap2, void* workspace, size_t workspace_bytes,
                                    hcm_stream_t stream) {
  return dense_impl(map1, map2, st, B, C, h, w, sample_ind, coord_ind, coord_w, keep, S, temperature, out4, gmap1, gmap2,
                    workspace, workspace_bytes, stream, 2);
}

static int dense_impl(const float* map1, const float* map2, hcm_strides4 st, int B, int C,
                      int h, int w, const int64_t* sample_ind, const int64_t* coord_ind,
                      int coord_w, const int32_t* keep, int S, float temperature, float* out4,
                      float* gmap1, float* gmap2, void* workspace, size_t workspace_bytes,
                      hcm_stream_t stream, int mode) {
  const bool bf16 = mode == 1, exact = mode == 2;      // 0: split-bf16 fp32 (default), 1: bf16 operands, 2: exact fp32 (r06)
  if (coord_ind == nullptr) { coord_ind = sample_ind; coord_w = w; }
  if (coord_w <= 0) return (int)hipErrorInvalidValue;
  if (C != kC || B <= 0 || S <= 0 || h <= 0 || w <= 0 || !(temperature > 0.f) || keep == nullptr)
    return (int)hipErrorInvalidValue;
  const DenseWs ws = carve_dense(workspace, B, S);
  if (workspace == nullptr || workspace_bytes < ws.bytes) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  const int rows = B * S;
  const MapView mv = view(st, w);
  dense_prep_kernel<<<(rows + 255) / 256, 256, 0, s>>>(keep, B, S, coord_ind, ws.meta, ws.gscale);
  HCM_CHECK_LAUNCH();
  gather_norm_kernel<<<dim3((rows + 3) / 4, 2), kWG, 0, s>>>(map1, map2, mv, sample_ind, S, rows,
                                                             keep, ws.F, ws.invn);
  HCM_CHECK_LAUNCH();
  StripArgs a;
  a.F = ws.F; a.invn = ws.invn; a.meta = ws.meta; a.keep = keep; a.S = S; a.nbatch = B;
  a.symmetric = 0; a.inv_tau = (float)(1.0 / (double)temperature); a.bounded = 2.0 / (double)temperature < 80.0 ? 1 : 0; a.gscale = ws.gscale;
  a.stat = ws.stat; a.rowloss = ws.rowloss; a.rowcorrect = ws.rowcorrect; a.dX = ws.dX;
  a.nkc = 1; a.pstat = nullptr; a.pdq = nullptr;
  const dim3 grid((S + 63) / 64, B, 2);
  DensePolicy pol{coord_w};
  {
    ProfSpan span(HCM_PROF_DENSE_STATS, s);
    if (a.bounded) {
      if (bf16) strip_kernel<DensePolicy, false, true, true><<<grid, kWG, 0, s>>>(a, pol);
      else if (exact) strip_kernel<DensePolicy, false, false, true, true><<<grid, kWG, 0, s>>>(a, pol);
      else strip_kernel<DensePolicy, false, false, true><<<grid, kWG, 0, s>>>(a, pol);
    } else {
      if (bf16) strip_kernel<DensePolicy, false, true, false><<<grid, kWG, 0, s>>>(a, pol);
      else if (exact) strip_kernel<DensePolicy, false, false, false, true><<<grid, kWG, 0, s>>>(a, pol);
      else strip_kernel<DensePolicy, false, false, false><<<grid, kWG, 0, s>>>(a, pol);
    }
    HCM_CHECK_LAUNCH();
    span.stop();
  }
  {
    ProfSpan span(HCM_PROF_DENSE_GRAD, s);
    if (bf16) strip_kernel<DensePolicy, true, true, false><<<grid, kWG, 0, s>>>(a, pol);
    else if (exact) strip_kernel<DensePolicy, true, false, false, true><<<grid, kWG, 0, s>>>(a, pol);
    else strip_kernel<DensePolicy, true, false, false><<<grid, kWG, 0, s>>>(a, pol);
    HCM_CHECK_LAUNCH();
    span.stop();
  }
  dense_finish_kernel<<<1, 1024, 0, s>>>(ws.rowloss, ws.rowcorrect, keep, B, S, ws.gscale, out4);
  HCM_CHECK_LAUNCH();
  scatter_rows_kernel<<<dim3((rows + 3) / 4, 2), kWG, 0, s>>>(ws.dX, sample_ind, S, rows, keep, mv,
                                                              gmap1, gmap2);
  HCM_CHECK_LAUNCH();
  return 0;
}

size_t hcm_scl_workspace_bytes(int B, int J, int C) {
  (void)C;
  return carve_scl(nullptr, B, J).bytes;
}

static int scl_impl(const float* map1, const float* map2, hcm_strides4 st, int B, int C, int h, int w,
                    const int64_t* pix, const int32_t* use_depth, const int32_t* use_rgb, int J,
                    float temperature, float* out1, float* gmap1, float* gmap2, void* workspace,
                    size_t workspace_bytes, hcm_stream_t stream, int mode);

int hcm_scl(const float* map1, const float* map2, hcm_strides4 st, int B, int C, int h, int w,
            const int64_t* pix, const int32_t* use_depth, const int32_t* use_rgb, int J,
            float temperature, float* out1, float* gmap1, float* gmap2, void* workspace,
            size_t workspace_bytes, hcm_stream_t stream) {
  return scl_impl(map1, map2, st, B, C, h, w, pix, use_depth, use_rgb, J, temperature, out1, gmap1, gmap2, workspace,
                  workspace_bytes, stream, 0);
}
int hcm_scl_bf16(const float* map1, const float* map2, hcm_strides4 st, int B, int C, int h, int w,
                 const int64_t* pix, const int32_t* use_depth, const int32_t* use_rgb, int J,
                 float temperature, float* out1, float* gmap1, float* gmap2, void* workspace,
                 size_t workspace_bytes, hcm_stream_t stream) {
  return scl_impl(map1, map2, st, B, C, h, w, pix, use_depth, use_rgb, J, temperature, out1, gmap1, gmap2, workspace,
                  workspace_bytes, stream, 1);
}
int hcm_scl_exact(const float* map1, const float* map2, hcm_strides4 st, int B, int C, int h, int w,
                  const int64_t* pix, const int32_t* use_depth, const int32_t* use_rgb, int J,
                  float temperature, float* out1, float* gmap1, float* gmap2, void* workspace,
                  size_t workspace_bytes, hcm_stream_t stream) {
  return scl_impl(map1, map2, st, B, C, h, w, pix, use_depth, use_rgb, J, temperature, out1, gmap1, gmap2, workspace,
                  workspace_bytes, stream, 2);
}

static int scl_impl(const float* map1, const float* map2, hcm_strides4 st, int B, int C, int h, int w,
                    const int64_t* pix, const int32_t* use_depth, const int32_t* use_rgb, int J,
                    float temperature, float* out1, float* gmap1, float* gmap2, void* workspace,
                    size_t workspace_bytes, hcm_stream_t stream, int mode) {
  const bool bf16 = mode == 1, exact = mode == 2;      // 0: split-bf16 fp32 (default), 1: bf16 operands, 2: exact fp32 (r06)
  if (C != kC || B <= 0 || J <= 0 || J > 0xffff || h <= 0 || w <= 0 || !(temperature > 0.f) ||
      use_depth == nullptr)
    return (int)hipErrorInvalidValue;
  const SclWs ws = carve_scl(workspace, B, J);
  if (workspace == nullptr || workspace_bytes < ws.bytes) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  const int rows = B * J, N = 2 * rows;
  const MapView mv = view(st, w);
  scl_prep_kernel<<<(N + 255) / 256, 256, 0, s>>>(use_depth, use_rgb, B, J, ws.meta, ws.gscale);
  HCM_CHECK_LAUNCH();
  gather_norm_kernel<<<dim3((rows + 3) / 4, 2), kWG, 0, s>>>(map1, map2, mv, pix, J, rows, nullptr,
                                                             ws.F, ws.invn);
  HCM_CHECK_LAUNCH();
  StripArgs a;
  a.F = ws.F; a.invn = ws.invn; a.meta = ws.meta; a.keep = nullptr; a.S = J; a.nbatch = B;
  a.symmetric = 1; a.inv_tau = (float)(1.0 / (double)temperature); a.bounded = 2.0 / (double)temperature < 80.0 ? 1 : 0; a.gscale = ws.gscale;
  a.stat = ws.stat; a.rowloss = ws.rowloss; a.rowcorrect = ws.rowcorrect; a.dX = ws.dX;
  a.nkc = ws.nkc; a.pstat = ws.pstat; a.pdq = ws.pdq;
  const dim3 grid((N + 63) / 64, ws.nkc, 1);
  SclPolicy pol;
  {
    ProfSpan span(HCM_PROF_SCL_STATS, s);
    if (a.bounded) {
      if (bf16) strip_kernel<SclPolicy, false, true, true><<<grid, kWG, 0, s>>>(a, pol);
      else if (exact) strip_kernel<SclPolicy, false, false, true, true><<<grid, kWG, 0, s>>>(a, pol);
      else strip_kernel<SclPolicy, false, false, true><<<grid, kWG, 0, s>>>(a, pol);
    } else {
      if (bf16) strip_kernel<SclPolicy, false, true, false><<<grid, kWG, 0, s>>>(a, pol);
      else if (exact) strip_kernel<SclPolicy, false, false, false, true><<<grid, kWG, 0, s>>>(a, pol);
      else strip_kernel<SclPolicy, false, false, false><<<grid, kWG, 0, s>>>(a, pol);
    }
    HCM_CHECK_LAUNCH();
    if (ws.nkc > 1) {
      strip_merge_stats_kernel<SclPolicy><<<(N + kWG - 1) / kWG, kWG, 0, s>>>(a, pol, N);
      HCM_CHECK_LAUNCH();
    }
    span.stop();
  }
  {
    ProfSpan span(HCM_PROF_SCL_GRAD, s);
    if (bf16) strip_kernel<SclPolicy, true, true, false><<<grid, kWG, 0, s>>>(a, pol);
    else if (exact) strip_kernel<SclPolicy, true, false, false, true><<<grid, kWG, 0, s>>>(a, pol);
    else strip_kernel<SclPolicy, true, false, false><<<grid, kWG, 0, s>>>(a, pol);
    HCM_CHECK_LAUNCH();
    if (ws.nkc > 1) {
      strip_merge_grad_kernel<<<(N + 3) / 4, kWG, 0, s>>>(a, N);
      HCM_CHECK_LAUNCH();
    }
    span.stop();
  }
  scl_finish_kernel<<<1, kWG, 0, s>>>(ws.rowloss, N, ws.gscale, out1);
  HCM_CHECK_LAUNCH();
  scatter_rows_kernel<<<dim3((rows + 3) / 4, 2), kWG, 0, s>>>(ws.dX, pix, J, rows, nullptr, mv,
                                                              gmap1, gmap2);
  HCM_CHECK_LAUNCH();
  return 0;
}

size_t hcm_joint_nce_workspace_bytes(int B, int J, int C) {
  (void)C;
  return carve_joint(nullptr, B, J).bytes;
}

int hcm_joint_nce(const float* map1, const float* map2, hcm_strides4 st, int B, int C, int h, int w,
                  const float* feat3, const int64_t* pix, const int32_t* joints_vis,
                  const int32_t* use_depth, int J, float temperature, float* out4, float* gmap1,
                  float* gmap2, float* gfeat3, void* workspace, size_t workspace_bytes,
                  hcm_stream_t stream) {
  if (C != kC || B <= 0 || J <= 0 || J > kJMax || h <= 0 || w <= 0 || !(temperature > 0.f))
    return (int)hipErrorInvalidValue;
  const JointWs ws = carve_joint(workspace, B, J);
  if (workspace == nullptr || workspace_bytes < ws.bytes) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  const MapView mv = view(st, w);
  const size_t lds = (size_t)(3 * J * kLS + 2 * J * (J + 1) + J) * sizeof(float);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(joint_nce_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  ProfSpan span(HCM_PROF_JOINT, s);
  joint_nce_kernel<<<dim3(B, 2), kJT, lds, s>>>(map1, map2, mv, feat3, pix, joints_vis, use_depth, B, J,
                                                (float)(1.0 / (double)temperature), ws.part, ws.dX, ws.gpart);
  HCM_CHECK_LAUNCH();
  joint_finish_kernel<<<(B * J + 3) / 4, kWG, 0, s>>>(ws.part, ws.gpart, feat3, B, J, out4, gfeat3);
  span.stop();
  HCM_CHECK_LAUNCH();
  const int rows = B * J;
  scatter_rows_kernel<<<dim3((rows + 3) / 4, 2), kWG, 0, s>>>(ws.dX, pix, J, rows, nullptr, mv,
                                                              gmap1, gmap2);
  HCM_CHECK_LAUNCH();
  return 0;
}

int hcm_upsample_bilinear2d(const float* in, int planes, int Hi, int Wi, int Ho, int Wo, float* out,
                            hcm_stream_t stream) {
  if (planes <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return (int)hipErrorInvalidValue;
  const int64_t total = (int64_t)planes * Ho * ((Wo + 3) / 4);
  int64_t blocks = (total + kWG - 1) / kWG;
  if (blocks > 16384) blocks = 16384;
  upsample_bilinear_kernel<false, false><<<(int)blocks, kWG, 0, (hipStream_t)stream>>>(
      in, nullptr, out, planes, Hi, Wi, Ho, Wo, (float)Hi / (float)Ho, (float)Wi / (float)Wo);
  HCM_CHECK_LAUNCH();
  return 0;
}

int hcm_upsample_bilinear2d_add(const float* in, const float* acc, int relu, int planes, int Hi, int Wi, int Ho, int Wo,
                                float* out, hcm_stream_t stream) {
  if (planes <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || !in || !acc || !out) return (int)hipErrorInvalidValue;
  const int64_t total = (int64_t)planes * Ho * ((Wo + 3) / 4);
  int64_t blocks = (total + kWG - 1) / kWG;
  if (blocks > 16384) blocks = 16384;
  const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
  if (relu) upsample_bilinear_kernel<true, true><<<(int)blocks, kWG, 0, (hipStream_t)stream>>>(in, acc, out, planes, Hi, Wi, Ho, Wo, sy, sx);
  else upsample_bilinear_kernel<true, false><<<(int)blocks, kWG, 0, (hipStream_t)stream>>>(in, acc, out, planes, Hi, Wi, Ho, Wo, sy, sx);
  HCM_CHECK_LAUNCH();
  return 0;
}

int hcm_upsample_bilinear2d_backward_relu(const float* grad_out, const float* y, int planes, int Hi, int Wi, int Ho, int Wo,
                                          float* grad_in, float* grad_masked, hcm_stream_t stream) {
  if (planes <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || !grad_out || !y || !grad_in || !grad_masked)
    return (int)hipErrorInvalidValue;
  const size_t lds = ((size_t)Ho * Wo + (size_t)Ho * Wi + 2 * (size_t)(Wo + Ho) + 2 * (size_t)(Wi + Hi)) * sizeof(float);
  if (lds > 60 * 1024) return (int)hipErrorInvalidValue;      // caller: threshold + hcm_upsample_bilinear2d_backward
  upsample_bilinear_bwd_plane_kernel<true><<<planes, kWG, lds, (hipStream_t)stream>>>(
      grad_out, y, grad_masked, grad_in, Hi, Wi, Ho, Wo, (float)Hi / (float)Ho, (float)Wi / (float)Wo);
  HCM_CHECK_LAUNCH();
  return 0;
}

int hcm_upsample_bilinear2d_backward(const float* grad_out, int planes, int Hi, int Wi, int Ho, int Wo,
                                     float* grad_in, hcm_stream_t stream) {
  if (planes <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || !grad_out || !grad_in) return (int)hipErrorInvalidValue;
  const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
  const size_t lds = ((size_t)Ho * Wi + 2 * (size_t)(Wo + Ho) + 2 * (size_t)(Wi + Hi)) * sizeof(float);
  if (lds <= 60 * 1024) {        // separable form, intermediate in LDS (all HRNet shapes)
    upsample_bilinear_bwd_plane_kernel<false><<<planes, kWG, lds, (hipStream_t)stream>>>(grad_out, nullptr, nullptr, grad_in,
                                                                                        Hi, Wi, Ho, Wo, sy, sx);
    HCM_CHECK_LAUNCH();
    return 0;
  }
  const int64_t total = (int64_t)planes * Hi * Wi;
  int64_t blocks = (total + kWG - 1) / kWG;
  if (blocks > 32768) blocks = 32768;
  upsample_bilinear_bwd_kernel<<<(int)blocks, kWG, 0, (hipStream_t)stream>>>(grad_out, grad_in, planes, Hi, Wi, Ho, Wo,
                                                                             sy, sx);
  HCM_CHECK_LAUNCH();
  return 0;
}

int hcm_upsample_bilinear2d_nhwc(const float* in, int N, int C, int Hi, int Wi, int Ho, int Wo,
                                 float* out, hcm_stream_t stream) {
  if (N <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return (int)hipErrorInvalidValue;
  const int64_t total = (int64_t)N * Ho * Wo * C;
  int64_t blocks = (total + kWG - 1) / kWG;
  if (blocks > 32768) blocks = 32768;
  upsample_bilinear_nhwc_kernel<<<(int)blocks, kWG, 0, (hipStream_t)stream>>>(
      in, out, N, C, Hi, Wi, Ho, Wo, (float)Hi / (float)Ho, (float)Wi / (float)Wo);
  HCM_CHECK_LAUNCH();
  return 0;
}

int hcm_sample_rows(const float* x, hcm_strides4 st, int B, int C, int hi, int wi, int h0, int w0,
                    const int64_t* pix, int R, float* out, int ldo, int col0, hcm_stream_t stream) {
  if (B <= 0 || C <= 0 || hi <= 0 || wi <= 0 || h0 <= 0 || w0 <= 0 || R <= 0 || ldo < col0 + C)
    return (int)hipErrorInvalidValue;
  const int nrows = B * R;
  sample_rows_kernel<<<(nrows + 3) / 4, kWG, 0, (hipStream_t)stream>>>(x, st, C, hi, wi, h0, w0, pix, R,
                                                                       nrows, out, ldo, col0);
  HCM_CHECK_LAUNCH();
  return 0;
}

int hcm_sample_rows_grad(const float* grad_rows, int ldo, int col0, hcm_strides4 st, int B, int C,
                         int hi, int wi, int h0, int w0, const int64_t* pix, int R, float* gx,
                         hcm_stream_t stream) {
  if (B <= 0 || C <= 0 || hi <= 0 || wi <= 0 || h0 <= 0 || w0 <= 0 || R <= 0 || ldo < col0 + C)
    return (int)hipErrorInvalidValue;
  const int nrows = B * R;
  if (hi == h0 && wi == w0)      // plain gather: deterministic owner-computes scatter
    scatter_sample_rows_same_grid_kernel<<<(nrows + 3) / 4, kWG, 0, (hipStream_t)stream>>>(
        grad_rows, ldo, col0, st, C, w0, pix, R, nrows, gx);
  else
    scatter_sample_rows_kernel<<<(nrows + 3) / 4, kWG, 0, (hipStream_t)stream>>>(
        grad_rows, ldo, col0, st, C, hi, wi, h0, w0, pix, R, nrows, gx);
  HCM_CHECK_LAUNCH();
  return 0;
}

int hcm_sampling_matrix(const int64_t* pix, int nrows, int hi, int wi, int h0, int w0, float* S,
                        hcm_stream_t stream) {
  if (nrows <= 0 || hi <= 0 || wi <= 0 || h0 <= 0 || w0 <= 0) return (int)hipErrorInvalidValue;
  sampling_matrix_kernel<<<(nrows + 255) / 256, 256, 0, (hipStream_t)stream>>>(pix, nrows, hi, wi, h0, w0, S);
  HCM_CHECK_LAUNCH();
  return 0;
}

int hcm_joint_pixels(const float* joints2d, int BJ, int h, int64_t* pix, hcm_stream_t stream) {
  if (BJ <= 0 || h <= 0) return (int)hipErrorInvalidValue;
  joint_pixels_kernel<<<(BJ + 255) / 256, 256, 0, (hipStream_t)stream>>>(joints2d, BJ, h, pix);
  HCM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"

#ifdef HCM_STRIP_TIMING
extern "C" int hcm_debug_strip_timing(unsigned long long* buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_strip_dbg), &buf, sizeof(buf));
}
#endif
