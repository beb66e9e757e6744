// Device-side helpers shared by the gfx950 kernels of libhcmoco_hip.so.
// CDNA4 only: 64-lane wavefronts, DPP row = 16 lanes.
#pragma once
// gfx950 ONLY: the kernels use CDNA4 instructions without a fallback -- ds_read_b64_tr_b16, v_mfma_f32_16x16x32_bf16,
// 160 KB of LDS per workgroup -- and are tuned for 256 CUs / wave64.  A build for another target must fail here, not at run time.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "hcmoco_amd/csrc targets gfx950 (MI355X, CDNA4) only: build with --offload-arch=gfx950 (csrc/Makefile ARCH)"
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>

#define HCM_LOG2E 1.4426950408889634f
#define HCM_LN2 0.6931471805599453f

#define HCM_CHECK_LAUNCH()                     \
  do {                                         \
    hipError_t e__ = hipGetLastError();        \
    if (e__ != hipSuccess) return (int)e__;    \
  } while (0)

namespace hcm {

// Opt-in hipEvent timing of individual kernels (bench.py's roofline objects): tags are the HCM_PROF_* values of
// include/hcmoco_hip.h.  Defined in bank.hip; the library's only process-global state, mutex-protected
// (kernels are launched from the trainer thread, autograd's device thread and the encoder runtime's helpers).
struct ProfSpan {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipStream_t st;
  int tag;
  // work: what this launch does in the tag's unit (flops, bytes or distance evaluations), summed per tag next to the time:
  // kernels that are launched at many shapes (the PointNet++ layers) still give ONE achieved rate, sum(work) / sum(time)
  ProfSpan(int tag, hipStream_t s, double work = 0.0);
  ~ProfSpan() { stop(); }
  void add_work(double work);      // more work for this span's tag (a span around several launches)
  void stop();
};

// One DPP-modified move: lane <- lane' of the same 16-lane row.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// All-reduce (sum / max) over the 16 lanes of a DPP row: xor-1, xor-2 inside the quad,
// then row_half_mirror (i <-> 7-i) and row_mirror (i <-> 15-i).
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  return v;
}

// All-reduce over the whole 64-lane wave (row16 + two cross-row exchanges).
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  v = row16_max(v);
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }

__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
  return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)));
}
__device__ __forceinline__ void fma4(float4& acc, float s, const float4& r) {
  acc.x = fmaf(s, r.x, acc.x);
  acc.y = fmaf(s, r.y, acc.y);
  acc.z = fmaf(s, r.z, acc.z);
  acc.w = fmaf(s, r.w, acc.w);
}
__device__ __forceinline__ void scale4(float4& acc, float s) {
  acc.x *= s; acc.y *= s; acc.z *= s; acc.w *= s;
}

// Philox4x32-10 (Random123): the counter-based generator behind every on-device draw of this library
// (negative rows: hcm_alias_draw; sampled pixels: hcm_pixel_sample).  Known-answer vectors: tests + oracle.
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

}  // namespace hcm
