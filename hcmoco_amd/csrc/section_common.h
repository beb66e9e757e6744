// Pieces shared by the loss-section kernels (section.hip, rowproj.hip): the eight-map argument block, the bilinear
// stencil of merge_all_res (networks/build_backbone.py:247-254) and the argument checks.
#pragma once
#include "hcm_common.h"
#include "../../include/hcmoco_hip.h"

namespace hcm {

// Both HRNets have the same branch geometry; the kernels take ONE argument block with the eight map pointers
// (selected by compare chains: indexing a by-value struct with a run-time index sends it to scratch).
struct Maps8 {
  const float* p[8];
  int C[4], H[4], W[4];
};
struct Maps8T {
  const float* t[8];      // channels-last copies [B, H W, C] of (modality, branch) maps, or NULL (csrc/rowproj.hip, r06)
};
struct Maps8Out {
  float* p[8];
  int C[4], H[4], W[4];
};
inline Maps8 pack8(const hcm_branches& a, const hcm_branches& b) {
  Maps8 o;
  for (int i = 0; i < 4; ++i) { o.p[i] = a.map[i]; o.p[4 + i] = b.map[i]; o.C[i] = a.C[i]; o.H[i] = a.H[i]; o.W[i] = a.W[i]; }
  return o;
}
inline Maps8Out pack8(const hcm_branches_out& a, const hcm_branches_out& b) {
  Maps8Out o;
  for (int i = 0; i < 4; ++i) { o.p[i] = a.map[i]; o.p[4 + i] = b.map[i]; o.C[i] = a.C[i]; o.H[i] = a.H[i]; o.W[i] = a.W[i]; }
  return o;
}
__device__ __forceinline__ int sel4(const int (&v)[4], int i) {
  return i == 0 ? v[0] : (i == 1 ? v[1] : (i == 2 ? v[2] : v[3]));
}
template <class P>
__device__ __forceinline__ P sel8(P const (&v)[8], int i) {
  const P lo = i == 0 ? v[0] : (i == 1 ? v[1] : (i == 2 ? v[2] : v[3]));
  const P hi = i == 4 ? v[4] : (i == 5 ? v[5] : (i == 6 ? v[6] : v[7]));
  return i < 4 ? lo : hi;
}

// ------------------------------------------------------------------------------------------
// bilinear stencil of a coarse branch for pixel (py, px) of the finest grid: the arithmetic of
// at::native::upsample_bilinear2d with align_corners=False (same as fmap.hip:bilinear_taps)
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

// r05: an encoder whose map[0] is NULL is ABSENT -- the second modality does not come from an HRNet (HRNetPN: a point
// cloud encoder); the entry points then work on modality 0 alone and leave modality 1's slices to the caller.
inline bool absent(const hcm_branches& e) { return e.map[0] == nullptr; }
inline bool absent(const hcm_branches_out& e) { return e.map[0] == nullptr; }

inline bool branches_ok(const hcm_branches& e, int Ctot) {
  int s = 0;
  for (int i = 0; i < 4; ++i) {
    if (e.map[i] == nullptr || e.C[i] <= 0 || e.H[i] <= 0 || e.W[i] <= 0) return false;
    s += e.C[i];
  }
  return s == Ctot;
}
inline bool branches_ok(const hcm_branches_out& e, int Ctot) {
  int s = 0;
  for (int i = 0; i < 4; ++i) {
    if (e.map[i] == nullptr || e.C[i] <= 0 || e.H[i] <= 0 || e.W[i] <= 0) return false;
    s += e.C[i];
  }
  return s == Ctot;
}


}  // namespace hcm
