/*
 * hcmoco_hip.h -- C ABI of libhcmoco_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the HCMoCo contrastive pre-training hot path
 * (SURVEY.md section 8).  Plain pointers and sizes only: every pointer is a
 * DEVICE pointer unless its comment says "host".  All entry points are
 * asynchronous on `stream` (a hipStream_t passed as void*), re-entrant, keep no
 * global state (sole exception: the opt-in timing hooks hcm_prof_*), never
 * allocate device memory, and return a hipError_t value as int
 * (0 == hipSuccess).  The caller owns every buffer, including workspaces whose
 * size is reported by the matching *_workspace_bytes() function.
 *
 * File:line citations name the reference interface (under
 * /root/reference/pycontrast) that each entry point replaces.
 */
#ifndef HCMOCO_HIP_H
#define HCMOCO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* hcm_stream_t; /* hipStream_t */

/* 2 (round 3): hcm_sgc_forward/backward take a workspace, the PointNet++ index/distance ops default to the
 * FMA arithmetic contract, new loss-section entry points.  3 (round 4): hcm_project_rows* added (row 8 on the matrix
 * cores); nothing removed.  4 (round 5): hcm_bn_relu_ballmax_*, hcm_ball_project_* (PointNet++ set abstraction without the
 * grouped tensors); nothing removed.  5 (round 6): hcm_ball_project_* take the relative offsets D and W_xyz instead of Q = W_xyz
 * centre (signature change: the coordinate half is no longer a difference of two projections).  A library and a caller must
 * agree on this number. */
#define HCM_ABI_VERSION 6
int hcm_abi_version(void);
/* hipGetErrorString() for a value returned by any entry point. */
const char* hcm_error_string(int err);

/* ------------------------------------------------------------------------ *
 * Row 1 -- alias sampler.  memory/alias_multinomial.py:7-42 (__init__),
 * :48-65 (draw).
 * ------------------------------------------------------------------------ */
/* HOST function.  Walker tables for `probs[n]` exactly as AliasMethod.__init__
 * builds them (fp32 arithmetic, LIFO work lists).  prob_out[n], alias_out[n]: host. */
int hcm_alias_build(const float* probs_host, int64_t n, float* prob_out_host, int64_t* alias_out_host);

/* idx[b*K1 + k] = draw for k>0, idx[b*K1] = y[b]  (mem_bank.py:176-177).
 * Counter based: element e = b*K1+k uses Philox4x32-10(ctr={e_lo,e_hi,off_lo,off_hi}, key=seed):
 *   kk = (r0<<32|r1) % n ; u = (r2>>8)*2^-24 ; out = u < prob[kk] ? kk : alias[kk].
 * y may be NULL (no positive column written: plain draw of B*K1 samples). */
int hcm_alias_draw(const float* prob, const int64_t* alias, int64_t n,
                   const int64_t* y, int B, int K1, uint64_t seed, uint64_t offset,
                   int64_t* idx, hcm_stream_t stream);

/* ------------------------------------------------------------------------ *
 * Rows 2+4 fused -- CMCMem3.forward gather + _compute_logit (memory/mem_bank.py:172-193,
 * :30-40) + _compute_loss_accuracy / CrossEntropyLoss / accuracy
 * (learning/contrast_trainer.py:212-253, learning/util.py:24-38) + their backward.
 *
 * banks: 3 x [n, D] fp32 row-major (D in {64,128}); idx [B, K1] int64 with idx[:,0] = y;
 * x1..x3 [B, D].  use_depth / use_rgb: [B] int32 or NULL, selecting the rows each of
 * the six logit sets averages over (contrast_trainer.py:223-250; sets are ordered
 * 12,21,23,32,13,31).  Outputs: losses[6], accs[6] (percent), gx1..gx3 [B, D]
 * (d sum(losses) / d x), all fp32.  logits are never materialised.
 * Every value of idx (and of all_y in hcm_bank_update) must lie in [0, n): like a raw gather the kernels do
 * not range-check -- torch's index_select / index_copy_ would device-assert.  The host mirror
 * (memory/mem_bank.py:CMCMem3) clamps what it passes and raises from check_indices().
 * ------------------------------------------------------------------------ */
size_t hcm_bank_nce_workspace_bytes(int B, int K1, int D);
int hcm_bank_nce_fused(const float* bank1, const float* bank2, const float* bank3, int64_t n,
                       const int64_t* idx, const float* x1, const float* x2, const float* x3,
                       const int32_t* use_depth, const int32_t* use_rgb,
                       int B, int K1, int D, float T,
                       float* losses6, float* accs6, float* gx1, float* gx2, float* gx3,
                       void* workspace, size_t workspace_bytes, hcm_stream_t stream);

/* API mode (materialised logits, the literal CMCMem3.forward contract, mem_bank.py:186-205):
 * logits [6, B, K1] fp32; backward takes d loss/d logits in the same layout. */
int hcm_bank_logits_fwd(const float* bank1, const float* bank2, const float* bank3, int64_t n,
                        const int64_t* idx, const float* x1, const float* x2, const float* x3,
                        int B, int K1, int D, float T, float* logits, hcm_stream_t stream);
int hcm_bank_logits_bwd(const float* bank1, const float* bank2, const float* bank3, int64_t n,
                        const int64_t* idx, const float* grad_logits,
                        int B, int K1, int D, float T, float* gx1, float* gx2, float* gx3,
                        void* workspace, size_t workspace_bytes, hcm_stream_t stream);

/* bf16 bank storage (BASELINE config 5): banks are [n, 128] bfloat16 (raw uint16 bits), every
 * product and sum stays fp32, the update rounds to nearest even.  Same contracts as the fp32
 * entry points above/below; D must be 128. */
int hcm_bank_nce_fused_bf16(const uint16_t* bank1, const uint16_t* bank2, const uint16_t* bank3, int64_t n,
                            const int64_t* idx, const float* x1, const float* x2, const float* x3,
                            const int32_t* use_depth, const int32_t* use_rgb,
                            int B, int K1, int D, float T,
                            float* losses6, float* accs6, float* gx1, float* gx2, float* gx3,
                            void* workspace, size_t workspace_bytes, hcm_stream_t stream);
int hcm_bank_logits_fwd_bf16(const uint16_t* bank1, const uint16_t* bank2, const uint16_t* bank3, int64_t n,
                             const int64_t* idx, const float* x1, const float* x2, const float* x3,
                             int B, int K1, int D, float T, float* logits, hcm_stream_t stream);
int hcm_bank_logits_bwd_bf16(const uint16_t* bank1, const uint16_t* bank2, const uint16_t* bank3, int64_t n,
                             const int64_t* idx, const float* grad_logits,
                             int B, int K1, int D, float T, float* gx1, float* gx2, float* gx3,
                             void* workspace, size_t workspace_bytes, hcm_stream_t stream);
int hcm_bank_update_bf16(uint16_t* bank1, uint16_t* bank2, uint16_t* bank3, int64_t n,
                         const float* all_x1, const float* all_x2, const float* all_x3,
                         const int64_t* all_y, int BW, int D, float momentum, hcm_stream_t stream);

/* ------------------------------------------------------------------------ *
 * Row 3 -- BaseMem._update_memory (memory/mem_bank.py:15-28), all three banks in one launch.
 * all_x* [BW, D], all_y [BW] int64 in gather (rank-major) order.  Reads use the
 * pre-update rows; duplicate y: the LAST occurrence writes (torch CPU index_copy_).
 * ------------------------------------------------------------------------ */
int hcm_bank_update(float* bank1, float* bank2, float* bank3, int64_t n,
                    const float* all_x1, const float* all_x2, const float* all_x3,
                    const int64_t* all_y, int BW, int D, float momentum, hcm_stream_t stream);

/* Range-checked forms used by the host mirror (memory/mem_bank.py): indices outside [0, n) are CLAMPED -- nothing
 * outside the banks is read or written -- and *oob_flag (device int32, caller-zeroed, sticky) is OR-ed with 1 when a
 * clamp changed a value (torch's index_select / index_copy_ would device-assert, memory/mem_bank.py:24-28, :179-184).
 * hcm_bank_update_checked also takes the row stride `ldx` of all_x* (>= D): the three feature blocks may be column
 * slices of one [BW, 3D (+2)] gathered matrix (learning/contrast_trainer.py:950-956). */
int hcm_alias_draw_checked(const float* prob, const int64_t* alias, int64_t n, const int64_t* y, int B, int K1,
                           uint64_t seed, uint64_t offset, int64_t* idx, int32_t* oob_flag, hcm_stream_t stream);
int hcm_bank_update_checked(float* bank1, float* bank2, float* bank3, int64_t n, const float* all_x1,
                            const float* all_x2, const float* all_x3, int64_t ldx, const int64_t* all_y, int BW,
                            int D, float momentum, int32_t* oob_flag, hcm_stream_t stream);
int hcm_bank_update_checked_bf16(uint16_t* bank1, uint16_t* bank2, uint16_t* bank3, int64_t n,
                                 const float* all_x1, const float* all_x2, const float* all_x3, int64_t ldx,
                                 const int64_t* all_y, int BW, int D, float momentum, int32_t* oob_flag,
                                 hcm_stream_t stream);

/* ------------------------------------------------------------------------ *
 * MoCo queue (secondary) -- memory/mem_moco.py:15-49.
 * logits [B, K+1] = cat(q.k, q @ queue^T)/T ; enqueue rows (index + j) % K.
 * ------------------------------------------------------------------------ */
int hcm_moco_logits(const float* q, const float* k, const float* queue, int B, int K, int D, float T,
                    float* logits, hcm_stream_t stream);
int hcm_moco_enqueue(float* queue, const float* all_k, int n_new, int K, int D, int64_t index,
                     hcm_stream_t stream);

/* ------------------------------------------------------------------------ *
 * Rows 5-7 -- feature-map losses (learning/contrast_trainer.py:642-892), forward + backward.
 * Feature maps are addressed through element strides {sN, sC, sH, sW} so both NCHW and
 * channels-last tensors are read in place; C must be 128.  Gradients wrt the maps are
 * ACCUMULATED (+=) into gmap1/gmap2 (same strides), which the caller zero-fills once per
 * step; all three losses add into the same gradient buffers.
 * ------------------------------------------------------------------------ */
typedef struct {
  int64_t sN, sC, sH, sW;
} hcm_strides4;

/* Row 5: _compute_soft_pri3d_loss_accuracy (:642-723).  sample_ind [B, S] int64 flat
 * pixel index (row*w + col) per image; keep [B] int32 (image has a non-empty mask, :677-682);
 * rows of sample_ind for dropped images are ignored.  out4 = {loss_r2d, loss_d2r, acc_r2d, acc_d2r}. */
size_t hcm_dense_soft_nce_workspace_bytes(int B, int S, int C);
int hcm_dense_soft_nce(const float* map1, const float* map2, hcm_strides4 st, int B, int C, int h, int w,
                       const int64_t* sample_ind, const int32_t* keep, int S, float temperature,
                       float* out4, float* gmap1, float* gmap2,
                       void* workspace, size_t workspace_bytes, hcm_stream_t stream);

/* Same, with the soft-target pixel coordinates decoupled from the gather index: coord_ind [B, S]
 * (row*coord_w + col) feeds the distance target (:702-706) while sample_ind addresses the map.
 * Used when the "map" is a matrix of already-sampled rows (hcm_sample_rows). */
int hcm_dense_soft_nce_coords(const float* map1, const float* map2, hcm_strides4 st, int B, int C, int h, int w,
                              const int64_t* sample_ind, const int64_t* coord_ind, int coord_w,
                              const int32_t* keep, int S, float temperature,
                              float* out4, float* gmap1, float* gmap2,
                              void* workspace, size_t workspace_bytes, hcm_stream_t stream);

/* Row 6: _compute_joints_pri3d_loss_accuracy (:744-828).  pix [B, J] int64 flat pixel of each
 * joint (clamp(floor(j/4))), feat3 [B, J, C] contiguous, joints_vis [B, J] int32,
 * use_depth [B] int32 or NULL.  out4 = {loss_rgb, loss_d, acc_rgb, acc_d}.
 * gfeat3 [B, J, C] is OVERWRITTEN. */
size_t hcm_joint_nce_workspace_bytes(int B, int J, int C);
int hcm_joint_nce(const float* map1, const float* map2, hcm_strides4 st, int B, int C, int h, int w,
                  const float* feat3, const int64_t* pix, const int32_t* joints_vis,
                  const int32_t* use_depth, int J, float temperature,
                  float* out4, float* gmap1, float* gmap2, float* gfeat3,
                  void* workspace, size_t workspace_bytes, hcm_stream_t stream);

/* Row 7: _compute_cross_subject_joints_pri3d_loss (:830-892; use_rgb==NULL follows
 * learning/segment_trainer.py:601-606).  out1 = {loss}. */
size_t hcm_scl_workspace_bytes(int B, int J, int C);
int hcm_scl(const float* map1, const float* map2, hcm_strides4 st, int B, int C, int h, int w,
            const int64_t* pix, const int32_t* use_depth, const int32_t* use_rgb, int J,
            float temperature, float* out1, float* gmap1, float* gmap2,
            void* workspace, size_t workspace_bytes, hcm_stream_t stream);

/* BASELINE config 5 ("bf16 feature-map GEMMs"): the same two losses with both contractions of the strip
 * kernel (Q K^T and G K) on the bf16 matrix cores -- operands rounded to bf16 in registers, fp32
 * accumulation, everything else fp32; same arguments, same workspaces.  Parity vs the fp32 path / oracle:
 * 1e-2 relative on the losses, 2e-2 relative L2 on the gradients (tests/test_fmap_gpu.py). */
int hcm_dense_soft_nce_coords_bf16(const float* map1, const float* map2, hcm_strides4 st, int B, int C,
                                   int h, int w, const int64_t* sample_ind, const int64_t* coord_ind,
                                   int coord_w, const int32_t* keep, int S, float temperature, float* out4,
                                   float* gmap1, float* gmap2, void* workspace, size_t workspace_bytes,
                                   hcm_stream_t stream);
/* The same two losses with EXACT fp32 contractions (r06, `--fmap_dtype fp32_exact`): every operand split into three bf16 pieces
 * (hi + mid + lo = x exactly), all nine piece products issued on the bf16 matrix cores with fp32 accumulation -- an fp32 dot
 * product whose only rounding is the accumulator's.  The default entry points above use two pieces and 3 (dense) / 4 (SCL) terms
 * (4e-6..6e-6 of float64 on the gradients, inside the 1e-5 / 1e-4 parity gate); this mode is 3 x their matrix work and exists so
 * that a true-fp32 contraction stays selectable.  Same arguments. */
int hcm_dense_soft_nce_coords_exact(const float* map1, const float* map2, hcm_strides4 st, int B, int C,
                                    int h, int w, const int64_t* sample_ind, const int64_t* coord_ind,
                                    int coord_w, const int32_t* keep, int S, float temperature, float* out4,
                                    float* gmap1, float* gmap2, void* workspace, size_t workspace_bytes,
                                    hcm_stream_t stream);
int hcm_scl_exact(const float* map1, const float* map2, hcm_strides4 st, int B, int C, int h, int w,
                  const int64_t* pix, const int32_t* use_depth, const int32_t* use_rgb, int J,
                  float temperature, float* out1, float* gmap1, float* gmap2, void* workspace,
                  size_t workspace_bytes, hcm_stream_t stream);
int hcm_scl_bf16(const float* map1, const float* map2, hcm_strides4 st, int B, int C, int h, int w,
                 const int64_t* pix, const int32_t* use_depth, const int32_t* use_rgb, int J,
                 float temperature, float* out1, float* gmap1, float* gmap2, void* workspace,
                 size_t workspace_bytes, hcm_stream_t stream);

/* joint_pixels: pix[b,j] = clamp(floor(j2d[b,j,0]/4),0,h-1)*h + clamp(floor(j2d[b,j,1]/4),0,h-1)
 * (contrast_trainer.py:757-761).  joints2d [B, J, 2] fp32. */
int hcm_joint_pixels(const float* joints2d, int BJ, int h, int64_t* pix, hcm_stream_t stream);

/* Row 8 helper (feature-map producer): F.interpolate(x, size=(Ho,Wo), mode='bilinear',
 * align_corners=False) forward on `planes` = N*C contiguous [Hi,Wi] planes -> [Ho,Wo] planes
 * (HRNet fuse layers, networks/official_hrnet/official_hrnet.py:231-236; merge_all_res,
 * networks/build_backbone.py:247-254). */
int hcm_upsample_bilinear2d(const float* in, int planes, int Hi, int Wi, int Ho, int Wo, float* out,
                            hcm_stream_t stream);
/* Same op on channels-last memory: in [N,Hi,Wi,C] -> out [N,Ho,Wo,C]. */
int hcm_upsample_bilinear2d_nhwc(const float* in, int N, int C, int Hi, int Wi, int Ho, int Wo,
                                 float* out, hcm_stream_t stream);
/* Backward of hcm_upsample_bilinear2d in gather form: grad_in [planes, Hi, Wi] = sum over the output pixels
 * whose stencil touches each input pixel, in a fixed order.  Deterministic, no atomics (ATen's
 * upsample_bilinear2d_backward scatters with float atomicAdd); overwrites grad_in. */
int hcm_upsample_bilinear2d_backward(const float* grad_out, int planes, int Hi, int Wi, int Ho, int Wo,
                                     float* grad_in, hcm_stream_t stream);
/* A term of an HRNet fuse layer in one launch (networks/official_hrnet.py:230-251: y = y + F.interpolate(f_ij(x_j));
 * after the last term, relu): out = relu?(acc + upsample(in)).  backward_relu is the backward of the relu form:
 * grad_masked = grad_out * (y > 0) (the gradient of `acc`), grad_in = upsample^T(grad_masked); it refuses planes
 * whose masked gradient does not fit LDS (> 60 KB with the separable form's tables): mask, then the plain backward. */
int hcm_upsample_bilinear2d_add(const float* in, const float* acc, int relu, int planes, int Hi, int Wi, int Ho, int Wo,
                                float* out, hcm_stream_t stream);
int hcm_upsample_bilinear2d_backward_relu(const float* grad_out, const float* y, int planes, int Hi, int Wi, int Ho, int Wo,
                                          float* grad_in, float* grad_masked, hcm_stream_t stream);

/* Row 8, sampled form: bilinear(x)[b, :, pix[b,r]] for one HRNet branch
 * x [B, C, hi, wi] (strides st), evaluated on the finest grid (h0 x w0, align_corners=False), written
 * to out[(b*R + r)*ldo + col0 + c].  Equals merge_all_res (build_backbone.py:247-254) restricted to
 * the sampled pixels; the 1x1 projection then runs on the [B*R, sum C] matrix.  The _grad form
 * scatter-adds (atomics) grad_rows back into gx (same strides as x, caller-zeroed). */
int hcm_sample_rows(const float* x, hcm_strides4 st, int B, int C, int hi, int wi, int h0, int w0,
                    const int64_t* pix, int R, float* out, int ldo, int col0, hcm_stream_t stream);
int hcm_sample_rows_grad(const float* grad_rows, int ldo, int col0, hcm_strides4 st, int B, int C,
                         int hi, int wi, int h0, int w0, const int64_t* pix, int R, float* gx,
                         hcm_stream_t stream);
/* Dense sampling matrix of a coarse branch: S[r, q] (caller-zeroed, [nrows, hi*wi]) += bilinear
 * weight of coarse pixel q for sampled pixel pix[r]; sampling is then bmm(S, x), its backward
 * bmm(S^T, g) -- deterministic library GEMMs instead of atomics. */
int hcm_sampling_matrix(const int64_t* pix, int nrows, int hi, int wi, int h0, int w0, float* S,
                        hcm_stream_t stream);

/* ------------------------------------------------------------------------ *
 * The serial section between the encoders' forward and backward as a few launches (SURVEY 8f-2, 8a rows 8-9;
 * csrc/section.hip).  `hcm_branches` = the four NCHW-contiguous fp32 maps an HRNet returns
 * (networks/official_hrnet/official_hrnet.py:411-454): map[i] is [B, C[i], H[i], W[i]], finest first.
 * ------------------------------------------------------------------------ */
typedef struct {
  const float* map[4];
  int C[4], H[4], W[4];
} hcm_branches;
typedef struct {
  float* map[4];
  int C[4], H[4], W[4];
} hcm_branches_out;
/* ABSENT second encoder (r05; ABI 4): an hcm_branches / hcm_branches_out whose map[0] is NULL.  The HRNetPN model
 * (networks/build_backbone.py:305-514) has ONE HRNet; its second modality is a point-cloud encoder that hands over a pooled
 * feature vector and an already projected [B, F, h, w] map.  With an absent enc2 / g2:
 *   hcm_heads_forward          pools modality 0 only; pooled[1] ([B, Ctot], e.g. [mean over the points | 0]) is an INPUT the
 *                              caller has filled on the same stream (W2 padded to [F, Ctot] accordingly);
 *   hcm_project_rows           writes rows[0] / xs[0] / zero-fills grows[0]; the [1] slices are the caller's
 *                              (rows[1] = a plain gather of the map, hcm_sample_rows; grows[1] zero-filled by the caller);
 *   hcm_project_rows_backward  modality 0's branch gradients and dWp1 / dbp1 only (Wp2, dWp2, dbp2 NULL; dpooled[0] is read);
 *   hcm_project_rows_dw        dWp2 == dbp2 == NULL selects the same. */

/* Heads (networks/build_backbone.py:265-288, networks/util.py:74-80): pooled[m] = cat_i mean_HW(enc_m.map[i])
 * [2, B, Ctot]; mean3 = mean_j feat3[b, j, :] [B, D3] (feat3 [B, J, D3]); ypre[h] = W_h x_h + b_h [3, B, F]
 * (W1, W2 [F, Ctot], W3 [F, D3], row-major like nn.Linear.weight); f[b, h*F + o] = ypre / max(|ypre|, 1e-12),
 * row stride ldf >= 3F; fT [3, B, F] = the same values, one contiguous [B, F] matrix per head (what the bank
 * entry points take).  index [B] int64 or NULL: when given, f[b, 3F], f[b, 3F+1] receive its bit pattern, so that
 * f IS the packed row of the feature/index all-gather (learning/contrast_trainer.py:950-951).  F <= 256. */
int hcm_heads_forward(hcm_branches enc1, hcm_branches enc2, const float* feat3, int B, int J, int Ctot, int D3,
                      int F, const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                      const float* b3, const int64_t* index, float* pooled, float* mean3, float* ypre, float* f,
                      int ldf, float* fT, hcm_stream_t stream);
/* Backward.  gfT [3, B, F] = d loss / d fT (the bank kernel's gx), multiplied by *scale (device scalar, NULL = 1).
 * Outputs: dW1, dW2 [F, Ctot], dW3 [F, D3], db1..3 [F]; dpooled [2, B, Ctot]; gfeat3 [B, J, D3] =
 * (*scale) * gfeat3_joint (the joint loss's own gradient w.r.t. feat3, NULL = 0) + head 3's share dX3[b] / J.
 * dyws: [3, B, F] scratch.  Sums over the batch run in ascending order (deterministic). */
int hcm_heads_backward(const float* gfT, const float* scale, const float* pooled, const float* mean3,
                       const float* ypre, int B, int J, int Ctot, int D3, int F, const float* W1, const float* W2,
                       const float* W3, const float* gfeat3_joint, float* dyws, float* dW1, float* db1, float* dW2,
                       float* db2, float* dW3, float* db3, float* dpooled, float* gfeat3, hcm_stream_t stream);

/* Pixel sampling of _compute_soft_pri3d_loss_accuracy (learning/contrast_trainer.py:671-685) and the joints'
 * pixels (:757-761) in one launch.  depth_mask [B, H, W] fp32 (valid = value > 0; the datasets produce 0/1 masks,
 * datasets/dataset.py:575, :600) is nearest-resized to h x w with torch's index rule; keep[b] = the resized mask is
 * non-empty AND (use_depth == NULL or some use_depth != 0) (:663-665, :677-682); for kept images S pixels are drawn
 * uniformly with replacement from the valid ones: element e = b*S + s uses Philox4x32-10(ctr = {e_lo, e_hi, off_lo,
 * off_hi}, key = seed); k = (r0 * count) >> 32; pixel = the (k+1)-th valid pixel in raster order.  (torch.multinomial
 * on the 0/1 weights draws from the same distribution with torch's own generator; parity tests inject indices.)
 * Outputs: pix [B, S+J] int64 (S sampled pixels, then the J joint pixels clamp(floor(j/4), 0, h-1), joints2d [B, J, 2]
 * fp32 (row, col)); coord [B, S] = the sampled pixels again (contiguous, for the soft target); keep [B] int32.
 * Dropped images get pixel 0.  h == w, h*w*4 <= 150 KiB. */
int hcm_pixel_sample(const float* depth_mask, int B, int H, int W, int h, int w, int S, const int32_t* use_depth,
                     const float* joints2d, int J, uint64_t seed, uint64_t offset, int64_t* pix, int64_t* coord,
                     int32_t* keep, hcm_stream_t stream);

/* merge_all_res (networks/build_backbone.py:247-254) at the sampled pixels, both modalities in one launch:
 * xs[m, b*R + r, :] = [enc_m.map[0][b, :, p] ; bilinear(map[1..3])[b, :, p] ; 1 ; 0...] for p = pix[b, r], row
 * stride ld = hcm_sample_branches_ld(Ctot) (Ctot + the bias column, rounded up to 4).  Also packs the 1x1 projections
 * (:243-245; Wp [F, Ctot], bp [F]) as Wpad [2, F, ld] = [W | b | 0], so that rows = xs Wpad^T is ONE batched GEMM
 * including the bias, and zero-fills grows [2, B*R, F] (the loss kernels accumulate the row gradients into it;
 * NULL = skip). */
int hcm_sample_branches_ld(int Ctot);
int hcm_sample_branches(hcm_branches enc1, hcm_branches enc2, int B, const int64_t* pix, int R, int Ctot, int F,
                        const float* Wp1, const float* bp1, const float* Wp2, const float* bp2, float* xs, float* Wpad,
                        float* grows, hcm_stream_t stream);
/* Backward of hcm_sample_branches + the average pooling of the heads, all eight branch gradients in one launch:
 * g_m.map[i][b, c, q] = dpooled[m, b, coff_i + c] / (H_i W_i) + (*scale) * sum over the rows r of image b whose
 * stencil holds q of weight * dxs[m, b*R + r, coff_i + c]   (dxs [2, B*R, ld] = grows Wpad; rows in ascending
 * order, taps in stencil order: owner computes, deterministic, every element written exactly once -- the maps need
 * no zero-fill).  dpooled NULL = 0.  keep [B] int32 / S (NULL / 0 = none): the first S rows of an image with
 * keep[b] == 0 (the dense samples of a dropped image, all on pixel 0) carry an exactly-zero gradient and are skipped.
 * Also unpacks dWpad [2, F, ld] (= grows^T xs) into dWp [F, Ctot], dbp [F] per
 * modality, times *scale (dWpad NULL = skip). */
int hcm_branch_grad(const float* dxs, const float* dpooled, const float* scale, const int64_t* pix, int R, int B,
                    int Ctot, hcm_branches_out g1, hcm_branches_out g2, const int32_t* keep, int S, const float* dWpad,
                    int F, float* dWp1, float* dbp1, float* dWp2, float* dbp2, hcm_stream_t stream);

/* Row 8 on the matrix cores (r04, csrc/rowproj.hip; replaces hcm_sample_branches + three library GEMMs + hcm_branch_grad
 * in the trainer's path).  Forward: rows[m, b*R + r, :] = [W_m | b_m] applied to the merged, sampled row
 * (merge_all_res + encoder{1,2}_linear at pixel pix[b, r]; networks/build_backbone.py:243-254, :290-300), fp32 MFMA,
 * F == 128.  xs [2, B*R, ld] (ld = hcm_sample_branches_ld(Ctot)) receives the sampled rows [x | 1 | 0] for the weight
 * gradient (NULL = inference, nothing saved); grows [2, B*R, F] is zero-filled (NULL = skip).  Ctot + 1 <= 724
 * (HRNet-w48: 720 channels). */
int hcm_project_rows(hcm_branches enc1, hcm_branches enc2, int B, const int64_t* pix, int R, int Ctot, int F,
                     const float* Wp1, const float* bp1, const float* Wp2, const float* bp2, float* xs, float* rows,
                     float* grows, hcm_stream_t stream);
/* Same, reading the branch maps it gathers from global memory CHANNELS-LAST (r06; north_star: "coalesced HBM reads of the
 * modality feature maps", SURVEY 8f-1: "channels-last so each gather is one line"; producer networks/build_backbone.py:247-254,
 * :290-300).  The HRNet writes NCHW, where the C_i values of a sampled pixel lie H_i W_i floats apart; with a workspace of
 * hcm_project_rows_nhwc_floats(...) floats the entry point first writes [B, H_i W_i, C_i] copies of the branches that are not
 * staged whole in LDS (one launch, 64-pixel tiles through LDS) and the row kernel reads ONE contiguous run of C_i floats per
 * stencil tap.  nhwc_ws NULL = hcm_project_rows.  Same results bit for bit (the same values enter the same sums). */
size_t hcm_project_rows_nhwc_floats(hcm_branches enc1, hcm_branches enc2, int B, int Ctot);
int hcm_project_rows_cl(hcm_branches enc1, hcm_branches enc2, int B, const int64_t* pix, int R, int Ctot, int F,
                        const float* Wp1, const float* bp1, const float* Wp2, const float* bp2, float* xs, float* rows,
                        float* grows, float* nhwc_ws, size_t nhwc_floats, hcm_stream_t stream);
/* d[W | b] = (*scale) * grows^T xs per modality -> dWp [F, Ctot], dbp [F]: MFMA partials over 128 row chunks, summed in
 * chunk order (deterministic).  workspace: hcm_project_rows_dw_workspace_bytes(B, Ctot). */
size_t hcm_project_rows_dw_workspace_bytes(int B, int Ctot);
int hcm_project_rows_dw(const float* grows, const float* xs, const float* scale, int B, int R, int Ctot, int F,
                        float* dWp1, float* dbp1, float* dWp2, float* dbp2, void* workspace, size_t workspace_bytes,
                        hcm_stream_t stream);
/* Backward of hcm_project_rows + the average pooling of the heads: the weight / bias gradients (above; dWp1 == NULL
 * skips them) and all eight branch gradients,
 *   g_m.map[i][b, c, q] = dpooled[m, b, coff_i + c] / (H_i W_i)
 *                         + (*scale) * sum_f W_m[f, coff_i + c] * ( sum over the stencil entries (r, tap) of image b that
 *                           land on pixel q of weight(r, tap) * grows[m, b*R + r, f] ),
 * entries in ascending (row, tap) order, f ascending: owner computes, no atomics, every element written exactly once.
 * The finest branch (a gather: one entry of weight 1 per row) is evaluated in the reference's own order since r06,
 *   sum over the entries of pixel q of weight * ( sum_f W_m[f, c] * grows[m, b*R + r, f] )
 * (the projected rows first, on the matrix cores; then offsets / entries / rows / one store pass per 256 pixels), whenever
 * H_0 W_0 is a multiple of 4; a workgroup whose pixel range holds more entries than its LDS stages takes the order above.
 * keep / S as in hcm_branch_grad.  Launches: the stencil plan (entries sorted by pixel, shared by both modalities), the
 * weight-gradient partials + reduction, the finest branch's projected rows, the branch tiles.  workspace:
 * hcm_project_rows_backward_workspace_bytes(B, R, Ctot, g1).  H_i W_i * 4R < 2^32, R <= 9600, C_i <= 384. */
size_t hcm_project_rows_backward_workspace_bytes(int B, int R, int Ctot, hcm_branches_out g1);
int hcm_project_rows_backward(const float* grows, const float* xs, const float* Wp1, const float* Wp2,
                              const float* dpooled, const float* scale, const int64_t* pix, int R, int B, int Ctot,
                              int F, hcm_branches_out g1, hcm_branches_out g2, const int32_t* keep, int S, float* dWp1,
                              float* dbp1, float* dWp2, float* dbp2, void* workspace, size_t workspace_bytes,
                              hcm_stream_t stream);

/* total[0] = sum(losses6) (+ meters9[0] + [1] + [4] + [5] + [8] when meters9 != NULL): the objective of
 * learning/contrast_trainer.py:980 (:594 for stage 1) from the kernels' own outputs, in one tiny launch. */
int hcm_section_total(const float* losses6, const float* meters9, float* total, hcm_stream_t stream);

/* ------------------------------------------------------------------------ *
 * SemGCN layer: SemGraphConv (networks/SGCN/sem_graph_conv.py:34-48) [+ BatchNorm1d +
 * ReLU, networks/SGCN/sem_gcn.py:8-28] after the library GEMM H = X [W0 | W1]  ([B*J, 2C]).
 * Graph: edges in the row-major order of the reference's `adj[self.m]` as CSR (row_ptr [J+1],
 * col_idx [E], edge_row [E]) and CSC (csc_ptr [J+1], csc_edge [E]); e [E] learned edge logits.
 * forward : A = row-softmax(e); Y = A_diag (.) H0 + A_off H1 + bias; out = relu(bn(Y)) (batch
 *           statistics when training, running statistics updated in place with `momentum`);
 *           saves xhat [B*J, C], invstd [C], A_out [E] for the backward.
 * backward: dH [B*J, 2C] (for dX = dH Wcat^T and dWcat = X^T dH), dgamma/dbeta/dbias [C], de [E].
 * J <= 32, C in {64, 128}, E <= 256.  One workgroup per batch element; the BatchNorm1d statistics split a
 * direction into two (forward) / three (backward) launches whose per-sample partial sums live in
 * `workspace` (caller-owned, hcm_sgc_workspace_floats(B, J, C, E) floats, contents undefined on entry;
 * the backward call may use a different buffer than the forward call).  Deterministic: partials are
 * merged in sample order, no atomics.
 * ------------------------------------------------------------------------ */
size_t hcm_sgc_workspace_floats(int B, int J, int C, int E);
int hcm_sgc_forward(const float* H, const float* e, const int* row_ptr, const int* col_idx,
                    const int* csc_ptr, const int* csc_edge, const int* edge_row, const float* bias,
                    const float* gamma, const float* beta, float* running_mean, float* running_var,
                    int B, int J, int C, int E, int has_bn, int relu, int training, float momentum,
                    float eps, float* out, float* xhat, float* invstd, float* A_out, float* workspace,
                    hcm_stream_t stream);
int hcm_sgc_backward(const float* dOut, const float* out, const float* xhat, const float* invstd,
                     const float* gamma, const float* A, const int* row_ptr, const int* col_idx,
                     const int* csc_ptr, const int* csc_edge, const int* edge_row, const float* H, int B,
                     int J, int C, int E, int has_bn, int relu, int training, float* dH, float* dgamma,
                     float* dbeta, float* dbias, float* de, float* workspace, hcm_stream_t stream);

/* ------------------------------------------------------------------------ *
 * Encoder normalisation: training-mode BatchNorm2d [+ residual add] [+ ReLU] on fp32 NCHW maps
 * (torch.nn.BatchNorm2d + `out += residual` + ReLU in networks/official_hrnet/official_hrnet.py:40-105,
 * :161-207, :287-347).  x, residual, y, dy, dz, dx: [N, C, HW] contiguous, HW % 4 == 0.
 * `stats` / `gstats`: caller-owned, hcm_bn_act_stats_floats(N, C, HW) floats each;
 *   stats  = [mean C][invstd C][scratch]   (written by forward, read by backward)
 *   gstats = [dgamma C][dbeta C][scratch]  (written by backward)
 * forward : y = relu?(gamma * (x - mean) * invstd + beta + residual?), biased batch variance;
 *           running_mean/var (nullable pair) <- (1 - momentum) * running + momentum * {mean, unbiased var}.
 * backward: dz = (dy + dy2) * [y > 0]; dy2 is an optional second gradient of the same tensor (NULL: none;
 *           saves the add kernel where a residual block's input collects its two gradients).  dz is
 *           also the gradient of `residual`; with neither relu nor dy2 that gradient is dy itself and
 *           y, dz may be NULL.  dx (NULL to skip), dgamma, dbeta.
 * Deterministic (fixed-order partial sums, no atomics).
 * ------------------------------------------------------------------------ */
size_t hcm_bn_act_stats_floats(int N, int C, int HW);
int hcm_bn_act_forward(const float* x, const float* residual, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, float momentum, float eps, int relu,
                       int N, int C, int HW, float* y, float* stats, hcm_stream_t stream);
int hcm_bn_act_backward(const float* dy, const float* dy2, const float* x, const float* y, const float* gamma,
                        const float* stats, int relu, int N, int C, int HW, float* dz, float* dx,
                        float* gstats, hcm_stream_t stream);
/* hcm_bn_act_forward whose statistics pass was done by the producer of x: partial_sums [2 * nslots + 1][C] = per-slot
 * sums of (x - k) and (x - k)^2 and, in the last row, the shift k[C] they were taken about
 * (hcm_conv3x3_forward_stats), added in a fixed order; mean = k + S1/M, var = S2/M - (S1/M)^2.  Mid-size and large maps only
 * (N*HW per channel > 8192): hipErrorInvalidValue otherwise.  stats as for hcm_bn_act_forward. */
int hcm_bn_act_forward_pre(const float* x, const float* residual, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float momentum, float eps, int relu,
                           int N, int C, int HW, float* y, float* stats, const float* partial_sums, int nslots,
                           hcm_stream_t stream);
/* 1x1 convolutions of the PointNet++ shared MLPs on ball tensors (r05; reference: networks/pointnet2/pytorch_utils.py:5-33,
 * nn.Conv2d(kernel_size=1, bias=False)).  x [N, C, P], w [K, C] (nn.Conv2d.weight viewed as a matrix), z [N, K, P], P = the
 * positions of one image (npoint * nsample), fp32, contiguous.  forward: z = w x per image; backward_data: dx = w^T dz.
 * Matrix-core products on the tensors as they lie (LDS only for the weights, no layout change).  Arithmetic (r06, ABI 6):
 * layers whose fp32 MFMA floor reaches their HBM floor (reduction length R % 32 == 0 and R * output channels >= 4096: 64 -> 128
 * channels and up) run on the bf16 matrix cores with split operands -- x = hi + mid, three terms, fp32 accumulation, 4.4e-6 of
 * float64 as a vector (max 7e-6 of the largest element), the scheme of hcm_dense_soft_nce -- the narrower ones as exact fp32 MFMA
 * (an fmaf chain per output element).  hcm_conv1x1_set_arith(1) makes every layer exact fp32, (0) restores the default; it
 * returns the previous mode (-1: bad argument); process-wide, what `--fmap_dtype fp32_exact` sets.  The *_exact entry points
 * are exact fp32 whatever the mode: the source-point projection of the implicit first layer (hcm_ball_project_* below) uses
 * them -- its output is gathered and normalised over a ball, and the first-layer accuracy contract of
 * tests/test_pointnet2_gpu.py::test_first_layer_on_the_implicit_grouped_tensor is an fp32 one.
 * hcm_conv1x1_supported: C % 4 == 0, K % 4 == 0, P % 64 == 0; anything else returns hipErrorInvalidValue and the caller keeps
 * its library path.
 * hcm_conv1x1_ball_wgrad: dw [K, C] = sum over n, p of dy[n][k][p] x[n][c][p] (same argument order as hcm_conv1x1_wgrad below:
 * H * W = P); fp32 MFMA over the positions as they lie, per-workgroup partials in the workspace, summed in fixed order
 * (deterministic).  Needs K % 16 == 0, C % 16 == 0, P % 256 == 0: _workspace_bytes returns 0 for anything else and the
 * caller keeps hcm_conv1x1_wgrad / its library path. */
int hcm_conv1x1_set_arith(int mode);
int hcm_conv1x1_supported(int C, int K, int P);
int hcm_conv1x1_forward(const float* x, const float* w, float* z, int N, int C, int K, int P, hcm_stream_t stream);
int hcm_conv1x1_backward_data(const float* dz, const float* w, float* dx, int N, int C, int K, int P, hcm_stream_t stream);
size_t hcm_conv1x1_ball_wgrad_workspace_bytes(int N, int C, int K, int H, int W);
int hcm_conv1x1_ball_wgrad(const float* x, const float* dy, int N, int C, int K, int H, int W, float* dw,
                           void* workspace, size_t workspace_bytes, hcm_stream_t stream);
int hcm_conv1x1_forward_exact(const float* x, const float* w, float* z, int N, int C, int K, int P, hcm_stream_t stream);
int hcm_conv1x1_backward_data_exact(const float* dz, const float* w, float* dx, int N, int C, int K, int P, hcm_stream_t stream);
int hcm_conv1x1_ball_wgrad_exact(const float* x, const float* dy, int N, int C, int K, int H, int W, float* dw,
                                 void* workspace, size_t workspace_bytes, hcm_stream_t stream);

/* BatchNorm2d (training mode) + ReLU + max over the ball in one piece (r05): the last layer of a PointNet++ SharedMLP
 * followed by F.max_pool2d(y, [1, nsample]) (networks/pointnet2/pointnet2_modules.py:44-55, pytorch_utils.py:5-33 of the
 * reference).  z [N, C, np, ns] fp32 contiguous (the 1x1 convolution's output); y = relu(bn(z)) is never written:
 *   forward : out [N, C, np] = max_j y, arg [N, C, np] = the FIRST j that attains it (ATen's rule, evaluated on y, so ties
 *             among clamped zeros resolve to the first zero), zsel [N, C, np] = z at arg; stats / running statistics as
 *             hcm_bn_act_forward.  stats, gstats: hcm_bn_relu_ballmax_stats_floats(N, C, np, ns) floats each.
 *   backward: dout [N, C, np] -> dz [N, C, np, ns] (the gradient w.r.t. z, dense: the batch statistics spread it), gstats =
 *             [dgamma C][dbeta C][scratch].  The sums of the normalisation backward come from the [N, C, np] tensors alone.
 * ns in {4, 8, 16, 32, 64}, np % 4 == 0; anything else hipErrorInvalidValue (the caller keeps the three-op path).
 * Deterministic (fixed-order partial sums, no atomics). */
size_t hcm_bn_relu_ballmax_stats_floats(int N, int C, int np, int ns);
int hcm_bn_relu_ballmax_forward(const float* z, const float* gamma, const float* beta, float* running_mean,
                                float* running_var, float momentum, float eps, int N, int C, int np, int ns, float* out,
                                int32_t* arg, float* zsel, float* stats, hcm_stream_t stream);
int hcm_bn_relu_ballmax_backward(const float* dout, const float* out, const int32_t* arg, const float* zsel,
                                 const float* z, const float* gamma, const float* stats, int N, int C, int np, int ns,
                                 float* dz, float* gstats, hcm_stream_t stream);
/* First layer of a PointNet++ SharedMLP WITHOUT the grouped tensor (r05; coordinate half restated in r06, ABI 5).
 * QueryAndGroup + the first 1x1 convolution (networks/pointnet2/pointnet2_utils.py:231-268, pytorch_utils.py:5-33 of the
 * reference) compute, for ball i of image b and its j-th member n = idx[b, i, j],
 *     z[b, :, i, j] = W [xyz_n - centre_i ; features_n] = W_xyz (xyz_n - centre_i) + W_f features_n
 *                   = Wxyz D[b, :, i, j] + P[b, :, n]
 * with D [B, 3, np, ns] = the reference's grouped_xyz (relative offsets: geometry, no features) and P = W_f features
 * [B, C, N] (hcm_conv1x1_forward, the caller's).  ABI 4 also commuted the coordinate half (P' = W [xyz ; features],
 * Q = W_xyz centre, z = P'[n] - Q[i]): a difference of two O(|xyz|) numbers worth O(radius), 40 x the reference's round-off on
 * 2.5 cm balls.  These entry points take z as that implicit tensor:
 *   forward : y [B, C, np, ns] = relu?(batchnorm(z)) with batch statistics over all B * np * ns members (running statistics
 *             updated like hcm_bn_act_forward); stats: hcm_ball_project_stats_floats(B, C, np, ns) floats.
 *   backward: dy [B, C, np, ns] (and y, for the ReLU mask) -> dz [B, C, np, ns] = d loss / d z (the caller scatters it into
 *             dP[b, c, idx], hcm_scatter_add_planned), dWxyz [C, 3] = sum_{b,i,j} dz D, gstats = [dgamma C][dbeta C][scratch]
 *             (same size as stats).  No gradient w.r.t. the coordinates (the reference's clouds carry none either:
 *             networks/build_backbone.py:379-445 builds them under no autograd-tracked input).
 * P may be NULL (no point features: the first SA level, z = Wxyz D; idx and N are then unused, and so is dz -- nothing
 * consumes it -- which may be NULL: the backward is then ONE pass over (dy, y, D), the forward reads D once per pass for all
 * channels, C in {16, 32}).  Wxyz [C, 3] row-major.
 * idx [B, np, ns] int32 in [0, N); ns in {4, 8, 16, 32, 64}, np % 4 == 0.  Deterministic. */
size_t hcm_ball_project_stats_floats(int B, int C, int np, int ns);
int hcm_ball_project_forward(const float* P, const float* D, const float* Wxyz, const int32_t* idx, const float* gamma,
                             const float* beta, float* running_mean, float* running_var, float momentum, float eps, int relu,
                             int B, int C, int N, int np, int ns, float* y, float* stats, hcm_stream_t stream);
int hcm_ball_project_backward(const float* dy, const float* y, const float* P, const float* D, const float* Wxyz,
                              const int32_t* idx, const float* gamma, const float* stats, int relu, int B, int C, int N,
                              int np, int ns, float* dz, float* dWxyz, float* gstats, hcm_stream_t stream);
/* Same, with the partial-sum scratch in its own buffer (hcm_bn_act_stats_floats - 2C floats): `gstats`
 * is then exactly [dgamma C][dbeta C], so a caller can lay every parameter gradient of a network out in
 * one dense buffer (what the encoder runtime hands to RCCL in place, csrc/torch_glue). */
int hcm_bn_act_backward_ws(const float* dy, const float* dy2, const float* x, const float* y, const float* gamma,
                           const float* stats, int relu, int N, int C, int HW, float* dz, float* dx,
                           float* gstats, float* scratch, hcm_stream_t stream);

/* ------------------------------------------------------------------------ *
 * Weight gradient of a 3x3 / stride 1 / pad 1 / bias-free convolution (the BasicBlock convolutions of
 * networks/official_hrnet/official_hrnet.py:40-70; torch.nn.Conv2d backward w.r.t. weight):
 *   dw[k][c][r][s] = sum_{n,y,x} dy[n][k][y][x] * x[n][c][y+r-1][x+s-1]
 * x [N,C,H,W], dy [N,K,H,W], dw [K,C,3,3] fp32 contiguous, W % 4 == 0.  workspace: caller-owned,
 * hcm_conv3x3_wgrad_workspace_bytes(...) bytes (per-workgroup partial sums; 0 = unsupported shape).
 * Deterministic (fixed-order reduction, no atomics).  Two launches.
 * ------------------------------------------------------------------------ */
size_t hcm_conv3x3_wgrad_workspace_bytes(int N, int C, int K, int H, int W);
int hcm_conv3x3_wgrad(const float* x, const float* dy, int N, int C, int K, int H, int W, float* dw,
                      void* workspace, size_t workspace_bytes, hcm_stream_t stream);
/* The same for a 1x1 / stride 1 / pad 0 convolution (fuse-layer and bottleneck 1x1s): dw [K,C,1,1]. */
size_t hcm_conv1x1_wgrad_workspace_bytes(int N, int C, int K, int H, int W);
int hcm_conv1x1_wgrad(const float* x, const float* dy, int N, int C, int K, int H, int W, float* dw,
                      void* workspace, size_t workspace_bytes, hcm_stream_t stream);
/* 3x3 / stride 2 / pad 1 (the down-sampling convolutions of the fuse layers and transitions):
 * x [N,C,2Ho,2Wo], dy [N,K,Ho,Wo], dw [K,C,3,3]; Ho, Wo are the arguments. */
size_t hcm_conv3x3s2_wgrad_workspace_bytes(int N, int C, int K, int Ho, int Wo);
int hcm_conv3x3s2_wgrad(const float* x, const float* dy, int N, int C, int K, int Ho, int Wo, float* dw,
                        void* workspace, size_t workspace_bytes, hcm_stream_t stream);
/* The two halves of the calls above, separately: hcm_conv_wgrad_partial (kind 3 = 3x3 stride 1, 1 = 1x1, 2 = 3x3
 * stride 2; same arguments, no dw) leaves the per-workgroup partial sums in `workspace` and returns their count in
 * *chunks; hcm_wgrad_reduce_batch sums the partials of MANY layers, each in the same fixed order as the single
 * calls, in one launch per 64 layers -- a caller walking a network backwards parks the partials of a stretch of
 * layers and reduces them together (dW feeds nothing but the optimizer). */
typedef struct { const float* partial; float* dw; int total; int chunks; } hcm_wgrad_reduce_desc;   /* total = K*C*taps */
int hcm_conv_wgrad_partial(int kind, const float* x, const float* dy, int N, int C, int K, int H, int W, void* workspace,
                           size_t workspace_bytes, int* chunks, hcm_stream_t stream);
int hcm_wgrad_reduce_batch(const hcm_wgrad_reduce_desc* descs, int n, hcm_stream_t stream);

/* ------------------------------------------------------------------------ *
 * Rows 12-17 -- PointNet++ ops.  Same argument order, ownership and layouts as the
 * reference's C launcher layer (networks/pointnet2/src/<name>_gpu.h), which the pybind
 * module `pointnet2_cuda` (src/pointnet2_api.cpp:10-24) wraps: the caller allocates and
 * pre-initialises every output (idx zero-filled, temp filled with 1e10, grads zero-filled;
 * networks/pointnet2/pointnet2_utils.py:25-26,55,67,94-95,128,146,172,190,218).
 * Launch failures are RETURNED (the reference calls exit(-1)).
 * ------------------------------------------------------------------------ */
/* src/sampling_gpu.h : furthest_point_sampling_kernel_launcher */
int hcm_furthest_point_sampling(int b, int n, int m, const float* dataset, float* temp, int* idxs,
                                hcm_stream_t stream);
/* src/ball_query_gpu.h:12-13 : ball_query_kernel_launcher_fast */
int hcm_ball_query(int b, int n, int m, float radius, int nsample, const float* new_xyz,
                   const float* xyz, int* idx, hcm_stream_t stream);
/* src/group_points_gpu.h : group_points_kernel_launcher_fast / group_points_grad_kernel_launcher_fast */
int hcm_group_points(int b, int c, int n, int npoints, int nsample, const float* points,
                     const int* idx, float* out, hcm_stream_t stream);
int hcm_group_points_grad(int b, int c, int n, int npoints, int nsample, const float* grad_out,
                          const int* idx, float* grad_points, hcm_stream_t stream);
/* src/sampling_gpu.h : gather_points_kernel_launcher_fast / gather_points_grad_kernel_launcher_fast */
int hcm_gather_points(int b, int c, int n, int npoints, const float* points, const int* idx,
                      float* out, hcm_stream_t stream);
int hcm_gather_points_grad(int b, int c, int n, int npoints, const float* grad_out, const int* idx,
                           float* grad_points, hcm_stream_t stream);
/* src/interpolate_gpu.h : three_nn_kernel_launcher_fast (dist2 = SQUARED distances) */
int hcm_three_nn(int b, int n, int m, const float* unknown, const float* known, float* dist2,
                 int* idx, hcm_stream_t stream);
/* src/interpolate_gpu.h : three_interpolate_kernel_launcher_fast / _grad_ */
int hcm_three_interpolate(int b, int c, int m, int n, const float* points, const int* idx,
                          const float* weight, float* out, hcm_stream_t stream);
int hcm_three_interpolate_grad(int b, int c, int n, int m, const float* grad_out, const int* idx,
                               const float* weight, float* grad_points, hcm_stream_t stream);

/* Arithmetic contract of the four ops above that do fp32 arithmetic.  The reference evaluates
 *   d = a*a + b*b + c*c   (src/sampling_gpu.cu:131, src/ball_query_gpu.cu:33, src/interpolate_gpu.cu:39)
 *   o = w0*p0 + w1*p1 + w2*p2   (src/interpolate_gpu.cu:96)
 * and is built by `nvcc -O2` (networks/pointnet2/setup.py:20; --fmad=true is nvcc's default), which
 * contracts the source form to  fma(c, c, fma(a, a, b*b)):  HCM_CONTRACT_FMA, what the plain entry
 * points above compute.  HCM_CONTRACT_IEEE is the un-fused ((a*a + b*b) + c*c) of an --fmad=false /
 * CPU build.  Indices can differ between the two on near-ties (duplicate points are routine:
 * networks/build_backbone.py:427 samples with replacement); each is bit-exact against the oracle
 * in the same mode. */
#define HCM_CONTRACT_IEEE 0
#define HCM_CONTRACT_FMA 1
int hcm_furthest_point_sampling_contract(int b, int n, int m, const float* dataset, float* temp, int* idxs,
                                         int contract, hcm_stream_t stream);
int hcm_ball_query_contract(int b, int n, int m, float radius, int nsample, const float* new_xyz,
                            const float* xyz, int* idx, int contract, hcm_stream_t stream);
int hcm_three_nn_contract(int b, int n, int m, const float* unknown, const float* known, float* dist2,
                          int* idx, int contract, hcm_stream_t stream);
int hcm_three_interpolate_contract(int b, int c, int m, int n, const float* points, const int* idx,
                                   const float* weight, float* out, int contract, hcm_stream_t stream);

/* LDS-resident backward of the three scatter-add ops above (they share one algebraic form):
 *   grad_points[b, c, j] = sum_{q : idx[b, q] == j} coef[b, q] * grad_out[b, c, q / div]
 * group_points_grad : idx [B, npoints*nsample], coef NULL, div 1, Qsrc = npoints*nsample
 * gather_points_grad: idx [B, npoints],         coef NULL, div 1, Qsrc = npoints
 * three_interpolate_grad: idx [B, n*3], coef = weight [B, n*3], div 3, Qsrc = n
 * One workgroup keeps grad_points[b, c0:c0+CBL, :] in LDS (m * 4 B <= 140 KB, i.e. every PointNet++
 * level), streams grad_out / idx / coef once (coalesced), accumulates with LDS float atomics and
 * OVERWRITES grad_points [B, C, m] (no zero-fill needed).  3-10x the reference-style global-atomic
 * kernels on MI355X (tools/bench_pointnet2.py).  Returns hipErrorInvalidConfiguration when m is too
 * large for LDS (then use the atomic kernels). */
int hcm_scatter_add_lds(const float* grad_out, const float* coef, const int* idx, int B, int C,
                        int Qsrc, int Q, int m, int div, float* grad_points, hcm_stream_t stream);

/* Deterministic form of the same backward (csrc/scatter.hip; r03): no atomics of any kind, every sum in an order fixed
 * by idx alone (bit-reproducible), and contributions that pile onto one target (an empty-mask image: all pixels
 * interpolate from points 0, 1, 2) are summed across the wave instead of serialised.
 *   hcm_scatter_plan: once per index tensor idx [B, Qsrc * div] (values in [0, m), m <= 65535; div = 1 or 3) and its
 *   weights coef [B, Qsrc * div] (NULL = 1): plan / plan_coef [hcm_scatter_plan_elems(B, Qsrc, div, m)] (int32 /
 *   float; 0 elements = unsupported shape) receive, in the order the kernel streams them ([b][step][slot < div][lane]
 *   [4 sources]; lane l owns sources [l R, (l+1) R), R = ceil(Qsrc / 64) rounded up to 4; the div pairs of a source
 *   ordered by target): target | class << 16 | any-flush << 20 -- class 5 = the lane's next source names the same
 *   target in this slot (the run is summed in registers), 0-3 = the run ends here and this is the k-th flush to that
 *   target among the 64 lanes, 4 = more than four such flushes (summed across the wave) -- and the weights.
 *   plan_coef NULL iff coef NULL.
 *   hcm_scatter_add_planned: grad_points [B, C, m] (overwritten, needs no zero-fill) from grad_out [B, C, Qsrc] and the
 *   plan; contribution q = src * div + t reads grad_out[b, c, src].  hipErrorInvalidConfiguration when m floats do not
 *   fit LDS (then use the atomic kernels). */
size_t hcm_scatter_plan_elems(int B, int Qsrc, int div, int m);
int hcm_scatter_plan(const int* idx, const float* coef, int B, int Qsrc, int div, int m, int* plan, float* plan_coef,
                     hcm_stream_t stream);
int hcm_scatter_add_planned(const float* grad_out, const int* plan, const float* plan_coef, int B, int C, int Qsrc, int m,
                            int div, float* grad_points, hcm_stream_t stream);

/* Max over the ball (F.max_pool2d(y, [1, nsample]) in PointnetSAModuleMSG.forward,
 * networks/pointnet2/pointnet2_modules.py:60-63): x [rows, ns] fp32 contiguous (rows = B*C*npoint) ->
 * y [rows], arg [rows] (first index of the maximum, ATen's tie rule); backward writes
 * dx[r][j] = dy[r] if j == arg[r] else 0 for every element (no zero-fill needed). */
int hcm_rowmax_forward(const float* x, long long rows, int ns, float* y, int* arg, hcm_stream_t stream);
int hcm_rowmax_backward(const float* dy, const int* arg, long long rows, int ns, float* dx, hcm_stream_t stream);

/* ------------------------------------------------------------------------ *
 * Measurement helper: launches `reps` back-to-back hcm_bank_nce_fused passes bracketed by
 * hipEvents on `stream` and returns the mean milliseconds per pass (host float).
 * ------------------------------------------------------------------------ */
int hcm_bank_nce_fused_timed(const float* bank1, const float* bank2, const float* bank3, int64_t n,
                             const int64_t* idx, const float* x1, const float* x2, const float* x3,
                             const int32_t* use_depth, const int32_t* use_rgb,
                             int B, int K1, int D, float T,
                             float* losses6, float* accs6, float* gx1, float* gx2, float* gx3,
                             void* workspace, size_t workspace_bytes, hcm_stream_t stream,
                             int reps, float* ms_per_pass_host);
int hcm_bank_nce_fused_timed_bf16(const uint16_t* bank1, const uint16_t* bank2, const uint16_t* bank3, int64_t n,
                                  const int64_t* idx, const float* x1, const float* x2, const float* x3,
                                  const int32_t* use_depth, const int32_t* use_rgb,
                                  int B, int K1, int D, float T,
                                  float* losses6, float* accs6, float* gx1, float* gx2, float* gx3,
                                  void* workspace, size_t workspace_bytes, hcm_stream_t stream,
                                  int reps, float* ms_per_pass_host);

/* In-library timing of selected kernels (bench.py's `roofline` objects): while enabled, every launch of
 * a tagged kernel is bracketed by hipEvents on its own stream.  hcm_prof_read_tag synchronises those
 * events and returns their summed duration and count (host pointers); hcm_prof_read = tag
 * HCM_PROF_BANK_PASS.  hcm_prof_enable(x) also clears what was recorded.  BENCH-ONLY: this is the
 * library's only process-global state (mutex-protected); leave it off in production. */
#define HCM_PROF_BANK_PASS 0    /* bank_pass_kernel of hcm_bank_nce_fused                    */
#define HCM_PROF_DENSE_STATS 1  /* strip_kernel<Dense, stats> of hcm_dense_soft_nce           */
#define HCM_PROF_DENSE_GRAD 2   /* strip_kernel<Dense, grad>                                  */
#define HCM_PROF_SCL_STATS 3    /* strip_kernel<Scl, stats> (+ its chunk merge) of hcm_scl    */
#define HCM_PROF_SCL_GRAD 4     /* strip_kernel<Scl, grad> (+ its chunk merge)                */
#define HCM_PROF_SGC_FWD 5      /* the kernels of hcm_sgc_forward                             */
#define HCM_PROF_SGC_BWD 6      /* the kernels of hcm_sgc_backward                            */
#define HCM_PROF_ROW8_FWD 7     /* project_rows_kernel of hcm_project_rows                     */
#define HCM_PROF_ROW8_DW 8      /* proj_dw_partial + proj_dw_reduce of hcm_project_rows_dw     */
#define HCM_PROF_ROW8_BWD 9     /* branch_grad_t_kernel of hcm_project_rows_backward           */
#define HCM_PROF_JOINT 10       /* the kernels of hcm_joint_nce                                */
/* BASELINE config 4 (HRNetPN): the PointNet++ kernels, launched at many shapes per step -- each launch adds its own
 * algorithmic work to the tag (hcm_prof_read_work), so sum(work) / sum(time) is ONE achieved rate per kernel (r06). */
#define HCM_PROF_CONV1X1_FWD 11   /* conv1x1_{split,rows}_kernel of hcm_conv1x1_forward: bytes 4 N (C + K) P (r06; r05: flops) */
#define HCM_PROF_CONV1X1_DX 12    /* the same kernels of hcm_conv1x1_backward_data: bytes 4 N (C + K) P              */
#define HCM_PROF_CONV1X1_DW 13    /* wgrad1x1_ball_kernel + reduce of hcm_conv1x1_ball_wgrad: bytes 4 N (C + K) P    */
#define HCM_PROF_BALL_FWD 14      /* ball_stats + ball_apply of hcm_ball_project_forward: bytes                      */
#define HCM_PROF_BALL_BWD 15      /* ball_bwd_reduce + ball_bwd_apply (+ merge) of hcm_ball_project_backward: bytes  */
#define HCM_PROF_BALLMAX_FWD 16   /* bn_stats + bn_relu_ballmax of hcm_bn_relu_ballmax_forward: bytes                */
#define HCM_PROF_BALLMAX_BWD 17   /* ballmax_bwd_reduce + ballmax_bwd_apply of hcm_bn_relu_ballmax_backward: bytes   */
#define HCM_PROF_FPS 18           /* fps kernels of hcm_furthest_point_sampling*: distance evaluations b m n          */
#define HCM_PROF_THREE_NN 19      /* three_nn kernels of hcm_three_nn*: distance evaluations b n m                    */
#define HCM_PROF_BALL_QUERY 20    /* ball_query kernel of hcm_ball_query*: distance evaluations b m n (upper bound)   */
#define HCM_PROF_ROW8_NHWC 21     /* nchw_to_nhwc_kernel of hcm_project_rows_cl: bytes (read + written)              */
#define HCM_PROF_NTAGS 24
int hcm_prof_enable(int enable);
int hcm_prof_read(double* total_ms_host, int64_t* launches_host);
int hcm_prof_read_tag(int tag, double* total_ms_host, int64_t* launches_host);
/* the work the tag's launches added since hcm_prof_enable(1) (0 for tags whose launchers state none) */
int hcm_prof_read_work(int tag, double* work_host);

/* Direct 3x3 convolution, stride 1, pad 1, fp32 NCHW, for the BasicBlocks of the two high-resolution HRNet
 * branches (reference: networks/official_hrnet.py:40-70 `conv3x3` -> nn.Conv2d(bias=False); replaces the
 * cudnn/MIOpen call behind F.conv2d and its data gradient).  x [N,C,H,W], w [K,C,3,3], y [N,K,H,W];
 * backward_data: dy [N,K,H,W] -> dx [N,C,H,W] = conv of dy with the transposed, flipped filter.
 * hcm_conv3x3_supported: 1 for the shapes with a kernel instance (C == K in 17..20 on 64-wide maps, 33..36 on
 * 32-wide maps, H % 4 == 0); anything else returns hipErrorInvalidValue and the caller keeps its library path. */
int hcm_conv3x3_supported(int C, int K, int H, int W);
int hcm_conv3x3_forward(const float* x, const float* w, float* y, int N, int C, int K, int H, int W, hcm_stream_t stream);
int hcm_conv3x3_backward_data(const float* dy, const float* w, float* dx, int N, int C, int K, int H, int W,
                              hcm_stream_t stream);
/* The forward with the statistics pass of the BatchNorm that follows it folded into its epilogue: partial_sums
 * [2 * hcm_conv3x3_stats_slots(N, H) + 1][K] receives, per workgroup (slot), the sums of (y - shift) and of
 * (y - shift)^2 over the slot's pixels for every output channel, and in its last row the shift itself.  shift [K]:
 * any estimate of the channel means (the BatchNorm's running mean), which keeps the variance from being a difference
 * of two large sums when |mean| >> std; NULL = 0.  hcm_bn_act_forward_pre consumes the buffer. */
int hcm_conv3x3_stats_slots(int N, int H);
int hcm_conv3x3_forward_stats(const float* x, const float* w, float* y, int N, int C, int K, int H, int W, const float* shift,
                              float* partial_sums, hcm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HCMOCO_HIP_H */
