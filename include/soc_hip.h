/*
 * soc_hip.h -- C ABI of libsoc_hip.so: the hand-written gfx950 (MI355X / CDNA4) kernels of
 * SOC's per-clip inference hot path (SURVEY.md section 8, rows a7 / a14 / a11+a16 / a19, plus
 * the fused add+LayerNorm of the "next" row 8f-1).
 *
 * Conventions (all entry points):
 *   - plain C, no torch types; every pointer is a DEVICE pointer unless stated otherwise;
 *   - inputs are borrowed, outputs are caller-allocated and fully overwritten;
 *   - asynchronous on `stream` (a hipStream_t passed as void*; NULL = the null stream);
 *   - stateless / thread-safe: no entry point sets or reads caller-visible process state -- every mode (e.g. the `split`
 *     arithmetic switch of K1 / K13) is an argument of the launch it applies to (ABI 16; ABI <= 15 had three process-wide
 *     setters).  What the library caches internally (CU count, kernel attributes, launch plans per geometry) is keyed per
 *     device and guarded;
 *   - forward only, except soc_msda_bwd_* (the reference's other backward paths are training-only);
 *   - return 0 on success, a negative SOC_E* code otherwise (soc_hip_error_string()).
 *
 * Each function cites the reference interface it replaces (paths relative to the reference repo).
 */
#ifndef SOC_HIP_H
#define SOC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SOC_HIP_ABI_VERSION 16

#define SOC_OK 0
#define SOC_EINVAL (-1)       /* null pointer / non-positive dimension */
#define SOC_EUNSUPPORTED (-2) /* shape outside what the kernel is built for */
#define SOC_ELAUNCH (-3)      /* hipGetLastError() != hipSuccess after the launch */
#define SOC_EWORKSPACE (-4)   /* workspace too small */

int soc_hip_abi_version(void);

/*
 * CUs a launch on `stream` may use: the device's CU count, cut down to the stream's CU mask when the caller created the stream
 * with hipExtStreamCreateWithCUMask.  The one-workgroup-per-CU kernels (K1's schedule, K13 / K13b, K20, K23, K24) size their
 * grids and plans from it, so a host that partitions the chip between two streams -- graph_runner.PartitionedClipGraph: the
 * head of clip i on one CU set, the tail of clip i-1 and the text encoder on the other -- gets single-round launches on both.
 * The plan / workspace queries below take the stream of the launch they describe for the same reason.  Reads the stream; keeps
 * nothing.
 */
int soc_stream_cus(void* stream);

const char* soc_hip_error_string(int code);

/*
 * K2 -- multi-scale deformable attention, forward.
 * Replaces MSDA.ms_deform_attn_forward (models/ops/src/vision.cpp:13-16,
 * models/ops/src/ms_deform_attn.h:20-39, models/ops/src/cuda/ms_deform_attn_cuda.cu:20-80,
 * kernel models/ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299).
 *   value          [N, S, M, D]        contiguous
 *   spatial_shapes [L, 2] int64 (H_l, W_l), level_start_index [L] int64  (device memory)
 *   sampling_loc   [N, Lq, M, L, P, 2] (x, y) normalised to [0,1]
 *   attn_weight    [N, Lq, M, L, P]
 *   out            [N, Lq, M*D]
 * im2col_step of the reference only chunks the batch and has no numerical effect; it is not
 * part of this ABI (the Python binding still validates it like the reference does).
 */
int soc_msda_fwd_f32(const float* value, const int64_t* spatial_shapes,
                     const int64_t* level_start_index, const float* sampling_loc,
                     const float* attn_weight, float* out, int N, int S, int M, int D, int L,
                     int Lq, int P, void* stream);
int soc_msda_fwd_f64(const double* value, const int64_t* spatial_shapes,
                     const int64_t* level_start_index, const double* sampling_loc,
                     const double* attn_weight, double* out, int N, int S, int M, int D, int L,
                     int Lq, int P, void* stream);

/*
 * K2 backward -- gradients of the op above with respect to value, sampling_loc and attn_weight
 * (training side of the reference's native module: MSDA.ms_deform_attn_backward,
 * models/ops/src/vision.cpp:13-16, src/cuda/ms_deform_attn_cuda.cu:83-153, kernels
 * src/cuda/ms_deform_im2col_cuda.cuh:301-921; called from functions/ms_deform_attn_func.py:31-38).
 *   grad_out [N, Lq, M*D]  ->  grad_value [N, S, M, D] (zeroed, then accumulated with atomics),
 *   grad_sampling_loc [N, Lq, M, L, P, 2], grad_attn_weight [N, Lq, M, L, P]; any D.
 */
int soc_msda_bwd_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                     const float* sampling_loc, const float* attn_weight, const float* grad_out,
                     float* grad_value, float* grad_sampling_loc, float* grad_attn_weight, int N, int S,
                     int M, int D, int L, int Lq, int P, void* stream);
int soc_msda_bwd_f64(const double* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                     const double* sampling_loc, const double* attn_weight, const double* grad_out,
                     double* grad_value, double* grad_sampling_loc, double* grad_attn_weight, int N, int S,
                     int M, int D, int L, int Lq, int P, void* stream);

/*
 * K2 (fused form) -- the same sampling core with the arithmetic MSDeformAttn.forward wraps around
 * it folded in (models/ops/modules/ms_deform_attn.py:95-112): softmax over the L*P attention
 * logits, sampling locations from reference points + raw offsets, and the value padding mask.
 *   value          [N, S, M, D]  value_proj output, NOT yet masked
 *   value_pad_mask [N, S] uint8 (non-zero = padded position samples as 0) and any_pad (one int32 in
 *                  device memory, non-zero iff the mask has any padding): both NULL or both set
 *   ref_points     [N, Lq, L, ref_dim], ref_dim 2: loc = ref + off / (W_l, H_l)
 *                                       ref_dim 4: loc = ref_xy + off / P * ref_wh * 0.5
 *   offsets        [N, Lq, M, L, P, 2]  raw sampling_offsets Linear output
 *   attn_logits    [N, Lq, M, L*P]      raw attention_weights Linear output (softmax in-kernel)
 * Built for D = 32, L = 4, P = 4 (every shipped config); other shapes: SOC_EUNSUPPORTED, use
 * soc_msda_fwd_f32.  Tap addresses are 32-bit byte offsets inside a frame formed with 24-bit multiplies:
 * S * M * 128 < 2^31, S < 2^24 and M * 128 < 2^24 are required (a 720x1280 clip has S = 19 160; the limit is
 * S ~ 2 million positions at 8 heads); beyond that SOC_EUNSUPPORTED -- soc_msda_fwd_f32 then takes its
 * 64-bit generic path for the same sizes.
 */
int soc_msda_fused_fwd_f32(const float* value, const uint8_t* value_pad_mask, const int32_t* any_pad,
                           const int64_t* spatial_shapes, const int64_t* level_start_index,
                           const float* ref_points, int ref_dim, const float* offsets,
                           const float* attn_logits, float* out, int N, int S, int M, int D, int L,
                           int Lq, int P, void* stream);


/*
 * K1 -- 3-D (shifted) window attention with relative position bias, fused with the
 * pad / cyclic-roll / window-partition / reverse / un-roll / crop index math.
 * Replaces WindowAttention3D.forward between `qkv` and `proj`
 * (models/video_swin_transformer.py:144-166) together with the data movement of
 * SwinTransformerBlock3D.forward_part1 (:219-249), window_partition/window_reverse (:40-68) and
 * compute_mask (:316-329).
 *   qkv        [B, D, H, W, 3*C]  qkv Linear applied to the un-padded, LayerNorm-ed tokens
 *   qkv_bias   [3*C]              what a zero (padded) token projects to
 *   bias_table [(2*tab_d-1)*(2*tab_h-1)*(2*tab_w-1), n_heads]
 *   out        [B, D, H, W, C]    attention output (before `proj`), token layout
 * win_* / shift_* are the values AFTER the reference's clamping (get_window_size :71-84);
 * tab_* is the module's nominal window (8,7,7) that sizes the bias table and defines the
 * `relative_position_index[:N,:N]` slicing rule (:151).  head_dim = C / n_heads must be 32;
 * win_d*win_h*win_w <= 400.  Shift mask value is -100 (:328), not -inf.
 * split: arithmetic of THIS launch (full 8x7x7 windows only, ignored otherwise): != 0 = scores and P.V on the bf16 matrix
 * cores with every f32 operand split exactly into three bf16 terms (six products, f32 accumulation: f32-level error, see
 * K20 below); 0 = the f32-input MFMA form.  Same results to f32 rounding.  Since round 6 the split arithmetic has two kernels:
 * 1 (and any value other than 0 and 2) = the streaming form (32-query tiles, key chunks of 32, softmax between the MFMAs; the
 * row max is subtracted only for tiles whose row sum says it is needed), 2 = round 3's form (a 16-query tile's whole score
 * block in registers), kept for A/B measurements.  Same entry point, same results to f32 rounding.
 */
int soc_win_attn3d_f32(const float* qkv, const float* qkv_bias, const float* bias_table,
                       float* out, int B, int D, int H, int W, int C, int n_heads, int win_d,
                       int win_h, int win_w, int shift_d, int shift_h, int shift_w, int tab_d,
                       int tab_h, int tab_w, int split, void* stream);

/*
 * K3 -- multi-head attention core softmax(q k^T / sqrt(d)) v on already-projected tensors,
 * sequence-first layout.  Replaces the core of torch.nn.MultiheadAttention as used by
 * MMF.forward (models/vla.py:20-23), VOC's Self/CrossAttentionLayer (models/voc.py:89-90,
 * 146-149) and the decoder self-attention (models/deformable_transformer.py:333).
 *   q [Lq, B, n_heads*head_dim], k / v [Lk, B, n_heads*head_dim]
 *   key_pad_mask [B, Lk] uint8, non-zero = ignore key (may be NULL)
 *   attn_mask    [B * attn_mask_heads, Lq, Lk] float, ADDED to the scaled logits (torch's float attn_mask;
 *                may be NULL); attn_mask_heads = 1 (one mask for all heads) or n_heads -- used by VOC's
 *                shifted temporal windows (models/voc.py:367-377,401-414)
 *   out [Lq, B, n_heads*head_dim]
 * batch_first != 0: q / out are [B, Lq, E] and k / v [B, Lk, E] instead (the decoder's native layout,
 * which the reference transposes around nn.MultiheadAttention, deformable_transformer.py:333).
 * head_dim must be 32.  `workspace` is reserved for mappings that need scratch memory:
 * soc_xattn_workspace_bytes() returns 0 for every shape in this build, so NULL / 0 is fine.
 */
size_t soc_xattn_workspace_bytes(int Lq, int Lk, int B, int n_heads, int head_dim);
int soc_xattn_f32(const float* q, const float* k, const float* v, const uint8_t* key_pad_mask,
                  const float* attn_mask, int attn_mask_heads, float* out, int Lq, int Lk, int B,
                  int n_heads, int head_dim, int batch_first, void* workspace, size_t workspace_bytes,
                  void* stream);

/*
 * K4 -- per-instance dynamic mask head (3 dynamic 1x1 conv layers over
 * [C feature channels, rel_x, rel_y]) with the relative-coordinate generation fused in.
 * Replaces SOC.dynamic_mask_with_coords + compute_locations + parse_dynamic_params +
 * mask_heads_forward (models/soc.py:399-483, 486-509, 536-549) for batch size 1.
 *   feats  [T, C, h, w]           FPN output (C = mask_kernels_dim = 8)
 *   params [T*Q, (C+2)*8+8*8+8+8+8+1] controller output, order w0 w1 w2 b0 b1 b2, row-major [out][in]
 *   refs   [T*Q, 2]               normalised (x, y) reference points, instance order (t, q)
 *   out    [T*Q, h, w]
 * rel = ref * (img_w, img_h) - (stride*x + stride/2, stride*y + stride/2).
 */
int soc_dyn_mask_f32(const float* feats, const float* params, const float* refs, float* out,
                     int T, int Q, int C, int h, int w, float img_h, float img_w, int stride,
                     void* stream);

/*
 * K5 -- fused residual add + LayerNorm over the last dimension ("next").
 * Replaces the add + nn.LayerNorm pairs of SwinTransformerBlock3D.forward
 * (models/video_swin_transformer.py:219,262-272), of the deformable encoder / decoder layers
 * (models/deformable_transformer.py:247-263,324-347) and of VOC's layers (models/voc.py:44-48,
 * 84-94,141-153).
 *   sum = x + y (y may be NULL); out_sum (may be NULL) receives sum;
 *   out_norm = (sum - mean) / sqrt(var + eps) * gamma + beta      (biased variance, as torch)
 * x, y, out_sum, out_norm: [rows, C] contiguous; gamma, beta: [C].  C % 4 == 0, C <= 2048.
 * out_sum may alias x or y; out_norm must not alias an input.
 */
int soc_add_layernorm_f32(const float* x, const float* y, const float* gamma, const float* beta,
                          float* out_sum, float* out_norm, long rows, int C, float eps,
                          void* stream);

/*
 * K6 -- fused bilinear up-sampling (align_corners = False) + threshold of mask logits
 * (SURVEY 8a row a21, 8f rank 3).  Replaces
 *   F.interpolate(pred_masks, size=(H0, W0), mode='bilinear', align_corners=False) followed by
 *   (pred_masks.sigmoid() > 0.5)       infer_refytb.py:230-231, models/postprocessing.py:222-224
 *   logits [T, h, w] f32 -> out [T, H0, W0] uint8 (1 where the up-sampled logit > threshold_logit;
 *   threshold_logit = 0 is sigmoid > 0.5).
 */
int soc_upsample_threshold_u8(const float* logits, uint8_t* out, int T, int h, int w, int H0, int W0,
                              float threshold_logit, void* stream);

/*
 * K6 (DAVIS form) -- up-sample O objects' mask logits, sigmoid, zero the scores below `threshold`, put a
 * constant `background` plane in front and take the argmax over {background, objects}: the multi-object
 * merge of infer_davis.py:248-272 (F.interpolate + sigmoid per object, then :264-268).
 *   logits [O, T, h, w] f32 -> out [T, H0, W0] uint8 labels (0 = background, o + 1 = object o); the
 *   first maximum wins, as torch.argmax.  O <= 255.
 */
int soc_upsample_merge_labels_u8(const float* logits, uint8_t* out, int O, int T, int h, int w, int H0,
                                 int W0, float threshold, float background, void* stream);

/*
 * K9 -- frame pre-processing (the input side of infer_refytb.py:193-201 /
 * infer_davis.py:214-226): decoded RGB frames -> resized, normalised model input, bit-identical to
 *   PIL.Image.resize((w, h), BILINEAR)            datasets/transforms.py:186-216 (F.resize on a PIL image)
 *   -> ToTensor (x / 255) -> Normalize((x - mean) / std)            infer_refytb.py:33-38
 *   frames [T, H0, W0, 3] uint8 (HWC RGB)  ->  out [T, 3, h, w] f32;  out_u8 [T, h, w, 3] (optional, may
 *   be NULL): the resized uint8 frames as PIL would return them.
 * bounds_* [n_out, 2] int32 (first source index, tap count) and coeffs_* [n_out, ksize_*] int32 are
 * Pillow's resampling tables (precompute_coeffs + normalize_coeffs_8bpc, 22 fractional bits) for the
 * horizontal (x: W0 -> w) and vertical (y: H0 -> h) pass, in DEVICE memory; mean / std are HOST
 * pointers to 3 floats.  workspace: soc_resize_workspace_bytes() bytes of device memory (the 8-bit
 * intermediate image of PIL's two-pass scheme).
 */
size_t soc_resize_workspace_bytes(int T, int H0, int W0, int h, int w);
int soc_resize_normalize_u8_f32(const uint8_t* frames, float* out, uint8_t* out_u8, int T, int H0, int W0,
                                int h, int w, const int* bounds_x, const int* coeffs_x, int ksize_x,
                                const int* bounds_y, const int* coeffs_y, int ksize_y, const float* mean,
                                const float* std, void* workspace, size_t workspace_bytes, void* stream);

/*
 * K10 -- GroupNorm over token-major activations (the layout copies around the
 * library GEMMs).  Replaces nn.GroupNorm(32, d_model) of input_proj (models/soc.py:107-125, applied at
 * :226-230) together with the NCHW <-> token-major copies on either side of it; the 1x1 Conv2d in
 * front becomes a GEMM over tokens.
 *   x, out [N, S, C]  (N = frames, S = h*w);  gamma, beta [C];  G groups of C/G consecutive channels;
 *   statistics per (frame, group) over S * C/G elements, biased variance, as torch.nn.GroupNorm.
 * C <= 256 with C/4 a power of two; C/G a multiple of 4 with (C/G)/4 a power of two, or C/G == 2; G <= 64.
 * relu != 0 applies max(., 0) to the result (the FPN spatial decoder's GroupNorm + ReLU, models/segmentation.py:57-72).
 * workspace: soc_groupnorm_tokens_workspace_bytes() bytes of device memory (per-chunk partial sums).
 */
size_t soc_groupnorm_tokens_workspace_bytes(int N, int S, int C, int G);
int soc_groupnorm_tokens_f32(const float* x, const float* gamma, const float* beta, float* out, int N, int S,
                             int C, int G, float eps, int relu, void* workspace, size_t workspace_bytes, void* stream);

/*
 * K11 -- Video-Swin patch merging: 2x2 spatial gather + LayerNorm(4C) in one pass.  Replaces
 * PatchMerging.forward up to its `reduction` Linear (models/video_swin_transformer.py:279-313: F.pad for odd
 * sizes, the four strided slices x0..x3, torch.cat, self.norm).
 *   x [BD, H, W, C] token-major (BD = batch * frames)  ->  out [BD, ceil(H/2), ceil(W/2), 4C];
 *   channel order of the reference: (h even, w even), (h odd, w even), (h even, w odd), (h odd, w odd).
 * C % 4 == 0, C <= 512.
 */
int soc_patch_merge_layernorm_f32(const float* x, const float* gamma, const float* beta, float* out, int BD,
                                  int H, int W, int C, float eps, void* stream);

/*
 * K23 -- a two-layer perceptron block in one launch, on the bf16 matrix cores (exact three-way operand split, f32-grade
 * results -- see soc_linear_split_f32):
 *     out = LN2(act(LN(x) W1^T + b1) W2^T + b2 + residual),   act 1 = ReLU, 2 = exact (erf) GELU;  LN, LN2, residual optional;
 *     with residual_ln != 0 the shortcut is LN(x) (the same LayerNorm as in front of W1) instead of x itself: `residual` must
 *     then be x (the same pointer), anything else is SOC_EINVAL.
 * Replaces  x + mlp(norm2(x))  of SwinTransformerBlock3D.forward_part2 (models/video_swin_transformer.py:262-272 with
 * Mlp.forward :24-37: norm2, fc1, nn.GELU, fc2, the residual add) and linear1 -> ReLU -> linear2 of
 * DeformableTransformerEncoderLayer.forward_ffn (models/deformable_transformer.py:253-263) with the residual add and norm2
 * behind it (post_gamma / post_beta: `src = norm2(src + dropout3(src2))`, :261-262) and norm1 in front (ln_gamma / ln_beta with
 * residual_ln: `src = norm1(src + dropout1(src2))`, :249-250, whose result is both the block's input and its shortcut).  The [M, F] hidden tensor is
 * never written.  Every row is taken: whole rounds of the chip stream both weight matrices once per 16 (C <= 96), 8
 * (C <= 256) or 4 (C = 384) row tiles of 16 per CU; a short last round, or a short input altogether, is cut over `nfs` hidden ranges whose
 * partial sums a second kernel adds in a fixed order (deterministic).
 *   x [M, C], w1 [F, C], b1 [F], w2 [C, F], b2 [C], ln_gamma / ln_beta [C] or both NULL, residual [M, C] or NULL,
 *   post_gamma / post_beta [C] or both NULL, out [M, C], out_sum [M, C] or NULL (with LN2: the sum in front of it, i.e. the
 *   shortcut the next Video-Swin block needs beside norm1 of it); C in {96, 128, 192, 256, 384, 512}, F % 32 == 0; every pointer 16-byte aligned.
 *   soc_mlp_split_packed_bytes / soc_mlp_split_pack_f32: split and lay out both weights ONCE (opaque image `packed`);
 *   re-pack after the weights change.  soc_mlp_split_workspace_bytes: scratch for the partial sums (0 when none are needed).
 *   soc_mlp_split_plan: the (workgroup rows, hidden ranges) cut chosen for M rows taken as ONE launch.
 *   soc_mlp_split_variant_f32: one launch with an explicit cut (workspace: nfs * M * C floats when nfs > 1); `variant` 0 is
 *   the shipped kernel, other values exist in diagnostic builds only (SOC_K23_VARIANTS) and return SOC_EUNSUPPORTED otherwise.
 */
size_t soc_mlp_split_packed_bytes(int C, int F);
int soc_mlp_split_pack_f32(const float* w1, const float* w2, void* packed, int C, int F, void* stream);
size_t soc_mlp_split_workspace_bytes(long M, int C, int F, void* stream);
int soc_mlp_split_plan(long M, int C, int F, int* nrg, int* nfs, void* stream);
int soc_mlp_split_f32(const float* x, const void* packed, const float* b1, const float* b2, const float* ln_gamma,
                      const float* ln_beta, float ln_eps, const float* residual, const float* post_gamma,
                      const float* post_beta, float post_eps, float* out, float* out_sum, float* workspace,
                      size_t workspace_bytes, long M, int C, int F, int act, int residual_ln, void* stream);
/*
 * K26 -- query selection + record packing of the inference loop in one launch, for every clip of a launch group (reference
 * infer_refytb.py:216-226: pred_cls.sigmoid().mean(0) -> max over classes -> argmax over queries -> the masks of that query):
 *   records[b] = [ idx_b, pred_cls[:, b, :, 0]  (T*Q logits),  pred_masks[:, b, idx_b]  (T*HW logits) ]        (float32)
 *   idx_b = argmax_q max_k mean_t sigmoid(pred_cls[t, b, q, k])     (the first maximum, as torch.argmax)
 * pred_cls [T, B, Q, K] with element strides (cls_stride_t, cls_stride_b, cls_stride_q), K contiguous (the model hands out a
 * transposed view); pred_masks [T, B, Q, HW] contiguous; records [B][record_stride], record_stride >= 1 + T*Q + T*HW.
 * Q <= 256; otherwise SOC_EUNSUPPORTED.
 */
int soc_select_pack_f32(const float* pred_cls, long cls_stride_t, long cls_stride_b, long cls_stride_q,
                        const float* pred_masks, float* records, long record_stride, int T, int B, int Q, int K, long HW,
                        void* stream);

/* Largest hidden width F soc_mlp_split_f32 takes at model width C (the b1 range of a workgroup shares the 160 KB of LDS with
 * the weight ring); 0 for a width K23 is not built for.  Pure function of C.  F beyond it: SOC_EUNSUPPORTED at launch. */
int soc_mlp_split_max_hidden(int C);
int soc_mlp_split_variant_f32(const float* x, const void* packed, const float* b1, const float* b2, const float* ln_gamma,
                              const float* ln_beta, float ln_eps, const float* residual, const float* post_gamma,
                              const float* post_beta, float post_eps, float* out, float* out_sum, float* workspace, long M,
                              int C, int F, int act, int residual_ln, int nrg, int nfs, int variant, void* stream);

/*
 * K24 -- an x-stationary linear layer on the bf16 matrix cores (exact three-way operand split, f32-grade results -- see
 * soc_linear_split_f32): out = act(LN(x) W^T + bias) + residual, act 0 = none, 1 = ReLU, 2 = exact (erf) GELU; LN, bias,
 * residual optional.  Replaces the pixel-sized nn.Linear calls of input widths 192 / 256 / 384 / 512 / 768 / 1024: qkv / proj of
 * WindowAttention3D in Video-Swin stages 1-3 (models/video_swin_transformer.py:144-166, norm1 :219 in front, the shortcut
 * :254-259 behind), the PatchMerging reduction into stage 2 (:277-312), value_proj / output_proj of MSDeformAttn
 * (models/ops/modules/ms_deform_attn.py:95,114), the query projection of the fusion blocks (models/vla.py:18-24) and
 * input_proj of level 1 (models/soc.py:226-230).  A wave keeps its 16 rows as split MFMA fragments (split once); the weights
 * stream through an LDS ring; a launch is cut into `nrg` workgroup rows x `ncr` column spans (0, 0: the library plans it).  A
 * span is N / ncr <= 2048 columns, walked as ranges of 4 / 6 / 8 / 12 / 16 / 18 column tiles by waves that keep their split rows;
 * every cut gives the same bits.  soc_xs_linear_plan returns the cut the library would take (nct = column tiles per span).
 *   x [M, K], w [N, K] (nn.Linear.weight layout), bias [N] or NULL, ln_gamma / ln_beta [K] or both NULL, residual [M, N] or
 *   NULL, out [M, N]; K in {192, 256, 384, 512, 768, 1024}, N % 32 == 0 and N / 16 / ncr divisible into ranges of 4 / 6 / 8 / 12 /
 *   16 / 18 (K = 1024: 4 / 6 / 8) column tiles (soc_xs_linear_plan says whether and how); every pointer 16-byte aligned.
 *   soc_xs_linear_packed_bytes / soc_xs_linear_pack_f32: split and lay out the weights ONCE (opaque image); re-pack after a
 *   weight update.
 */
size_t soc_xs_linear_packed_bytes(int N, int K);
int soc_xs_linear_pack_f32(const float* w, void* packed, int N, int K, void* stream);
int soc_xs_linear_plan(long M, int N, int K, int* nrg, int* ncr, int* nct, void* stream);
int soc_xs_linear_f32(const float* x, const void* packed, const float* bias, const float* ln_gamma, const float* ln_beta,
                      float ln_eps, const float* residual, float* out, long M, int N, int K, int act, int nrg, int ncr,
                      void* stream);

/*
 * K25 -- multi-head self-attention core for short sequences: softmax(q k^T * scale + mask) v per (batch, head), L <= 64 tokens,
 * head dim 32 or 64.  Replaces torch.nn.functional.scaled_dot_product_attention as HuggingFace's RobertaSelfAttention calls it
 * inside the text encoder SOC.forward_text runs per clip (models/soc.py:167-181) -- PyTorch's AOTriton kernel on ROCm.
 *   q, k, v, out [B, L, H * D] token-major (heads are D-wide column blocks: what the q / k / v projections write),
 *   mask additive float (e.g. -inf on padding) addressed mask[b * mask_batch_stride + i * mask_query_stride + j] (query stride 0
 *   for a key-padding mask broadcast over queries) or NULL; a fully masked row yields zeros.
 */
int soc_small_attn_f32(const float* q, const float* k, const float* v, const float* mask, float* out, int B, int L, int H, int D,
                       float scale, long mask_batch_stride, long mask_query_stride, void* stream);

/*
 * K21 -- Video-Swin patch embedding: the (1,4,4) / stride (1,4,4) convolution + LayerNorm(C) in one pass.  Replaces
 * PatchEmbed3D.forward (models/video_swin_transformer.py:438-456: F.pad to multiples of the patch, the Conv3d `proj`,
 * flatten(2).transpose(1, 2), `norm`) for the patch size SOC builds the backbone with (:676, temporal patch 1).
 *   frames [N, 3, H, W] (N = batch * frames), weight [C, 3, 1, 4, 4] as stored in the reference's state_dict,
 *   bias [C] or NULL, gamma / beta [C]  ->  out [N, ceil(H/4), ceil(W/4), C] token-major;
 *   pixels past the right / bottom edge count as zeros (the reference's F.pad).
 * C = 96 (Swin-T / -S) or 128 (Swin-B); anything else returns SOC_EUNSUPPORTED.
 */
int soc_patch_embed_layernorm_f32(const float* frames, const float* weight, const float* bias, const float* gamma,
                                  const float* beta, float* out, int N, int H, int W, int C, float eps, void* stream);

/*
 * K12 -- tiled fp32 MFMA GEMM with a fused activation: out = act(x W^T + bias), act 0 = none, 1 = ReLU,
 * 2 = exact (erf) GELU.  Replaces fc1 + nn.GELU of the Video-Swin MLP (models/video_swin_transformer.py:24-37)
 * where the fused form beats the library GEMM + a separate GELU pass.
 *   x [M, K], w [N, K] (nn.Linear.weight layout), bias [N] or NULL, out [M, N]; K % 16 == 0, N % 4 == 0,
 *   16-byte aligned pointers.
 */
int soc_linear_act_f32(const float* x, const float* w, const float* bias, float* out, int M, int N, int K, int act,
                       void* stream);
/*
 * The same with the input formed as x + x_add (x_add [M, K] or NULL: the positional term of with_pos_embed) and
 * one to four layers that read it (HOST arrays of length nseg <= 4, as in soc_linear_small_multi_f32): the
 * deformable encoder's sampling_offsets + attention_weights on `src + pos`
 * (models/ops/modules/ms_deform_attn.py:96-97, models/deformable_transformer.py:245-249) as one launch.
 */
int soc_linear_act_multi_f32(const float* x, const float* x_add, int nseg, const float* const* w,
                             const float* const* bias, float* const* out, const int* N, int M, int K, int act,
                             void* stream);

/*
 * K7 -- small-M linear layer out = act((x [+ x_add]) W^T + bias)  ("next": the
 * library-GEMM share; here the latency-bound query-side layers).  Replaces nn.Linear / F.linear on
 * the frame-query / video-query / word tensors: DeformableTransformerDecoderLayer
 * (models/deformable_transformer.py:308-347: MultiheadAttention in/out projections, linear1/2),
 * MSDeformAttn.sampling_offsets / attention_weights on the decoder queries
 * (models/ops/modules/ms_deform_attn.py:96-97), VOC's attention projections and FFNs
 * (models/voc.py:44-48,66,123), the MLP heads (models/soc.py:552-563) and the "tensor + pos" adds
 * in front of them (with_pos_embed, deformable_transformer.py:318-320; voc.py:84-90,141-149).
 *   x      [M, K] contiguous         w [N, K] (nn.Linear.weight layout)    bias [N] or NULL
 *   x_add  [add_mod, K] or NULL: row m of the input is x[m] + x_add[(m / add_div) % add_mod]
 *   out    [M, N];  relu: 0 none, 1 max(., 0), 2 exact (erf) GELU (the text encoder's intermediate activation)
 * K % 16 == 0, M <= 4096, 16-byte aligned x / x_add / w; otherwise SOC_EUNSUPPORTED (use the
 * library GEMM).  fp32 MFMA accumulation, K split 4-ways per output tile.
 */
int soc_linear_small_f32(const float* x, const float* x_add, int add_div, int add_mod, const float* w,
                         const float* bias, float* out, int M, int N, int K, int relu, void* stream);
/*
 * The same for nseg <= 4 layers that read the same input (q / k / v of one attention:
 * torch.nn.MultiheadAttention's in_proj; MSDeformAttn's sampling_offsets + attention_weights) in ONE
 * launch.  w / bias / out / N / use_add are HOST arrays of length nseg (bias and use_add may be NULL;
 * bias[i] may be NULL); segment i computes out[i] [M, N[i]] from x (+ x_add iff use_add[i] != 0).
 */
int soc_linear_small_multi_f32(const float* x, const float* x_add, int add_div, int add_mod, int nseg,
                               const float* const* w, const float* const* bias, float* const* out,
                               const int* N, const int* use_add, int M, int K, int relu, void* stream);

/*
 * K8 -- iterative box refinement between decoder layers, one launch.  Replaces
 * DeformableTransformerDecoder.forward's `new = (tmp + inverse_sigmoid(reference_points)).sigmoid()`
 * and the next layer's `reference_points_input` (models/deformable_transformer.py:358-381,
 * inverse_sigmoid: util/misc.py) and the box head of SOC.forward (models/soc.py:330-340).
 *   delta [N*Q, 4]  bbox_embed output     ref [N*Q, ref_dim], ref_dim 2 (only x, y are refined) or 4
 *   new_ref [N*Q, 4]
 *   ref_in  [N*Q, L, 4] = new_ref * (vr_x, vr_y, vr_x, vr_y) with valid_ratios [N, L, 2]; ref_in and
 *   valid_ratios may both be NULL.
 */
int soc_box_refine_f32(const float* delta, const float* ref, int ref_dim, const float* valid_ratios,
                       float* new_ref, float* ref_in, int N, int Q, int L, void* stream);

/*
 * K13 -- weight-stationary linear layer for the tall, short-K GEMMs of the Video-Swin blocks, with the LayerNorm
 * in front and the residual add behind folded in (reference models/video_swin_transformer.py:219 norm1, :144-147
 * qkv, :163-164 proj, :254-259 residual, :262-267 norm2 -> mlp.fc1 -> GELU -> mlp.fc2 -> residual; Mlp :24-37):
 *   out[M, N] = act( LN(x)[M, K] . w[N, K]^T + bias ) + residual[M, N]
 * ln_gamma / ln_beta NULL: no LayerNorm (ln_eps ignored); bias / residual may be NULL; act 0 none, 1 ReLU,
 * 2 exact (erf) GELU as nn.GELU().  The weights are staged once per workgroup in LDS and the rows of x stream
 * through in MFMA operand layout.  K in {96, 128, 192, 256, 384, 512} (LayerNorm: K <= 256), N % 16 == 0, 16-byte
 * aligned pointers; otherwise SOC_EUNSUPPORTED (use the library GEMM).  out may alias residual.
 * split: arithmetic of THIS launch: != 0 = the widths K13b covers (K = 96 / 128 / 192 / 256 ...) run on the bf16 matrix cores
 * with the exact three-way operand split (f32-grade results, see soc_linear_split_f32); 0 = f32-input MFMA for every width.
 */
int soc_ws_linear_f32(const float* x, const float* ln_gamma, const float* ln_beta, float ln_eps, const float* w,
                      const float* bias, const float* residual, float* out, long M, int N, int K, int act, int split,
                      void* stream);

/*
 * K15 -- the cross-attention block of a deformable-decoder layer in one launch (reference
 * models/deformable_transformer.py:335-341: tgt2 = cross_attn(with_pos_embed(tgt, query_pos), reference_points, src,
 * ...); tgt = norm1(tgt + tgt2); MSDeformAttn.forward models/ops/modules/ms_deform_attn.py:79-117):
 *   out = LayerNorm(tgt + output_proj(MSDeformAttn(tgt + query_pos, ref_points, value_proj(memory))))
 * The 256-wide memory rows are sampled first and value_proj is applied to the sampled sums (value_proj is linear; the
 * bias is weighted by the sum of the in-range tap coefficients), so the whole-memory value projection of the reference
 * is never computed.  For few queries (one workgroup per (frame, query) row reads the four weight matrices).
 *   tgt [N, Lq, 256]; query_pos [N, Lq, 256] (query_pos_per_frame != 0) or [Lq, 256] shared by the frames
 *   ref_points [N, Lq, 4, ref_dim] (ref_dim 2 or 4, as soc_msda_fused_fwd_f32); memory [N, S, 256] (un-masked)
 *   memory_pad_mask [N, S] uint8 + any_pad int32[1]: both NULL or both set (padded positions sample as 0)
 *   w_off [256,256] b_off [256] sampling_offsets; w_att [128,256] b_att [128] attention_weights;
 *   w_val / b_val value_proj; w_out / b_out output_proj; ln_gamma / ln_beta / ln_eps norm1;  out [N, Lq, 256]
 * Built for d_model 256, 8 heads, 4 levels, 4 points (every shipped config); otherwise SOC_EUNSUPPORTED.
 */
int soc_decoder_cross_attn_f32(const float* tgt, const float* query_pos, int query_pos_per_frame,
                               const float* ref_points, int ref_dim, const float* memory,
                               const uint8_t* memory_pad_mask, const int32_t* any_pad, const int64_t* spatial_shapes,
                               const int64_t* level_start_index, const float* w_off, const float* b_off,
                               const float* w_att, const float* b_att, const float* w_val, const float* b_val,
                               const float* w_out, const float* b_out, const float* ln_gamma, const float* ln_beta,
                               float ln_eps, float* out, int N, int Lq, int S, int d_model, int n_heads, int n_levels,
                               int n_points, void* stream);

/*
 * K16 -- up to three 256-wide linear layers on few rows, optionally closed by residual add + LayerNorm, in one launch:
 *   h = x (+ x_add, rows broadcast as (m / add_div) % add_mod);  h = relu(h w[i]^T + bias[i]) for i < n_layers - 1;
 *   y = h w[last]^T + bias[last]   ([n_out, 256], n_out <= 256);
 *   out = y (+ residual)                          when ln_gamma == NULL
 *   out = LayerNorm(residual + y; gamma, beta)    otherwise (n_out == 256; residual may be NULL)
 * Replaces MLP.forward of bbox_embed / controller (reference models/soc.py:552-564, 3 Linear + ReLU) and the
 * out_proj -> residual -> LayerNorm tails of the decoder's self-attention (models/deformable_transformer.py:330-334)
 * and of VOC's attention layers (models/voc.py:44-48,84-94).  K (input and hidden width) must be 256, n_layers <= 3;
 * otherwise SOC_EUNSUPPORTED.  One workgroup per row: meant for M <= a few hundred rows.
 */
int soc_row_mlp_f32(const float* x, const float* x_add, int add_div, int add_mod, int n_layers, const float* const* w,
                    const float* const* bias, int n_out, const float* residual, const float* ln_gamma,
                    const float* ln_beta, float ln_eps, float* out, int M, int K, void* stream);

/*
 * K17 -- GroupNorm (+ the preceding convolution's bias, + ReLU) on an NCHW map in one launch: the
 * `F.relu(gn(lay(x)))` steps of the FPN spatial decoder (reference models/segmentation.py:57-72) without the
 * convolution itself:   y = act(GroupNorm(x + bias[c]; groups, gamma, beta, eps)),  act = ReLU when relu != 0.
 * x, y [N, C, HW] contiguous (y may alias x); bias [C] or NULL; biased variance as torch.  HW % 4 == 0 and at most 32 768
 * values per (sample, group); otherwise SOC_EUNSUPPORTED.
 */
int soc_groupnorm_nchw_f32(const float* x, const float* bias, const float* gamma, const float* beta, float* y, int N,
                           int C, int HW, int groups, float eps, int relu, void* stream);

/*
 * K18 -- lateral connection of the FPN spatial decoder (reference models/segmentation.py:62-72:
 * `cur_fpn + F.interpolate(x, size=cur_fpn.shape[-2:], mode="nearest")`):
 *   y[n,c,h,w] = lateral[n,c,h,w] + bias[c] + prev[n, c, min(floor(h * Hp / H), Hp-1), min(floor(w * Wp / W), Wp-1)]
 * (the float index rule of F.interpolate(mode="nearest")).  lateral, y [N,C,H,W]; prev [N,C,Hp,Wp]; bias [C] or NULL
 * (the adapter convolution's bias when that was run without one).
 */
int soc_upsample_add_nchw_f32(const float* lateral, const float* bias, const float* prev, float* y, int N, int C, int H,
                              int W, int Hp, int Wp, void* stream);

/*
 * K18 (token-major form) -- the same lateral connection on channels-last maps:
 *   y[n, h, w, c] = lateral[n, h, w, c] + bias[c] + prev[n, min(floor(h * Hp / H), Hp-1), min(floor(w * Wp / W), Wp-1), c]
 * lateral, y [N, H*W, C] contiguous (y may alias lateral); prev [N, Hp*Wp, C] contiguous; C % 4 == 0.
 */
int soc_upsample_add_tokens_f32(const float* lateral, const float* bias, const float* prev, float* y, int N, int C, int H,
                                int W, int Hp, int Wp, void* stream);

/*
 * K19 -- 3x3 / stride 1 / zero-pad 1 convolution on a token-major (channels-last) map as an implicit GEMM on f32 MFMA:
 * the convolutions of the FPN spatial decoder (reference models/segmentation.py:24-33,57-74: lay1..lay5, out_lay).
 *   out[(n, y, x), co] = act(bias[co] + sum_{ky,kx,ci} in[n, y+ky-1, x+kx-1, ci] * w[co, ci, ky, kx])
 *   in      [N][H*W][Cin], frame n at in + n * in_frame_stride (elements; >= H*W*Cin: a level of the encoder memory is
 *           read in place)
 *   w_taps  [Cout][9 * Cin] = the Conv2d weight [Cout, Cin, 3, 3] permuted to [Cout, ky, kx, Cin]
 *   bias    [Cout] or NULL;  relu != 0: max(., 0)
 *   out     token-major [N*H*W][Cout] (out_nchw == 0) or NCHW [N][Cout][H][W] (out_nchw != 0)
 * Cin % 16 == 0 and 16-byte aligned in / w_taps; otherwise SOC_EUNSUPPORTED.
 */
int soc_conv3x3_tokens_f32(const float* in, long in_frame_stride, const float* w_taps, const float* bias, float* out,
                           int N, int H, int W, int Cin, int Cout, int out_nchw, int relu, void* stream);

/*
 * K20 -- the pixel-sized f32 linear layers on the bf16 matrix cores by EXACT operand splitting (f32 in, f32 out, f32
 * accumulation): nn.Linear as called by the Video-Swin blocks (reference models/video_swin_transformer.py:144-147 qkv,
 * :163-164 proj, :24-37 Mlp fc1 / GELU / fc2, :254-274 the LayerNorms in front and the residual adds behind), PatchMerging
 * (:309-311), input_proj (models/soc.py:56-77), the MMF projections (models/vla.py:20-24, incl. `tgt *`) and the
 * deformable encoder (models/deformable_transformer.py:253-263 FFN, models/ops/modules/ms_deform_attn.py:93-116):
 *   out[M, N] = mul[M, N] * act( LN(x [+ x_add])[M, K] . w[N, K]^T + bias[N] ) + residual[M, N]
 * Every f32 operand is split into three bf16 numbers whose sum is the operand exactly; six of the nine bf16 x bf16
 * products (each exact in f32) are accumulated in f32, the three dropped ones are <= 2^-23 of |a b| -- f32-level error
 * (no larger than the f32 library GEMM's against an f64 reference) at 6/16 of the f32 MFMA time.
 *   soc_linear_split_packed_bytes / soc_linear_split_pack_f32: split w [N][K] ONCE into the kernel's weight image
 *     (opaque; w_packed below); re-pack after the weights change.
 *   soc_row_stats_f32: stats[M][2] = (mean, 1 / sqrt(biased variance + eps)) of the rows of x [M][K] (K % 4 == 0,
 *     K <= 1024), as nn.LayerNorm computes them.
 *   soc_linear_split_f32: x_add, row_stats (+ w_colsum), bias, residual, mul may be NULL; act 0 none, 1 ReLU, 2 exact
 *     (erf) GELU; K % 8 == 0, N % 4 == 0; every pointer 16-byte aligned; out may alias residual or mul; tile_cfg 0..4 =
 *     128x256, 256x128, 128x128, 256x96, 128x64 output tiles (speed only).  A LayerNorm in front of the layer is applied
 *     behind it, exactly: LN(x) w^T + b = rstd (x (w diag(gamma))^T - mean colsum) + (b + w beta).  The CALLER packs
 *     w diag(gamma), passes bias = b + w beta, w_colsum[n] = sum_k w[n][k] gamma[k] and row_stats = soc_row_stats_f32(x);
 *     the kernel computes rstd[m] * (acc[m][n] - mean[m] * w_colsum[n]) + bias[n] in front of act / mul / residual.
 *     out2 != NULL: two layers that read the same input (rows of w stacked): columns [0, n_split) go to out [M, n_split],
 *     columns [n_split, N) to out2 [M, N - n_split] (n_split % 4 == 0; no residual / mul) -- the deformable encoder's
 *     sampling_offsets | attention_weights on src + pos (models/ops/modules/ms_deform_attn.py:95-96).
 */
size_t soc_linear_split_packed_bytes(int N, int K);
int soc_linear_split_pack_f32(const float* w, void* packed, int N, int K, void* stream);
int soc_row_stats_f32(const float* x, float* stats, long M, int K, float eps, void* stream);
int soc_linear_split_f32(const float* x, const float* x_add, const float* row_stats, const float* w_colsum,
                         const void* w_packed, const float* bias, const float* residual, const float* mul, float* out,
                         float* out2, int n_split, long M, int N, int K, int act, int tile_cfg, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SOC_HIP_H */
