/*
 * hipie_mi355.h -- C ABI of libhipie_mi355.so: the hand-written HIP (gfx950 / CDNA4) kernels of HIPIE's
 * single-image inference hot path.
 *
 * Conventions:
 *   - plain C, no exceptions, no torch types; every pointer is a DEVICE pointer unless it says "host";
 *   - the caller owns all buffers (inputs are borrowed, outputs are pre-allocated), nothing is allocated here;
 *   - every entry point takes the hipStream_t to launch on (as void*) and returns 0 on success or a negative
 *     HIPIE_E* code; hipie_last_error() gives the text for the calling thread;
 *   - stateless and thread-safe; launches are asynchronous (no sync inside), so they can be captured in hipGraphs;
 *   - tensors are dense row-major with the last index fastest unless explicit strides are given.
 *
 * Each function names the reference interface it replaces (paths relative to projects/HIPIE/hipie/).
 */
#ifndef HIPIE_MI355_H
#define HIPIE_MI355_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIPIE_ABI_VERSION 13

/* element types of activations */
#define HIPIE_F32 0
#define HIPIE_F16 1
#define HIPIE_BF16 2
#define HIPIE_F64 3          /* hipie_msda_forward only (the reference op dispatches float | double) */
#define HIPIE_HL8 4          /* SPLIT fp16 ("hi + lo in groups of 8"): a logical row of K values is stored as K/8 groups of 32 bytes,
                                8 fp16 hi[0..7] then 8 fp16 lo[0..7], with hi = fp16(x), lo = fp16(x - hi): hi + lo == x to 2^-22.
                                Same bytes as fp32; the operand format of the fp32-class products on the 16-bit matrix pipe
                                (hipie_gemm, hipie_vit_attn_rel with HIPIE_ATTN_SPLIT).  Row strides are given in fp16 elements (>= 2K). */

#define HIPIE_OUT_F32 0x100   /* or-ed into the `dtype` of hipie_flash_attn / hipie_bi_xattn(_ws): q, k, v stay 16 bit, the OUTPUT(S) are written as
                                fp32 (output strides in fp32 elements): the attention result enters the next linear unrounded */
#define HIPIE_K_HL8_HI 0x200  /* or-ed into the `dtype` of hipie_flash_attn: `k` points at an HIPIE_HL8 buffer and the keys are its `hi` halves (hi = fp16(x)
                                by construction): the 8-element chunks of a head row are 16 elements apart; k_sb / k_st / k_sh are strides of
                                the HL8 buffer in fp16 elements (k_sh = 2 * head_dim).  Saves the strided copy that extracts them. */

/* flags of the attention entry points that take a `flags` argument */
#define HIPIE_ATTN_FAST 1    /* deferred running max (rescale only when a row maximum grows by > 2^8) and, where the head dim
                                leaves a padded MFMA row, softmax denominators from a ones column of V (same 16-bit-rounded
                                probabilities as the numerator).  Without it: classic running max, fp32 sums of the
                                unrounded probabilities (the parity policy). */

/* error codes */
#define HIPIE_OK 0
#define HIPIE_EINVAL (-22)   /* bad shape / dtype / null pointer / unsupported geometry */
#define HIPIE_ELAUNCH (-5)   /* hipLaunchKernel failed (hipie_last_error() has hipGetErrorString) */

int hipie_version(void);
/* text of the last error raised on the calling thread ("" when none). */
const char* hipie_last_error(void);

/*
 * Multi-scale deformable attention sampling, forward.
 * Replaces: MultiScaleDeformableAttention.ms_deform_attn_forward (pybind, models/deformable_detr/ops/src/vision.cpp:13-16)
 *           = ms_deform_attn_cuda_forward (ops/src/cuda/ms_deform_attn_cuda.cu:20-80)
 *           -> ms_deformable_im2col_gpu_kernel (ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299),
 *           and the identical copy under models/maskdino/pixel_decoder/ops/.
 *   value          (B, S, M, D)        dtype `value_dtype` (f32 as the reference; f16/bf16 halve the gather traffic; f64: the
 *                                      reference's double instantiation -- then sampling_loc, attn_weight and out are f64 too)
 *   spatial_shapes (L, 2) int64 (H, W) device
 *   level_start    (L,)   int64        device
 *   sampling_loc   (B, Lq, M, L, P, 2) f32 (f64 with an f64 value), (x, y) normalised to [0,1] (values outside sample zero padding)
 *   attn_weight    (B, Lq, M, L, P)    f32 (f64 with an f64 value)
 *   out            (B, Lq, M*D)        dtype `value_dtype`; fully overwritten (the reference at::zeros + writes all)
 * im2col_step of the reference is a batching detail of its host loop and has no equivalent here.
 */
int hipie_msda_forward(const void* value, const int64_t* spatial_shapes, const int64_t* level_start,
                       const void* sampling_loc, const void* attn_weight, void* out,
                       int B, int S, int M, int D, int L, int Lq, int P, int value_dtype, void* stream);

/*
 * Backward of the operator above -- the plugin's second entry point.
 * Replaces: ms_deform_attn_backward of the `MultiScaleDeformableAttention` extension (ops/src/vision.cpp:13-16; called by
 *           MSDeformAttnFunction.backward, ops/functions/ms_deform_attn_func.py:32-41)
 *           = ms_deform_attn_cuda_backward (ops/src/cuda/ms_deform_attn_cuda.cu:83-153)
 *           -> ms_deformable_col2im_gpu_kernel_* (ops/src/cuda/ms_deform_im2col_cuda.cuh:301-1320).  Training only.
 *   value, spatial_shapes, level_start, sampling_loc, attn_weight: as hipie_msda_forward;  dtype HIPIE_F32 | HIPIE_F64 for all
 *   grad_output       (B, Lq, M*D)
 *   grad_value        (B, S, M, D)         zeroed here, then accumulated with hardware atomics (arrival order, as the reference)
 *   grad_sampling_loc (B, Lq, M, L, P, 2)  fully written, deterministic
 *   grad_attn_weight  (B, Lq, M, L, P)     fully written, deterministic
 */
int hipie_msda_backward(const void* value, const int64_t* spatial_shapes, const int64_t* level_start, const void* sampling_loc,
                        const void* attn_weight, const void* grad_output, void* grad_value, void* grad_sampling_loc,
                        void* grad_attn_weight, int B, int S, int M, int D, int L, int Lq, int P, int dtype, void* stream);

/*
 * The same operator with a caller-provided workspace: for fp32 and D = 32 (every MSDA of the model) grad_value is computed from the
 * destination side -- sampling corners are binned by (pixel, head) with integer atomics, then every destination sums its own list in
 * registers and is written once -- instead of B*Lq*M*L*P*4*D floating-point atomics (which bound the form above: ~460 G adds/s).
 * Other dtypes / widths run the kernel above (the workspace is ignored).  Same arguments and results; grad_value needs no zeroing and
 * is summed in slot (arrival) order.  workspace: at least hipie_msda_backward_workspace(...) bytes, 16-byte aligned, contents
 * irrelevant before and after.  Replaces the same reference entry as hipie_msda_backward (ms_deform_attn_cuda.cu:83-153).
 */
int64_t hipie_msda_backward_workspace(int B, int S, int M, int L, int Lq, int P);
int hipie_msda_backward_ws(const void* value, const int64_t* spatial_shapes, const int64_t* level_start, const void* sampling_loc,
                           const void* attn_weight, const void* grad_output, void* grad_value, void* grad_sampling_loc,
                           void* grad_attn_weight, int B, int S, int M, int D, int L, int Lq, int P, int dtype, void* workspace,
                           int64_t workspace_bytes, void* stream);

/*
 * Fused form used by the product path: sampling locations and the 16-way softmax are computed in the kernel from
 * the raw projections, so neither (B,Lq,M,L,P,2) locations nor normalised weights ever reach HBM.
 * Replaces: MSDeformAttn.forward lines 99-114 (ops/modules/ms_deform_attn.py) + the op above.
 *   offsets: row (b,q) at offsets + (b*Lq+q)*off_row_stride holds (M, L, P, 2) = sampling_offsets(query)
 *   logits:  row (b,q) at logits  + (b*Lq+q)*logit_row_stride holds (M, L*P) = attention_weights(query)
 *            (both `aux_dtype` f32|f16|bf16; the strides let one concatenated projection GEMM feed both)
 *   ref     (B, Lq, L, ref_dim) f32, ref_dim 2: loc = ref + off/(W_l,H_l); ref_dim 4: loc = ref_xy + off/P*ref_wh*0.5
 */
int hipie_msda_fused_forward(const void* value, const int64_t* spatial_shapes, const int64_t* level_start,
                             const float* ref, const void* offsets, const void* logits, void* out,
                             int B, int S, int M, int D, int L, int Lq, int P, int ref_dim, int value_dtype,
                             int aux_dtype, int64_t off_row_stride, int64_t logit_row_stride, void* stream);

/*
 * hipie_msda_fused_forward on a value tensor whose pixel rows are `value_row_stride` elements apart (>= M*D, a multiple of 8):
 * the value projections of all decoder layers read the same memory, so they run as ONE GEMM on the concatenated weights
 * (deformable_transformer_dino.py:418-450, one value_proj per layer) and every layer samples its own column block in place.
 */
int hipie_msda_fused_forward_strided(const void* value, int64_t value_row_stride, const int64_t* spatial_shapes,
                                     const int64_t* level_start, const float* ref, const void* offsets, const void* logits,
                                     void* out, int B, int S, int M, int D, int L, int Lq, int P, int ref_dim, int value_dtype,
                                     int aux_dtype, int64_t off_row_stride, int64_t logit_row_stride, void* stream);

/*
 * Fused (flash-style) attention core shared by the ViT blocks and the VL fusion:
 *     out[b,i,h,:] = softmax_j( clamp(scale * q[b,i,h,:].k[b,j,h,:], +-clamp) + bias_h[bh,j / kw,i] + bias_w[bh,i,j % kw]
 *                               + (key_mask[b,j] ? 0 : -inf) ) . v[b,j,h,:]
 * with fp32 scores / softmax / accumulation, 16-bit MFMA operands, nothing of size Nq x Nk in HBM.
 * q,k,v,out are addressed with explicit element strides (batch, token, head); head_dim contiguous.
 *   dtype      HIPIE_F16 or HIPIE_BF16 (q, k, v, out)
 *   head_dim   80 (ViT-H), 64 (ViT-B/L), 256 (VL fusion), 32
 *   bias_h     (B*H, kh, Nq) f32 (key-row major: one coalesced line per key row) or NULL;  bias_w (B*H, Nq, kw) f32 or
 *              NULL  (both or none; needs Nk == kh*kw)
 *   key_mask   (B, Nk) uint8 (1 = keep) or NULL
 *   clamp      <= 0 disables the clamp
 * Replaces: Attention.forward's (q*scale)@k^T -> add_decomposed_rel_pos -> softmax -> @v (backbone/vit.py:72-80,
 *           backbone/utils.py:96-125) and the two softmax(QK^T)V products of BiMultiHeadAttention.forward
 *           (models/deformable_detr/fuse_helper.py:69-121).
 */
int hipie_flash_attn(const void* q, const void* k, const void* v, void* out,
                     int B, int H, int Nq, int Nk, int head_dim,
                     int64_t q_sb, int64_t q_st, int64_t q_sh, int64_t k_sb, int64_t k_st, int64_t k_sh,
                     int64_t v_sb, int64_t v_st, int64_t v_sh, int64_t o_sb, int64_t o_st, int64_t o_sh,
                     const float* bias_h, const float* bias_w, int kh, int kw,
                     const uint8_t* key_mask, float scale, float clamp, int dtype, void* stream);

/*
 * ViT attention with decomposed relative position bias on a packed qkv tensor.
 * Replaces: Attention.forward between the qkv and proj Linears (backbone/vit.py:69-80) for both the 14x14 windowed
 * blocks (x already window-partitioned, B = batch*windows, gh=gw=14) and the global blocks (gh x gw token grid).
 *   qkv    (B, gh*gw, 3, heads, hd) 16-bit;  rel_h (B*heads, gh, gh*gw) f32 = q.Rh[hq,hk] (key row major);
 *          rel_w (B*heads, gh*gw, gw) f32 = q.Rw[wq,wk]
 *   out    (B, gh*gw, heads*hd) 16-bit
 */
int hipie_vit_attn(const void* qkv, const float* rel_h, const float* rel_w, void* out,
                   int B, int gh, int gw, int heads, int hd, float scale, int dtype, void* stream);

/*
 * Bi-directional text<->vision cross attention core of the VL early-fusion block.
 * Replaces: BiMultiHeadAttention.forward lines 69-121 (models/deformable_detr/fuse_helper.py), i.e. everything between
 * the four input projections and the two output projections; clamp +-50000, text key mask, no mask on the image side.
 *   q (B,Nv,H,hd) = v_proj(v)*scale, k (B,L,H,hd) = l_proj(l), vv (B,Nv,H,hd), vl (B,L,H,hd): 16-bit
 *   text_mask (B,L) uint8;  out_v (B,Nv,H*hd), out_l (B,L,H*hd) 16-bit
 */
int hipie_bi_xattn(const void* q, const void* k, const void* vv, const void* vl, const uint8_t* text_mask,
                   void* out_v, void* out_l, int B, int H, int Nv, int L, int hd, float clamp, int dtype, void* stream);

/*
 * hipie_bi_xattn with a caller-provided workspace (the library never allocates).  For one short text per image
 * (64 < L <= 224, head dim 256: the class captions of the detection task) both directions run specialised kernels: image -> text
 * as one softmax window over the resident text, text -> image with the text block in registers and the image range split over
 * workgroups, whose partial (max, sum, accumulator) triples live in `workspace` until a combine kernel reduces them.
 * hipie_bi_xattn_workspace returns the bytes needed (0: the shape takes the generic kernel and needs none); a null / short
 * workspace is not an error -- the text -> image direction then runs the generic kernel.
 */
int64_t hipie_bi_xattn_workspace(int B, int H, int Nv, int L, int head_dim);
int hipie_bi_xattn_ws(const void* q, const void* k, const void* vv, const void* vl, const uint8_t* text_mask,
                      void* out_v, void* out_l, void* workspace, int64_t workspace_bytes, int B, int H, int Nv, int L,
                      int head_dim, float clamp, int dtype, void* stream);

/*
 * Mask-logit contraction  out[b,q,p] = sum_c embed[b,q,c] * feats[b,c,p].
 * Replaces: torch.einsum("bqc,bchw->bqhw") in MaskDINODecoder.forward_prediction_heads
 *           (models/maskdino/transformer_decoder/maskdino_decoder.py:520-529).
 *   embed (B,Q,C) f32, feats (B,C,HW) f32, out (B,Q,HW) `out_dtype` (f32 | f16 | bf16).
 *   precision 0: exact fp32 (v_mfma_f32_32x32x2_f32, bitwise an fmaf chain); 1: bf16x3 split (3 bf16 MFMAs per
 *   product, ~2^-16 relative); 2: single bf16 MFMA.   C % 16 == 0.
 */
int hipie_mask_einsum(const float* embed, const float* feats, void* out, int B, int Q, int C, int HW,
                      int precision, int out_dtype, void* stream);

/*
 * The same contraction with a caller-provided workspace, plus a per-row constant:
 *   out[b,q,p] = row_bias[b,q] + sum_c embed[b,q,c] * feats[b,c,p],   row_bias (B,Q) f32 or NULL, precision 1 or 2.
 * The embedding is split ONCE per call into the workspace (bf16 hi + lo parts in the layout of the kernel's LDS tile) and staged
 * by LDS-DMA, instead of being split by every workgroup (hipie_mask_einsum): the faster form.  hipie_mask_einsum_workspace
 * returns the bytes needed; the workspace is scratch (nothing is kept between calls), 16-byte aligned.
 * With embed' = embed . W and row_bias = embed . b the last 1x1 convolution of the pixel decoder's mask_features head
 * (Conv2d(256,256,1), maskdino_encoder.py:283-300) folds into the contraction of maskdino_decoder.py:527: the (B,256,H/4,W/4)
 * map is neither convolved nor written a second time.
 */
int64_t hipie_mask_einsum_workspace(int B, int Q, int C);
int hipie_mask_einsum_ws(const float* embed, const float* feats, const float* row_bias, void* out, void* workspace,
                         int64_t workspace_bytes, int B, int Q, int C, int HW, int precision, int out_dtype, void* stream);

/*
 * The same contraction on 16-bit features (the activation dtype of the 16-bit policies): out[b,q,p] = sum_c embed[b,q,c] *
 * feats[b,c,p] with feats (B,C,HW) `dtype` (f16 | bf16), the query embedding pre-split by the caller into embed_hi + embed_lo
 * (both (B,Q,C) `dtype`; embed_lo NULL: single product), out (B,Q,HW) `out_dtype` (= dtype, or f32).  fp32 accumulation.
 * row_bias (B,Q) f32 or NULL is added to every pixel of a query row: with embed' = embed . W and row_bias = embed . b the last
 * 1x1 convolution of the pixel decoder's mask_features head (maskdino_encoder.py:283-300) folds into this contraction.
 * Replaces the same torch.einsum (maskdino_decoder.py:527).  Q <= 320, C % 16 == 0, HW % 8 == 0 (HW even for f32 output).
 */
int hipie_mask_einsum16(const void* embed_hi, const void* embed_lo, const void* feats, const float* row_bias, void* out, int B,
                        int Q, int C, int HW, int dtype, int out_dtype, void* stream);

/*
 * Fused CondInst dynamic mask head: relative-coordinate generation + per-instance 10->8->8->1 MLP (ReLU, ReLU, none)
 * + aligned_bilinear x`up`, without materialising the (1, N*10, H, W) input or running a grouped conv.
 * Replaces: DDETRSegmUniDN.dynamic_mask_with_coords (models/ddetrs_dn.py:1411-1502) incl. compute_locations (:1857),
 *           parse_dynamic_params (:1806), mask_heads_forward (:1390) and aligned_bilinear (:1832).
 *   feats (B,8,H,W) f32;  refs (B*Q,2) f32 pixels (x,y);  params (B*Q,169) f32 = [w0 8x10 | w1 8x8 | w2 1x8 | b0 8 | b1 8 | b2 1]
 *   out (B*Q, up*H, up*W) `out_dtype`;  instance n uses feats[n / Q];  up in {1,2};  stride = mask feature stride (8)
 */
int hipie_dynamic_mask(const float* feats, const float* refs, const float* params, void* out,
                       int B, int Q, int H, int W, int stride, int up, int out_dtype, void* stream);

/*
 * Backward of hipie_dynamic_mask (fp32, up in {1,2}): gradients of a scalar loss with respect to the mask features, the reference
 * points and the 169 controller parameters per instance, given grad_out (B*Q, up*H, up*W) f32 = d loss / d out.
 * Replaces: torch.autograd through DDETRSegmUniDN.dynamic_mask_with_coords / mask_heads_forward / aligned_bilinear
 *           (models/ddetrs_dn.py:1411-1502, 1390-1408, 1832-1854) in the training forward (coco_forward, :264-750).
 *   grad_feats (B,8,H,W), grad_refs (B*Q,2), grad_params (B*Q,169) f32, overwritten (zeroed here, then accumulated: fp32 atomics, so the
 *   sums over pixels / instances run in arrival order like the reference's cuDNN / atomics-based conv backward).
 *   Ragged instance counts per image (the matched instances of training): one call per image with B = 1.
 */
int hipie_dynamic_mask_backward(const float* feats, const float* refs, const float* params, const float* grad_out, float* grad_feats,
                                float* grad_refs, float* grad_params, int B, int Q, int H, int W, int stride, int up, void* stream);

/*
 * hipie_dynamic_mask (up = 2) with the three layers on the matrix pipe: 16-bit operands (`dtype` f16 / bf16: the features,
 * the layer weights and the hidden activations are rounded to it; the coordinate weights are split hi + lo), fp32 accumulate,
 * fp32 biases and reference-point terms.  Same replaced reference code and argument meaning as hipie_dynamic_mask.
 * W % 4 == 0, stride % 4 == 0, stride * max(H, W) <= 8192 (pixel coordinates exact in 16 bit); else HIPIE_EINVAL.
 */
int hipie_dynamic_mask16(const float* feats, const float* refs, const float* params, void* out,
                         int B, int Q, int H, int W, int stride, int dtype, int out_dtype, void* stream);

/*
 * Decomposed relative-position bias tables of one ViT block, straight from the packed 16-bit qkv tensor.
 * Replaces: add_decomposed_rel_pos's two einsums + get_rel_pos gathers (backbone/utils.py:63-125) feeding hipie_vit_attn.
 *   qkv (B, gh*gw, 3, heads, hd) 16-bit;  tab_h (2*gh-1, hd), tab_w (2*gw-1, hd) 16-bit = rel_pos_h / rel_pos_w after the
 *   reference's linear re-interpolation to length 2*size-1;  rel_h (B*heads, gh, gh*gw) f32,  rel_w (B*heads, gh*gw, gw) f32.
 *   hd in {64, 80}; gh, gw <= 96.
 */
int hipie_vit_relpos(const void* qkv, const void* tab_h, const void* tab_w, float* rel_h, float* rel_w,
                     int B, int gh, int gw, int heads, int hd, int dtype, void* stream);

/*
 * Fused residual add + LayerNorm + cast:  s = x + delta;  res_out = s (optional);  norm_out = LN(s) * gamma + beta.
 * Replaces the add -> nn.LayerNorm -> cast chains of Block.forward (backbone/vit.py:212-230, eps 1e-6) and of the post-norm
 * residuals in DeformableTransformerEncoderLayer.forward (models/deformable_detr/deformable_transformer_dino.py:384-394).
 *   x (rows, C) x_dtype; delta (rows, C) delta_dtype or NULL; gamma, beta (C) f32; res_out (rows, C) x_dtype or NULL;
 *   norm_out (rows, C) norm_dtype.  C % 4 == 0, C <= 2048.  fp32 statistics (two-pass mean / centred variance).
 */
int hipie_add_layernorm(const void* x, const void* delta, const float* gamma, const float* beta, void* res_out,
                        void* norm_out, int64_t rows, int C, float eps, int x_dtype, int delta_dtype, int norm_dtype,
                        void* stream);

/*
 * hipie_vit_attn with the decomposed relative-position bias computed INSIDE the kernel from the (re-interpolated) tables
 * (get_rel_pos + add_decomposed_rel_pos, hipie/backbone/utils.py:63-125): bias_w[q, kx] = q . Rw[qx - kx + gw - 1] and
 * bias_h[q, ky] = q . Rh[qy - ky + gh - 1] are two MFMA products per wave in the prologue, so neither hipie_vit_relpos nor
 * its (B*heads, N, gh + gw) fp32 outputs are needed.  tab_h (2*gh-1, hd), tab_w (2*gw-1, hd) in `dtype`, entry
 * [q - k + size - 1].  Geometry: gw == 64, gh <= 64, hd in {64, 80} (the 1024-pixel global blocks); else HIPIE_EINVAL and the
 * caller uses hipie_vit_relpos + hipie_vit_attn.
 */
int hipie_vit_attn_fused(const void* qkv, const void* tab_h, const void* tab_w, void* out, int B, int gh, int gw, int heads,
                         int hd, float scale, int dtype, void* stream);

/*
 * ViT attention of one block with the decomposed relative-position bias computed inside the kernel -- global blocks (one
 * key tile = one key row of the gh x gw token grid) and the 14 x 14 windowed blocks (B = batch * windows; two key rows per
 * tile) alike.  Replaces: Attention.forward between the qkv and proj Linears (backbone/vit.py:69-80) including
 * get_rel_pos + add_decomposed_rel_pos (backbone/utils.py:63-125).
 * Operand contract (the two constants are folded into the weights once by the host, hipie_amd/modeling/vit.py):
 *   qkv   (B, gh*gw, 3, heads, hd) 16-bit with the q rows PRE-SCALED:  q' = scale * log2(e) * q
 *   tab_h (2*gh-1, hd), tab_w (2*gw-1, hd) 16-bit = rel_pos_h / rel_pos_w after the reference's linear re-interpolation to
 *         2*size-1 rows, DIVIDED by scale (entry [q - k + size - 1])
 *   so that q'.k + q'.tab_h' + q'.tab_w' is the attention logit in the exp2 domain.
 *   out   (B, gh*gw, heads*hd) 16-bit.   hd in {64, 80};  gw <= 96;  gh <= 84 when gw > 64 (LDS);  flags: HIPIE_ATTN_FAST.
 */
int hipie_vit_attn_rel(const void* qkv, const void* tab_h, const void* tab_w, void* out, int B, int gh, int gw, int heads,
                       int hd, int dtype, int flags, void* stream);

/*
 * hipie_add_layernorm with row maps, so that window_partition / window_unpartition (hipie/backbone/utils.py:16-60) around
 * the windowed ViT blocks (backbone/vit.py:214-225) cost no pass of their own:
 *   for every OUTPUT row j < out_rows:  r = out_src ? out_src[j] : j;   r < 0: norm_out[j] = 0 (a pad token);  otherwise
 *   s = x[r] + delta[delta_row ? delta_row[r] : r];  res_out[r] = s (optional);  norm_out[j] = LN(s) * gamma + beta.
 *   out_src (out_rows) int32: token row feeding each row of the (padded, window-ordered) output, or NULL;
 *   delta_row (rows of x) int32: where the residual branch of token r lives (the window-ordered attention output), or NULL.
 */
int hipie_add_layernorm_rows(const void* x, const void* delta, const float* gamma, const float* beta, void* res_out,
                             void* norm_out, int64_t out_rows, int C, float eps, int x_dtype, int delta_dtype,
                             int norm_dtype, const int32_t* delta_row, const int32_t* out_src, void* stream);

/*
 * hipie_add_layernorm with a second output  sum_out = norm_out + addend  (both `norm_dtype`, (rows, C)): the post-norm residual of
 * a deformable ENCODER layer that also prepares the next layer's query `src + pos`
 * (deformable_transformer_dino.py:384-394, with_pos_embed :380) in the same pass.
 */
int hipie_add_layernorm_sum(const void* x, const void* delta, const float* gamma, const float* beta, void* res_out,
                            void* norm_out, const void* addend, void* sum_out, int64_t rows, int C, float eps,
                            int x_dtype, int delta_dtype, int norm_dtype, void* stream);

/*
 * The post-norm residual of a DINO decoder layer (deformable_transformer_dino.py:418-450 == dino_decoder.py:222-268) when the
 * query stream is fp32 and the GEMMs take 16-bit operands:  n = LayerNorm(x + delta) is written once in fp32 (norm_out, the next
 * residual) and, optionally and in the same pass, as norm16_out = (aux)n and sum16_out = (aux)(n + addend)  (addend = the
 * positional query, `aux_dtype`): the inputs of the next value / FFN / box GEMMs and of the next attention's query GEMM.
 *   x (rows, C) f32; delta (rows, C) `delta_dtype`; aux_dtype f16 | bf16; C % 4 == 0, C <= 2048; fp32 two-pass statistics.
 */
int hipie_add_layernorm_dec(const float* x, const void* delta, const float* gamma, const float* beta, float* norm_out,
                            void* norm16_out, const void* addend, void* sum16_out, int64_t rows, int C, float eps,
                            int delta_dtype, int aux_dtype, void* stream);

/*
 * The two small MLP heads of a decoder layer, one launch each (they were a sine kernel + 2 GEMMs, and 3 GEMMs + the refinement):
 *   hipie_ref_point_mlp: query_pos = ref_point_head(get_sine_pos_embed(ref)) -- deformable_transformer_dino.py:484-490 / dino_decoder.py
 *     :126-131; ref (n, ref_stride >= 4) f32 boxes (x, y, w, h); dim_t (128) as in hipie_sine_embed; w1t (512, 256), w2t (256, 256) =
 *     the layers' weights TRANSPOSED to (in, out), biases, output (n, 256) all in `dtype`; sine features and the hidden activations are
 *     rounded to `dtype` where the GEMM path rounded them, fp32 accumulation.
 *   hipie_box_head: new_ref = sigmoid(bbox_embed(x) + inverse_sigmoid(ref)) -- :502-520 with MLP(256, 256, 4, 3); everything f32;
 *     w1t, w2t (256, 256), w3t (256, 4) transposed weights; x (n, 256), ref, out (n, 4).
 */
int hipie_ref_point_mlp(const float* ref, const float* dim_t, const void* w1t, const void* b1, const void* w2t, const void* b2,
                        void* out, int64_t n, int ref_stride, float scale, int dtype, void* stream);
int hipie_box_head(const float* x, const float* ref, const float* w1t, const float* b1, const float* w2t, const float* b2,
                   const float* w3t, const float* b3, float* out, int64_t n, float eps, void* stream);

/*
 * GroupNorm with 8 channels per group (GroupNorm(32, 256) of every conv + GN block after the backbone: input_proj of both heads,
 * deformable_detr.py:139-160; the pixel decoder's lateral / output convs and mask_features head, maskdino_encoder.py:262-300),
 * on the layout the tensor arrives in, with an optional per-channel pre-bias (y = GN(x + prebias[c])) and an optional ReLU.
 *   x, out (B, C, H*W) when channels_last == 0, (B, H*W, C) when 1;  channels_last == 2: x (B, H*W, C), out (B, C, H*W) -- the map
 *   leaves pixel-fastest (the mask_features operand of hipie_mask_einsum) through an LDS tile, no transposing copy; H*W % 64 == 0;
 *   workspace: 2 * B * groups * 512 floats (partial sums);
 *   gamma, beta (C) f32;  statistics fp32 (sum / sum of squares per group, biased variance, eps inside the sqrt).
 *   C == 8 * groups, 256 % groups == 0;  NCHW: H*W % 8 == 0.  Two launches, no atomics (deterministic).
 */
int hipie_group_norm(const void* x, const float* prebias, const float* gamma, const float* beta, void* out, float* workspace,
                     int B, int C, int HW, int groups, int channels_last, float eps, int relu, int x_dtype, int out_dtype,
                     void* stream);

/* out = (dtype)(a + b): a (n) f32, b (n) `dtype` f16 | bf16 -- `tgt + query_pos` rounded once to the GEMM operand type. */
int hipie_add_cast(const float* a, const void* b, void* out, int64_t n, int dtype, void* stream);

/*
 * Batched (class-aware) NMS for one batch of images, the device form of the per-image
 *   keep_indices = torchvision.ops.batched_nms(box_cxcywh_to_xyxy(box_pred), nms_scores, idxs, 0.7)
 * in HIPIE_IMG.inference (projects/HIPIE/hipie/hipie_img.py:626-629).
 *   boxes (B, Q, 4) f32 normalised cxcywh; classes (B, Q) int64; order (B, Q) int32 = stable descending argsort of the
 *   NMS scores (positions into Q); keep (B, Q) int32 out: surviving query indices in decreasing-score order, -1 padded;
 *   count (B) int32 out.  coordinate_trick 1: torchvision's <=4000-coordinate path (every box offset by
 *   class * (max_coordinate + 1), all pairs compared); 0: only same-class pairs on the raw boxes (its >4000 path).
 *   IoU = inter / (area_i + area_j - inter) > iou_threshold suppresses, evaluated with single IEEE roundings (no FMA),
 *   so keep/count are bit-exact against the CPU algorithm.  Q <= 1024 (the Q x Q bit matrix lives in LDS).
 */
int hipie_batched_nms(const float* boxes, const int64_t* classes, const int32_t* order, int32_t* keep, int32_t* count,
                      int B, int Q, float iou_threshold, int coordinate_trick, void* stream);

/*
 * Instance-mask finalisation for one image:  F.interpolate(x`up`, bilinear, align_corners=False) -> sigmoid -> "> threshold"
 * -> crop to (crop_h, crop_w)  (hipie_img.py:693-699)  -> F.interpolate(size=(out_h, out_w), mode="nearest") -> byte
 * (segmentation_postprocess, hipie/models/ddetrs.py:1065-1070), in one pass over the stride-`up` logits.
 *   masks (*, hm, wm) `dtype`; qidx (n) int32 rows of `masks` to process (NULL: rows 0..n-1); out (n, out_h, out_w) uint8.
 */
int hipie_mask_finalize(const void* masks, int dtype, const int32_t* qidx, int n, int hm, int wm, int up, int crop_h,
                        int crop_w, int out_h, int out_w, float threshold, uint8_t* out, void* stream);

/*
 * Fused semantic + panoptic maps of one image -- the tensor part of the detection tail of HIPIE_IMG.inference
 * (hipie_img.py:716-748), semantic_inference (:870-878) and panoptic_inference (:473-505):
 *   sig[q] = sigmoid(resize(masks[q]))  with resize = bilinear x`up` -> crop (crop_h, crop_w) -> bilinear to (out_h, out_w);
 *   sem[c] = sum_q cls[q, c] * sig[q];   pan_idx = argmax_q pscore[q] * sig[q] over pscore > 0 (first index on ties, -1 if
 *   none);  pan_own = sig[pan_idx] >= 0.5;   area[q] += #pixels with sig[q] >= 0.5.
 * The N x out_h x out_w sigmoid tensor never exists: each value is produced in registers as an MFMA B-operand element.
 *   masks (N, hm, wm) f32; cls_hi / cls_lo (ceil32(C) rounded to 32/96/160, Npad) bf16: class probabilities TRANSPOSED,
 *   zero padded, split as value = hi + lo (cls_lo unused and may be NULL for precision 1); pscore (Npad) f32 (<= 0: query
 *   not kept; padding -1); sem (C, out_h, out_w) f32; pan_idx (out_h, out_w) int32; pan_own (out_h, out_w) uint8;
 *   area (Npad) int32, zero-initialised by the caller.  Npad % 16 == 0, C <= 160.
 *   precision 0: bf16 hi+lo operands, 3 MFMAs per product (~2^-16);  1: plain bf16.
 */
int hipie_sem_pan(const float* masks, const void* cls_hi, const void* cls_lo, const float* pscore, float* sem,
                  int32_t* pan_idx, uint8_t* pan_own, int32_t* area, int N, int Npad, int C, int hm, int wm, int up,
                  int crop_h, int crop_w, int out_h, int out_w, int precision, void* stream);

/*
 * Sine embedding of reference points / boxes: the query_pos input of both DINO decoders.
 * Replaces: get_sine_pos_embed (models/deformable_detr/deformable_transformer_dino.py:636-670) == gen_sineembed_for_position
 *           (models/maskdino/utils/utils.py:74-100), ~21 eager launches per decoder layer.
 *   ref (n, ref_stride >= n_coord) f32, n_coord 2 | 4 coordinates (x, y[, w, h]);  dim_t (num_pos_feats) f32 = T^(2*floor(i/2)/F)
 *   (computed by the caller with the reference's formula);  out (n, n_coord * num_pos_feats) `out_dtype`, coordinate blocks in
 *   the reference's order (y, x, w, h);  value = sin | cos (even | odd feature) of ref * scale / dim_t, fp32.
 */
int hipie_sine_embed(const float* ref, const float* dim_t, void* out, int64_t n, int n_coord, int num_pos_feats, int ref_stride,
                     float scale, int out_dtype, void* stream);

/*
 * Iterative box refinement: out = sigmoid(delta + inverse_sigmoid(ref)), inverse_sigmoid with the reference's clamps (eps 1e-5).
 * Replaces: deformable_transformer_dino.py:502-520 / dino_decoder.py:150-160 + util/misc.py:493-497 (8 eager launches per layer).
 *   delta (n) `delta_dtype`, ref (n) f32, out (n) f32 (n = boxes * 4).
 */
int hipie_box_refine(const void* delta, const float* ref, float* out, int64_t n, float eps, int delta_dtype, void* stream);

/*
 * The linears of the path as one hand-written MFMA GEMM:   out = epilogue( alpha * A (M x K) . W^T + bias ),  W (N x K) as
 * torch.nn.Linear stores it.  Replaces: nn.Linear / F.linear of Attention.qkv / proj and Mlp.fc1 / fc2 (hipie/backbone/vit.py:
 * 67-83, 193-197, 212-230), BertLayer's dense layers (transformers BertModel via models/deformable_detr/bert_model.py:54-58), the
 * FFNs and projections of the deformable encoder / decoder layers (models/deformable_detr/deformable_transformer_dino.py:378-394,
 * 418-450; ops/modules/ms_deform_attn.py:95-99) -- which the reference runs in fp32.
 *   in_fmt   HIPIE_F16: A (M, K) and W (N, K) fp16, one MFMA per product;
 *            HIPIE_HL8: both operands split fp16 (rows of 2K fp16); the product is W_lo.A_hi + W_hi.A_lo + W_hi.A_hi with fp32
 *            accumulation = fp32-class results (every fp16 x fp16 product is exact in fp32; the dropped lo x lo term is 2^-22).
 *            HIPIE_F32: A (M, K) plain fp32 rows (lda in fp32 elements, a multiple of 4) against an HL8 W: the kernel splits the A
 *            fragments into hi / lo after the LDS read (an fp32 group of 8 takes the 32 bytes of an HL8 group) -- the same results
 *            as hipie_to_hl8 followed by the HL8 form, bit for bit, without the conversion launch.
 *   lda, ldw row strides in fp16 elements (multiples of 8);  K a multiple of 64 (F16) / 32 (HL8);  N a multiple of 8
 *   bias     (N) f32 or NULL;   resid (M, N) f32 with row stride ldr, or NULL
 *   epilogue y = alpha * acc + bias;  act 1: exact-erf GELU(y), 2: ReLU(y), 3: QuickGELU y * sigmoid(1.702 y) (the OpenAI CLIP towers of
 *            MaskCLIP, hipie/open_vocab/clip.py via open_clip);  y = (y + resid) * oscale
 *   out      (M, N) in out_fmt HIPIE_F32 | HIPIE_F16 | HIPIE_HL8 (row stride ldo in elements of that format; HL8: fp16 elements >= 2N):
 *            the HL8 form is directly the A operand of a following hipie_gemm.  out may alias resid (in-place residual update).
 *   out_row  (M) int32 or NULL: product row m is written to output row out_row[m] and takes its residual from resid row out_row[m];
 *            negative entries are dropped -- window_unpartition (hipie/backbone/utils.py:40-60) as the store index of the projection.
 * All pointers 16-byte aligned.  Deterministic (fixed accumulation order).
 */
int hipie_gemm(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* resid, int64_t ldr,
               void* out, int64_t ldo, const int32_t* out_row, int M, int N, int K, int in_fmt, int out_fmt, int act, float alpha,
               float oscale, void* stream);

/*
 * hipie_gemm (split operands, N = 256) with the post-norm residual LayerNorm of the deformable encoder layer in its epilogue:
 *     y = LayerNorm_256( resid + alpha * A . W^T + bias ) * gamma + beta;   out = y as fp32 rows;  out_hl8 (or NULL) = y as HIPIE_HL8 rows
 * Replaces `src = norm1(src + dropout1(output_proj(msda)))` (models/deformable_detr/deformable_transformer_dino.py:387-389 with
 * ops/modules/ms_deform_attn.py:114) as ONE launch: N = 256 is one column tile, so the workgroup that owns 256 rows holds them whole and the
 * row statistics (two passes, fp32, as hipie_add_layernorm_dec) are two LDS exchanges away; the projection never exists in memory.
 * A in_fmt HIPIE_HL8 | HIPIE_F32 (lda as in hipie_gemm); W (256, 2K) HL8; bias (256) or NULL; resid (M, 256) fp32, required, may alias out
 * (every residual value of a row is read before any value of that row is stored); gamma, beta (256); ldo >= 256 (fp32 elements),
 * ldo_hl8 >= 512 (fp16 elements).  K a multiple of 32.  All pointers 16-byte aligned.
 */
int hipie_gemm_ln(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* resid, int64_t ldr,
                  const float* gamma, const float* beta, float eps, float* out, int64_t ldo, void* out_hl8, int64_t ldo_hl8, int M, int K,
                  int in_fmt, float alpha, void* stream);

/*
 * hipie_gemm whose product row m READS operand row a_row[m] (0 <= a_row[m] < a_rows; A is a_rows x K; split formats only; the whole operand
 * below 4 GiB): the linears of the windowed ViT blocks run over the REAL tokens only and pick them out of / scatter them into (out_row) the
 * zero-padded window layout (window_partition pads 64 x 64 tokens to 70 x 70: 19.6 % more rows, whose qkv is the bias and whose projection
 * is discarded -- hipie/backbone/utils.py:16-60, hipie/backbone/vit.py:212-230).
 */
int hipie_gemm_gather(const void* A, int64_t lda, int64_t a_rows, const int32_t* a_row, const void* W, int64_t ldw, const float* bias,
                      const float* resid, int64_t ldr, void* out, int64_t ldo, const int32_t* out_row, int M, int N, int K, int in_fmt,
                      int out_fmt, int act, float alpha, float oscale, void* stream);

/*
 * hipie_vit_attn_rel on SPLIT operands (fp32-class logits): qkv (B, gh*gw, 3, heads, hd) as HIPIE_HL8 rows (2 * 3 * heads * hd fp16
 * per token) with the q rows pre-scaled by scale * log2(e); tab_h (2*gh-1, hd), tab_w (2*gw-1, hd) = the rel-pos tables / scale, HL8.
 * Scores and both bias terms are three-product sums (q_lo.k_hi + q_hi.k_lo + q_hi.k_hi), the probabilities one fp16, V both halves;
 * classic running maximum, fp32 row sums.  out (B, gh*gw, heads*hd) as HIPIE_HL8: the A operand of the projection hipie_gemm.
 * Replaces: Attention.forward between the qkv and proj Linears (hipie/backbone/vit.py:69-80) + add_decomposed_rel_pos / get_rel_pos
 * (hipie/backbone/utils.py:63-125) at the reference's fp32 accuracy.  Geometry: head_dim 64 | 80; 14-wide windows, grids up to 64 wide.
 */
int hipie_vit_attn_split(const void* qkv, const void* tab_h, const void* tab_w, void* out, int B, int gh, int gw, int heads,
                         int hd, void* stream);

/*
 * Fused attention for the TRAINING step's global ViT blocks: O = softmax(q' k'^T) v and its gradients, no (heads, N, N) tensor
 * in HBM.  Replaces Attention.forward between the qkv and proj Linears (hipie/backbone/vit.py:69-80) and its autograd, with
 * add_decomposed_rel_pos (hipie/backbone/utils.py:96-125) folded into the operands by the caller:
 *   q' = [scale q, rel_h(q, :), rel_w(q, :), 0..], k' = [k, onehot(key row), onehot(key column), 0..]   -- 224 columns; v, O: 80 columns
 * Every operand is an fp16 pair given as two planes (hi = fp16(x), lo = fp16(x - hi), row-major, contiguous); products are three fp16 MFMAs
 * accumulated in fp32 (the library's split form).  N a multiple of 128; BH = batch x heads <= 65535.
 *   forward:   q', k' (BH, N, 224) pairs, v (BH, N, 80) pair -> out (BH, N, 80) f32, lse (BH, N) f32 (log of the softmax denominator + row max)
 *   backward:  the same operands, v and dO (scaled by the caller into fp16's range) as (BH, N, 96) pairs (columns 80.. zero), lse, delta (BH, N)
 *              f32 = rowsum(dO * out) -> dq' (BH, N, 224) f32 (columns 80.. are d rel_h | d rel_w), dk (BH, N, 80) f32, dv (BH, N, 80) f32
 */
int hipie_attn_train_forward(const void* q_hi, const void* q_lo, const void* k_hi, const void* k_lo, const void* v_hi, const void* v_lo,
                             void* out, void* lse, int BH, int N, void* stream);
/* fp32 rows (row stride ldx elements) -> the two fp16 planes (rows x Cp, contiguous) of an operand of the two entries below: hi = fp16(s x),
 * lo = fp16(s x - hi), columns C.. zero; s = *scale (a DEVICE float, e.g. the power of two that lifts dO into fp16's range) or 1 when NULL. */
int hipie_to_f16_pair(const void* x, int64_t ldx, void* hi, void* lo, int64_t rows, int C, int Cp, const void* scale, void* stream);
int hipie_attn_train_backward(const void* q_hi, const void* q_lo, const void* k_hi, const void* k_lo, const void* v_hi, const void* v_lo,
                              const void* do_hi, const void* do_lo, const void* lse, const void* delta, void* dq, void* dk, void* dv,
                              int BH, int N, void* stream);

/* dst + rows[i] * ld_bytes <- the row_bytes bytes at src_row, for i < n_rows (negative entries are skipped): one constant row into a
 * listed set of rows.  Used to write the HL8 qkv BIAS row into the padding rows of a window-layout qkv buffer -- the qkv of a padding token
 * of window_partition is the bias, its LayerNorm output being zero (hipie/backbone/utils.py:29-37, vit.py:67-71).  16-byte units. */
int hipie_fill_rows(void* dst, int64_t ld_bytes, const int32_t* rows, int64_t n_rows, const void* src_row, int64_t row_bytes, void* stream);

/* rows of `x_dtype` (HIPIE_F32 | HIPIE_F16) values -> HIPIE_HL8 rows of scale * x (K a multiple of 8; ldx in elements of x, ldo in
 * fp16 elements >= 2K): the generic producer of split operands (the LayerNorm / GEMM epilogues emit HL8 directly). */
int hipie_to_hl8(const void* x, int64_t ldx, void* out, int64_t ldo, int64_t rows, int K, int x_dtype, float scale, void* stream);

/* x (rows x C fp32 values, row stride ldx elements) -> out (C x 2 rows_p) HIPIE_HL8 rows of scale * x^T: the transpose as a split operand in ONE
 * pass (a strided transpose copy + hipie_to_hl8 otherwise), columns rows <= m < rows_p zero (rows_p a multiple of 8; 32 for a GEMM operand).
 * The producer of both operands of the weight gradient dW = dy^T . x of a Linear (torch.nn.functional.linear's backward in the reference's
 * training step, hipie/backbone/vit.py:67-83 and every other Linear): the contraction runs over the token rows. */
int hipie_to_hl8_t(const void* x, int64_t ldx, void* out, int64_t ldo, int64_t rows, int C, int64_t rows_p, float scale, void* stream);

/*
 * hipie_gemm on n_outer x n_inner independent problems in ONE launch (split-fp16 HL8 operands only, no bias / residual / activation):
 * problem (o, i) reads A + o * a_outer + i * a_inner (fp16 elements), W + o * w_outer + i * w_inner, writes out + o * o_outer +
 * i * o_inner (elements of out_fmt: fp32, or fp16 units for HL8).  Used for the image -> text direction of the vision-language fusion
 * in the split policy: per (image, head)  S = Q_h . K_h^T  and  out_h = P_h . V_h  (models/deformable_detr/fuse_helper.py:77-121).
 * n_outer * n_inner <= 65535; offsets keep 16-byte alignment.
 */
int hipie_gemm_batched(const void* A, int64_t lda, int64_t a_outer, int64_t a_inner, const void* W, int64_t ldw, int64_t w_outer,
                       int64_t w_inner, void* out, int64_t ldo, int64_t o_outer, int64_t o_inner, int n_outer, int n_inner, int M, int N,
                       int K, int out_fmt, float alpha, void* stream);

/*
 * hipie_gemm_batched whose epilogue is the masked row softmax (N <= 256: the whole row is one column tile) -- the logits of the image -> text
 * fusion attention never reach HBM:  P[b][h] = softmax_j( clamp(alpha * A . W^T, +-clamp) over the columns j < L with mask[b][j] ) written as
 * HIPIE_HL8 (the A operand of the P . V product that follows); masked and padding columns (L <= j < N) are 0, a row without a valid column is 0.
 * mask (n_outer, L) uint8 or NULL.  Replaces hipie_gemm_batched(F32 out) + hipie_softmax_hl8 for texts of up to 256 tokens
 * (attn_weights_v of BiMultiHeadAttention.forward, models/deformable_detr/fuse_helper.py:77-111).
 */
int hipie_gemm_batched_softmax(const void* A, int64_t lda, int64_t a_outer, int64_t a_inner, const void* W, int64_t ldw, int64_t w_outer,
                               int64_t w_inner, void* out, int64_t ldo, int64_t o_outer, int64_t o_inner, int n_outer, int n_inner, int M,
                               int N, int K, const unsigned char* mask, int L, float clamp, float alpha, void* stream);

/*
 * The image -> text direction of BiMultiHeadAttention (models/deformable_detr/fuse_helper.py:62-139) with the two visual-side projections
 * FOLDED into the text side, so that nothing of width embed_dim (2048) is ever computed per visual token:
 *     logits[b,h][i,j] = (W_q,h x_i + b_q,h) . k_{b,j,h} = x_i . M_{b,h,j} + c_{b,h,j},      M = k_h W_q,h  (L x 256),  c = k_h . b_q,h
 *     out_v[b][i]      = W_o concat_h( P_h[i,:] V_h ) + b_o = sum_h P_h[i,:] . U_{b,h} + b_o,   U = V_h W_o,h^T  (L x 256)
 * hipie_gemm_batched_softmax_bias: hipie_gemm_batched_softmax with col_bias (n_outer * n_inner, N) fp32 added to the scaled accumulator BEFORE
 * the clamp (the reference clamps q . k including the bias); A may be shared by the inner index (a_inner = 0: every head reads x).
 * hipie_gemm_batched_resid: hipie_gemm_batched (fp32 out) with hipie_gemm's bias (shared) and residual epilogue; resid moves with the
 * problem index like out (r_outer / r_inner in fp32 elements).  One launch each per fusion layer.
 */
int hipie_gemm_batched_softmax_bias(const void* A, int64_t lda, int64_t a_outer, int64_t a_inner, const void* W, int64_t ldw, int64_t w_outer,
                                    int64_t w_inner, void* out, int64_t ldo, int64_t o_outer, int64_t o_inner, int n_outer, int n_inner,
                                    int M, int N, int K, const unsigned char* mask, int L, const float* col_bias, float clamp, float alpha,
                                    void* stream);
int hipie_gemm_batched_resid(const void* A, int64_t lda, int64_t a_outer, int64_t a_inner, const void* W, int64_t ldw, int64_t w_outer,
                             int64_t w_inner, const float* bias, const float* resid, int64_t ldr, int64_t r_outer, int64_t r_inner, float* out,
                             int64_t ldo, int64_t o_outer, int64_t o_inner, int n_outer, int n_inner, int M, int N, int K, float alpha,
                             void* stream);

/*
 * Row softmax of fp32 logits written as an HL8 operand:  P[r, :Lp] = softmax over the L valid columns of clamp(S[r, :L], +-clamp) with
 * columns masked by mask[r / rows_per_batch, :] (uint8, 1 = keep; NULL = all) or beyond L set to 0.  S rows lds floats apart, P rows
 * ldp fp16 elements apart (>= 2 * Lp);  Lp a multiple of 8, <= 4096.  A row with no valid column gives zeros (DEVIATION, never reached
 * on the path: the reference's additive -9e15 mask makes such a row a UNIFORM average over all L columns; a caption always keeps [CLS]
 * and [SEP], so every text has valid tokens -- same convention in hipie_attn_f32 / hipie_attn_split / hipie_flash_attn).
 * Replaces: attn_weights_v = softmax(clamp(attn_weights) + attention_mask) of BiMultiHeadAttention.forward (fuse_helper.py:97-111).
 */
int hipie_softmax_hl8(const float* S, int64_t lds, void* P, int64_t ldp, int64_t rows, int L, int Lp, const unsigned char* mask,
                      int64_t rows_per_batch, float clamp, void* stream);

/*
 * EXACT fp32 softmax attention for the small attentions of the path (fp32 operands, fp32 FMA products, fp32 softmax):
 *     out[b,i,h,:] = softmax_j( scale * q[b,i,h,:].k[b,j,h,:] + (key_mask[b,j] ? 0 : -inf) ) . v[b,j,h,:]
 * q, k, v fp32 with element strides (batch, token); head h is the head_dim contiguous elements at h * head_dim (so the three may be
 * column blocks of ONE projection output); out (B, Nq, H * head_dim) fp32 contiguous; head_dim 32 | 64; key_mask (B, Nk) uint8 or NULL
 * (a row whose keys are all masked gives zeros).  Deterministic.
 * Replaces: BertSelfAttention's matmul -> softmax -> matmul (transformers, behind models/deformable_detr/bert_model.py:54-58) and
 *           nn.MultiheadAttention's core in the decoder layers (models/deformable_detr/deformable_transformer_dino.py:418-432,
 *           models/maskdino/transformer_decoder/dino_decoder.py:222-240), which the reference computes in fp32.
 */
int hipie_attn_f32(const float* q, const float* k, const float* v, const unsigned char* key_mask, float* out, int B, int H, int Nq, int Nk,
                   int head_dim, int64_t q_sb, int64_t q_st, int64_t k_sb, int64_t k_st, int64_t v_sb, int64_t v_st, float scale,
                   void* stream);

/*
 * hipie_attn_f32 with a mask per QUERY row: query_mask (B, Nq, Nk) uint8, 1 = row i may attend to key j (NULL = all).  The mask tokens of
 * MaskCLIP: every mask token sees only the image patches its mask covers, plus the class token.
 * Replaces: the attn_mask path of open_clip's ResidualAttentionBlock as the reference's MaskCLIP drives it (hipie/clip.py:158-215, :249-262) --
 * logits, masked_fill(-inf), softmax, P . V of the mask tokens' rows (hipie_amd/open_vocab.py: ResidualAttentionBlock.forward_mask_rows).
 */
int hipie_attn_f32_rows(const float* q, const float* k, const float* v, const unsigned char* query_mask, float* out, int B, int H, int Nq,
                        int Nk, int head_dim, int64_t q_sb, int64_t q_st, int64_t k_sb, int64_t k_st, int64_t v_sb, int64_t v_st, float scale,
                        void* stream);
/* the same on the matrix pipe at fp32-class accuracy (the arithmetic of hipie_attn_split; what the product runs: one thread per query row of the
 * exact kernel is 0.83 ms per layer at 150 mask tokens per image, this one 0.05 ms) */
int hipie_attn_split_rows(const float* q, const float* k, const float* v, const unsigned char* query_mask, float* out, int B, int H, int Nq,
                          int Nk, int head_dim, int64_t q_sb, int64_t q_st, int64_t k_sb, int64_t k_st, int64_t v_sb, int64_t v_st, float scale,
                          void* stream);

/*
 * The same attention (same operands, same output, same mask semantics) on the matrix pipe at fp32-CLASS accuracy: q (pre-multiplied by
 * scale * log2 e), k, v are split in the kernel into fp16 pairs (22 mantissa bits), logits = three-product sums with fp32 accumulation, the
 * probabilities an fp16 pair, O += V_hi.(P_hi + P_lo) + V_lo.P_hi -- the arithmetic of hipie_vit_attn_split.  Within 2e-6 of
 * hipie_attn_f32; what the split policy runs (27 launches per step: 4.4 -> ~1 ms).  Replaces: the same reference code as hipie_attn_f32.
 */
int hipie_attn_split(const float* q, const float* k, const float* v, const unsigned char* key_mask, float* out, int B, int H, int Nq, int Nk,
                   int head_dim, int64_t q_sb, int64_t q_st, int64_t k_sb, int64_t k_st, int64_t v_sb, int64_t v_st, float scale,
                   void* stream);

/*
 * The FFN of a deformable encoder layer in ONE launch at the split policy's accuracy:  out = W2 . relu(W1 . x + b1) + b2.
 *   x    (M, 2 D) HIPIE_HL8 rows (row stride ldx in fp16 elements);  w1 (F, 2 D) HL8;  b1 (F) fp32;
 *   w2p  (D, 2 F) HL8 of W2 with its COLUMNS permuted inside every block of 16 into the order 0-3, 8-11, 4-7, 12-15 (the order in which
 *        the MFMA C layout of the hidden tile hands its rows to the next product);  b2 (D) fp32;  out (M, D) fp32, row stride ldo.
 *   D = 256, F = 2048 (the shipped sizes of both deformable encoders).
 * Each product is the three-term split sum with fp32 accumulation; the hidden activations (ReLU, then split into an fp16 pair exactly as
 * hipie_gemm's HL8 epilogue does) never leave the registers -- no (M, F) tensor exists.
 * Replaces: linear2(dropout(relu(linear1(src)))) of DeformableTransformerEncoderLayer.forward_ffn
 *           (models/deformable_detr/deformable_transformer_dino.py:378-394) and MSDeformAttnTransformerEncoderLayer.forward_ffn
 *           (models/maskdino/pixel_decoder/maskdino_encoder.py:142-157), fp32 nn.Linear in the reference.
 */
int hipie_ffn_fused(const void* x, int64_t ldx, const void* w1, const float* b1, const void* w2p, const float* b2, float* out, int64_t ldo,
                    int M, int D, int F, void* stream);

/*
 * 3 x 3 convolution (stride 1, padding 1, no groups) at the split policy's accuracy as an IMPLICIT GEMM of hipie_gemm's kernel: K = 9 taps x C.
 * The caller provides the input on a zero-PADDED pixel grid: x = row 0 of a (B, H + 2, W + 2, C) channels-last tensor (rows = pixels, fp32 or
 * HIPIE_HL8, row stride ldx elements) with at least Wp + 1 readable rows before and after it (Wp = W + 2); tap (dy, dx) of output row r reads
 * input row r + (dy - 1) * Wp + (dx - 1).  w (N, 9 * C) as HIPIE_HL8 with k = (3 * dy + dx) * C + c, bias (N) fp32 or NULL; out = `rows`
 * rows on the SAME padded grid (the border rows are meaningless and cropped by the caller), fp32 or HL8, row stride ldo; act 0 | 1 GELU | 2 ReLU.
 * C a multiple of 32.  Three products per tap and channel with fp32 accumulation, like every split linear.
 * Replaces: the fp32 nn.Conv2d 3 x 3 of the MaskDINO pixel decoder's FPN output (detectron2 Conv2d + GN, maskdino_encoder.py:294-310) and of
 * MaskHeadSmallConv (ddetrs_dn.py:1633-1689), MIOpen implicit-GEMM fp32 kernels before (1.3 ms per 256 -> 256 map at 128 x 128, bs 8).
 */
int hipie_conv3x3_split(const void* x, int64_t ldx, const void* w, const float* bias, void* out, int64_t ldo, int64_t rows, int Wp, int C,
                        int N, int in_fmt, int out_fmt, int act, void* stream);

/*
 * Row-wise top-k of fp32 scores, k <= 1024: idx_out (rows, k) int64 in descending value order (ascending index among equal values;
 * NaN sorts as the largest value), val_out (rows, k) f32 or NULL.  One launch, hipGraph-replay safe.
 * Replaces: torch.topk in the two-stage query selections (models/deformable_detr/deformable_transformer_dino.py:222-230, 900 of Nv;
 * models/maskdino/transformer_decoder/maskdino_decoder.py:413-426, 300 of Nv) and in HIPIE_IMG.inference (hipie_img.py:640-648).
 */
int hipie_topk(const float* x, int64_t row_stride, int rows, int n, int k, int64_t* idx_out, float* val_out, void* stream);

/* device-side self-test helpers used by tests/ to pin the MFMA / LDS-transpose lane layouts this library assumes.
 *   which 0: D = A(32x16) . B(16x32) with v_mfma_f32_32x32x16_bf16, operands loaded with the layouts documented in
 *            csrc/mfma.h; out (32,32) f32.   which 1: ds_read_b64_tr_b16 of a (64,16) bf16 tile; out (64,4) f32 per lane.
 *   a, b: bf16 (as uint16) inputs. */
int hipie_selftest(int which, const uint16_t* a, const uint16_t* b, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HIPIE_MI355_H */
