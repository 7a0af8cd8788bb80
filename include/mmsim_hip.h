/* mmsim_hip.h -- C ABI of libmmsim_hip.so, the MI355X (gfx950) compute library under the
 * MultimodalSimilar drop-in modules (arcface / nlp_classifier / cv_classifier / multimodal_classifier).
 *
 * The reference (forrestsocool/MultimodalSimilar) has no native layer: its hot path is Python over
 * torch / transformers / timm.  Each entry point below names the reference lines whose arithmetic it
 * replaces.  Conventions:
 *   - every function returns 0 on success, non-zero on error; mmsim_last_error() gives the message;
 *     arguments are validated BEFORE anything is launched;
 *   - all pointers are device pointers (HBM) unless stated; nothing is allocated or freed here;
 *   - `stream` is a hipStream_t (0 = default stream); launches are asynchronous on it;
 *   - bf16 buffers are row-major with a leading dimension in ELEMENTS; rows are 16-byte aligned;
 *   - dropout uses a counter-based generator keyed by (seed, stream_id, element index): the same triple
 *     in forward and backward reproduces the mask, nothing is stored.
 * The Python binding (multimodalsimilar_amd/_lib.py) is generated from this file; tests check that the
 * library exports every symbol declared here.
 */
#ifndef MMSIM_HIP_H
#define MMSIM_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

/* ---- library state -------------------------------------------------------------------------- */
/* 300 = this header (fp16 forward tensors in the image-tower entry points, mmsim_embed_ln_bwd2, mmsim_adamw_step2); a binding built
 * from an older header must check mmsim_version() < 300 before passing bf16 image tensors. */
int mmsim_version(void);
int mmsim_device_count(void);
const char* mmsim_last_error(void);
/* Deterministic (verification) mode, process-wide: every cross-workgroup sum in a fixed order (partial slabs reduced by one
 * workgroup per output, no split-K, single-slice pooling, serial embedding scatter).  Results are then bit-identical from run
 * to run; the default mode keeps the faster forms whose fp32 atomic adds arrive in varying order.  Costs step time. */
/* hipGraph replay of a training step: dropout seeds are kernel arguments, which a captured graph freezes.  With a device word
 * registered here every dropout kernel (embeddings, hidden / attention dropout of modeling_bert.py:68-108,111-136,289-293, the
 * Dropout(0.5) of cv_classifier.py:52) adds *dev_ptr to its seed argument at run time.  NULL (default) = seeds are the arguments. */
int mmsim_set_step_seed_ptr(const unsigned long long* dev_ptr);
int mmsim_set_deterministic(int on);
int mmsim_get_deterministic(void);


/* ---- dense products: torch F.linear / nn.Linear / 1x1 conv and their backward -------------------
 * C[M,N] = alpha * op(A)[M,K] op(B)[K,N] (+bias[N]) with a fused epilogue, bf16 inputs, fp32 accumulate.
 *   trans_a = 0: A stored [M][K] (lda >= K)      trans_a = 1: A stored [K][M] (lda >= M)
 *   b_kmajor = 1: B stored [N][K] (nn.Linear weight layout)   b_kmajor = 0: B stored [K][N]
 * Replaces: BERT dense layers (transformers modeling_bert.py:174-176, 289-293, 334-337, 347-351, 457-463
 * as called from transformer_emb.py:20-24), F.linear in arcface.py:47, timm 1x1 convs under
 * cv_classifier.py:49, nn.Linear cv_classifier.py:53 -- and autograd's dgrad / wgrad of each.
 * epilogue: 0 none | 1 GELU(erf): aux_out <- pre-activation, C <- gelu | 2 C <- acc * gelu'(aux_in)
 *           | 3 C <- acc + aux_in | 4 tanh | 5 row-fix (f32 C, see mmsim_arcface_rowfix; bias then holds [2][M] row vectors)
 *           | 6 GELU(erf): aux_out <- gelu'(pre-activation), C <- gelu | 7 C <- acc * aux_in  (6 / 7: the pair the text tower
 *             trains with: BertIntermediate, modeling_bert.py:334-337, and its backward as one multiply).
 * split_k > 1 adds atomically into an f32 C (gradient buffers);
 * accumulate != 0 makes C += result for split_k == 1 as well (f32 C only).
 * K-major operands must be zero in [K, round_up(K,8)) of each row when K % 8 != 0. */
int mmsim_gemm_bf16(int trans_a, int b_kmajor, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                    void* C, int ldc, int c_is_f32, const float* bias, int epilogue, const void* aux_in,
                    void* aux_out, int ld_aux, float alpha, int split_k, int accumulate, void* stream);

/* Two weight gradients of one layer in ONE launch: C1[M1,N] += A1^T B1 and C2[M2,N] += A2^T B2 (A = dY stored [K][M],
 * B = X stored [K][N], f32 C, split-K atomics; M1, M2, N multiples of 256, K of 64).  The attention-output (16 tiles) and
 * q|k|v (48 tiles) weight gradients of a BERT layer (autograd of modeling_bert.py:174-176, 289) fill the chip together. */
int mmsim_gemm_bf16_wgrad_pair(int M1, int M2, int N, int K, const void* A1, int lda1, const void* B1, int ldb1, float* C1,
                               int ldc1, const void* A2, int lda2, const void* B2, int ldb2, float* C2, int ldc2, int split_k,
                               void* stream);

/* Weight gradient AND bias gradient of a dense layer from one pass over dY (autograd of nn.Linear, modeling_bert.py:334-351 for
 * intermediate.dense):  C[M,N] (fp32) += A^T B  with A = dY stored [K][M], B = X stored [K][N], split-K atomics, and
 * colsum[M] (fp32) += sum_k A[k][m]  (two extra MFMAs per 32-deep step against an all-ones operand in the tile-column-0 blocks)
 * -- instead of mmsim_gemm_bf16 + a separate mmsim_colsum_bf16 pass over dY.  Shapes: mmsim_gemm_bf16_wgrad_colsum_eligible
 * (M, N multiples of 256, K / split multiples of 64, enough tiles for the pipelined 256 x 256 kernel). */
int mmsim_gemm_bf16_wgrad_colsum_eligible(int M, int N, int K, int split_k);
int mmsim_gemm_bf16_wgrad_colsum(int M, int N, int K, const void* A, int lda, const void* B, int ldb, float* C, int ldc,
                                 float* colsum, int split_k, void* stream);

/* 1x1 conv whose input is the previous BatchNorm + SiLU (+ squeeze-excite gate) applied while the operand is
 * staged: x -> silu(xf_scale[c] x + xf_shift[c]) * xf_gate[pixel / xf_hw, c]  (gate may be NULL); with xf_scale = xf_shift =
 * NULL the operand is already activated (mmsim_pool_bn_act_store) and only x -> x * gate remains.
 *   xf_operand 1 (forward): C[P,Cout] = xf(A)[P,Cin] B[Cout,Cin]^T;  2 (wgrad): C[Cout,Cin] (+)= A[P,Cout]^T xf(B)[P,Cin].
 * Replaces timm's conv_pwl / conv_pw after bn+act+se inside the MBConv blocks under cv_classifier.py:49.
 * Element types (image-tower contract, see the EfficientNet section): xf_operand 1 -- A (activation), B (weight shadow) and a
 * non-f32 C are fp16, the MFMAs run on fp16; xf_operand 2 -- A = the bf16 gradient, B = the fp16 activation (converted to bf16
 * while it is staged), C f32. */
int mmsim_gemm_bf16_xf(int xf_operand, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C,
                       int ldc, int c_is_f32, const float* xf_scale, const float* xf_shift, const float* xf_gate,
                       int xf_hw, int split_k, int accumulate, void* stream);

/* mmsim_gemm_bf16 with an explicit element format: fmt 0 = bf16 (identical to mmsim_gemm_bf16); fmt 1 = fp16 A and B, C fp16 or
 * f32, forward layout only (trans_a 0, b_kmajor 1, no split-K; an fp16 C takes no bias / epilogue) -- the image tower's 1x1
 * convs and the fc layer of cv_classifier.py:53 on fp16 activations; fmt 2 = weight-gradient layout only (trans_a 1, b_kmajor 0,
 * f32 C): A is the bf16 gradient, B the fp16 activation, converted to bf16 while it is staged (autograd of the same layers). */
int mmsim_gemm_fmt(int fmt, int trans_a, int b_kmajor, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                   void* C, int ldc, int c_is_f32, const float* bias, int epilogue, const void* aux_in, void* aux_out,
                   int ld_aux, float alpha, int split_k, int accumulate, void* stream);

/* Paired launch of a layer's two backward products.  Between _begin and _end (per host thread, not nestable) the
 * mmsim_gemm_bf16 / mmsim_gemm_bf16_xf calls that would run the generic (ragged-shape) kernel are parked instead of launched
 * (at most two); _end launches them: a weight gradient (trans_a = 1, b_kmajor = 0, optionally xf_operand 2) followed by a
 * data gradient (trans_a = 0, b_kmajor = 0) on the same stream as ONE launch (the two grids concatenated), anything else one
 * by one in program order.  Results are identical to the separate launches.  The 1x1 convs of the 14x14 / 7x7 MBConv stages
 * under cv_classifier.py:49 each fill the chip for only 1.3-2.5 rounds of tiles; their dW and dX products share one launch. */
int mmsim_gemm_group_begin(void);
int mmsim_gemm_group_end(void);

/* Forward 1x1 conv (xf_operand 0: plain, 1: BN + SiLU (+ gate) applied to A while staged) that also ACCUMULATES the
 * train-mode BatchNorm statistics of its bf16 output into sums [2][N] (sum, sum of squares; pre-zeroed by the caller):
 * replaces conv + the statistics pass of the following nn.BatchNorm2d (timm conv_pw / conv_pwl / conv_head + bn).
 * scratch: >= ceil(M/128) * 2 * N floats.  A, B and C are fp16 (image-tower forward tensors; the name keeps the family prefix). */
int mmsim_gemm_bf16_bnstats(int xf_operand, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C,
                            int ldc, const float* xf_scale, const float* xf_shift, const float* xf_gate, int xf_hw,
                            float* sums, float* scratch, unsigned long long scratch_floats, void* stream);

/* ---- self-attention (modeling_bert.py:111-136, 164-203), head_dim 64, S in {32, 64, 128} ----------
 * qkv: bf16 [B*S, ld_qkv] with q | k | v column blocks of width H; mask: int64 [B,S] (1 keep / 0 pad)
 * or NULL; ctx: bf16 [B*S, ld_ctx]; lse: fp32 [B*heads*S] log-sum-exp saved for backward. */
int mmsim_attn_fwd(const void* qkv, int ld_qkv, const long long* mask, void* ctx, int ld_ctx, float* lse, int B,
                   int S, int heads, int H, float dropout_p, unsigned long long seed, unsigned int stream_id,
                   void* stream);
int mmsim_attn_bwd(const void* qkv, int ld_qkv, const long long* mask, const void* ctx, const void* dctx, int ld_ctx,
                   const float* lse, void* dqkv, int B, int S, int heads, int H, float dropout_p,
                   unsigned long long seed, unsigned int stream_id, void* stream);
/* mmsim_attn_bwd that also accumulates dbias [3H] += dqkv.sum(0) (bias gradients of the fused q|k|v projection, of the
 * rounded values) without re-reading dqkv.  scratch: >= B * 3H floats. */
int mmsim_attn_bwd_dbias(const void* qkv, int ld_qkv, const long long* mask, const void* ctx, const void* dctx, int ld_ctx,
                         const float* lse, void* dqkv, float* dbias, int B, int S, int heads, int H, float dropout_p,
                         unsigned long long seed, unsigned int stream_id, float* scratch, unsigned long long scratch_floats,
                         void* stream);

/* ---- BERT embeddings: LayerNorm(word[ids] + type[tt] + pos[position_ids]) then dropout (modeling_bert.py:68-108;
 * position ids as forwarded by nlp_classifier.py:23-27 / transformer_emb.py:20-24).
 * ids / token_types / position_ids: int64 [B*S] (token_types NULL = zeros; position_ids NULL = arange(S) per row);
 * tables and gamma/beta fp32; out bf16 [B*S,H].  vocab_size / type_vocab_size (<= 2) / max_positions are the table
 * heights: an index outside its table sets *err_flag (device int, never cleared by the kernels; nn.Embedding raises an
 * IndexError there) and is clamped, so neither the gather nor the backward's scatter-add leaves its table.
 * Backward accumulates (atomically) into dword [V,H], dpos [P,H], dtype [type_vocab_size,H], dgamma, dbeta (fp32, pre-zeroed).
 * `scratch` (mmsim_embed_ln_bwd2; optional, >= mmsim_embed_ln_bwd_scratch_floats(B, S, H) floats; smaller or NULL: ignored): the sums
 * every wave adds to -- dgamma, dbeta, the token-type rows -- then leave through one slab row per workgroup and a small reduction
 * instead of thousands of same-address atomics. */
int mmsim_embed_ln_fwd(const long long* ids, const long long* token_types, const long long* position_ids,
                       const float* word, const float* pos, const float* type, const float* gamma, const float* beta,
                       void* out, int B, int S, int H, int vocab_size, int type_vocab_size, int max_positions,
                       int* err_flag, float eps, float dropout_p, unsigned long long seed, unsigned int stream_id,
                       void* stream);
int mmsim_embed_ln_bwd2(const void* dout, const long long* ids, const long long* token_types, const long long* position_ids,
                        const float* word, const float* pos, const float* type, const float* gamma, float* dword,
                        float* dpos, float* dtype, float* dgamma, float* dbeta, int B, int S, int H, int vocab_size,
                        int type_vocab_size, int max_positions, int* err_flag, float eps, float dropout_p,
                        unsigned long long seed, unsigned int stream_id, float* scratch, unsigned long long scratch_floats,
                        void* stream);
/* floats of `scratch` the slab path of mmsim_embed_ln_bwd2 needs for this shape (the library's own batch split; 0 for an empty shape) */
int mmsim_embed_ln_bwd_scratch_floats(int B, int S, int H);
/* The version-200 signature (no scratch: atomics only), kept so that a binding generated from the older header keeps working. */
int mmsim_embed_ln_bwd(const void* dout, const long long* ids, const long long* token_types, const long long* position_ids,
                       const float* word, const float* pos, const float* type, const float* gamma, float* dword,
                       float* dpos, float* dtype, float* dgamma, float* dbeta, int B, int S, int H, int vocab_size,
                       int type_vocab_size, int max_positions, int* err_flag, float eps, float dropout_p,
                       unsigned long long seed, unsigned int stream_id, void* stream);

/* ---- y = dropout(t) + resid ; h = LayerNorm(y)  (BertSelfOutput / BertOutput, modeling_bert.py:289-293, 347-351).
 * t, resid, y, h: bf16 [M,H]; mean/rstd: fp32 [M] saved for backward. */
int mmsim_add_ln_fwd(const void* t, const void* resid, const float* gamma, const float* beta, void* y, void* h,
                     float* mean, float* rstd, int M, int H, float eps, float dropout_p, unsigned long long seed,
                     unsigned int stream_id, void* stream);
/* dh = dh_a (+ dh_b if not NULL).  dy: gradient of y (residual branch).  dt: gradient of t, written only when
 * dropout_p > 0 (otherwise dt == dy and the caller reuses dy).  dgamma/dbeta/dbias: fp32 [H], accumulated.
 * scratch: >= ceil(M/(4*ceil(M/2048)))*3*H floats for the per-block partial column sums. */
int mmsim_ln_bwd(const void* dh_a, const void* dh_b, const void* y, const float* mean, const float* rstd,
                 const float* gamma, void* dy, void* dt, float* dgamma, float* dbeta, float* dbias, int M, int H,
                 float dropout_p, unsigned long long seed, unsigned int stream_id, float* scratch,
                 unsigned long long scratch_floats, void* stream);

/* out[n] += sum_m x[m,n]  (bias gradients) */
int mmsim_colsum_bf16(const void* x, int ld, float* out, int M, int N, void* stream);

/* ---- F.normalize(x, p=2, dim=1) (arcface.py:47, multimodal_classifier.py:54-55) and its backward.
 * out[r, col_off + c] = post_scale * x[r,c] / max(||x_r||, eps); f32 and/or bf16 copies; inv_norm[r] saved.
 * backward: dx (+)= (g - xh (xh.g)) * inv with g = pre_scale * dxh[r, col_off + c], xh = x * inv. */
int mmsim_l2norm_fwd(const void* x, int x_is_bf16, int ldx, float* out_f32, void* out_bf16, int ldo, int col_off,
                     float* inv_norm, int R, int D, float eps, float post_scale, void* stream);
int mmsim_l2norm_bwd(const void* x, int x_is_bf16, int ldx, const float* inv_norm, const float* dxh, int ldd,
                     int col_off, float* dx, int lddx, int R, int D, float pre_scale, int accumulate, void* stream);

/* ---- ArcFace margin + scale (arcface.py:49-61) on cosine logits, in place: fp32 [B, ld].
 * err_flag (device int) is set to 1 when a label is outside [0, C) (the reference raises there). */
int mmsim_arcface_margin(float* logits, int ld, const long long* label, int B, int C, float s, float m,
                         int easy_margin, int* err_flag, void* stream);
/* Fused margin + scale + nn.CrossEntropyLoss (multimodal_classifier_train.py:188) + argmax (:191) + the
 * gradient wrt the cosines: loss[b], argmax[b], dcos bf16 [B, ld] = grad_scale * dLoss_b/dcos (pad zeroed).
 * dcos and argmax may be NULL (evaluation). */
int mmsim_arcface_ce(const float* cosm, int ld, const long long* label, float* loss, long long* argmax, void* dcos,
                     int B, int C, float s, float m, int easy_margin, float grad_scale, int* err_flag, void* stream);
/* The fused head without extra passes over [B, C] (arcface.py:47-61 + nn.CrossEntropyLoss of multimodal_classifier_train.py:188):
 * cos [B, ld] (f32) = x_hat [B, D] w_hat [ld, D]^T (bf16; rows >= C of w_hat zero), and per row lse / loss / argmax / the target's logit and
 * margin slope from per-column-segment online-softmax statistics -- left by the cosine product's own epilogue when the pipelined
 * 256 x 256 kernel takes the shape (B % 256 == 0, ld % 256 == 0, D % 64 == 0, >= 128 tiles), by one statistics pass otherwise --
 * and loss_mean[0] = mean_b loss_b (NULL: skipped).  part: scratch, >= B * ceil(ld / 64) * 4 floats.  rowst [B][4] = {lse, target
 * logit, margin slope, -}.  A label outside [0, C) sets *err_flag and contributes a zero loss.
 * mmsim_arcface_dcos_rowfix (backward, ld % 256 == 0): ONE pass over the cosines writes dcos (bf16 [B, ld], pad zero) =
 * dloss_dev[0] * grad_scale * dLoss_b/dcos AND the two row vectors of the weight gradient's row-fix epilogue (what
 * mmsim_arcface_rowfix computes in a second pass): rowvec[c] = inv_w[c], rowvec[C + c] = sum_b dcos[b][c] cos[b][c].
 * dloss_dev: the upstream gradient of the mean loss as a DEVICE scalar (NULL = 1). */
int mmsim_arcface_fwd_fused(const void* x_hat, const void* w_hat, float* cosm, int ld, const long long* label, float* part,
                            unsigned long long part_floats, float* rowst, float* loss_b, long long* argmax, float* loss_mean,
                            int B, int C, int D, float s, float m, int easy_margin, int* err_flag, void* stream);
int mmsim_arcface_dcos_rowfix(const float* cosm, int ld, const long long* label, const float* rowst, const float* dloss_dev,
                              float grad_scale, const float* inv_w, void* dcos, float* rowvec, int B, int C, float s, float m,
                              int easy_margin, void* stream);
/* Backward of mmsim_arcface_margin for externally supplied dlogits (torch CrossEntropyLoss on the logits). */
int mmsim_arcface_dlogits_to_dcos(const float* dlogits, int ld_dl, const float* cosm, int ld, const long long* label,
                                  void* dcos, int B, int C, float s, float m, int easy_margin, void* stream);

/* ---- class-sharded ArcFace head (data parallelism with the [C, D] weight split by class range over the ranks; the loss
 * semantics kept are the reference's: softmax over ALL classes, nlp_classifier_train_daodian_v2_dist.py:139-144).
 * cos: the local columns [class_offset, class_offset + C_local) of the cosine matrix for all B rows of the global batch; labels
 * are global class indices.  mmsim_arcface_ce_partial leaves per row stats[b][4] = {local max logit, sum exp(logit - max),
 * target logit or 0, 1 if the target is local} and arg[b] = global index of the local argmax; after the ranks have exchanged the
 * stats, mmsim_arcface_dcos_from_lse writes dcos (bf16 [B, ld], pad zeroed) of the local columns from the row's global
 * log-sum-exp: dcos = row_scale[b] * s * slope * (softmax - onehot). */
int mmsim_arcface_ce_partial(const float* cosm, int ld, const long long* label, float* stats, long long* arg, int B,
                             int C_local, long long class_offset, long long C_total, float s, float m, int easy_margin,
                             int* err_flag, void* stream);
int mmsim_arcface_dcos_from_lse(const float* cosm, int ld, const long long* label, const float* lse, const float* row_scale,
                                void* dcos, int B, int C_local, long long class_offset, float s, float m, int easy_margin,
                                void* stream);

/* Weight-gradient path of the head without the fp32 dW_hat round trip: rowvec [2][C] <- (inv_w[c], sum_b dcos[b][c] cos[b][c]);
 * mmsim_gemm_bf16 with epilogue 5 (row-fix: C[m][n] (+)= bias[m] * (acc - aux_in[m][n] * bias[M + m]), bias = rowvec,
 * aux_in = W_hat bf16) then writes dW = (dW_hat - w_hat (w_hat . dW_hat)) / ||w|| straight into the gradient buffer: the
 * backward of F.normalize(self.weight) (arcface.py:47) folded into dcos^T x_hat. */
int mmsim_arcface_rowfix(const void* dcos, const float* cosm, int ld, const float* inv_w, float* rowvec, int B, int C,
                         void* stream);

/* ---- small glue ------------------------------------------------------------------------------- */
int mmsim_cast_f32_to_bf16(const float* x, void* y, unsigned long long n, void* stream);
int mmsim_cast_f32_to_f16(const float* x, void* y, unsigned long long n, void* stream);       /* saturating at +-65504 */
int mmsim_cast_bf16_to_f32(const void* x, float* y, unsigned long long n, void* stream);
int mmsim_gather_cls(const void* h, void* out, int B, int S, int H, void* stream);     /* h[:,0] (modeling_bert.py:460) */
int mmsim_scatter_cls(const void* src, void* dh, int B, int S, int H, void* stream);   /* its backward */
int mmsim_tanh_bwd(const float* dpooled, const float* pooled, void* dpre, unsigned long long n, void* stream);

/* ---- EfficientNet image tower (timm efficientnet_b0/b4 under cv_classifier.py:23-27,49), NHWC activations.
 * ELEMENT TYPES of every entry point of this section: FORWARD tensors -- conv outputs z, activated tensors a / act_out, block
 * inputs / outputs x, the operand `in` of the depthwise kernels, and the weight shadow the FORWARD 1x1 convs read (w*_f16) -- are
 * fp16 (IEEE half, round-to-nearest-even, saturating); GRADIENT tensors -- dy, dz, dpre, dx, da, resid of a backward entry point,
 * `other` of mmsim_pool_bn_act -- and the weight shadow the data-gradient products read (w*_bf16) are bf16.  Why: with bf16
 * storage of the ~5 stored tensors per MBConv block the image embedding missed north_star's 1e-2 by 3.5-6x (8-bit significand);
 * fp16 has the same bytes and MFMA rate and an 11-bit significand; gradients need bf16's range.
 * Train-mode BatchNorm2d (eps 1e-5, momentum 0.1): bn_stats accumulates per-channel sum / sum of squares into
 * sums [2][C] (fp32, pre-zeroed); bn_finalize turns them into mean, rstd, scale = gamma*rstd,
 * shift = beta - mean*scale and updates the running statistics; bn_apply writes act(scale*z+shift) (+resid).
 * Per-channel reductions write per-block partial slabs into `scratch` (fp32, >= 4M floats is always enough for
 * the supported towers; each call states its need in the error message) and sum them with a second tiny launch:
 * results are bitwise reproducible and no launch leaves a thousand blocks queueing on the same atomic address. */
int mmsim_bn_stats(const void* z, float* sums, int P, int C, float* scratch, unsigned long long scratch_floats, void* stream);
int mmsim_bn_finalize(const float* sums, const float* gamma, const float* beta, float* mean, float* rstd, float* scale,
                      float* shift, float* run_mean, float* run_var, int C, float count, float eps, float momentum,
                      void* stream);
/* The same finalisation ATTACHED to the reduction of the statistics: after _arm, the next partial-sum reduction into `sums` issued on
 * this host thread (inside mmsim_gemm_bf16_bnstats, mmsim_dwtile_fwd, mmsim_pw_project_fwd, mmsim_pw_expand_fwd, mmsim_stem_fwd,
 * mmsim_dwconv_fwd, mmsim_bn_stats) is done by a kernel that sums the slab in a fixed order (no atomics) and writes mean / rstd /
 * scale / shift and the running statistics as well: one launch instead of two per BatchNorm.  _flush: launches mmsim_bn_finalize if
 * nothing consumed the request; a no-op otherwise.  Results agree with the two-launch form to summation order. */
int mmsim_bn_finalize_arm(float* sums, const float* gamma, const float* beta, float* mean, float* rstd, float* scale, float* shift,
                          float* run_mean, float* run_var, int C, float count, float eps, float momentum);
int mmsim_bn_finalize_flush(void* stream);
int mmsim_bn_apply(const void* z, const float* scale, const float* shift, const void* resid, void* out, int P, int C,
                   int act_silu, void* stream);
/* out[b,c] = mul * sum_hw act(scale*z+shift) * (other ? other : 1): SE squeeze / global pool (mul = 1/HW), SE dgate. */
int mmsim_pool_bn_act(const void* z, const float* scale, const float* shift, const void* other, float* out, int B,
                      int HW, int C, int act_silu, float mul, void* stream);
/* The SE squeeze that also keeps the activated tensor: act_out [B*HW][C] bf16 = silu(scale*z+shift), out = mul * sum_hw of it.
 * The projection conv and its weight gradient then take act_out with a gate-only operand transform (mmsim_gemm_bf16_xf /
 * _bnstats with xf_scale = xf_shift = NULL): one multiply per element instead of BN + SiLU + gate in the GEMM's staging. */
int mmsim_pool_bn_act_store(const void* z, const float* scale, const float* shift, void* act_out, float* out, int B,
                            int HW, int C, float mul, void* stream);

/* Backward of the SE squeeze and, in the same pass over (z, dy), the partial sums from which the depthwise BatchNorm's
 * backward batch sums follow once the SE backward has produced dsq (timm SqueezeExcite + bn2 + act under cv_classifier.py:49):
 * out5 [5][B][C] fp32: [0] dgate = sum_hw silu(bn z) dy; [1..4] = sum_hw dy a', a', dy a' zhat, a' zhat  (a' = silu'(bn z)).
 * mmsim_bn_bwd_sums_from_pool then accumulates into sums [2][C] (pre-zeroed) the batch sums of da and da*zhat, da = (dy gate + dsq/HW) a', which
 * mmsim_bn_bwd consumes with sums_ready = 1. */
int mmsim_pool_bn_bwd(const void* z, const float* scale, const float* shift, const float* mean, const float* rstd,
                      const void* dy, float* out5, int B, int HW, int C, void* stream);
int mmsim_bn_bwd_sums_from_pool(const float* out5, const float* gate, const float* dsq, float* sums, int B, int HW, int C,
                                void* stream);
/* Squeeze-excite: hr = W_reduce s + b_reduce (saved pre-activation; hs = silu(hr) saved too); gate = sigmoid(W_expand silu(hr) + b_expand).
 * weT [RD][C] receives conv_expand.weight transposed (kept for the backward of the same step).  Backward: from
 * dgate [B,C] produces dr [B,RD], ds [B,C] (gradient of the squeezed input) and accumulates the four parameter
 * gradients (dweT [RD][C]: scratch).  w_expand = NULL (forward): weT already holds the transposed weights; dw_expand = NULL
 * (backward): dweT is the caller's ZEROED buffer and receives the transposed expand-weight gradient (accumulated; the caller
 * transposes it into the parameter's gradient, e.g. for all blocks at once with mmsim_dw_tap_major_batch). */
int mmsim_se_mlp_fwd(const float* s, const float* w_reduce, const float* b_reduce, const float* w_expand,
                     const float* b_expand, float* weT, float* hr, float* hs, float* gate, int B, int C, int RD,
                     void* stream);
int mmsim_se_mlp_bwd(const float* dgate, const float* gate, const float* hr, const float* hs, const float* s,
                     const float* w_reduce, const float* weT, float* dr, float* ds, float* dweT, float* dw_reduce,
                     float* db_reduce, float* dw_expand, float* db_expand, int B, int C, int RD, void* stream);
/* BatchNorm (+SiLU, +SE gate) backward: da = (gate ? dy*gate + dsq/hw : dy) * (act ? silu'(scale*z+shift) : 1);
 * dz = scale*(da - mean(da) - zhat*mean(da*zhat)); dgamma += sum da*zhat; dbeta += sum da.
 * sums [2][C] must be zero on entry unless sums_ready (already produced by mmsim_dwconv_bwd_data). */
int mmsim_bn_bwd(const void* dy, const void* z, const float* mean, const float* rstd, const float* scale,
                 const float* shift, const float* gate, const float* dsq, int hw, int act_silu, float* sums,
                 int sums_ready, void* dz, float* dgamma, float* dbeta, int P, int C, float* scratch,
                 unsigned long long scratch_floats, void* stream);
/* The two layout transforms below for ALL depthwise layers of a network in one launch: segs_dev = device array of nseg x 4 int64
 * {src offset, dst offset, C, K*K} (offsets in floats from src_base / dst_base).  to_tap_major 1: weights [C][K*K] -> [K*K][C];
 * 0: tap-major gradient ACCUMULATED into the [C][K*K] gradient.  max_elems = the largest C * K*K (grid sizing). */
int mmsim_dw_tap_major_batch(const void* segs_dev, int nseg, const float* src_base, float* dst_base, int to_tap_major,
                             int max_elems, void* stream);

/* Depthwise k3/k5 stride 1/2 conv, pad k/2.  Weights in tap-major fp32 [K*K][C] (see the two converters).
 * fwd also accumulates the output's BN sums; bwd_data also applies silu'(bn(z1)) of the producer and
 * accumulates that BatchNorm's backward sums (z1 == NULL: plain transposed conv, + resid if given);
 * bwd_weight accumulates into a tap-major gradient. */
int mmsim_dw_weight_to_tap_major(const float* w, float* wT, int C, int K, void* stream);
int mmsim_dw_grad_from_tap_major(const float* gT, float* g, int C, int K, void* stream);
int mmsim_dwconv_fwd(const void* a, const float* w_tap_major, void* z, float* sums, int B, int Hi, int Wi, int C, int K,
                     int S, float* scratch, unsigned long long scratch_floats, void* stream);
int mmsim_dwconv_bwd_data(const void* dz, const float* w_tap_major, const void* z1, const float* mean, const float* rstd,
                          const float* scale, const float* shift, const void* resid, void* dpre, float* sums, int B,
                          int Hi, int Wi, int C, int K, int S, float* scratch, unsigned long long scratch_floats,
                          void* stream);
int mmsim_dwconv_bwd_weight(const void* dz, const void* a, float* g_tap_major, int B, int Hi, int Wi, int C, int K, int S,
                            float* scratch, unsigned long long scratch_floats, void* stream);
/* The same with the operand formed on the fly: a = silu(xf_scale * z + xf_shift) (rounded to bf16 as a stored copy would be), z the
 * pre-BatchNorm tensor -- the stride-2 MBConv blocks then never store the activated expansion. */
int mmsim_dwconv_bwd_weight_xf(const void* dz, const void* z, const float* xf_scale, const float* xf_shift, float* g_tap_major, int B,
                               int Hi, int Wi, int C, int K, int S, float* scratch, unsigned long long scratch_floats, void* stream);
/* ---- LDS-tiled depthwise kernels (csrc/mbconv.hip): the timm MBConv depthwise stage under cv_classifier.py:49 with the
 * passes either side of it folded in.  All activations NHWC bf16, w_tap_major [K*K][C] fp32 (mmsim_dw_weight_to_tap_major),
 * scratch >= the per-block partial slabs (checked), sums pre-zeroed by the caller as for every BN-sum producer.
 * mmsim_dwtile_fwd: z [B,Ho,Wo,C] = dwconv_KxK_stride_S(a), a = silu(xf_scale * in + xf_shift) formed while the input tile
 *   (with halo) is staged in LDS (xf_scale / xf_shift NULL: `in` is already activated), + sums [2][C] += per-channel sum and
 *   sum of squares of the rounded outputs (the following BatchNorm's batch statistics).
 * mmsim_dwtile_bwd (stride 1): the whole backward of  z1 -> bn1 -> silu -> dwconv -> bn2 -> silu -> x gate  in one kernel.
 *   dy = dLoss/d(gated activation), dsq [B,C] = dLoss/d(squeeze) from the SE backward, sums2 [2][C] = the depthwise
 *   BatchNorm's backward sums (mmsim_bn_bwd_sums_from_pool).  Produces out = dLoss/d(z1 after bn1) x silu' (the input of
 *   bn1's backward) with its sums1 [2][C] (+=), the tap-major depthwise weight gradient g_tap_major (+=), and
 *   dgamma2 / dbeta2 (+=).  scale1 == NULL: DS block, z1 is the block input itself (no bn1 / silu on the data path), `resid`
 *   (optional) is added to out, sums1 unused. */
int mmsim_dwtile_fwd(const void* in, const float* xf_scale, const float* xf_shift, const float* w_tap_major, void* z,
                     float* sums, int B, int Hi, int Wi, int C, int K, int S, float* scratch,
                     unsigned long long scratch_floats, void* stream);
int mmsim_dwtile_bwd(const void* dy, const void* z2, const float* scale2, const float* shift2, const float* mean2,
                     const float* rstd2, const float* sums2, const float* gate, const float* dsq, const void* z1,
                     const float* scale1, const float* shift1, const float* mean1, const float* rstd1,
                     const void* resid, const float* w_tap_major, void* out, float* sums1, float* g_tap_major,
                     float* dgamma2, float* dbeta2, int B, int H, int W, int C, int K, float* scratch,
                     unsigned long long scratch_floats, void* stream);
/* mmsim_dwtile_bwd_s2 (round 4): the same one-kernel backward for the STRIDE-2 depthwise conv of an IR block (the first block of a
 * stage: timm conv_dw with symmetric padding K/2 under cv_classifier.py:49).  H x W is the conv's INPUT plane (z1, out, sums1); dy, z2,
 * gate, dsq and the depthwise BatchNorm's state live on the output plane ((H + 2 (K/2) - K) / 2 + 1 per side).  Replaces
 * mmsim_bn_bwd_apply + mmsim_dwconv_bwd_weight(_xf) + mmsim_dwconv_bwd_data for those blocks. */
int mmsim_dwtile_bwd_s2(const void* dy, const void* z2, const float* scale2, const float* shift2, const float* mean2,
                        const float* rstd2, const float* sums2, const float* gate, const float* dsq, const void* z1,
                        const float* scale1, const float* shift1, const float* mean1, const float* rstd1,
                        const float* w_tap_major, void* out, float* sums1, float* g_tap_major,
                        float* dgamma2, float* dbeta2, int B, int H, int W, int C, int K, float* scratch,
                        unsigned long long scratch_floats, void* stream);

/* The 5 x 5 stride-1 depthwise convolution on the matrix cores (csrc/dwmfma.hip, round 4): same contracts as mmsim_dwtile_fwd / _bwd
 * for K = 5, S = 1 (timm conv_dw of the k5 MBConv stages under cv_classifier.py:49), whole-plane tiles of 16 channels held PLANAR in
 * LDS, the taps as banded 4 x 4 operands of v_mfma_f32_4x4x4_16b (block = channel).  Eligible: C % 16 == 0, H, W <= 28.
 * scratch: >= B * 2 C floats (forward); backward: >= B * (2 + 25) C floats. */
int mmsim_dw5m_eligible(int B, int H, int W, int C, int K, int S);
int mmsim_dw5m_fwd(const void* in, const float* xf_scale, const float* xf_shift, const float* w_tap_major, void* z, float* sums,
                   int B, int H, int W, int C, float* scratch, unsigned long long scratch_floats, void* stream);
int mmsim_dw5m_bwd(const void* dy, const void* z2, const float* scale2, const float* shift2, const float* mean2, const float* rstd2,
                   const float* sums2, const float* gate, const float* dsq, const void* z1, const float* scale1, const float* shift1,
                   const float* mean1, const float* rstd1, const float* w_tap_major, void* out, float* sums1, float* g_tap_major,
                   float* dgamma2, float* dbeta2, int B, int H, int W, int C, float* scratch, unsigned long long scratch_floats,
                   void* stream);
/* Projection 1x1 conv of the early MBConv stages (timm conv_pwl after bn2 + SiLU + SE under cv_classifier.py:49; conv_pw of the
 * depthwise-separable blocks) as one streaming pass:  z3[P,cout] (bf16) = (a2[P,mid] * gate[P / HW, mid]) W3[cout,mid]^T, and the
 * train-mode BatchNorm statistics of the bf16 output ACCUMULATED into sums [2][cout] (pre-zeroed by the caller) -- the same contract
 * as mmsim_gemm_bf16_bnstats(xf_operand 1, gate only), for the shapes mmsim_pw_project_fwd_eligible accepts (mid <= 384,
 * cout <= 64, HW >= the strip height: the 112^2 / 56^2 / 28^2 stages of B0-B4).  scratch: >= 128 * cout floats. */
int mmsim_pw_project_fwd_eligible(int P, int HW, int mid, int cout);
int mmsim_pw_project_fwd(const void* a2, const float* gate, const void* w3_f16, void* z3, float* sums, int P, int HW, int mid,
                         int cout, float* scratch, unsigned long long scratch_floats, void* stream);

/* Expansion 1x1 conv of the 56^2 MBConv stage (timm conv_pw under cv_classifier.py:49) as one streaming pass:
 * z1[P,mid] (bf16) = x[P,cin] W1[mid,cin]^T and the train-mode BatchNorm statistics of the bf16 output ACCUMULATED into
 * sums [2][mid] (pre-zeroed by the caller): the contract of mmsim_gemm_bf16_bnstats(xf_operand 0) for cin <= 32, 64 <= mid <= 192,
 * P % 64 == 0 (mmsim_pw_expand_fwd_eligible).  scratch: >= 128 * mid floats. */
int mmsim_pw_expand_fwd_eligible(int P, int mid, int cin);
int mmsim_pw_expand_fwd(const void* x, const void* w1_f16, void* z1, float* sums, int P, int mid, int cin, float* scratch,
                        unsigned long long scratch_floats, void* stream);

/* The two streaming projection kernels with the operand formed on the fly: `z2` is the pre-BatchNorm depthwise output and the operand
 * is silu(xf_scale z2 + xf_shift) * gate -- the activated tensor a2 is then never stored (forward) nor read (backward). */
int mmsim_pw_project_fwd_xf(const void* z2, const float* xf_scale, const float* xf_shift, const float* gate, const void* w3_f16,
                            void* z3, float* sums, int P, int HW, int mid, int cout, float* scratch,
                            unsigned long long scratch_floats, void* stream);
int mmsim_pw_project_bwd_xf(const void* dz3, const void* z2, const float* xf_scale, const float* xf_shift, const float* gate,
                            const void* w3_bf16, void* da, float* dw3, int P, int HW, int mid, int cout, float* scratch,
                            unsigned long long scratch_floats, void* stream);

/* Backward of the same conv as one streaming pass (autograd of conv_pwl / conv_pw after bn + SiLU + SE):
 *   da[P,mid] (bf16) = dz3[P,cout] W3[cout,mid]            -- the gradient w.r.t. the GATED activation a2 * gate
 *   dw3[cout,mid] (fp32) += dz3^T (a2 * gate[P / HW])
 * replacing mmsim_gemm_bf16_xf(2, ...) + mmsim_gemm_bf16(dgrad) for the shapes mmsim_pw_project_bwd_eligible accepts (the 112^2
 * and 56^2 stages: mid <= 192, cout <= 32).  scratch: >= 64 * cout * mid floats. */
int mmsim_pw_project_bwd_eligible(int P, int HW, int mid, int cout);
int mmsim_pw_project_bwd(const void* dz3, const void* a2, const float* gate, const void* w3_bf16, void* da, float* dw3, int P,
                         int HW, int mid, int cout, float* scratch, unsigned long long scratch_floats, void* stream);

/* mmsim_pw_expand_bwd: the backward of an MBConv block's expand stage  x -> conv_pw (W1) -> z1 -> bn1 (train mode)  in one
 * streaming pass (timm conv_pw + bn1 under cv_classifier.py:49): dpre = dLoss/d(bn1 output) [P, mid] (what mmsim_dwtile_bwd
 * leaves), sums1 [2][mid] = bn1's backward sums.  dx [P, cin] = dz1 W1 (+ resid), dw1 [mid, cin] += dz1^T x,
 * dgamma1 / dbeta1 += the sums, with dz1 = scale1 (dpre - S1/P - zhat S2/P) formed on the way into LDS.  Replaces
 * mmsim_bn_bwd + two mmsim_gemm_bf16 calls.  Shapes: mmsim_pw_expand_bwd_eligible (W1 and dW1 must fit LDS / registers). */
int mmsim_pw_expand_bwd_eligible(int P, int mid, int cin);
int mmsim_pw_expand_bwd(const void* dpre, const void* z1, const void* x, const void* resid, const void* w1_bf16,
                        const float* scale1, const float* mean1, const float* rstd1, const float* sums1, void* dx,
                        float* dw1, float* dgamma1, float* dbeta1, int P, int mid, int cin, float* scratch,
                        unsigned long long scratch_floats, void* stream);
/* Stem: 3x3 stride-2 pad-1 conv on the NCHW fp32 image -> NHWC bf16, with the output's BN sums; and its wgrad. */
int mmsim_stem_fwd(const float* x, const float* w, void* z, float* sums, int B, int Hi, int Wi, int Co, float* scratch,
                   unsigned long long scratch_floats, void* stream);
int mmsim_stem_wgrad(const void* dz, const float* x, float* dw, int B, int Hi, int Wi, int Co, float* scratch,
                     unsigned long long scratch_floats, void* stream);
/* Tower top (cv_classifier.py:50-54): dropout -> bf16, BatchNorm1d on fp32 [B,C], pool backward broadcast. */
int mmsim_bn1d_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                   float* run_mean, float* run_var, int B, int C, float eps, float momentum, int training, void* stream);
int mmsim_bn1d_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, float* dx,
                   float* dgamma, float* dbeta, int B, int C, void* stream);
int mmsim_dropout_cast(const float* x, void* y_f16, unsigned long long n, float p, unsigned long long seed,
                       unsigned int stream_id, void* stream);
int mmsim_dropout_bwd(const float* dy, float* dx, unsigned long long n, float p, unsigned long long seed,
                      unsigned int stream_id, void* stream);
int mmsim_broadcast_pool_grad(const float* dpool, void* dy, int B, int HW, int C, void* stream);

/* ---- torch.optim.AdamW.step() (multimodal_classifier_train.py:152-156,161,195,199) over a flat fp32 buffer.
 * g is multiplied by grad_scale first (1/world_size after an all-reduce sum); bf16_shadow (may be NULL)
 * receives the updated parameters rounded to bf16 for the next forward.  n % 4 == 0.  step is 1-based.  * dev_hyper (both AdamW entry points; NULL = use the arguments): device float[3] = {lr, 1 - beta1^step, 1 / sqrt(1 - beta2^step)}
 * read by the kernel at run time instead of lr / step -- the step-dependent scalars of a launch captured in a hipGraph. */
int mmsim_adamw_step(float* p, const float* g, float* m, float* v, void* bf16_shadow, unsigned long long n, float lr,
                     float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                     const float* dev_hyper, void* stream);
/* The same with a second, fp16 shadow (may be NULL): the flat buffers of the image tower keep both -- its forward products read
 * the weights as fp16, its data-gradient products as bf16 (EfficientNet section). */
int mmsim_adamw_step2(float* p, const float* g, float* m, float* v, void* bf16_shadow, void* f16_shadow, unsigned long long n,
                      float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                      const float* dev_hyper, void* stream);

/* The same update for a row-normalised weight matrix p [R][D] (the ArcFace head), one wave per row, which also leaves
 * w_hat bf16 [R][D] = p / max(||p_row||, l2_eps) and inv_norm [R] for the NEXT forward: F.normalize(self.weight) (arcface.py:47)
 * then needs no pass of its own.  D % 4 == 0, D <= 4096. */
int mmsim_adamw_rows_l2norm(float* p, const float* g, float* m, float* v, void* w_hat, float* inv_norm, int R, int D,
                            float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                            float l2_eps, const float* dev_hyper, void* stream);

/* ---- exhaustive inner-product top-k search (nlp_infer.py:139-152, daodian_infer.py:225-230, 295-302:
 * faiss.normalize_L2 + IndexFlat(METRIC_INNER_PRODUCT).add / .search) ------------------------------------------------
 * mmsim_split_bf16_cat: x fp32 [R, D] (* scale) -> bf16 [R, 3D]: [hi | lo | hi] for queries (db_side 0), [hi | hi | lo] for the
 * database (db_side 1), hi = bf16(x), lo = bf16(x - hi): one bf16 GEMM over K = 3D (mmsim_gemm_bf16, fp32 result) then
 * gives the fp32 inner products to ~2^-16 relative.
 * mmsim_topk_merge: folds scores [nq, n] (columns = database rows col_offset ..) into the running lists best_val / best_idx
 * [nq, k] (descending score, equal scores by ascending index; first != 0: the lists are empty on entry).  k <= 64. */
int mmsim_split_bf16_cat(const float* x, int ldx, void* out, int ldo, int R, int D, int db_side, float scale, void* stream);
int mmsim_topk_merge(const float* scores, int ld, int nq, int n, long long col_offset, int k, float* best_val,
                     long long* best_idx, int first, void* stream);

/* ---- image input stage (multimodal_infer.py:86-91, cv_classifier_train.py:39-40: timm create_transform =
 * torchvision Resize(bicubic, PIL) -> CenterCrop -> ToTensor -> Normalize) -------------------------------------------
 * One image: img [H][W][3] uint8 (row pitch in bytes) -> out [3][S][S] fp32.  bx/by [out][2] = (first input index, tap
 * count) and kx/ky [out][ks] = Pillow's 22-bit fixed-point bicubic kernels for the whole resized axis (built on the host,
 * multimodalsimilar_amd/preprocess.py); (left, top, S) is the crop window in the resized image; pass 1 covers input rows
 * [row0, row0 + nrows) into tmp (>= nrows * S * 4 bytes).  Bit-exact with Pillow's 8-bit two-pass resampling; ToTensor /
 * Normalize in fp32 with correctly rounded divisions. */
int mmsim_preprocess_image(const void* img, int H, int W, int pitch, const int* bx, const int* kx, int ksx, int out_w,
                           const int* by, const int* ky, int ksy, int out_h, int left, int top, int S, int row0, int nrows,
                           void* tmp, unsigned long long tmp_bytes, float* out, float mean0, float mean1, float mean2,
                           float std0, float std1, float std2, void* stream);

#ifdef __cplusplus
}
#endif
#endif
