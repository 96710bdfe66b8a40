/* C ABI of libclipself_hip.so -- the drop-in seam under the CLIPSelf hot path.
 *
 * The reference (wusize/CLIPSelf) is pure Python and has no FFI of its own; the native code it reaches lives in
 * third-party wheels (cuBLAS/cuDNN via torch, torchvision roi_align, xformers attention, apex LayerNorm).  Each
 * entry point below names the reference call site(s) (file:line under /root/reference) whose kernel it replaces.
 *
 * Conventions: plain pointers + sizes, no torch types; every buffer (incl. workspaces) is caller-allocated device
 * memory; launches are asynchronous on `stream`; return 0 on success, negative on error (message via
 * cs_last_error(), thread-local); no global mutable state; re-entrant from any thread that owns the stream.
 * bf16 tensors are passed as `void*` / `const void*`.  Row-major everywhere; `ld*` are row strides in elements.
 */
#ifndef CLIPSELF_HIP_H
#define CLIPSELF_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* cs_stream_t; /* == hipStream_t */

const char* cs_last_error(void);

/* --- matmuls: F.linear at src/open_clip/eva_clip/eva_vit_model.py:99-103 (SwiGLU w1,w2,w3), :177-179 (q/k/v proj),
 *     :219/:253 (attn proj), :250 (v-only proj of the last dense block), :585/:617 (head), nn.Conv2d patch embed :348,355
 *     (as im2row GEMM), and their autograd backward (dgrad / wgrad).
 * C[M,N] (+epilogue) = A[M,K] . B[N,K]^T, bf16 operands, fp32 accumulation.  K % 64 == 0, lda/ldb % 8 == 0.
 * epi: 0 bf16 = acc+bias | 1 f32 = acc+bias | 2 f32 = extra(residual)+acc+bias (in-place allowed) |
 *      3 fused SwiGLU: B=[W1;W2] [2*group,K], bias [2*group], out bf16 [M,group] = silu(x1)*x2 |
 *      4 f32 atomic accumulate (split-K allowed: `splits` >= 1, or <= 0 = chosen by the library) |
 *      5 patch embed: out row = row + row/group + 1, value += extra[(row%group+1)*ldc + col]   (cls/pos layout :540-543) |
 *      7 bf16 = GELU(acc+bias), 8 bf16 = QuickGELU(acc+bias): c_fc + activation of the OpenAI-CLIP ViT MLP
 *        (src/open_clip/transformer.py:209-213 `mlp`, :31-34 `QuickGELU`)
 * flags bit0: use register staging instead of the global_load_lds DMA path; bits 4-7: force a tile schedule (0 = heuristic; 11 = the
 * streaming persistent kernel with register-level epilogues, the default for the bf16 / QuickGELU / SwiGLU epilogues of large problems);
 * bits 20-27: compute units the persistent kernels leave free (0 = use all 256; data-parallel runs reserve a few for RCCL's kernels). */
int cs_gemm_nt(const void* A, const void* B, void* C, const float* bias, const float* extra, int M, int N, int K,
               int lda, int ldb, int ldc, int epi, int splits, int group, int flags, cs_stream_t stream);

/* fp8 operands (BASELINE configs[4] "fp8 MFMA weights"; reference call sites: the F.linear calls of eva_vit_model.py:99-103,177-179,218-219 under
 * src/training/region_clip.py:28-67).  cs_quant_rows_fp8: bf16 [M,K] -> OCP e4m3 [M,Kp] (Kp = K rounded up to 128, padding zero; ldq bytes per
 * row) with one fp32 scale per row (amax/448).  cs_gemm_nt_f8: C = row_scale[m] * col_scale[n] * (A8 . B8^T) + bias (+ extra), contracted
 * with the block-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, unit block scales), fp32 accumulate; epi 0 = bf16 out, 2 = fp32 residual. */
int cs_quant_rows_fp8(const void* x_bf16, long ldx, void* q_e4m3, long ldq, float* scale, int M, int K, cs_stream_t stream);
int cs_gemm_nt_f8(const void* A8, const void* B8, void* C, const float* bias, const float* extra, const float* row_scale, const float* col_scale,
                  int M, int N, int K8, int lda, int ldb, int ldc, int epi, int flags, cs_stream_t stream);

/* Weight gradient of a Linear (autograd of F.linear at the call sites above): dW[M,N] (f32, row stride ldc) += A[M,K] . B[N,K]^T with
 * A = dY^T, B = X^T (bf16, contraction = tokens, zero padded to K % 64 == 0).  The K range is split into slices whose partial
 * products go through `workspace` (>= cs_gemm_wgrad_workspace bytes, 16-byte aligned) and one pass adds them into dW. */
size_t cs_gemm_wgrad_workspace(int M, int N, int K);
int cs_gemm_wgrad(const void* A, const void* B, float* dW, void* workspace, int M, int N, int K, int lda, int ldb, int ldc,
                  cs_stream_t stream);

/* The same weight gradient without transposed copies: dW[N,K] (f32) += dY[tokens,N]^T . X[tokens,K], both operands token-major as the
 * backward left them (MFMA operands through the transposing LDS read ds_read_b64_tr_b16).  Any token count; N and K multiples of 8
 * (ragged tiles re-read valid rows / columns, the last token tile's missing tokens are zeroed in the fragments): cs_gemm_wgrad_tn_workspace
 * returns 0 and cs_gemm_wgrad_tn returns 1 (nothing launched) otherwise -- use cs_transpose_bf16 + cs_gemm_wgrad then. */
size_t cs_gemm_wgrad_tn_workspace(int N, int K, int tokens);
int cs_gemm_wgrad_tn(const void* dY, const void* X, float* dW, void* workspace, int N, int K, int tokens, int ldy, int ldx, int ldc,
                     cs_stream_t stream);

/* cs_gemm_nt with a LayerNorm folded in (frozen teacher): the GEMM reads the *un-normalised* bf16 rows, B = gamma (.) W,
 * ln_colsum[n] = sum_k B[n,k], bias = W.beta + b, and the epilogue applies rstd[m] * (acc - mean[m] * ln_colsum[n]) + bias[n]:
 *   epi 6: residual form, C = extra + that value                  (SwiGLU.ffn_ln -> w3 eva_vit_model.py:102-103, inner_attn_ln -> proj :218-219)
 *   epi 0 / 3 with ln_mean != NULL: before the bf16 store / before SiLU*mul   (Block.norm1 -> q|k|v :306,176-179; norm2 -> w1|w2 :307,99-101)
 * and the epilogues can emit what the NEXT folded GEMM needs, so that no LayerNorm pass over the activations is left:
 *   epi 3 with stats_part != NULL: per 32-hidden-unit slice s and row m, (sum, sum of squares) of the rounded outputs at
 *          stats_part[(s*M + m)*2 ..]  (4*ceil(group/128) slices);
 *   epi 2 / 6 with stats_part != NULL: the same per 64-column slice of the fp32 outputs (ceil(N/64) slices); with xb_out != NULL a bf16
 *          copy of the fp32 output (row stride ldxb).  cs_ln_stats_finalize turns the partials into mean/rstd.
 * All other arguments as in cs_gemm_nt. */
int cs_gemm_nt_ln(const void* A, const void* B, void* C, const float* bias, const float* extra, const float* ln_mean,
                  const float* ln_rstd, const float* ln_colsum, float* stats_part, void* xb_out, int ldxb, int M, int N, int K, int lda,
                  int ldb, int ldc, int epi, int splits, int group, int flags, cs_stream_t stream);

/* Epilogue 6 of cs_gemm_nt_ln on the "split stream": the frozen tower's fp32 residual stream x (Block.forward, eva_vit_model.py:306-307:
 * x = x + attn(norm1(x)); x = x + mlp(norm2(x))) kept as two 16-bit planes of the word y = bits(x) + 0x8000:
 *   hi[m, n] = y >> 16     -- x rounded to bf16 (halves away from zero), i.e. the operand the next folded GEMM (norm1 -> q|k|v,
 *                             norm2 -> w1|w2) reads as A, row stride ldxb elements;
 *   lo[m, n] = y & 0xffff  -- the rest; x = bits^-1(((hi << 16) | lo) - 0x8000) exactly.
 * A residual GEMM then moves 8 bytes per stream element (4 in, 4 out) instead of 10 (fp32 in, fp32 out, bf16 copy) and nothing is lost.
 *   x_in  != NULL: the stream is read as fp32 [M, ldc] (first block, after the stem);  NULL: as (hi, lo)
 *   x_out != NULL: it is written as fp32 [M, ldc] (last folded block; hi / lo are only read);  NULL: back to (hi, lo), in place
 *   stats_part: per 64-column slice (sum, sum of squares) of the fp32 outputs as in cs_gemm_nt_ln; required when the stream leaves
 *               split (x_out == NULL), rejected with x_out != NULL.
 * out = x + rstd[m] * (A.B^T - mean[m] * ln_colsum[n]) + bias[n];  N % 32 == 0, K % 64 == 0; hi / lo 16-byte aligned with ldxb % 8 == 0
 * (the planes move in 16-byte pieces); flags bits 20-27 as cs_gemm_nt. */
int cs_gemm_nt_ln_split(const void* A, const void* B, const float* bias, const float* ln_mean, const float* ln_rstd,
                        const float* ln_colsum, const float* x_in, float* x_out, void* hi, void* lo, int ldxb, float* stats_part,
                        int M, int N, int K, int lda, int ldb, int ldc, int flags, cs_stream_t stream);

/* --- LayerNorm(eps, biased var): src/open_clip/eva_clip/transformer.py:52-58 used at eva_vit_model.py:306-307 (norm1/2),
 *     :218 (inner_attn_ln), :102 (ffn_ln), :565/:616 (final norm); replaces apex FusedLayerNorm / F.layer_norm.
 * x_dtype 0=f32 1=bf16; y bf16 (NULL = statistics only); mean/rstd [M] f32 (nullable when no backward is needed). */
int cs_layernorm_fwd(const void* x, int x_dtype, long ldx, const float* gamma, const float* beta, void* y, long ldy,
                     float* mean, float* rstd, int M, int C, float eps, cs_stream_t stream);
/* the same, and the e4m3 copy of y for cs_gemm_nt_f8 (precision amp_fp8, src/training/region_clip.py:28-67 under BASELINE configs[4]):
 * q8 [M, ldq >= C rounded up to 128] bytes with zero padding, q_scale [M] = row amax / 448 -- bit-identical to cs_quant_rows_fp8(y) */
int cs_layernorm_fwd_q8(const void* x, int x_dtype, long ldx, const float* gamma, const float* beta, void* y, long ldy,
                        float* mean, float* rstd, void* q8, long ldq, float* q_scale, int M, int C, float eps, cs_stream_t stream);
/* part [P][M][2] f32 = per-slice (sum, sum of squares) over npp columns each (slices past C ignored) -> LayerNorm mean/rstd [M]
 * of a C-wide row; producers: cs_gemm_nt_ln epi 3 (npp 32) and cs_attn_fwd_stats (npp 64, P = heads). */
int cs_ln_stats_finalize(const float* part, int P, int npp, int C, int M, float eps, float* mean, float* rstd, cs_stream_t stream);
/* f32 in -> f32 out: `ln_pre` of the OpenAI-CLIP ViT (src/open_clip/transformer.py:371,477), whose output is the residual stream */
int cs_layernorm_fwd_f32(const float* x, long ldx, const float* gamma, const float* beta, float* y, long ldy, float* mean, float* rstd,
                         int M, int C, float eps, cs_stream_t stream);
size_t cs_layernorm_bwd_workspace(int M, int C);
/* dx_mode 0: bf16 write, 1: f32 write, 2: f32 accumulate (residual gradient stream).  dgamma/dbeta nullable (frozen).
 * dx_copy (modes 1 / 2, nullable): bf16 copy of the dx rows after the write / accumulate (row stride ldcopy) -- the operand of the next
 * dgrad / wgrad GEMMs -- and copy_colsum[C] (nullable) (+)= its column sums = the bias gradient of the linear layer that feeds this
 * residual branch (autograd of x = x + Linear(..) in Block.forward, eva_vit_model.py:306-307): no cast / column-sum pass of their own. */
int cs_layernorm_bwd(const void* dy, long lddy, const void* x, int x_dtype, long ldx, const float* gamma, const float* mean,
                     const float* rstd, void* dx, int dx_mode, long lddx, float* dgamma, float* dbeta, int accumulate_params,
                     void* workspace, void* dx_copy, long ldcopy, float* copy_colsum, int M, int C, cs_stream_t stream);
/* cs_layernorm_bwd whose bf16 copy also leaves as e4m3 bytes (q8 [M, ldq >= C rounded up to 128], zero padding) + fp32 row scales: the A operand
   of an fp8 dgrad through cs_gemm_nt_f8; bit-identical to cs_quant_rows_fp8(dx_copy).  dx_copy required. */
int cs_layernorm_bwd_q8(const void* dy, long lddy, const void* x, int x_dtype, long ldx, const float* gamma, const float* mean,
                        const float* rstd, void* dx, int dx_mode, long lddx, float* dgamma, float* dbeta, int accumulate_params,
                        void* workspace, void* dx_copy, long ldcopy, float* copy_colsum, void* q8, long ldq, float* q_scale, int M, int C,
                        cs_stream_t stream);

/* --- F.normalize(x, dim=-1) of the dense token map: eva_vit_model.py:620 (eps 1e-12) and its backward. */
int cs_l2norm_fwd(const float* x, float* y, float* inv_norm, int M, int C, float eps, cs_stream_t stream);
int cs_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, void* dx_bf16, int M, int C, cs_stream_t stream);

/* --- attention core: xops.memory_efficient_attention / softmax math branch at eva_vit_model.py:198-243 together with
 *     VisionRotaryEmbeddingFast.forward (src/open_clip/eva_clip/rope.py:148-164) on q,k tokens 1.. ; head dim 64.
 * qkv [B*Ntok, ldqkv] bf16 = q|k|v (bias added, not rotated); cos/sin [(Ntok-1),64] f32; out [B*Ntok, ldo] bf16;
 * lse [B*H, Ntok] f32 (nullable in inference).  The forward kernels read the tables separably, exactly as rope.py:118-142 builds
 * them (Ntok-1 = g*g; dims [0,32) depend on the grid row only, dims [32,64) on the grid column only): row r*g supplies the
 * row part, row c the column part.  Round 6: the short-sequence kernel (Ntok <= 197) keeps ONE entry per frequency in LDS, so the rest of
 * that construction is a precondition too: the row part of grid row i equals the column part of grid column i (both come from the same
 * `freqs` tensor, rope.py:134-138) and the two dims of a rotation pair share their entry (`repeat(..., r = 2)`, :137) -- i.e.
 * cos_t[(i*g)*64 + 2j] == cos_t[(i*g)*64 + 2j+1] == cos_t[i*64 + 32 + 2j], same for sin_t.  Identity tables (cos 1, sin 0: the OpenAI-CLIP
 * family) satisfy it.  The Python wrapper (clipself_amd/hip.py) verifies the tables once per tensor and raises otherwise. */
int cs_attn_fwd(const void* qkv, const float* cos_t, const float* sin_t, void* out, float* lse, int B, int Ntok, int H,
                int ldqkv, int ldo, float scale, cs_stream_t stream);
/* CLS-query attention for the frozen teacher's last block: VisionTransformer.forward_features returns x[:, 0]
 * (eva_vit_model.py:505-519), so only that query row of the last block is live.  q [B, ldq] bf16 (CLS queries, never rotated);
 * kv [B*Ntok, ldkv] bf16 = k|v; out [B, ldo] bf16. */
int cs_attn_cls_fwd(const void* q, const void* kv, const float* cos_t, const float* sin_t, void* out, int B, int Ntok, int H,
                    int ldq, int ldkv, int ldo, float scale, cs_stream_t stream);
/* Extra query tokens of the OpenAI-CLIP family's mask-attention pooling: VisionTransformer.mask_attn_pool / _mask_attn_pool
 * (src/open_clip/transformer.py:736-834), reached through extract_type='v1' (:660-671) and CLIP.encode_masks(mask_attn=True)
 * (src/open_clip/model.py:245-247).  Q query rows per image attend the image's own keys / values of the same depth: key j of query row r
 * is allowed iff allow[r * Ntok + j] != 0 (the reference's bool attn_mask, inverted; key 0 = the CLS token).  q [B*Q, ldq] bf16;
 * kv [B*Ntok, ldkv] bf16 = k|v; out [B*Q, ldo] bf16.  No rotary embedding in this family.  Inference only (no backward).  A row that allows
 * no key at all (the reference always allows key 0) yields a zero output row, not NaN. */
int cs_attn_query_fwd(const void* q, const void* kv, const unsigned char* allow, void* out, int B, int Q, int Ntok, int H,
                      int ldq, int ldkv, int ldo, float scale, cs_stream_t stream);
/* cs_attn_fwd that also emits stats_part [H][B*Ntok][2] f32 = per head (sum, sum of squares) of each output row's 64 values. */
int cs_attn_fwd_stats(const void* qkv, const float* cos_t, const float* sin_t, void* out, float* lse, float* stats_part, int B, int Ntok,
                      int H, int ldqkv, int ldo, float scale, cs_stream_t stream);
size_t cs_attn_bwd_workspace(int B, int Ntok, int H);
int cs_attn_bwd(const void* qkv, const void* o, const void* dout, const float* lse, const float* cos_t, const float* sin_t,
                void* dqkv, void* workspace, int B, int Ntok, int H, int ldqkv, int ldo, float scale, cs_stream_t stream);

/* --- SwiGLU elementwise: eva_vit_model.py:101  hidden = silu(x1) * x2   (x12 = [x1 | x2], each Hd wide) */
int cs_swiglu_fwd(const void* x12, long ldx, void* h, long ldh, int M, int Hd, cs_stream_t stream);
int cs_swiglu_bwd(const void* dh, long lddh, const void* x12, long ldx, void* dx12, long lddx, int M, int Hd, cs_stream_t stream);
/* the same + colsum[2*Hd] += the column sums of dx12 = the bias gradients of w1 | w2 (autograd of F.linear's bias at eva_vit_model.py:99-100),
   bit-identical to cs_swiglu_bwd followed by cs_colsum_bf16(dx12) without its pass over the matrix; workspace >= cs_colsum_workspace(M, 2*Hd) */
int cs_swiglu_bwd_colsum(const void* dh, long lddh, const void* x12, long ldx, void* dx12, long lddx, float* colsum, void* workspace,
                         int M, int Hd, cs_stream_t stream);
/* the same + the e4m3 copy of dx12 (row-wise amax / 448 scales, bytes [dx1 | dx2 | zero padding to a multiple of 128]) for an fp8 dgrad through
   cs_gemm_nt_f8 -- bit-identical to cs_quant_rows_fp8(dx12), without its pass over the matrix; Hd <= 4096 */
int cs_swiglu_bwd_q8(const void* dh, long lddh, const void* x12, long ldx, void* dx12, long lddx, void* q8, long ldq, float* q_scale,
                     int M, int Hd, cs_stream_t stream);

/* --- GELU / QuickGELU elementwise on bf16 [M,N] (training path keeps the c_fc output): src/open_clip/transformer.py:31-34,211
 *     y = act(x); dx = dy * act'(x); quick 0 = nn.GELU (erf), 1 = x*sigmoid(1.702x) */
int cs_gelu_fwd(const void* x, long ldx, void* y, long ldy, int M, int N, int quick, cs_stream_t stream);
int cs_gelu_bwd(const void* dy, long lddy, const void* x, long ldx, void* dx, long lddx, int M, int N, int quick, cs_stream_t stream);

/* --- data movement helpers of the step */
int cs_cast_f32_bf16(const float* x, void* y, long n, cs_stream_t stream);
int cs_transpose_bf16(const void* in, long ld_in, void* out, long ld_out, int R, int Cc, cs_stream_t stream); /* out[c,r]; zero pad r in [R, ld_out) */
/* `count` such transposes in one launch (the W^T shadows the dgrad GEMMs read are rebuilt after every AdamW step: eva_vit_model.py:99-103,
 * 177-179,218-219 are the linears; 49 matrices per step for B/16).  desc = DEVICE array of 48-byte records
 *   { const void* in; void* out; long ld_in; long ld_out; int R; int Cc; int tile0; int tiles_x; }
 * with tiles_x = ceil(Cc / 64), tile0 = sum over the preceding records of tiles_x * ceil(ld_out / 64); total_tiles = that sum over all. */
int cs_transpose_bf16_batched(const void* desc, int count, int total_tiles, cs_stream_t stream);
size_t cs_colsum_workspace(int M, int N);
int cs_colsum_bf16(const void* x, long ldx, float* out, void* workspace, int M, int N, cs_stream_t stream);   /* out[n] += sum_m x[m,n] (bias grads); fixed summation order, no atomics */
int cs_im2row(const void* img, int img_dtype, void* out, int B, int S, int p, int ldo, cs_stream_t stream);    /* PatchEmbed unfold, eva_vit_model.py:355 */
int cs_cls_row(float* x, const float* cls, const float* pos, int B, int Ntok, int C, cs_stream_t stream);      /* x[b,0,:] = cls + pos[0], :540-543 */

/* --- RoIAlign 1x1 / aligned / adaptive sampling on the token-major map: torchvision.ops.roi_align called at
 *     eva_vit_model.py:628-629 with boxes from _denormalize_boxes :655-664 (boxes given normalised to [0,1]). */
int cs_roialign_fwd(const float* feat, const float* rois, float* pooled, int K, int Ntok, int grid_h, int grid_w, int E,
                    int tok_off, cs_stream_t stream);
/* backward: dfeat [B, Ntok, E] += the boxes' gradients; a gather per map cell over the image's boxes in ascending box order (no atomics:
 * bit-reproducible, unlike torchvision's atomicAdd scatter); any E: 1024 channels per workgroup, wider maps take more workgroups) */
int cs_roialign_bwd(const float* dpooled, const float* rois, float* dfeat, int K, int B, int Ntok, int grid_h, int grid_w, int E,
                    int tok_off, cs_stream_t stream);

/* --- cosine distillation loss: src/training/clipself.py:42-47.  stats [K,3] f32 workspace kept for the backward. */
int cs_cosine_loss_fwd(const float* student, const float* teacher, float* stats, float* loss, int K, int E, float weight,
                       cs_stream_t stream);
/* upstream: optional device scalar d(total)/d(loss) multiplied in on the device (nullable = 1). */
int cs_cosine_loss_bwd(const float* student, const float* teacher, const float* stats, float* dstudent, int K, int E,
                       float weight, float grad_scale, const float* upstream, cs_stream_t stream);

/* --- RegionCLIP federated BCE over the sampled noun columns: src/training/region_clip.py:47-56
 *     (F.binary_cross_entropy_with_logits(...).sum(-1).mean() on logits * temp; one-hot target at tgt[k], -1 = none).
 * bwd writes d(logits) as bf16 [K, ldd] with zeroed padding columns (operand of the d(features) GEMM). */
int cs_fed_bce_fwd(const float* logits, long ldz, const int* tgt, float* rowloss, float* loss, int K, int ns, float temp,
                   float weight, cs_stream_t stream);
int cs_fed_bce_bwd(const float* logits, long ldz, const int* tgt, void* dz_bf16, long ldd, int K, int ns, float temp,
                   float weight, const float* upstream, cs_stream_t stream);

/* --- input pipeline on the GPU: the region crops of GridDistillDataset._obtain_image_crops (src/training/data.py:226-245,
 *     transforms[1] = ResizeMaxSize + ToTensor + Normalize, src/open_clip/transform.py:26-49,93-99) and the det image itself
 *     (ResizeLongest, transform.py:169-191) from ONE decoded RGB image resident in HBM.  Pillow's bicubic resampling (PIL.Image.crop +
 *     Image.resize as reached through torchvision.transforms.functional.resize) is restated bit-exactly: fixed-point coefficients,
 *     horizontal pass then vertical pass, uint8 after each.
 * src [H,W,3] u8; boxes [K,4] f32 pixel (x0,y0,x1,y1), device; mean3/std3: HOST arrays of 3 floats; out [K,3,S,S] f32;
 * pad_center 1 = centred zero padding (crop transform), 0 = right/bottom (det transform); workspace >= cs_crop_resize_workspace bytes. */
size_t cs_crop_resize_workspace(int H, int K, int S);
int cs_crop_resize_u8(const void* src, int H, int W, const float* boxes, int K, int S, int pad_center, const float* mean3,
                      const float* std3, float* out, void* workspace, cs_stream_t stream);

/* --multiscale (src/training/clipself.py:17-27): F.interpolate(images, size=(t, t), mode='bilinear') of the student batch.
 * in [planes,H,W] f32 -> out [planes,Ho,Wo] f32, align_corners=False arithmetic of torch. */
int cs_resize_bilinear_f32(const float* in, float* out, int planes, int H, int W, int Ho, int Wo, cs_stream_t stream);

/* --- sharing one GPU between the two towers of a step.  src/training/clipself.py:36-40 runs the frozen teacher (`dist_model.encode_image`
 * under no_grad) and the student one after the other on one CUDA stream; here the teacher's pass over the NEXT batch runs beside the
 * student's step (src/training/train.py:90-115), each on its own share of the compute units: persistent GEMM grids sized by cs_gemm_nt's
 * flags bits 20-27 and, optionally, queues restricted by a CU mask.  cs_num_compute_units: CUs of the current device (256 on MI355X).
 * cs_stream_create_cu_mask: a stream whose kernels only run on mask bits [first_cu, first_cu + n_cus) (multiples of 8: the driver deals
 * bit i to XCD i % 8, so such a range takes n_cus / 8 CUs from every XCD); destroy it with cs_stream_destroy. */
int cs_num_compute_units(void);
int cs_stream_create_cu_mask(int first_cu, int n_cus, cs_stream_t* out);
int cs_stream_destroy(cs_stream_t stream);

/* --- optimizer: torch.optim.AdamW built at src/training/main.py:198-213, stepped at src/training/train.py:115.
 * Flat fp32 master/grad/moment buffers; flags[n/64]: bit0 = tensor has a gradient this step, bit1 = weight decay applies. */
int cs_adamw_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, const uint8_t* flags, long n, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, cs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
