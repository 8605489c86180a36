/*
 * vjepa_hip.h -- C ABI of libvjepa_hip.so: the MI355X (gfx950 / CDNA4) kernels of the V-JEPA pretraining step.
 *
 * The reference (facebookresearch/jepa) has no FFI layer: its device work is the ATen operator stream issued by
 * app/vjepa/train.py:414-498 through the modules of src/models.  This header is the boundary a maintainer would bind instead of
 * those ATen calls (see INTEGRATION.md for the ctypes stub).  Each entry point names the reference call site it
 * replaces.  Conventions:
 *   - every function returns 0 on success, a negative value for an argument error, or a positive hipError_t;
 *     vj_last_error() returns a thread-local description.  No C++ exception crosses this boundary.
 *   - all pointers are DEVICE pointers owned by the caller (borrowed for the duration of the launch);
 *     the library never allocates, frees or retains them.  Workspaces are passed in (vj_*_ws_bytes()).
 *   - every launch goes to the explicit hipStream_t (last argument); entry points are re-entrant.
 *   - "bf16" buffers are raw bfloat16 bits (uint16_t); indices are int64 as produced by the reference collator
 *     (src/masks/multiblock3d.py:155-203).
 *   - no torch types appear here; PyTorch is only the allocator/stream provider on the host side.
 */
#ifndef VJEPA_HIP_H
#define VJEPA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* vj_stream_t; /* == hipStream_t */

int vj_abi_version(void);
const char* vj_last_error(void);

/* ---- token-row movement (bit-exact) -------------------------------------------------------------------------
 * apply_masks: torch.gather(x, 1, idx[..., None].repeat(1,1,D))            src/masks/utils.py:11-23
 * dst[b,k,:] = src[b*src_batch_stride_rows + idx[b,k], :]; rows are row_bytes wide (multiple of 4).
 * src_batch_stride_rows = N for a [B,N,D] source, 0 to broadcast a [1,N,D] table (predictor.py:199,214). */
int vj_gather_rows(const void* src, void* dst, const int64_t* idx, int64_t B, int64_t K, int64_t row_bytes,
                   int64_t src_batch_stride_rows, vj_stream_t stream);
/* backward of the gather (aten::gather_backward -> scatter_add_ into zeros; indices unique per sample):
 * dst[B,N,row] is zero-filled, then dst[b, idx[b,k], :] = src[b,k,:]. */
int vj_scatter_rows(const void* src, void* dst, const int64_t* idx, int64_t B, int64_t N, int64_t K,
                    int64_t row_bytes, vj_stream_t stream);
/* bf16 row-slice copy: dst[b, dst_off+j, :] = src[b, src_off+j, :], j < n  (x[:, N_ctxt:], predictor.py:236);
 * src == NULL zero-fills those destination rows instead (autograd's zero gradient of the rows x[:, :N_ctxt] that the slice drops) */
int vj_copy_rows(const void* src, void* dst, int64_t B, int64_t src_rows, int64_t src_off, int64_t dst_rows,
                 int64_t dst_off, int64_t n, int64_t D, vj_stream_t stream);

/* ---- tubelet PatchEmbed3D front end -------------------------------------------------------------------------
 * Conv3d(3->D, k=s=(tub,p,p)) + flatten(2).transpose(1,2)                  src/models/utils/patch_embed.py:31-57
 * is a GEMM over non-overlapping tubelets; this packs fp32 clips [B,C,T,H,W] into its bf16 A operand
 * [B,K,C*tub*p*p] (element order c,dt,dh,dw = Conv3d weight order).  idx (nullable, [B,K]) fuses the context
 * mask gather (only kept tubelets are packed); NULL packs all K = N tokens in (t,h,w) order. */
int vj_tubelet_pack(const float* clips, void* out_bf16, const int64_t* idx, int64_t B, int64_t C, int64_t T,
                    int64_t H, int64_t W, int64_t tubelet, int64_t patch, int64_t K, vj_stream_t stream);
/* x += pos_embed (vision_transformer.py:172-174); with idx: x[b,k] += pos[idx[b,k]] (gather fused). */
int vj_add_pos(void* x_bf16, const float* pos, const int64_t* idx, int64_t B, int64_t K, int64_t D,
               vj_stream_t stream);

/* ---- LayerNorm ----------------------------------------------------------------------------------------------
 * nn.LayerNorm(eps=1e-6) (modules.py:97,106,115,119; vision_transformer.py:193; predictor.py:233).
 * bf16 in/out, fp32 statistics; mean/rstd (nullable pair) are saved for the backward. */
int vj_layernorm_fwd(const void* x_bf16, const float* gamma, const float* beta, void* y_bf16, float* mean,
                     float* rstd, int64_t rows, int64_t D, float eps, vj_stream_t stream);
int64_t vj_layernorm_bwd_ws_bytes(int64_t D);
/* dx = LN'(dy) [+ dres]; dgamma/dbeta = alpha*sum + beta_acc*old (fp32, into the gradient arena). */
int vj_layernorm_bwd(const void* dy_bf16, const void* x_bf16, const float* gamma, const float* mean,
                     const float* rstd, const void* dres_bf16, void* dx_bf16, float* dgamma, float* dbeta,
                     float alpha, float beta_acc, int64_t rows, int64_t D, void* ws, int64_t ws_bytes,
                     vj_stream_t stream);

/* the same, plus dxsum[D] (nullable) = alpha * column sums of dx + beta_acc * old: in a Block, dx of norm2's backward is the dY
 * of attn.proj and dx of norm1's backward the dY of the previous block's mlp.fc2, so autograd's bias gradients of those two
 * Linears (sum over tokens of dY, modules.py:34,76) come out of this pass instead of a separate read of dY. */
int vj_layernorm_bwd_colsum(const void* dy_bf16, const void* x_bf16, const float* gamma, const float* mean,
                            const float* rstd, const void* dres_bf16, void* dx_bf16, float* dgamma, float* dbeta,
                            float* dxsum, float alpha, float beta_acc, int64_t rows, int64_t D, void* ws,
                            int64_t ws_bytes, vj_stream_t stream);

/* ---- bf16 MFMA GEMM  C[M,N] = A[M,K] * B[N,K]^T (+ fused epilogue) ------------------------------------------
 * nn.Linear fwd/bwd (modules.py:31-34,63,76; predictor.py:194,237) and the Conv3d GEMM (patch_embed.py:56).
 * epilogue: 0 bf16 out = acc [+bias] [+residual]      (qkv / proj+residual / fc2+residual / dgrads)
 *           1 bf16 out = gelu(u), u = bf16(acc+bias); aux_out(nullable) = bf16(gelu'(u))   (fc1 + nn.GELU, modules.py:31-32;
 *             the derivative is saved INSTEAD of the pre-activation: it shares the erfc evaluation of the forward and is
 *             the only thing the backward needs from u)
 *           2 bf16 out = acc * aux_in, aux_in = what epilogue 1 saved   (fc2 dgrad fused with the GELU backward)
 *           3 fp32 out = alpha*acc + beta*C            (wgrad straight into the fp32 gradient arena)
 *           4 bf16 out = (acc + bias) * (n < N/3 ? alpha : 1)   (the qkv projection of Attention, modules.py:63, whose q part carries
 *             the soft-max scale alpha = head_dim^-0.5 * log2(e) with ONE rounding; the attention entry points are then called with
 *             a NEGATIVE scale = "q is pre-scaled"; N % 12 == 0, no residual)
 * K % 32 == 0, N % 4 == 0, lda/ldb % 8 == 0.  flags bits 4-8: kernel selection (0 = automatic).  The GELU epilogues (1) evaluate
 * Phi(-|x|) as exp2 of a degree-6 polynomial: 5 of the 24 222 bf16 inputs in (-5, 2^127) round differently from the correctly
 * rounded bf16 erf-GELU of nn.GELU() (tests/test_gelu_poly.py). */
int vj_gemm_bf16_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M,
                    int64_t N, int64_t K, const float* bias, const void* residual, int64_t ldr, const void* aux_in,
                    void* aux_out, int64_t ldaux, int epilogue, float alpha, float beta, int flags,
                    vj_stream_t stream);
/* fc2 dgrad (epilogue 2) that also produces the bias gradient of fc1 (autograd of Mlp.fc1's bias, modules.py:31-34: the sum
 * over tokens of the dY this GEMM writes): when the persistent 256 x 256 kernel takes the problem, colpart
 * [vj_gemm_colsum_rows(M)][N] receives fp32 column sums of C (before the bf16 rounding) per (row tile, wave row) and *fused = 1;
 * otherwise the plain GEMM runs and *fused = 0 (sum C with vj_colsum_bf16).  C is bit-identical to vj_gemm_bf16_nt's. */
int64_t vj_gemm_colsum_rows(int64_t M);
int vj_gemm_bf16_nt_dgelu_colsum(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M,
                                 int64_t N, int64_t K, const void* aux_in, int64_t ldaux, float* colpart,
                                 int64_t colpart_rows, int flags, int* fused, vj_stream_t stream);
/* wgrad form of the same GEMM: C (fp32) = alpha*A*B^T + beta*C where K (= tokens) is long and the [M,N] tile grid
 * alone cannot fill 256 CUs: K is split across workgroups, partials combined deterministically from ws
 * (ws_bytes >= M*N*4; more workspace allows more slices). */
int vj_gemm_bf16_nt_splitk(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int64_t M,
                           int64_t N, int64_t K, float alpha, float beta, int flags, void* ws, int64_t ws_bytes,
                           vj_stream_t stream);
/* weight gradient WITHOUT operand transposes (autograd of nn.Linear, modules.py:31-34,63,76):
 *   dW[N1,N2] (fp32) = alpha * dY[T,N1]^T * X[T,N2] + beta * dW,   T = tokens (any positive count; the last 64-token
 * tile is completed with zero rows inside the kernel).  Both operands are read row-major as the backward pass left
 * them; N1 % 8 == 0, N2 % 8 == 0, ldy/ldx % 8 == 0 and < 2^24, 16-byte aligned bases; split-K over the tokens as
 * above (ws_bytes >= N1*N2*4). */
int vj_gemm_bf16_tn_splitk(const void* dY, int64_t ldy, const void* X, int64_t ldx, float* dW, int64_t ldw, int64_t T,
                           int64_t N1, int64_t N2, float alpha, float beta, void* ws, int64_t ws_bytes,
                           vj_stream_t stream);
/* The weight gradients of n <= 4 Linear layers over the SAME T tokens in ONE launch -- a transformer block's qkv, proj,
 * fc1, fc2 (the four nn.Linear backward nodes of Block.forward, modules.py:31-34,63,76).  Per problem the contract of
 * vj_gemm_bf16_tn_splitk; one split factor for the group; ws_bytes >= 4 * sum N1*N2 (more allows split-K). */
typedef struct vj_tn_problem {
  const void* dY; int64_t ldy; /* [T, N1] bf16 */
  const void* X;  int64_t ldx; /* [T, N2] bf16 */
  float* dW;      int64_t ldw; /* [N1, N2] fp32 */
  int64_t N1, N2;
} vj_tn_problem_t;
int vj_gemm_bf16_tn_grouped(const vj_tn_problem_t* problems, int64_t n, int64_t T, float alpha, float beta, void* ws,
                            int64_t ws_bytes, vj_stream_t stream);
/* out[N, Mpad] = in[M,N]^T (zero padded): the wgrad operands dY^T, X^T; weight shadows W^T for dgrad. */
int vj_transpose_bf16(const void* in, void* out, int64_t M, int64_t N, int64_t ld_in, int64_t Mpad,
                      vj_stream_t stream);
/* one launch for many transposes (all W^T dgrad shadows after an optimizer step): desc = device array of
 * {src, dst, M, N, ld_in, Mpad} (6 x int64 per tensor), blocks = device int32[4*n_blocks] {tensor, tile_m, tile_n, 0} */
int vj_transpose_multi(const void* desc, const void* blocks, int64_t n_blocks, vj_stream_t stream);
/* transpose + bias gradient in ONE pass over dY: colsum[n] = alpha*sum_m in[m][n] + beta*colsum[n] */
int64_t vj_transpose_colsum_ws_bytes(int64_t M, int64_t N);
int vj_transpose_colsum_bf16(const void* in, void* out, int64_t M, int64_t N, int64_t ld_in, int64_t Mpad,
                             float* colsum, float alpha, float beta, void* ws, int64_t ws_bytes, vj_stream_t stream);
/* bias / mask-token gradients: out[n] = alpha * sum_{m: row_lo <= m % group < row_hi} in[m][n] + beta*out[n] */
int64_t vj_colsum_ws_bytes(int64_t N);
int vj_colsum_bf16(const void* in, int64_t M, int64_t N, int64_t ld, int64_t group, int64_t row_lo, int64_t row_hi,
                   float* out, float alpha, float beta, void* ws, int64_t ws_bytes, vj_stream_t stream);
int vj_reduce_partials(const float* part, float* out, int64_t P, int64_t N, float alpha, float beta,
                       vj_stream_t stream);
/* n <= 16 such reductions in ONE launch: out_s[n] = alpha * sum_p part_s[p*stride_s + n] + beta * out_s[n]; same summation
 * order (hence the same bits) as vj_reduce_partials.  The backward of a Block ends with one of these: LayerNorm dgamma / dbeta,
 * the proj / fc2 bias sums (modules.py:34,76) and the qkv / fc1 bias partials of the kernels that produce their dY. */
typedef struct vj_reduce_seg {
  const float* part; /* [P, stride] fp32 partials */
  float* out;        /* [N] */
  int64_t P, N, stride;
} vj_reduce_seg_t;
int vj_reduce_segments(const vj_reduce_seg_t* segs, int64_t n_segs, float alpha, float beta, vj_stream_t stream);

/* ---- attention ----------------------------------------------------------------------------------------------
 * F.scaled_dot_product_attention(q,k,v) (modules.py:66-69): dense, non-causal, scale = head_dim^-0.5.
 * qkv: packed qkv-Linear output [B,S,3,H,hd] (modules.py:63 before the permute); o: [B,S,H*hd];
 * lse2: [B,H,S] fp32 log2-sum-exp saved for the backward (nullable in inference).  hd % 8 == 0, hd <= 128. */
/* scale > 0: the kernels fold scale*log2(e) into the stationary operand of the score product themselves (one more bf16 rounding
 * of q / k); scale < 0: the caller's qkv GEMM already stored q * |scale| * log2(e) (epilogue 4 above) -- same for the backward. */
int vj_attn_fwd(const void* qkv, void* o, float* lse2, int64_t B, int64_t S, int64_t H, int64_t hd, float scale,
                vj_stream_t stream);
/* The same for n_segs <= 4 segments of ONE token-major activation in one launch -- the rows of the masks of a V-JEPA batch
 * are concatenated (MultiMaskWrapper loops over them, src/models/utils/multimask.py:17-27): qkv [M, 3*H*hd], o [M, H*hd],
 * lse2 [H*M] with segment i's block [B_i, H, S_i] at offset H*row0_i.  The short segment's workgroups fill the tail of the
 * long one instead of paying their own launch. */
typedef struct vj_seg {
  int64_t row0; /* first token row of the segment */
  int64_t B, S; /* rows row0 .. row0 + B*S are B sequences of S tokens */
} vj_seg_t;
int vj_attn_fwd_segs(const void* qkv, void* o, float* lse2, const vj_seg_t* segs, int64_t n_segs, int64_t H, int64_t hd,
                     float scale, vj_stream_t stream);
int64_t vj_attn_bwd_ws_bytes(int64_t B, int64_t S, int64_t H);   /* the two-kernel form's delta [B,H,S] only: prefer the next one */
/* workspace bytes of vj_attn_bwd / vj_attn_bwd_segs / vj_attn_bwd_colsum for a segment list: delta = rowsum(dO . O) of the softmax
 * backward (autograd of modules.py:66-69), 4*H bytes per token row up to the last row of the list */
int64_t vj_attn_bwd_segs_ws_bytes(const vj_seg_t* segs, int64_t n_segs, int64_t H, int64_t hd);
/* dqkv [B,S,3,H,hd] <- (dout [B,S,H*hd], saved qkv, o, lse2) */
int vj_attn_bwd(const void* qkv, const void* o, const void* dout, const float* lse2, void* dqkv, int64_t B,
                int64_t S, int64_t H, int64_t hd, float scale, void* ws, int64_t ws_bytes, vj_stream_t stream);
/* the same, plus the column sums of dqkv over this [B,S] segment as fp32 partials -- the qkv bias gradient (autograd of
 * Attention.qkv's bias, modules.py:63) from the kernels that produce dqkv instead of a second pass over it:
 * colq [rows_q][H*hd] (dQ kernel: one row per (sample, 128-query block)), colkv [rows_kv][2*H*hd] (dK/dV kernel: one row per
 * (sample, key block)); row counts from vj_attn_bwd_colsum_rows (they depend on the head-dim class); every element is
 * written; dqkv is bit-identical to vj_attn_bwd's.  Reduce with vj_reduce_segments. */
int vj_attn_bwd_colsum_rows(int64_t B, int64_t S, int64_t hd, int64_t* rows_q, int64_t* rows_kv);
int vj_attn_bwd_colsum(const void* qkv, const void* o, const void* dout, const float* lse2, void* dqkv, int64_t B,
                       int64_t S, int64_t H, int64_t hd, float scale, void* ws, int64_t ws_bytes, float* colq,
                       float* colkv, vj_stream_t stream);
/* backward over n_segs <= 4 segments in one launch pair (dQ, then dK/dV); ws >= 4*H*M bytes; colq / colkv (both or neither,
 * nullable): the column partials, segment after segment in list order */
int vj_attn_bwd_segs(const void* qkv, const void* o, const void* dout, const float* lse2, void* dqkv, const vj_seg_t* segs,
                     int64_t n_segs, int64_t H, int64_t hd, float scale, void* ws, int64_t ws_bytes, float* colq,
                     float* colkv, vj_stream_t stream);

/* ---- cross-attention of a few learned queries against frozen-encoder tokens (attentive probe, row f4 widened) ------
 * CrossAttention.forward (src/models/utils/modules.py:140-157) as used by CrossAttentionBlock (modules.py:177-181) inside
 * AttentivePooler / AttentiveClassifier (src/models/attentive_pooler.py:96-136; num_queries = 1 there).
 * q: [B or 1, NQ, H*hd] bf16 (q_bstride = elements between samples, 0 when every sample shares the projected query);
 * kv: packed kv-Linear output [B,N,2,H,hd] (modules.py:145 before the permute); resid (nullable): [NQ, H*hd] bf16 added to
 * the result (the block's q + xattn(...)); out: [B,NQ,H*hd] bf16; lse2: [B,H,NQ] fp32 (nullable in inference).
 * Matrix-vector work, HBM-bound: 4*N*hd bytes per (sample, head, query) forward, 12*N*hd backward.  The reference's
 * `proj` Linear of CrossAttention is never applied (modules.py:156-157) and is not applied here. */
int vj_xattn_fwd(const void* q, int64_t q_bstride, const void* kv, const void* resid, void* out, float* lse2, int64_t B,
                 int64_t NQ, int64_t N, int64_t H, int64_t hd, float scale, vj_stream_t stream);
/* dq [B,H*hd] bf16 (per sample; the caller sums over the batch when the query is shared), dkv [B,N,2,H,hd] bf16
 * <- (dy [B,1,H*hd] bf16, saved q, kv, lse2).  NQ must be 1. */
int vj_xattn_bwd(const void* q, int64_t q_bstride, const void* kv, const void* dy, const float* lse2, void* dq, void* dkv,
                 int64_t B, int64_t NQ, int64_t N, int64_t H, int64_t hd, float scale, vj_stream_t stream);

/* ---- predictor token assembly (predictor.py:194-221) --------------------------------------------------------
 * out[b, j<Ke] = embed[b,j] + pos[idx_e[b,j]];  out[b, Ke+j] = mask_token + pos[idx_p[b,j]] */
int vj_pred_assemble_fwd(const void* e_bf16, const float* mask_token, const float* pos, const int64_t* idx_e,
                         const int64_t* idx_p, void* out_bf16, int64_t B, int64_t Ke, int64_t Kp, int64_t D,
                         vj_stream_t stream);

/* ---- targets and loss (app/vjepa/train.py:419-459) ----------------------------------------------------------
 * h[b,k,:] = F.layer_norm( norm(x)[b, idx[b,k], :] ), fp32 out: final encoder LayerNorm (eps_norm), the
 * affine-free F.layer_norm (eps_ln = 1e-5, train.py:426) and apply_masks fused over the predicted rows only. */
int vj_target_rows(const void* x_bf16, const float* gamma, const float* beta, const int64_t* idx, float* h,
                   int64_t B, int64_t N, int64_t K, int64_t D, float eps_norm, float eps_ln, vj_stream_t stream);
/* loss_out (+)= out_scale * sum(|z-h|^p / p); dz (nullable) = sign(z-h)|z-h|^(p-1) * gscale  (train.py:440-446) */
int64_t vj_latent_loss_ws_bytes(void);
int vj_latent_loss(const void* z_bf16, const float* h, void* dz_bf16, int64_t numel, float p, float gscale,
                   float out_scale, int accumulate, float* loss_out, void* ws, int64_t ws_bytes,
                   vj_stream_t stream);
/* reg_fn (train.py:448-449,458): pstd[b,d] (+)= sqrt(var_k z[b,k,d] + 1e-4); reg = mean(relu(1 - pstd/n_masks)) */
int vj_token_pstd(const void* z_bf16, float* pstd, float* stats /* nullable [B,D,2] = {mean, sqrt(var+eps)} */,
                  int64_t B, int64_t K, int64_t D, int accumulate, vj_stream_t stream);
/* backward of the regulariser: dz -= coef * 1[pstd_sum/n_masks < 1] * (z - mean) / ((K-1) * sqrt(var+eps)) */
int vj_reg_grad(const void* z_bf16, const float* pstd_sum, const float* stats, void* dz_bf16, int64_t B, int64_t K,
                int64_t D, int64_t n_masks, float coef, vj_stream_t stream);
int vj_reg_finish(const float* pstd_sum, int64_t n, int64_t n_masks, float* out, vj_stream_t stream);

/* ---- parameter update over flat fp32 arenas (train.py:461-487; app/vjepa/utils.py:156-210) -------------------
 * torch.optim.AdamW step (decoupled wd, bias correction from `step`), EMA of the target encoder
 * (param_k = m*param_k + (1-m)*param_q, train.py:486-487) and the bf16 re-cast of both weight sets, one pass.
 * p_bf16 / tgt / tgt_bf16 are nullable; gscale pre-multiplies the gradient (clip coefficient). n % 4 == 0. */
int vj_adamw_ema(float* p, const float* g, float* exp_avg, float* exp_avg_sq, void* p_bf16, float* tgt,
                 void* tgt_bf16, int64_t n, float lr, float wd, float beta1, float beta2, float eps, int64_t step,
                 float gscale, float ema, vj_stream_t stream);
/* Device-guarded form of the same update: nothing between backward and update needs the host.
 *   gstat = [sumsq_enc, nonfinite_enc, sumsq_pred, nonfinite_pred] (device; two vj_sqnorm_f32 results);
 *   any non-finite gradient -> AdamW is skipped for every range (GradScaler.step, train.py:471; utils.py:209) while the
 *   EMA still runs; clip > 0 applies torch.nn.utils.clip_grad_norm_'s coefficient min(1, clip/(norm+1e-6)) with
 *   norm = sqrt(gstat[2*sel]) * norm_scale (train.py:468-470); the Adam step count t is the device float *step_dev,
 *   advanced by vj_step_advance only when the step is not skipped. */
int vj_step_advance(const float* gstat, float* step_dev, vj_stream_t stream);
int vj_adamw_ema_guarded(float* p, const float* g, float* exp_avg, float* exp_avg_sq, void* p_bf16, float* tgt,
                         void* tgt_bf16, int64_t n, float lr, float wd, float beta1, float beta2, float eps, float gscale,
                         float ema, const float* gstat, int sel, float clip, float norm_scale, const float* step_dev,
                         vj_stream_t stream);
/* Logging statistics without per-tensor host syncs (grad_logger / adamw_logger, src/utils/logging.py:91-118;
 * train.py:476-481): desc = device int64 [n_tensors][2] {element offset (multiple of 4), numel} into the flat arenas;
 * out = device fp32 [n_tensors][vj_grad_stats_chunks()][3] partial sums of {g^2, |exp_avg|, |exp_avg_sq|}
 * (M1 = M2 = NULL: gradients only).  One launch; evaluated by the caller only when a log line is due. */
int64_t vj_grad_stats_chunks(void);
int vj_grad_stats_multi(const float* G, const float* M1, const float* M2, const int64_t* desc, int64_t n_tensors,
                        float* out, vj_stream_t stream);
int vj_ema_update(float* tgt, const float* src, void* tgt_bf16, int64_t n, float m, vj_stream_t stream);
int vj_cast_f32_to_bf16(const float* src, void* dst_bf16, int64_t n, vj_stream_t stream);
/* out2[0] (+)= sum g^2, out2[1] (+)= count of non-finite values  (clip_grad_norm_, GradScaler inf check) */
int64_t vj_sqnorm_ws_bytes(void);
int vj_sqnorm_f32(const float* g, int64_t n, float* out2, int accumulate, void* ws, int64_t ws_bytes,
                  vj_stream_t stream);

/* ---- whole-trunk launch chains (host-side sequencing only; no arithmetic of their own) ------------------------
 * N transformer blocks, forward or backward, enqueued by ONE call: Block.forward / Attention.forward / MLP.forward
 * (src/models/utils/modules.py:30-36,61-78,114-120) as the loops of VisionTransformer.forward
 * (src/models/vision_transformer.py:181-184) and VisionTransformerPredictor.forward (src/models/predictor.py:231-232)
 * run them, and the autograd graph that loss.backward() (app/vjepa/train.py:461-464) walks over them.
 * Token rows of several equal-length sequence groups (one per mask) are concatenated along M; `segs` describes them
 * (attention is launched per segment, everything else over all M rows).
 * All buffers are borrowed; workspaces are caller-allocated, 256-byte aligned, sized by the *_ws_bytes queries. */
typedef struct vj_linear {
  const void* w;    /* bf16 [n_out, k_in]                                   (forward operand)              */
  const float* b;   /* fp32 [n_out] or NULL                                                                  */
  const void* wT;   /* bf16 [k_in, ldwT >= n_out] transposed shadow          (dgrad operand; NULL = fwd only) */
  int64_t ldwT;
  float* gw;        /* fp32 [n_out, k_in] weight-gradient view               (NULL = fwd only)                */
  float* gb;        /* fp32 [n_out] bias-gradient view or NULL                                               */
  int64_t n_out, k_in;
} vj_linear_t;
typedef struct vj_norm {
  const float* g;
  const float* b;
  float* gg; /* gradient views, NULL = fwd only */
  float* gb;
} vj_norm_t;
typedef struct vj_block {
  vj_norm_t norm1;
  vj_linear_t qkv, proj;
  vj_norm_t norm2;
  vj_linear_t fc1, fc2;
} vj_block_t;
/* (vj_seg_t: declared with the attention entry points above) */
typedef void (*vj_layer_cb_t)(void* user, int layer);

/* save != 0: every block keeps what its backward needs (16*D bf16 per token: x, LN1 out, qkv, o, x1, LN2 out,
 * pre-GELU u, GELU out + row statistics + softmax lse) in `ws`; save == 0: one such set is reused by all blocks. */
int64_t vj_blocks_fwd_ws_bytes(int64_t M, int64_t D, int64_t Dh, int64_t heads, int64_t n_blocks, int save);
/* x_out [M,D] bf16 = blocks[n-1](...blocks[0](x_in)) ; x_in must stay valid until the backward has run.
 * gemm_flags: kernel selection for the four Linear GEMMs of every block, as in vj_gemm_bf16_nt (0 = automatic;
 * 0x100 = the two-workgroups-per-CU kernel, the better choice when this stream has the GPU to itself: inference). */
int vj_blocks_fwd(const vj_block_t* blocks, int64_t n_blocks, const void* x_in, void* x_out, int64_t M, int64_t D,
                  int64_t heads, const vj_seg_t* segs, int64_t n_segs, float ln_eps, int save, int gemm_flags, void* ws,
                  int64_t ws_bytes, vj_stream_t stream);
/* The same trunk for blocks that never run backward (the EMA target encoder of train.py:419-429, frozen-encoder inference) with both
 * LayerNorms of every block FOLDED into the Linear that consumes them (modules.py:115,119 -> 63 / 31): a statistics pass reads the
 * residual stream (vj_ln_rowstats), the qkv / fc1 GEMM reads it too and applies rstd_m * (acc - mean_m * c_n) + b'_n in its epilogue
 * (vj_gemm_bf16_nt_lnfold) -- the LayerNorm output is never written or re-read.  folds[i]: block i's folded weights, prepared by
 * vj_ln_fold_weights from the fp32 weights once per optimizer step.  Workspace: vj_blocks_fwd_ws_bytes(..., save = 0). */
typedef struct vj_lnfold {
  const void* w_qkv;    /* bf16 [3D, D] = bf16(W_qkv * diag(norm1.weight)) */
  const float* c_qkv;   /* [3D]  row sums of w_qkv */
  const float* b_qkv;   /* [3D]  qkv.bias + W_qkv norm1.bias */
  const void* w_fc1;    /* bf16 [Dh, D] = bf16(W_fc1 * diag(norm2.weight)) */
  const float* c_fc1;   /* [Dh] */
  const float* b_fc1;   /* [Dh]  fc1.bias + W_fc1 norm2.bias */
} vj_lnfold_t;
int vj_blocks_fwd_lnfold(const vj_block_t* blocks, const vj_lnfold_t* folds, int64_t n_blocks, const void* x_in, void* x_out,
                         int64_t M, int64_t D, int64_t heads, const vj_seg_t* segs, int64_t n_segs, float ln_eps, int gemm_flags,
                         void* ws, int64_t ws_bytes, vj_stream_t stream);
/* the pieces (also usable alone): rowstats[m] = {rstd_m, -mean_m * rstd_m} of bf16 rows (fp32, two passes in registers, biased
 * variance + eps: nn.LayerNorm's statistics); folded weights Wf = bf16(W diag(gamma)), cvec[n] = sum_k Wf[n,k], bf = b + W beta;
 * C = LayerNorm(X) W^T + b from the raw rows X (epilogue 0 bf16, 1 GELU, 4 bf16 with the first N/3 columns times alpha) */
int vj_ln_rowstats(const void* x_bf16, float* rowstats, int64_t rows, int64_t D, float eps, vj_stream_t stream);
int vj_ln_fold_weights(const float* W, const float* b, const float* gamma, const float* beta, void* Wf_bf16, float* cvec, float* bf,
                       int64_t N, int64_t K, vj_stream_t stream);
int vj_gemm_bf16_nt_lnfold(const void* X, int64_t ldx, const void* Wf, int64_t ldw, void* C, int64_t ldc, int64_t M, int64_t N,
                           int64_t K, const float* bias_f, const float* rowstats, const float* colsum_w, int epilogue, float alpha,
                           int flags, vj_stream_t stream);
int64_t vj_blocks_bwd_ws_bytes(int64_t M, int64_t D, int64_t Dh, int64_t heads);
/* dx_out [M,D] bf16 = d loss / d x_in given dout = d loss / d x_out; parameter gradients are written as
 * alpha * grad + beta_acc * old into the fp32 views of `blocks` (beta_acc = 1 accumulates micro-batches).
 * The dgrad chain runs on `stream`; weight / bias gradients (and the operand transposes feeding them) run on `side`
 * (NULL = same stream), ordered by events; on return `side` may still hold pending work -- make the consumer of the
 * gradients wait on it.  on_layer_done(user, l) (nullable) is called from the enqueueing thread as soon as block l's
 * backward has been enqueued on both streams (gradient-bucket launch hook, DDP equivalent of train.py:295-297).
 * flags bit0: transpose-free weight gradients (vj_gemm_bf16_tn_splitk; with option "wgrad_group" the four of a block
 * in one vj_gemm_bf16_tn_grouped launch once the block's last dY exists).
 * flags bit1: the caller already wrote the LAST block's fc2 bias gradient (the column sums of dout; it comes out of the
 * LayerNorm backward that produced dout, vj_layernorm_bwd_colsum) -- with the transpose-free route only.
 * flags bits 2-3: bit 3 set = "bit 2 tells whether the forward stored q pre-scaled by scale*log2(e)" (what vj_blocks_fwd does under
 * option "attn_softmax" = 2 when D % 4 == 0); callers record the convention at forward time and pass it here, so that an option change
 * between a forward and its backward cannot make the backward read the saved qkv with the wrong convention.  Bit 3 clear: the option is
 * read again at backward time (only right if it did not change).
 * With the transpose-free route and option "bias_fuse" (default) the bias gradients of qkv and fc1 come from column partials of
 * the kernels that produce dqkv / du (vj_attn_bwd_colsum, vj_gemm_bf16_nt_dgelu_colsum) and every partial reduction of a block
 * (both LayerNorms', those two) is ONE vj_reduce_segments launch at the end of the block. */
int vj_blocks_bwd(const vj_block_t* blocks, int64_t n_blocks, const void* x_in, const void* dout, void* dx_out, int64_t M,
                  int64_t D, int64_t heads, const vj_seg_t* segs, int64_t n_segs, float alpha, float beta_acc,
                  const void* save_ws, int64_t save_ws_bytes, void* tmp_ws, int64_t tmp_ws_bytes, int flags,
                  vj_stream_t stream, vj_stream_t side, vj_layer_cb_t on_layer_done, void* user);

/* ---- per-launch timing of the chains' GEMM / attention launches (HIP events on the launch stream) -------------
 * vj_prof_enable(1) starts recording; vj_prof_collect synchronises and sums per family (0 GEMM, 1 attention forward,
 * 2 attention backward): ms[3], flop[3], launches[3]; csv_path (nullable) receives one line per launch. */
int vj_prof_enable(int on);
int vj_prof_collect(double* ms, double* flop, int64_t* launches, const char* csv_path);

/* ---- gradient collective over RCCL / xGMI -------------------------------------------------------------------
 * DistributedDataParallel's gradient averaging (app/vjepa/train.py:295-297; src/utils/distributed.py:18-47) for hosts
 * without torch.distributed: one process per GPU, SUM all-reduce of slices ("buckets") of the flat fp32 gradient arena
 * on a communication stream the caller orders against the backward with events (the mean is folded into the fused
 * AdamW kernel's gscale = 1/world, see vj_adamw_ema_guarded).  librccl.so is dlopen()ed on first use (the copy already
 * mapped by the process, e.g. PyTorch's, is reused).  Rank 0 calls vj_comm_unique_id and ships the
 * vj_comm_unique_id_bytes() = 128 bytes to the other ranks out of band; every rank then calls vj_comm_init on its own
 * current HIP device.  The Python engine (jepa_amd/engine/dp.py) keeps using torch.distributed's RCCL backend by
 * default -- same library, same collectives -- so that it shares the launcher's process group. */
typedef struct vj_comm_opaque* vj_comm_t;
int64_t vj_comm_unique_id_bytes(void);
int vj_comm_unique_id(void* id_out);
int vj_comm_init(vj_comm_t* comm_out, int rank, int world, const void* id);
int vj_comm_allreduce_bucket(vj_comm_t comm, float* grad, int64_t count, vj_stream_t stream);   /* in place, SUM */
int vj_comm_broadcast(vj_comm_t comm, float* buf, int64_t count, int root, vj_stream_t stream);  /* parameter sync */
int vj_comm_destroy(vj_comm_t comm);

/* ---- run-time tuning switches ------------------------------------------------------------------------------
 * Named integer options that choose between kernels computing the same result (A/B measurements interleaved in one
 * process, tools/abab.py): "gemm_fwd_flags", "gemm_dgrad_flags", "gemm_4w", "gemm_persist", "wgrad_tn",
 * "wgrad_group", "gemm_dbg", "attn_softmax", "bias_fuse", "gemm_raster", "ws_guard", "gemm_epi_pre" (twelve; meaning, default
 * and accepted values of each: jepa_amd/csrc/options.hpp / options.cpp; forms that were measured and removed live as patches
 * under lab/patches/).  Initial value: environment variable VJ_<NAME IN UPPER CASE>, else
 * the built-in default.  Unknown names are an argument error.  The reference has no counterpart (it has no kernels). */
int vj_set_option(const char* name, int value);
int vj_get_option(const char* name, int* value);

/* Diagnostics (tests only).  With option "ws_guard" = 1 every member of the two chain workspaces (saved activations, backward
 * temporaries, column partials, split-K partials) is followed by a 256-byte gap that vj_blocks_fwd / vj_blocks_bwd fill with a byte
 * pattern before their first kernel; vj_ws_guard_check synchronises the device and reports how many distinct gaps were poisoned
 * since the last call and how many of them no longer hold the pattern (a write past the end of a workspace member).  No reference
 * counterpart. */
int vj_ws_guard_check(int64_t* n_checked, int64_t* n_bad);

/* ---- hardware probes (tests / profiles only) ---------------------------------------------------------------- */
int vj_probe_tr16(uint32_t* out256, int addr_scale, vj_stream_t stream);
int vj_probe_copy(const void* src, void* dst, int64_t bytes, vj_stream_t stream);
/* LDS read throughput of a CU, 8 waves x iters x 8 back-to-back reads: mode 0 ds_read_b128, 1 ds_read_b64_tr_b16 (the TN
 * GEMM's fragment addressing), 2 ds_read_b64; out[wg*8 + wave] = clock64 cycles (100 MHz timer ticks on gfx9) */
int vj_probe_lds_bw(long long* out, int mode, int iters, int n_wgs, vj_stream_t stream);
/* one wave idling for `ticks` periods of the 100 MHz timer: two of them on two streams take one spin time iff the streams are
 * mapped to different hardware queues (used once per process to pick independent side / update / communication streams) */
int vj_probe_spin(int64_t ticks, vj_stream_t stream);
/* the same kernel leaving {start, end} of its spin on the chip-wide 100 MHz timer in stamps[0..1] (device memory): two of them on two
 * streams overlap ON THE DEVICE'S CLOCK iff the streams sit on different hardware queues; no host timing involved (what
 * engine/layers.py streams_concurrent uses since round 6).  No reference counterpart. */
int vj_probe_spin_stamped(int64_t ticks, int64_t* stamps, vj_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VJEPA_HIP_H */
