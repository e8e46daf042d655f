/* libvlb C-ABI: the MI355X (gfx950) compute boundary of the phantom_vlb fine-tuning hot path.
 *
 * The reference (courtois-neuromod/phantom_vlb) is pure Python and has NO FFI of its own
 * (SURVEY.md 2.1): every entry point below replaces arithmetic that the reference reaches
 * through torch / transformers / peft / timm / flash_attn calls inside
 *   src/litmodule/videollama2_vlb_litmodule.py:229-306   (forward + training_step)
 *   src/utils.py:40-73                                   (HRFConvolveLayer, RidgeRegressionLayer)
 * and each declaration cites the call it stands in for.  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / C++ types cross the boundary.
 *   - All pointers are DEVICE pointers unless the name ends in _host.  The caller owns every
 *     buffer (inputs, outputs, workspaces); the library never allocates or frees.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued there, nothing syncs.
 *   - Return 0 on success, negative VLB_ERR_* otherwise; vlb_last_error() gives a
 *     thread-local message.  Nothing throws or aborts across the boundary.
 *   - "bf16" buffers are raw 16-bit bfloat16; row-major; `ld*` are row strides in ELEMENTS.
 *   - Functions are re-entrant and may be called from several host threads: the only process state is the
 *     one-time LDS-size attribute of a few kernels, set through thread-safe static initialisation on first use
 *     (it applies to the current device: one process per GPU).  The exported symbols are exactly the functions
 *     declared here - kernel-variant switches and timing-only ablations exist only in the separate tools build
 *     (libvlb_tools.so, `make -C phantom_vlb_amd/csrc tools`), which the package, bench.py and tests never load.
 */
#ifndef VLB_H
#define VLB_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libvlb.so is built with -fvisibility=hidden: the declarations between this push and the pop at the end of the file
 * are the ONLY dynamic symbols it exports (no mangled helpers, no kernel launch stubs). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define VLB_OK 0
#define VLB_ERR_INVALID (-1) /* bad shape / alignment / argument */
#define VLB_ERR_LAUNCH (-2)  /* HIP launch error */

/* 2 (round 4): symbols are hidden unless declared here; vlb_gemm_masked_pair_swiglu_bwd requires 16-byte-aligned
 * [gate | up] rows (round 3, also checked by VLB_REQUIRE at the call); entry points added since 1 are listed in
 * INTEGRATION.md.  A binder checks `vlb_abi_version() == VLB_ABI_VERSION`. */
#define VLB_ABI_VERSION 2

/* epilogue activations for vlb_gemm_bf16 / vlb_layernorm_fwd */
#define VLB_ACT_NONE 0
#define VLB_ACT_QUICK_GELU 1 /* x*sigmoid(1.702x): CLIP MLP (transformers modeling_clip.py CLIPMLP) */
#define VLB_ACT_GELU 2       /* erf GELU: STC connector readout */
#define VLB_ACT_SILU 3       /* SE fc1, sampler */
#define VLB_ACT_SWIGLU_PAIR 4 /* MistralMLP act_fn(gate)*up fused into the gate/up GEMM: W rows interleaved in
                                 16-row blocks [gate_b | up_b | gate_b+1 | ...]; C gets N/2 columns; no bias/residual */
#define VLB_ACT_SWIGLU_BWD 5  /* internal: epilogue of vlb_gemm_masked_pair_swiglu_bwd */

int vlb_abi_version(void);
const char* vlb_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Dense contraction (MFMA).  Replaces every nn.Linear / 1x1-conv / Conv3d-as-GEMM on the path:
 *   transformers modeling_mistral.py:41-47,134-137 (q/k/v/o, gate/up/down),
 *   modeling_clip.py CLIPAttention/CLIPMLP linears, patch-embed conv (as im2col GEMM),
 *   timm RegStage 1x1 convs, STCConnector sampler/readout,
 *   and peft LoRA  y += s*B(A(x))  through the second operand pair (A2 = s*x*A^T, W2 = B).
 *
 *   C[M,N] = act( A[M,K] . W[N,K]^T  +  A2[M,K2] . W2[N,K2]^T  +  bias[N] ) + residual[M,N]
 *
 * A, W, A2, W2, C, residual bf16; bias bf16 or NULL; fp32 accumulate.  K, K2 multiples of 8.
 * The library picks a 256x256 / 256x128 LDS-DMA MFMA kernel when N and K allow it, else a
 * bounds-checked 64x64 MFMA kernel.  residual may alias C.
 */
int vlb_gemm_bf16(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K,
                  const void* bias, const void* residual, int ldr, int act, const void* A2, int lda2,
                  const void* W2, int ldw2, int K2, void* stream);

/* Same contraction with a caller-owned scratch buffer (>= vlb_gemm_workspace_bytes(), 16-byte aligned, may be shared
 * by every GEMM issued on one stream).  With it the tiles of a partial last wave (tile count not a multiple of the
 * 256 CUs) are each cut into 2-8 contiguous K ranges that run on otherwise idle CUs; fp32 partial tiles go through
 * the workspace and are summed in a fixed order (deterministic for a given shape; the fp32 summation order of those
 * tiles differs from the unsplit kernel's).  ws == NULL behaves exactly like vlb_gemm_bf16. */
int vlb_gemm_bf16_ws(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K,
                     const void* bias, const void* residual, int ldr, int act, const void* A2, int lda2,
                     const void* W2, int ldw2, int K2, void* ws, int64_t ws_bytes, void* stream);
/* The LoRA / full fine-tune form of the fused MistralMLP gate/up projection (modeling_mistral.py:169-170
 * `down_proj(act_fn(gate_proj(x)) * up_proj(x))`): W_il / W2_il rows interleaved as for VLB_ACT_SWIGLU_PAIR; writes
 * H[M, N/2] = silu(gate) * up AND the pre-activations GU[M, N] = [gate | up] (plain column order) that the backward
 * pass reads - one GEMM epilogue instead of a GEMM plus an elementwise pass over [M, N]. */
int vlb_gemm_swiglu_save(const void* A, int lda, const void* W_il, int ldw, void* H, int ldh, void* GU, int ldgu, int M, int N,
                         int K, const void* A2, int lda2, const void* W2_il, int ldw2, int K2, void* ws, int64_t ws_bytes,
                         void* stream);
int64_t vlb_gemm_workspace_bytes(void);
/* How a long-K GEMM (K + K2 >= 4096, N % 256 == 0) is cut on the four-wave kernel: tile rows (192|256) * 1000 +
 * tail mode * 100 + K splits; tail mode 0 = whole tiles, 1 = partial wave re-cut into 256x128 halves, 2 = split-K
 * through the workspace.  0 for other shapes.  with_workspace: 0 none, 1 vlb_gemm_bf16_ws, 3 vlb_gemm_bf16_masked_pair_ws
 * (K2 = 64).  Pure host arithmetic (tests). */
int vlb_gemm_plan(int M, int N, int K, int K2, int with_workspace);

/* Which kernel vlb_gemm_bf16 picks for a shape: 0 = generic 64x64, 1 = 256x256, 2 = 256x128. */
/* LoRA backward through dropout in ONE GEMM (peft: dx = dy.W + dropout_mask * (u.A) / (1-p)):
 *   C[M,N] = A[M,K].W[N,K]^T + keep(m,n)/(1-p) * (A2[M,64].W2[N,64]^T)
 * keep(m,n) is the counter-based mask of vlb_lora_down / vlb_lora_dx_masked for an [M,N] activation under `seed`
 * (same bits, so forward and backward agree).  The second pair runs first and the accumulators are masked in
 * place before the main K loop.  N % 256 == 0, K % 64 == 0, K >= 128, M*N/2 < 2^32; no bias/residual/activation. */
int vlb_gemm_bf16_masked_pair(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K,
                              const void* A2, int lda2, const void* W2, int ldw2, float drop_p, uint32_t seed,
                              void* stream);

/* ... with the split-K workspace of vlb_gemm_bf16_ws (the masked pair rides in the first K range). */
int vlb_gemm_bf16_masked_pair_ws(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K,
                                 const void* A2, int lda2, const void* W2, int ldw2, float drop_p, uint32_t seed,
                                 void* ws, int64_t ws_bytes, void* stream);

/* LoRA backward of the MLP down projection fused with the SwiGLU backward (replaces vlb_gemm_bf16_masked_pair +
 * vlb_swiglu_bwd; MistralMLP, transformers modeling_mistral.py:41-47):
 *   d_h[M,ff]   = dY[M,K].Wt[ff,K]^T + keep/(1-p) * (U[M,64].At[ff,64]^T)        (never written)
 *   dgu[:, :ff] = d_h * up * silu'(gate),   dgu[:, ff:] = d_h * silu(gate)       gate = gu[:, :ff], up = gu[:, ff:]
 * in the GEMM epilogue, from the fp32 accumulators.  Shape rules of vlb_gemm_bf16_masked_pair with N = ff; gu and dgu rows
 * 16-byte aligned (base pointers % 16 == 0, ldgu % 8 == 0, lddgu % 8 == 0, ff % 8 == 0): the epilogue moves 16 bytes per lane. */
int vlb_gemm_masked_pair_swiglu_bwd(const void* dY, int lddy, const void* Wt, int ldw, const void* gu, int ldgu, void* dgu,
                                    int lddgu, int M, int ff, int K, const void* U, int ldu, const void* At, int ldat,
                                    float drop_p, uint32_t seed, void* ws, int64_t ws_bytes, void* stream);

int vlb_gemm_kernel_choice(int M, int N, int K, int K2);

/* out[C,R] = in[R,C]^T (bf16).  Used once per frozen weight to lay down W^T for dgrad. */
int vlb_transpose_bf16(const void* in, void* out, int R, int C, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Attention (flash-style, LDS-staged K/V tiles, fp32 online softmax).
 * Replaces flash_attention_2 / eager attention of modeling_mistral.py:96-178 (causal, GQA,
 * key padding mask) and modeling_clip.py CLIPAttention (non-causal, no mask).
 *   q: [B,S,Hq,D] with row stride ldq (elements between consecutive tokens), k/v likewise with
 *   Hkv heads; out [B,S,Hq,D] stride ldo.  key_mask: [B,S] bytes (1 = attend) or NULL.
 *   lse: [B,Hq,S] fp32 log-sum-exp (natural log, of scaled scores) or NULL; needed by bwd.
 *   D in {64,128}.
 *   Packed (unpadded) rows, the flash-attn varlen path of modeling_mistral.py (_upad_input): when
 *   cu_rows != NULL (int32 [B+1] device prefix sums), clip b owns rows [cu_rows[b], cu_rows[b+1]) of
 *   q/k/v/out/key_mask and S is the LONGEST clip (grid size; lse/delta stay [B,Hq,S]).  NULL = dense.
 */
int vlb_attention_fwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out, int ldo,
                      float* lse, const uint8_t* key_mask, int B, int S, int Hq, int Hkv, int D, int causal,
                      float scale, const int* cu_rows, void* stream);

/* Backward of the above. dq/dk/dv have the layout/strides of q/k/v (dk/dv rows 16-byte aligned).
 * delta: [B,Hq,S] fp32 workspace.  dq_acc: workspace of rows*Hq*D fp32 elements (rows = total_rows or B*S); with
 * grouped-query heads it holds the per-q-head bf16 partials of dK/dV (may be NULL when Hq == Hkv).
 * Launches: delta = rowsum(dout*out); dK/dV with one workgroup per (128 keys, q-head) - the causal triangle is
 * spread over Hq/Hkv times more workgroups than a per-kv-head sweep; a fixed-order sum of each GQA group's partials
 * (bf16 partials, as autograd's repeat_kv backward produces); dQ (query-block outer loop, accumulated in registers).
 * No atomics: results are reproducible.  total_rows = cu_rows[B] (host copy; ignored when cu_rows == NULL). */
int vlb_attention_bwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* out,
                      int ldo, const void* dout, int lddo, const float* lse, const uint8_t* key_mask, void* dq,
                      int lddq, void* dk, int lddk, void* dv, int lddv, float* delta, float* dq_acc, int B, int S,
                      int Hq, int Hkv, int D, int causal, float scale, const int* cu_rows, int total_rows,
                      void* stream);

/* ---------------------------------------------------------------------------------------------
 * Row normalisations (HBM-bound, 16-byte vector loads, fp32 statistics).
 */
/* MistralRMSNorm, modeling_mistral.py:182-196: y = w * bf16(x * rsqrt(mean(x^2)+eps)). */
int vlb_rmsnorm_fwd(const void* x, const void* w, void* y, int rows, int dim, float eps, void* stream);
/* dx = rmsnorm backward (+ optional accumulate into dx_accum_in: dx = dx_in + ...); dw not needed
 * (norm weights are frozen in every LoRA configuration). */
int vlb_rmsnorm_bwd(const void* x, const void* w, const void* dy, const void* dx_in, void* dx, int rows, int dim,
                    float eps, void* stream);
/* nn.LayerNorm / timm LayerNorm2d on channels-last rows, fused tail:
 *   y = act( LN(x)*w + b + residual ).  residual may be NULL.  Replaces CLIP layer_norm1/2,
 *   pre_layrnorm, and ConvNormAct(norm=LayerNorm2d, act=SiLU) of timm RegStage. */
int vlb_layernorm_fwd(const void* x, const void* w, const void* b, const void* residual, void* y, int rows, int dim,
                      float eps, int act, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Decoder element-wise ops.
 */
/* apply_rotary_pos_emb, modeling_mistral.py:51-81, in place on `heads` heads of width D starting at x
 * (token stride ld). cos/sin: [S, D/2] fp32 tables (positions 0..S-1).  sign=+1 forward, -1 backward.
 * Row r sits at position pos[r] (int32 [rows], packed layout) or r % S when pos == NULL. */
int vlb_rope_inplace(void* x, int ld, const float* cos_t, const float* sin_t, int rows, int S, int heads, int D,
                     int sign, const int* pos, void* stream);
/* MistralMLP gate: out[r, j] = silu(gu[r, j]) * gu[r, ff + j]   (gu = [gate | up], row stride 2*ff). */
int vlb_swiglu_fwd(const void* gu, void* out, int rows, int ff, void* stream);
/* dgu from dout, recomputing silu from gu. */
int vlb_swiglu_bwd(const void* gu, const void* dout, void* dgu, int rows, int ff, void* stream);
/* y = a + b (bf16), n elements (multiple of 8). */
int vlb_add_bf16(const void* a, const void* b, void* y, int64_t n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Vision ingest + CLIP embeddings (modeling_clip.py:138-219; litmodule :267 per-sample .to(bf16)).
 */
/* vision fp32 [N,3,H,W] -> patch rows bf16 [N*(H/P)*(W/P), Kpad] with K = 3*P*P zero-padded to Kpad;
 * fuses the fp32->bf16 cast of training_step:267 into the im2col of the stride-P patch conv. */
int vlb_patchify(const float* vision, void* patches, int N, int H, int W, int P, int Kpad, void* stream);
/* tokens[n, 0] = cls + pos[0]; tokens[n, 1+i] = patch_emb[n*G+i] + pos[1+i]   (all bf16, width D) */
int vlb_vit_assemble(const void* patch_emb, const void* cls, const void* pos, void* tokens, int N, int G, int D,
                     void* stream);
/* copy rows dropping the CLS token: out[n, i] = tokens[n, 1+i] */
int vlb_drop_cls(const void* tokens, void* out, int N, int G, int D, void* stream);

/* ---------------------------------------------------------------------------------------------
 * STC connector pieces (VideoLLaMA2 projector + timm RegStage; channels-last rows).
 */
/* depthwise 3x3, stride 1, pad 1, no bias.  x,y: [N,H,W,C] bf16; w: [C,9] bf16 (= conv2.conv.weight). */
int vlb_dwconv3x3(const void* x, const void* w, void* y, int N, int H, int W, int C, void* stream);
/* squeeze: mean over H*W -> [N,C] bf16. */
int vlb_se_pool(const void* x, void* pooled, int N, int HW, int C, void* stream);
/* excite: y = x * sigmoid(gate[n,c]) */
int vlb_se_scale(const void* x, const void* gate, void* y, int N, int HW, int C, void* stream);
/* im2col for Conv3d(k=2,s=2,p=1): x [B,T,H,W,C] -> cols [B*T2*H2*W2, 8*C], tap order (kt,kh,kw,c). */
int vlb_im2col3d_k2s2p1(const void* x, void* cols, int B, int T, int H, int W, int C, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Token splice (VideoLLaMA2 prepare_inputs_labels_for_multimodal) and the HRF weight mask
 * (VLBLitModule.make_weight_mask, litmodule :178-203) - one launch each instead of Python loops.
 */
/* ids: [B,L] int64 with one video_id per row. embeds out [B, L-1+Nv, D]; key_mask out [B, L-1+Nv] bytes
 * (ids != 0, left-extended with ones).  Returns VLB_ERR_INVALID via err_flag[0] != 0 if a row lacks
 * exactly one video token (flag is device memory, checked by the caller lazily).
 * Packed output: with cu_rows != NULL (int32 [B+1]) only the first cu_rows[b+1]-cu_rows[b] tokens of
 * clip b are emitted, at rows cu_rows[b].. of embeds / key_mask; row_pos (int32 [rows], optional)
 * receives each emitted row's position in its clip (the rotary position). */
int vlb_splice_embed(const int64_t* ids, const void* embed_w, const void* video_tokens, void* embeds,
                     uint8_t* key_mask, int* err_flag, int B, int L, int Nv, int D, int64_t video_id, int vocab,
                     const int* cu_rows, int* row_pos, void* stream);
/* wmask[b, :] = [left zeros][vis_w[b,f] x tokens_per_frame][2+inst zeros][lang_w[b,:dialog]][4+pad zeros]
 * padvals int64 [B,3] = (pad_len, inst_len, dialog_len); vis_w f64 [B,F]; lang_w f64 [B,Lw]; out f32 [B,S].
 * round_bf16 != 0 rounds every weight to bf16 first, as the reference does (.to(self.config.dtype), :190-194). */
int vlb_weight_mask(const int64_t* padvals, const double* vis_w, const double* lang_w, float* wmask, int B, int F,
                    int Lw, int tokens_per_frame, int S, int round_bf16, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Brain head (litmodule :245-254,302; utils.py:56,66-71), fused:
 *   hidden bf16 [B,S,E] --LN1--> --sum_s w[b,s]*.--> pooled --LN2--> z (x keep_mask/(1-p)) --Linear--> pred
 *   loss = mean((pred-y)^2) + lambda*||W||_F^2
 * The normalised [B,S,E] tensor is never written; tokens whose weight is zero are skipped.
 * Caller-allocated buffers (fp32 unless noted):
 *   ws          scratch of vlb_head_ws_floats(B,S,E,V) floats (shared by fwd and bwd)
 *   saved for backward: stats [B,S,2] (LN1 mean, rstd; rows with w==0 untouched), pooled_raw [B,E],
 *   sumw [B], zhat [B,E] (LN2 output before affine), ln2_rstd [B], z [B,E] bf16 (head input of the
 *   ridge layer, rounded to bf16 as autocast does), pred [B,V].
 *   keep_scale: [B,E] dropout keep-mask already divided by (1-p), or NULL (eval / p=0).
 *   y: [B,V] fp32 targets.  loss_terms[3] = {mse, l2, mse+l2}.
 */
int vlb_head_partial_rows(int S);
int64_t vlb_head_ws_floats(int B, int S, int E, int V);
int vlb_head_fwd(const void* hidden, const float* wmask, const void* ln1_w, const void* ln1_b, const void* ln2_w,
                 const void* ln2_b, const void* ridge_w, const void* ridge_b, const float* y, const float* keep_scale,
                 float* ws, float* stats, float* pooled_raw, float* sumw, float* zhat, float* ln2_rstd, void* z,
                 float* pred, float* loss_terms, int B, int S, int E, int V, float eps, float l2_lambda, const int* cu_rows,
                 void* stream);
/* Both head entry points accept packed hidden rows: cu_rows != NULL (int32 [B+1]) means clip b's
 * hidden/dhidden rows are [cu_rows[b], cu_rows[b+1]); wmask and stats stay dense [B,S].  total_rows =
 * cu_rows[B] on the host. */
/* Gradients of loss wrt head parameters (fp32 outputs, overwritten) and, when dhidden != NULL, wrt
 * hidden (bf16 [B,S,E]).  loss_scale multiplies the mse term's gradient (1/world under data
 * parallelism); l2_scale the ridge penalty's.  dz_ws, dpooled_ws: [B,E] fp32 scratch. */
int vlb_head_bwd(const void* hidden, const float* wmask, const void* ln1_w, const void* ln2_w, const void* ridge_w,
                 const float* y, const float* keep_scale, const float* stats, const float* pooled_raw,
                 const float* sumw, const float* zhat, const float* ln2_rstd, const void* z, const float* pred,
                 float* d_ridge_w, float* d_ridge_b, float* d_ln2_w, float* d_ln2_b, float* d_ln1_w, float* d_ln1_b,
                 float* ws, float* dz_ws, float* dpooled_ws, void* dhidden, int B, int S, int E, int V, float eps,
                 float l2_lambda, float loss_scale, float l2_scale, const int* cu_rows, int total_rows, void* stream);

/* ---------------------------------------------------------------------------------------------
 * LoRA adapters (peft LoraConfig(r, lora_alpha, lora_dropout) + get_peft_model, litmodule :113-120):
 *   y = x W^T + s * B(A(dropout_p(x))),  s = alpha/r; only A [r,in] and B [out,r] train.
 * The base GEMM carries the adapter through vlb_gemm_bf16's second operand pair (A2 = t, W2 = B).
 * Dropout masks are counter-based hashes of (seed, row, column) - regenerated in backward, never
 * stored; every 16-rank group (= one adapted projection) has its own seed (peft: one Dropout each).
 */
/* t[M,R] = scale/(1-p) * keep(x) . A^T ; A: [R,K] bf16 (R = 16 * projections sharing x, <= 48);
 * seeds_host: R/16 host uint32 (may be NULL when drop_p == 0). */
int vlb_lora_down(const void* x, int ldx, const void* A, void* t, int ldt, int M, int K, int R, float scale,
                  float drop_p, const uint32_t* seeds_host, void* stream);
/* dx[M,K] += sum_g keep_g/(1-p) * (u[:, 16g:16g+16] . A_g) ; At: [K, >=R] bf16 (transposed adapters, row stride ldat).
 * K % 64 == 0, dx 16-byte aligned with lddx % 8 == 0. */
int vlb_lora_dx_masked(const void* u, int ldu, const void* At, int ldat, void* dx, int lddx, int M, int K, int R, float drop_p,
                       const uint32_t* seeds_host, void* stream);
/* dB^T and u in one pass over dY (LoRA backward of one projection, rank 16):
 *   dW[16,K] (fp32) = alpha * G[:, :16]^T . X + beta * dW        (G = t, X = dY  ->  dB^T = t^T dY)
 *   u[m, 0:16] (bf16, row stride ldu) = u_scale * X[m,:] . Bt^T    (Bt = B^T [16,K] bf16  ->  u = s dY B)
 * ws as for vlb_wgrad_skinny; u_ws: fp32 [vlb_wgrad_u_ws_floats(M,K)] partials per 256-column block, summed
 * in a fixed order.  K % 64 == 0.  Replaces a vlb_lora_down + vlb_wgrad_skinny pair (two sweeps over dY). */
int64_t vlb_wgrad_u_ws_floats(int M, int K);
int vlb_wgrad_skinny_u(const void* G, int ldg, const void* X, int ldx, float* dW, float* ws, int M, int K, float alpha,
                       float beta, const void* Bt, float u_scale, void* u, int ldu, float* u_ws, void* stream);
/* The same pass for the 1..3 projections that share one dY = X [M, sum cols] (q|k|v, gate|up; reference
 * find_all_linear_names targets, litmodule :36-55): projection j owns cols[j] columns (multiples of 256), its G is
 * t[:, 16j:16j+16), its gradient dW[j] [16, cols[j]] fp32, its Bt[j] [16, cols[j]] bf16, and u[:, 16j:16j+16) receives
 * u_scale * dY_j . Bt[j]^T.  cols / dW / Bt are HOST arrays (of ints / device pointers).  One sweep and one reduce
 * launch instead of one pair per projection; ws: splits * 16 * sum(cols) floats, u_ws: vlb_wgrad_u_ws_floats(M, sum cols). */
int vlb_wgrad_skinny_u_multi(const void* G, int ldg, const void* X, int ldx, int M, int nproj, const int* cols, float* const* dW,
                             const void* const* Bt, float* ws, float alpha, float beta, float u_scale, void* u, int ldu,
                             float* u_ws, void* stream);
/* Rebuild derived adapter layouts after an optimiser step in ONE launch.  jobs: device array of n_jobs
 * records {const bf16* src; bf16* dst; int32 n; int32 n0; int32 ld; int32 il} (32 bytes; il = 1: destination row (i/16)*32 + i%16 instead of i, the gate/up interleave of VLB_ACT_SWIGLU_PAIR): rows
 * [n0, n0+256) of the transpose of the row-major [16, n] matrix `src` are written to dst[i*ld + 0..15]
 * (dst points at the first row/column of the 16-column band, 16-byte aligned, ld % 8 == 0).
 * Used for At[:, 16j:16j+16] = A_j^T and Bpad[row0:row0+N, 16j:16j+16] = B_j (peft: lora_A / lora_B). */
int vlb_transpose16_scatter(const void* jobs, int n_jobs, void* stream);
/* Skinny weight gradient (MFMA, both operands read transposed from LDS):
 *   dW[N,K] (fp32) = alpha/(1-p) * sum_m G[m,n] * keep_g(X[m,k]) + beta * dW ;  N in {16,32,48}, g = n/16.
 * dA of all projections sharing x in one launch (G = [u_q|u_k|u_v], X = x, one dropout seed per
 * 16-rank group) and dB^T = t^T dY (G = t, X = dY, p = 0).  ws: fp32 [vlb_wgrad_splits(M), N, K]
 * partial slabs summed in a fixed order (reproducible).  seeds_host: N/16 host uint32, NULL if p == 0. */
int vlb_wgrad_splits(int M);
int vlb_wgrad_skinny(const void* G, int ldg, const void* X, int ldx, float* dW, float* ws, int M, int N, int K,
                     float alpha, float beta, float drop_p, const uint32_t* seeds_host, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Optimiser (litmodule :345-379 AdamW + CosineAnnealingLR; Trainer gradient_clip_val).
 */
/* sumsq[0] += sum(g^2)   (fp32 grads; sumsq zeroed by caller once per step).  Two-stage fixed-order
 * reduction through ws (vlb_sumsq_ws_floats() floats): bit-reproducible, so data-parallel ranks that
 * hold identical gradients compute identical clip coefficients and stay in lock-step. */
int vlb_sumsq_ws_floats(void);
int vlb_grad_sumsq(const float* g, int64_t n, float* sumsq, float* ws, void* stream);
/* fused clip + AdamW on an fp32 master with fp32 moments; writes the bf16 compute copy when
 * param_bf16 != NULL.  clip coefficient = min(1, max_norm/(sqrt(sumsq[0])+1e-6)) read on device
 * (max_norm <= 0 disables clipping).  step counts from 1. */
int vlb_adamw_step(float* master, void* param_bf16, const float* grad, float* m, float* v, int64_t n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int step, const float* sumsq,
                   float max_norm, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Full-parameter fine-tuning (BASELINE configs[4]; litmodule :86-99 with freeze_backbone=False, use_lora=False:
 * everything but the vision tower trains).  What torch autograd does for the reference behind loss.backward():
 * weight gradients of every linear / 1x1 conv are the SAME TN GEMM (vlb_gemm_bf16) on transposed activations,
 * dW[N,K] = dy^T[N,M] . (x^T[K,M])^T; the entries below are the HBM-bound pieces around it.  Gradients of backbone
 * parameters are bf16 (the reference's parameters and therefore their autograd gradients are bf16, litmodule :155).
 */
/* out[C, ld_out] = in[R, C]^T with columns R..Rpad-1 written as zeros (token axis padded to the GEMM K granule) */
int vlb_transpose_pad(const void* in, int ld_in, void* out, int ld_out, int R, int C, int Rpad, void* stream);
/* workspace (floats) of the two norm backward entries below and of vlb_colsum */
int64_t vlb_norm_bwd_ws_floats(int rows, int dim);
int64_t vlb_colsum_ws_floats(int rows, int dim);
/* RMSNorm weight gradient: dw[c] = sum_rows dy * x * rsqrt(mean(x^2)+eps)  (modeling_mistral.py:182-196 backward) */
int vlb_rmsnorm_bwd_dw(const void* x, const void* dy, void* dw_bf16, float* ws, int rows, int dim, float eps, void* stream);
/* RMSNorm's whole backward in one sweep: dx = vlb_rmsnorm_bwd's result (+ dx_in when given) AND dw as above */
int64_t vlb_rmsnorm_bwd_full_ws_floats(int rows, int dim);
int vlb_rmsnorm_bwd_full(const void* x, const void* w, const void* dy, const void* dx_in, void* dx, void* dw_bf16, float* ws,
                         int rows, int dim, float eps, void* stream);
/* backward of vlb_layernorm_fwd (y = act(LN(x; w, b) + residual)): dx, d residual (may be NULL), dw, db */
int vlb_layernorm_bwd(const void* x, const void* w, const void* b, const void* residual, const void* dy, void* dx, void* dres,
                      void* dw_bf16, void* db_bf16, float* ws, int rows, int dim, float eps, int act, void* stream);
/* y = act(x): the training forward keeps pre-activations, so activations the inference path fuses into GEMM
 * epilogues (sampler SiLU, readout GELU, SE SiLU) run as their own pass (n % 8 == 0) */
int vlb_act_fwd(const void* x, void* y, int64_t n, int act, void* stream);
/* dx = dy * act'(x) for VLB_ACT_SILU / VLB_ACT_GELU / VLB_ACT_QUICK_GELU (n % 8 == 0) */
int vlb_act_bwd(const void* x, const void* dy, void* dx, int64_t n, int act, void* stream);
/* bias gradients: out[c] = sum_r x[r, c] */
int vlb_colsum(const void* x, int ld, void* out_bf16, float* ws, int rows, int dim, void* stream);
/* embed_tokens gradient (splice backward): dE[tok[j]] = sum of d_embeds rows rows[beg[j] .. beg[j+1]) (host-built lists) */
int vlb_embed_grad(const void* d_embeds, int ld, const int* tok, const int* beg, const int* rows, int n_tokens, void* dE, int D,
                   void* stream);
/* clip norm / AdamW on bf16 gradients (n % 8 == 0); same contract as vlb_grad_sumsq / vlb_adamw_step */
int vlb_grad_sumsq_bf16(const void* g, int64_t n, float* sumsq, float* ws, void* stream);
int vlb_adamw_step_g16(float* master, void* param_bf16, const void* grad_bf16, float* m, float* v, int64_t n, float lr,
                       float beta1, float beta2, float eps, float weight_decay, int step, const float* sumsq, float max_norm,
                       void* stream);
/* STC connector backward pieces: depthwise 3x3 weight gradient [9,C]; squeeze-excite gate-logit and input gradients;
 * inverse of vlb_im2col3d_k2s2p1 (every input element sits in exactly one 2x2x2 window) */
int64_t vlb_dwconv3x3_bwd_w_ws_floats(int N, int H, int C);
int vlb_dwconv3x3_bwd_w(const void* x, const void* dy, void* dw9_bf16, float* ws, int N, int H, int W, int C, void* stream);
int vlb_se_bwd_gate(const void* x, const void* dy, const void* s, void* ds, int N, int HW, int C, void* stream);
int vlb_se_bwd_x(const void* dy, const void* s, const void* dpool, void* dx, int N, int HW, int C, void* stream);
int vlb_col2im3d_k2s2p1(const void* dcols, void* dx, int B, int T, int H, int W, int C, void* stream);

/* ---------------------------------------------------------------------------------------------
 * fp8 MFMA GEMMs (BASELINE configs[4]): OCP e4m3 elements with one E8M0 scale (2^(s-127)) per 32 consecutive K
 * elements - the MX block format v_mfma_scale_f32_16x16x128_f8f6f4 consumes at twice the bf16 MFMA rate.
 */
/* x bf16 [rows, K] -> q uint8 [rows, K] (e4m3, round-to-nearest-even, saturating) + scales uint8 [rows, K/32];
 * the shared exponent of a block is ceil(log2(amax / 448)).  K % 32 == 0; ld* in elements of the respective array. */
int vlb_quantize_mxfp8(const void* x_bf16, int ldx, void* q, int ldq, void* scales, int lds, int rows, int K, void* stream);
/* The same quantisation of x^T in one pass: q uint8 [C, Rpad], scales uint8 [C, Rpad/32] with the MX blocks running along
 * x's ROW axis (rows R..Rpad-1 quantise as zeros) - the operands of the dgrad (W^T) and wgrad (dy^T, x^T) products. */
int vlb_transpose_quantize_mxfp8(const void* x_bf16, int ldx, void* q, int ldq, void* scales, int lds, int R, int C, int Rpad,
                                 void* stream);
/* Both of the above from ONE read of x (what the backward of an fp8 linear needs of its dy: row-wise for dgrad, transposed for
 * wgrad); outputs bit-identical to the two separate calls.  C % 64 == 0. */
int vlb_quantize_dual_mxfp8(const void* x_bf16, int ldx, void* q, int ldq, void* scales, int lds, void* qt, int ldqt, void* scales_t,
                            int ldst, int R, int C, int Rpad, void* stream);
/* Producers that emit the quantisation of their output next to the bf16 tensor (q / scales bit-identical to
 * vlb_quantize_mxfp8 of the bf16 output): vlb_rmsnorm_fwd (dim % 32 == 0, dim <= 4096) and vlb_swiglu_fwd (ff % 32 == 0). */
int vlb_rmsnorm_fwd_mxfp8(const void* x, const void* w, void* y, void* q, int ldq, void* scales, int lds, int rows, int dim, float eps,
                          void* stream);
int vlb_swiglu_fwd_mxfp8(const void* gu, void* out, void* q, int ldq, void* scales, int lds, int rows, int ff, void* stream);
/* C[M,N] bf16 = dequant(Aq,sA)[M,K] . dequant(Wq,sW)[N,K]^T + residual (optional); fp32 accumulate.
 * N % 256 == 0, K % 128 == 0, any M. */
int vlb_gemm_mxfp8(const void* Aq, int lda, const void* sA, int ldsa, const void* Wq, int ldw, const void* sW, int ldsw, void* C,
                   int ldc, int M, int N, int K, const void* residual, int ldr, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Exchange steps of the sharded training step over RCCL / xGMI - what an fsdp.yaml-driven run of the reference
 * (fsdp.yaml:5-14 FULL_SHARD; never loaded by its mainline) would issue through torch FSDP's NCCL calls:
 * parameter all-gather, gradient reduce-scatter, the clip norm's scalar all-reduce.  "direct" = all-pairs
 * schedules: every shard crosses its own xGMI link (grouped ncclSend/ncclRecv), and the reduce-scatter sums the
 * staged slices in RANK ORDER in one local kernel, so reduced gradients are bit-reproducible.
 * RCCL (librccl.so.1) is bound with dlopen at first use; one communicator per process (= per GPU); calls enqueue on
 * `stream` and return; the handle is created from a 128-byte unique id that rank 0 makes and the caller
 * distributes (torch.distributed's store / a broadcast).  Buffers are caller-owned device memory.
 */
int vlb_comm_unique_id(void* id128_host);                                     /* host buffer of 128 bytes (rank 0) */
int vlb_comm_init(int rank, int world, const void* id128_host, void** comm_out);
int vlb_comm_destroy(void* comm);
/* Loopback transport (testing the exchange schedules without a second GPU): `world` communicators in ONE process whose
 * ncclSend / ncclRecv calls are replaced by device copies paired through an in-process mailbox with the same stream
 * ordering (receiver waits for the sender's stream, sender waits for the copy).  vlb_allgather_direct,
 * vlb_reducescatter_direct(_bf16) and vlb_allreduce_scalar then run unchanged with world 2..8 on one GPU - every rank
 * MUST be driven by its own host thread (a group blocks until its peers arrive; 60 s without them is an error, not a
 * hang).  `stage`: caller-owned device memory of vlb_comm_loopback_stage_bytes(world) bytes for the all-reduce, alive
 * until the last communicator is destroyed.  comms_out[world].  tests/test_gpu_comm.py drives it. */
int64_t vlb_comm_loopback_stage_bytes(int world);
int vlb_comm_init_loopback(int world, void* stage, int64_t stage_bytes, void** comms_out);
int vlb_comm_rank(void* comm);
int vlb_comm_world(void* comm);
/* full[r*shard_bytes ...] = rank r's shard, for every r.  In place when shard == full + rank*shard_bytes. */
int vlb_allgather_direct(void* comm, const void* shard, void* full, int64_t shard_bytes, void* stream);
/* out[n_per_rank] = sum over ranks r (in rank order) of rank r's send[rank*n_per_rank ...]; fp32; `stage` holds
 * vlb_reducescatter_stage_floats(n_per_rank, world) floats; n_per_rank % 4 == 0, 16-byte aligned buffers. */
int64_t vlb_reducescatter_stage_floats(int64_t n_per_rank, int world);
int vlb_reducescatter_direct(void* comm, const float* send, float* out, int64_t n_per_rank, float* stage, void* stream);
/* The same exchange for bf16 buffers (the bf16 gradients of the full fine-tune's backbone store, configs[4]): slices
 * travel as bf16, are widened to fp32, summed in rank order and rounded to bf16 ONCE.  `stage` holds
 * n_per_rank * world bf16 elements; n_per_rank % 8 == 0, 16-byte aligned buffers. */
int vlb_reducescatter_direct_bf16(void* comm, const void* send, void* out, int64_t n_per_rank, void* stage, void* stream);
/* The local half of both reduce-scatters on its own: out[i] = sum over r = 0..world-1, IN THAT ORDER, of
 * stage[r*n_per_rank + i] (is_bf16 = 0: fp32 in / out; 1: bf16 in, fp32 accumulate, bf16 out).  No communicator: the
 * fixed-order reduction can be verified for any world size on one GPU with hand-staged slices. */
int vlb_reduce_slices(const void* stage, void* out, int64_t n_per_rank, int world, int is_bf16, void* stream);
/* values[0..count) summed over ranks in place (the clip norm's sum of squares, logged losses) */
int vlb_allreduce_scalar(void* comm, float* values, int count, void* stream);

/* The reference's two exported head layers on their own (src/__init__.py:3-13; on the training path they run fused inside
 * vlb_head_fwd / vlb_head_bwd):
 *   HRFConvolveLayer.forward (src/utils.py:44-56)   out[B,E] = einsum('bse,bs->be', embeddings[B,S,E], weights[B,S])
 *     embeddings bf16 (emb_is_f32 = 0, out bf16) or fp32 (1, out fp32); weights fp32; fp32 accumulation in a fixed order;
 *     ws: vlb_hrf_pool_ws_floats(B, E) floats.
 *   RidgeRegressionLayer.forward (src/utils.py:59-73)   pred[B,V] = x[B,E] . W[V,E]^T + b  (fp32 out), *l2_out = lambda * ||W||_F^2
 *     x, W, b bf16; ws: vlb_ridge_ws_floats(V) floats. */
int64_t vlb_hrf_pool_ws_floats(int B, int E);
int vlb_hrf_pool(const void* embeddings, int emb_is_f32, const float* weights, void* out, float* ws, int B, int S, int E, void* stream);
int64_t vlb_ridge_ws_floats(int V);
int vlb_ridge_fwd(const void* x, const void* ridge_w, const void* ridge_b, float* pred, float* l2_out, float* ws, int B, int E, int V,
                  float l2_lambda, void* stream);

/* head dropout mask (litmodule :226,251 nn.Dropout(p) in training): out[i] = keep_i / (1-p), keep_i from a
 * counter-based hash of (seed, i) with 16 random bits per element (same mixer as the LoRA masks).  The result is
 * what vlb_head_fwd / vlb_head_bwd take as `keep_scale`. */
int vlb_dropout_keep_scale(float* out, int64_t n, float p, uint32_t seed, void* stream);

/* misc */
/* One empty single-wave launch named `vlb_profile_marker_kernel`: bench.py brackets its timed region with two of them so
 * that tools/profile_tables.py can cut model construction and warm-up steps out of a rocprofv3 kernel trace. */
int vlb_profile_marker(void* stream);
int vlb_cast_f32_to_bf16(const float* in, void* out, int64_t n, void* stream);
int vlb_cast_bf16_to_f32(const void* in, float* out, int64_t n, void* stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* VLB_H */
