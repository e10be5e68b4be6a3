/* dgtd.h — C ABI of libdgtd.so: the MI355X (gfx950) kernels behind the depth-guided
 * texture-diffusion segmentation hot path (reference: twig/model/cod.py).
 *
 * Conventions (they mirror the reference's own native-op boundary, twig/ops):
 *   - plain device pointers + explicit sizes; no framework types      (twig/ops/src/ms_deform_attn.h:20-27)
 *   - tensors are contiguous in the stated layout and live on the device that owns `stream`
 *                                                                      (twig/ops/src/cuda/ms_deform_attn_cuda.cu:28-38)
 *   - the callee never allocates: outputs, gradients and workspaces are caller-owned
 *                                                                      (ms_deform_attn_cuda.cu:54, :121-123)
 *   - kernels are enqueued on the caller's stream and never synchronise (ms_deform_attn_cuda.cu:65)
 *   - every entry returns 0 on success; on failure a non-zero code, with the text available
 *     from dgtd_last_error() (thread-local).  Launch errors are returned, not printed
 *     (contrast ms_deform_im2col_cuda.cuh:948-952).
 *   - re-entrant: no mutable global state; forward runs on the Python thread, backward on
 *     autograd's device thread.
 *
 * dtype: DGTD_F32 = exact-fp32 kernels (parity mode, fp32 MFMA); DGTD_BF16 = bf16 I/O with
 * fp32 accumulation (throughput mode, bf16 MFMA); DGTD_F16 = IEEE half I/O with fp32 accumulation
 * (fp16 MFMA) - the reference's own AMP recipe, AmpOptimWrapper = fp16 autocast + fp32 masters +
 * dynamic loss scaling (config/sod.yml:57, config/cod.yml:58).  Wherever DGTD_BF16 is accepted
 * DGTD_F16 is too.  Statistics (lse, mean, rstd) are always fp32.
 */
#ifndef DGTD_H
#define DGTD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { DGTD_F32 = 0, DGTD_BF16 = 1, DGTD_F64 = 2 /* ms_deform_attn only */, DGTD_F16 = 3 } dgtd_dtype;
typedef void* dgtd_stream; /* hipStream_t */

int dgtd_version(void);
const char* dgtd_last_error(void);

/* ---- opt-in per-call device timing (measurement only; bench.py's roofline leg) ---------------------------------------------
 * enable(1) clears earlier records and makes every entry point below bracket what it enqueues with a HIP event pair on the caller's
 * stream, tagged with a key (entry name + shape) and the call's algorithmic HBM bytes or MFMA flops; enable(0) stops
 * recording.  dump() waits for the events and writes one line per call: "key<TAB>hbm|mfma<TAB>amount<TAB>milliseconds"; it returns
 * the buffer size the full text needs.  The only mutable global state of the library, behind a mutex, inert unless enabled.      */
int dgtd_profile_enable(int on);
int64_t dgtd_profile_dump(char* buf, int64_t cap);
/* an EMPTY bracket under key "dgtd_profile_empty": the event pair's own floor, which bench.py subtracts from every per-call figure */
int dgtd_profile_empty(dgtd_stream stream);

/* ---- LayerNorm over the last dim of a [rows, C] token matrix ---------------------------------
 * replaces nn.LayerNorm / F.layer_norm at twig/model/cod.py:979,881,929,936,1367-1391,1043.
 * gamma/beta fp32 [C].  mean/rstd fp32 [rows] are written for the backward.                    */
int dgtd_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y,
                       float* mean, float* rstd, int64_t rows, int C, float eps,
                       dgtd_dtype dt, dgtd_stream s);
/* dgamma/dbeta fp32 [C] are OVERWRITTEN.  workspace: dgtd_layernorm_bwd_workspace(C) bytes.  */
int64_t dgtd_layernorm_bwd_workspace(int C);
int dgtd_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                       const float* rstd, void* dx, float* dgamma, float* dbeta, void* workspace,
                       int64_t rows, int C, dgtd_dtype dt, dgtd_stream s);

/* Same, with the gradient of a residual branch that forks off x added in the same pass: dx = LayerNorm'(dy) + dx_add (dx_add may be
 * NULL).  Pre-norm residual blocks (x feeds the norm AND the skip connection, cod.py:958-959) otherwise pay a separate add.    */
int dgtd_layernorm_bwd_add(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                           const void* dx_add, void* dx, float* dgamma, float* dbeta, void* workspace, int64_t rows,
                           int C, dgtd_dtype dt, dgtd_stream s);
/* First stage only: workspace receives [*nblocks][2C] = { dgamma | dbeta } partial rows (see dgtd_multi_reduce).                 */
int dgtd_layernorm_bwd_partial(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                               const void* dx_add, void* dx, void* workspace, int64_t rows, int C, dgtd_dtype dt, int* nblocks,
                               dgtd_stream s);

/* ---- Spatial-reduction multi-head attention core: softmax(Q K^T * scale) V, head_dim = 64 -----
 * replaces twig/model/cod.py:913-917 (the q/kv/proj Linears stay GEMM calls).
 * q   [B, N,   heads*64]      = output of Attention.q   (cod.py:902) — head h in cols [64h, 64h+64)
 * kv  [B, Nkv, 2*heads*64]    = output of Attention.kv  (cod.py:908) — K then V, same head split
 * out [B, N,   heads*64]      = (attn @ v).transpose(1,2).reshape(B,N,C) (cod.py:917)
 * lse [B, heads, N] fp32      = log-sum-exp of the scaled scores (saved for the backward)       */
int dgtd_sra_attn_fwd(const void* q, const void* kv, void* out, float* lse,
                      int B, int N, int Nkv, int heads, float scale, dgtd_dtype dt, dgtd_stream s);
/* dq [B,N,C] (dt);  dkv_f32 [B,Nkv,2C] fp32.  DGTD_F32: must be ZEROED by the caller (atomically accumulated).  DGTD_BF16 / DGTD_F16:
 * every element is overwritten (the query chunks of the key-stationary dK/dV kernel write plain partial slabs into the workspace and
 * one reduce launch sums them; no atomics), so the result is bit-reproducible and the buffer needs no initialisation.
 * workspace: dgtd_sra_attn_bwd_workspace(B,N,heads) bytes = delta = rowsum(dO*O) (computed inside the dQ kernel) + 64 MB of slabs.   */
int64_t dgtd_sra_attn_bwd_workspace(int B, int N, int heads);
int dgtd_sra_attn_bwd(const void* q, const void* kv, const void* out, const void* dout,
                      const float* lse, void* dq, float* dkv_f32, void* workspace,
                      int B, int N, int Nkv, int heads, float scale, dgtd_dtype dt, dgtd_stream s);

/* ---- Depthwise KxK convolution, stride 1, "same" zero padding, NHWC / token-major ----------------
 * replaces convnext_Block.dwconv 7x7 (twig/model/cod.py:1095,1106) and DWConv 3x3 + bias fused with Mlp's
 * exact GELU (cod.py:1523-1531, :854-855).  x, y [B,H,W,C]; w_t fp32 [K*K, C] (= Conv2d weight [C,1,K,K]
 * transposed so that tap rows are contiguous in C); bias fp32 [C] or NULL.  K in {3, 7}.
 * mode 0: y = conv(x)+bias   mode 1: y = gelu(conv(x)+bias)   mode 2: y = aux * gelu'(conv(x)+bias)
 * (mode 2 is the backward through the fused GELU: aux = upstream gradient, pre-activation recomputed).
 * mode 3: y = conv(x)+bias+aux (the input gradient of a residual block: aux = gradient arriving through the skip connection).
 * Backward w.r.t. x = mode 0 applied to the gradient with the spatially flipped filter.            */
int dgtd_dwconv_fwd(const void* x, const float* w_t, const float* bias, const void* aux, void* y,
                    int B, int H, int W, int C, int K, int mode, dgtd_dtype dt, dgtd_stream s);
/* Conv2d-layout parameters -> kernel layout, one launch: w [C,1,K,K] and bias [C] (dtype wdt; bias may be NULL)
 * -> packed fp32 [(2*K*K + 1) * C] = { w_t | w_t spatially flipped (for the backward w.r.t. x) | bias }.      */
int dgtd_dwconv_pack(const void* w, const void* bias, float* packed, int C, int K, dgtd_dtype wdt, dgtd_stream s);
/* The same packing for n layers in one launch (HOST arrays of n; bias: NULL or n pointers, entries may be NULL): weights change only
 * between optimizer steps, so a training step re-packs every depthwise layer once, together.                                      */
int dgtd_dwconv_pack_batched(const void* const* w, const void* const* bias, float* const* packed, const int* C, const int* K, int n,
                             dgtd_dtype wdt, dgtd_stream s);
/* grads fp32 [(K*K + 1) * C] = { dw_t | db } -> dw [C,1,K,K], db [C] in dtype wdt (db may be NULL).           */
int dgtd_dwconv_unpack_grads(const float* grads, void* dw, void* db, int C, int K, dgtd_dtype wdt, dgtd_stream s);
/* grads fp32 [(K*K + 1) * C] = { dw_t | db } is OVERWRITTEN (db = 0 when has_bias == 0); C % 128 == 0.
 * workspace: dgtd_dwconv_bwd_weight_workspace(...) bytes (per-workgroup partial sums, reduced in a fixed order). */
int64_t dgtd_dwconv_bwd_weight_workspace(int B, int H, int W, int C, int K);
int dgtd_dwconv_bwd_weight(const void* x, const void* du, float* grads, int has_bias, void* workspace,
                           int B, int H, int W, int C, int K, dgtd_dtype dt, dgtd_stream s);
/* First stage only: workspace receives [*nblocks][(K*K + 1) * C] = { dw_t | db } partial rows; finish with a dgtd_multi_reduce entry
 * with tr_rows = K*K, tr_cols = C (writes dw [C,1,K,K] and db [C] in the parameters' dtype: no separate unpack launch).          */
/* Deferred weight-gradient phase: the same first stage for n same-shaped layers in ONE launch.  x, du: HOST arrays of n device
 * pointers.  workspace receives [n][blocks][(K*K + 1) * C] partial rows, blocks = dgtd_dwconv_bwd_weight_batched_blocks(n, ...)
 * (fp32; size n * blocks * (K*K+1) * C * 4 bytes); finish each layer with a transposing dgtd_multi_reduce entry as above.        */
int dgtd_dwconv_bwd_weight_batched_blocks(int n, int B, int H, int W, int C, int K);
int dgtd_dwconv_bwd_weight_batched(const void* const* x, const void* const* du, int n, int has_bias, void* workspace, int B, int H, int W,
                                   int C, int K, dgtd_dtype dt, dgtd_stream s);
int dgtd_dwconv_bwd_weight_partial(const void* x, const void* du, int has_bias, void* workspace, int B, int H, int W, int C, int K,
                                   dgtd_dtype dt, int* nblocks, dgtd_stream s);

/* ---- Second stage of the two-stage column reductions, for MANY reductions in one launch --------------------------------------
 * The backward kernels below that also produce a per-column sum (LayerNorm dgamma/dbeta, the bias gradient of every nn.Linear, the
 * layer-scale gradient) write one partial row per workgroup and need a second pass over [nblocks][ncols].  The plain entry points
 * run that pass themselves; their `_partial` siblings stop after the first stage and report `nblocks`, so that a caller can run the
 * second stage of a whole backward pass (≈ 290 reductions per training step) in a handful of launches with dgtd_multi_reduce.
 * One entry: out columns [0, nA) of the row sum go to outA as fp32 (skipped when outA is NULL), columns [nA, ncols) to outB in dtB.
 * tr_rows > 0 (depthwise-conv weight gradients): the outB columns are a [tr_rows][tr_cols] matrix (tap-major partials { dw_t }) that
 * is written TRANSPOSED, outB[c * tr_rows + t] (= Conv2d weight layout [C,1,K,K]); the columns after it ({ db }) go to outC in dtB
 * (skipped when outC is NULL).
 * Fixed summation order (deterministic), identical to what the plain entry points produce.  entries: HOST array.                  */
typedef struct {
  const float* ws;   /* partial rows [nblocks][ncols], fp32 */
  int32_t nblocks, ncols;
  float* outA;       /* fp32 [nA] or NULL */
  int32_t nA;
  void* outB;        /* [ncols - nA] in dtB, or NULL when nA == ncols */
  int32_t dtB;       /* dgtd_dtype */
  int32_t tr_rows, tr_cols;   /* 0, 0 = no transpose */
  void* outC;
} dgtd_reduce_entry;
int dgtd_multi_reduce(const dgtd_reduce_entry* entries, int n, dgtd_stream s);

/* ---- Fused residual epilogue and column sums on [rows, C] token matrices -------------------------
 * out = x + s[b] * gamma[c] * y : convnext_Block tail (gamma*x, DropPath, residual; twig/model/cod.py:1112-1116) and the
 * Block residuals x + DropPath(.) (cod.py:958-959).  s fp32 [B] = per-sample stochastic-depth scale (0 or 1/keep) or NULL;
 * gamma fp32 [C] or NULL; row r belongs to sample r / rows_per_sample.                                                */
int dgtd_scale_residual_fwd(const void* x, const void* y, const float* s, const float* gamma, void* out,
                            int64_t rows, int C, int64_t rows_per_sample, dgtd_dtype dt, dgtd_stream st);
int64_t dgtd_colsum_workspace(int C);
/* dy = s*gamma*g (overwritten); dgamma[c] = sum_r s*g*y (overwritten; NULL iff gamma is NULL).  d(out)/dx is the identity. */
int dgtd_scale_residual_bwd(const void* g, const void* y, const float* s, const float* gamma, void* dy,
                            float* dgamma, void* workspace, int64_t rows, int C, int64_t rows_per_sample,
                            dgtd_dtype dt, dgtd_stream st);
/* Backward of a Linear fused with the op that consumes its output: the bias gradient of the Linear is the column sum of the very
 * gradient these kernels write, so both come from one pass (workspace: dgtd_colsum2_workspace(C) bytes).
 * residual epilogue (cod.py:1112-1116 after pwconv2 :1099; cod.py:958-959 after attn.proj :874 / Mlp.fc2 :832):
 *   dy = s*gamma*g (overwritten), dgamma = sum s*g*y (fp32; NULL iff gamma is NULL), dbias[C] (dtype bias_dt) = sum_r dy.
 * GELU (cod.py:1098 between pwconv1 and pwconv2): dpre = g * gelu'(pre) (overwritten), dbias[C] = sum_r dpre.                    */
int64_t dgtd_colsum2_workspace(int C);
int dgtd_scale_residual_bias_bwd(const void* g, const void* y, const float* s, const float* gamma, void* dy, float* dgamma,
                                 void* dbias, dgtd_dtype bias_dt, void* workspace, int64_t rows, int C,
                                 int64_t rows_per_sample, dgtd_dtype dt, dgtd_stream st);
int dgtd_gelu_bias_bwd(const void* g, const void* pre, void* dpre, void* dbias, dgtd_dtype bias_dt, void* workspace,
                       int64_t rows, int C, dgtd_dtype dt, dgtd_stream st);
/* First stages only: workspace receives [*nblocks][2C] = { dgamma partials (zero columns when gamma is NULL / for the GELU form) |
 * dbias partials }; [*nblocks][C] for dgtd_colsum_partial.                                                                        */
int dgtd_scale_residual_bias_bwd_partial(const void* g, const void* y, const float* s, const float* gamma, void* dy, void* workspace,
                                         int64_t rows, int C, int64_t rows_per_sample, dgtd_dtype dt, int* nblocks, dgtd_stream st);
int dgtd_gelu_bias_bwd_partial(const void* g, const void* pre, void* dpre, void* workspace, int64_t rows, int C, dgtd_dtype dt,
                               int* nblocks, dgtd_stream st);
int dgtd_colsum_partial(const void* x, void* workspace, int64_t rows, int C, dgtd_dtype dt, int* nblocks, dgtd_stream st);
/* out [C] (dtype out_dt, fp32 accumulation) = column sums of x [rows, C]: the bias gradient of nn.Linear
 * (cod.py:829,832,872-875,1097,1099), written in the dtype of the bias it belongs to.                                   */
int dgtd_colsum(const void* x, void* out, dgtd_dtype out_dt, void* workspace, int64_t rows, int C, dgtd_dtype dt,
                dgtd_stream st);

/* ---- The Linear layers as a hand-written MFMA GEMM with this model's epilogues (bf16 / fp16; csrc/gemm.hip) ----------------------
 * D[M,N] = A[M,K] . B[N,K]^T, both operands K-contiguous (nn.Linear's x [tokens, in] and weight [out, in]), fp32 accumulation,
 * all epilogue arithmetic in fp32 with one rounding per output.  Replaces the library GEMM plus the elementwise passes around it at
 * twig/model/cod.py:900-921 (q / kv / proj), :852-859 (fc1 / fc2), :1097-1099 + :1112-1116 (pwconv1 + GELU, pwconv2 + gamma + DropPath
 * + residual).  Shapes: M % 128 == 0, N % 64 == 0, K % 64 == 0 (dgtd_gemm_supported); everything 16-byte aligned.
 *   gemm_bias:          d = a b^T + bias                                   (bias [N] in dt or NULL)
 *   gemm_bias_gelu:     pre = a b^T + bias (stored when pre != NULL: the backward needs it), h = gelu_erf(pre)
 *   gemm_bias_residual: y = a b^T + bias (stored when y != NULL: the layer-scale gradient needs it),
 *                       out = x + scale[row / rows_per_sample] * gamma[col] * y   (scale / gamma fp32 or NULL)
 *   gemm_gelu_bwd:      dpre = (dy w_t^T) * gelu_erf'(pre): the input gradient of the Linear that FOLLOWS a GELU taken through the GELU,
 *                       w_t = that Linear's weight TRANSPOSED ([in, out]: K-contiguous in the reduction dim); colsum_ws (bytes:
 *                       dgtd_gemm_gelu_bwd_workspace) receives [*nblocks][N] fp32 column partials of dpre = first stage of the bias
 *                       gradient of the Linear in front of the GELU (second stage: dgtd_multi_reduce); NULL: no partials.
 * An input gradient dX[M,K] = dY[M,N] . W[N,K] is gemm_bias(dy, w_t, NULL, dx, M, K, N): the same kernel on the transposed weight copy;
 * dgtd_transpose_batched refreshes those copies (dst[i] [cols][rows] = src[i] [rows][cols]^T) for a whole model in one launch.      */
int dgtd_gemm_supported(int M, int N, int K, dgtd_dtype dt);
int dgtd_gemm_bias(const void* a, const void* b, const void* bias, void* d, int M, int N, int K, dgtd_dtype dt, dgtd_stream s);
int dgtd_gemm_bias_gelu(const void* a, const void* b, const void* bias, void* pre, void* h, int M, int N, int K, dgtd_dtype dt,
                        dgtd_stream s);
int dgtd_gemm_bias_residual(const void* a, const void* b, const void* bias, const void* x, const float* scale, const float* gamma,
                            void* y, void* out, int M, int N, int K, int64_t rows_per_sample, dgtd_dtype dt, dgtd_stream s);
int64_t dgtd_gemm_gelu_bwd_workspace(int M, int N);
int dgtd_gemm_gelu_bwd(const void* dy, const void* w_t, const void* pre, void* dpre, void* colsum_ws, int* nblocks, int M, int N, int K,
                       dgtd_dtype dt, dgtd_stream s);
int dgtd_transpose_batched(const void* const* src, void* const* dst, const int* rows, const int* cols, int n, dgtd_dtype dt,
                           dgtd_stream s);

/* ---- Fused structure loss of the five deep-supervision heads (fp32) -------------------------------
 * loss = sum_k mix[k] * cal_loss(bilinear_x(S/hs)(lo[k]), label): twig/model/cod.py:76-85 (weighted BCE + weighted IoU with
 * weit = 1 + 5|avgpool31(gt) - gt|), :137-142 (mix = 0, .2, .4, .6 for P1[0..3], 1 for P2) and the x8 align_corners=False
 * up-sampling of cod.py:796, :806.  lo [5,B,hs,hs] low-res logits; label [B,1,S,S]; mix [5]; loss [1].
 * workspace (dgtd_seg_loss_workspace bytes) carries the weight map and the per-(map, sample) sums to the backward.   */
int64_t dgtd_seg_loss_workspace(int B, int S);
int dgtd_seg_loss_fwd(const float* lo, const float* label, const float* mix, float* loss, void* workspace,
                      int B, int S, int hs, dgtd_stream s);
/* dlo [5,B,hs,hs] overwritten = gout[0] * d loss / d lo.                                                                */
int dgtd_seg_loss_bwd(const float* lo, const float* label, const float* mix, const float* gout, float* dlo,
                      const void* workspace, int B, int S, int hs, dgtd_stream s);

/* SSIM value of the high-pass image (cod.py:143-144 with SSIM._ssim cod.py:330-348; it has no gradient path to any parameter):
 * out[0] = mean(clamp((1 - ssim_n/ssim_d)/2, 0, 1)) over reflect-padded 3x3 windows of e = (x_hp - min x_hp)/(max x_hp - min x_hp + 1e-8)
 * and `image`; x_hp, image fp32 [B,C,S,S] (NCHW).  Three launches instead of ~30 elementwise passes over [B,3,S,S].
 * workspace: dgtd_ssim_workspace() bytes.                                                                                */
int64_t dgtd_ssim_workspace(void);
int dgtd_ssim_value(const float* x_hp, const float* image, float* out, void* workspace, int B, int C, int S, dgtd_stream s);

/* ---- Texture diffuser front end (fp32) -------------------------------------------------------
 * replaces twig/model/cod.py:1295-1298 (nearest 12x12 sample of the FFT high-pass image, 1x1 conv 3->1176,
 * sigmoid; depth 1x1 conv 1->24 + bilinear to 12x12) and MessagePassing.forward cod.py:1193-1205
 * (random-walk normalisation + 4 zero-padded 7x7 propagation steps).
 * x_hp [B,3,S,S], depth [B,1,S,S]; reg_w [1176,3], reg_b [1176] (regressor channel = c*49 + ky*7 + kx);
 * enc_w [24], enc_b [24]  ->  x4 [B,24,12,12].                                                   */
int dgtd_diffuser_fwd(const float* x_hp, const float* depth, const float* reg_w, const float* reg_b,
                      const float* enc_w, const float* enc_b, float* x4, int B, int S, dgtd_stream s);
/* parameter gradients only (the inputs carry none); d_* must be ZEROED by the caller (atomics over batch). */
int dgtd_diffuser_bwd(const float* x_hp, const float* depth, const float* reg_w, const float* reg_b,
                      const float* enc_w, const float* enc_b, const float* g4, float* d_reg_w,
                      float* d_reg_b, float* d_enc_w, float* d_enc_b, int B, int S, dgtd_stream s);
/* out [B,3,S,S] = bilinear_{12->S, align_corners=False}(conv1x1(x4; cw [3,24], cb [3])) + image
 * replaces cod.py:1206-1207 and the `+ image` of cod.py:1302.                                     */
int dgtd_diffuse_tail_fwd(const float* x4, const float* cw, const float* cb, const float* image,
                          float* out, int B, int S, dgtd_stream s);
int64_t dgtd_diffuse_tail_bwd_workspace(int B);
/* g4 [B,24,12,12] overwritten; d_cw [3,24], d_cb [3] ZEROED by the caller.                        */
int dgtd_diffuse_tail_bwd(const float* gout, const float* x4, const float* cw, float* g4, float* d_cw,
                          float* d_cb, void* workspace, int B, int S, dgtd_stream s);

/* ---- Hitnet CAB decoder glue on channels_last maps viewed as [B, HW, C] ------------------------------
 * PReLU with ONE shared slope a[1] (the default-argument nn.PReLU() of twig/model/cod.py:686, applied at cod.py:444-446):
 * y = x > 0 ? x : a*x over n elements (any dense layout).                                                             */
int dgtd_prelu_fwd(const void* x, const float* a, void* y, int64_t n, dgtd_dtype dt, dgtd_stream s);
/* dx = g * (x > 0 ? 1 : a) overwritten; da[1] += sum_{x<=0} g*x, ZEROED by the caller.                                   */
int dgtd_prelu_bwd(const void* x, const void* g, const float* a, void* dx, float* da, int64_t n,
                   dgtd_dtype dt, dgtd_stream s);
/* out = res * sigmoid(W2 relu(W1 mean_hw(res))) + x : CALayer.forward (cod.py:428-431) plus the residual of CAB.forward
 * (cod.py:449-451).  res, x, out [B,HW,C]; w1 fp32 [R,C], w2 fp32 [C,R] (the bias-free 1x1 convs of cod.py:421-425);
 * C <= 128, R <= 32.  stats fp32 [2*B*C + B*R + 64*B*C] = { pooled sums | gate | hidden | per-slice partial sums }: the first three
 * are kept for the backward; nothing has to be zeroed (fixed-order sums, deterministic).                                          */
int dgtd_ca_gate_fwd(const void* res, const void* x, const float* w1, const float* w2, void* out, float* stats,
                     int B, int HW, int C, int R, dgtd_dtype dt, dgtd_stream s);
/* dres [B,HW,C], dw1 [R,C], dw2 [C,R] overwritten (d out / d x is the identity); scratch fp32 [B*C + 64*B*C + B*2*R*C]; nothing to zero. */
int dgtd_ca_gate_bwd(const void* g, const void* res, const float* w1, const float* w2, const float* stats, void* dres,
                     float* dw1, float* dw2, float* scratch, int B, int HW, int C, int R, dgtd_dtype dt, dgtd_stream s);
/* The same with the weight gradients left as one row per sample: dw_rows fp32 [B][2*R*C] = { dh_b x mean_b | dz_b x hidden_b } overwritten;
 * the caller sums the rows (no serial walk over the batch inside the launch).                                                      */
int dgtd_ca_gate_bwd_rows(const void* g, const void* res, const float* w1, const float* w2, const float* stats, void* dres,
                          float* dw_rows, float* scratch, int B, int HW, int C, int R, dgtd_dtype dt, dgtd_stream s);

/* SAM (twig/model/cod.py:454-506) over NHWC maps x_h, x_l [B,HW,C]: out = x_h G(x_h) + x_l G(x_l) with
 * G(x)[b][c] = sigmoid(W2 relu(W1 y))[c] * sigmoid(V2 relu(V1 y)), y = mean_hw(x)[b]; W1 [R,C], W2 [C,R] = SAM.fc (cod.py:459-464),
 * V1 [R,C], V2 [R] = SAM.fc_wight (cod.py:465-470), fp32.  C <= 128, R <= 32, 3 R C + R <= 1024.  stats fp32
 * [dgtd_sam_stats_floats(B, C, R)] is written by the forward and read by the backward; nothing has to be zeroed (fixed-order sums).
 * Backward: dxh, dxl [B,HW,C] and dw fp32 { dW1 [R,C] | dW2 [C,R] | dV1 [R,C] | dV2 [R] } overwritten; scratch fp32
 * [dgtd_sam_scratch_floats(B, C)].                                                                                               */
int dgtd_sam_supported(int B, int HW, int C, int R, dgtd_dtype dt);
int64_t dgtd_sam_stats_floats(int B, int C, int R);
int64_t dgtd_sam_scratch_floats(int B, int C);
int dgtd_sam_fwd(const void* xh, const void* xl, const float* w1, const float* w2, const float* v1, const float* v2, void* out,
                 float* stats, int B, int HW, int C, int R, dgtd_dtype dt, dgtd_stream s);
int dgtd_sam_bwd(const void* g, const void* xh, const void* xl, const float* w1, const float* w2, const float* v1, const float* v2,
                 const float* stats, void* dxh, void* dxl, float* dw, float* scratch, int B, int HW, int C, int R, dgtd_dtype dt,
                 dgtd_stream s);

/* nn.BatchNorm2d of BasicConv2d (twig/model/cod.py:359, :366) over an NHWC map x [N = B*H*W, C]; gamma, beta, running statistics fp32
 * (NULL gamma / beta = no affine).  training != 0: batch mean and biased variance per channel, y = (x - mean) rstd gamma + beta,
 * running_mean / running_var (NULL = not tracked) moved by `momentum` with the unbiased variance, *num_batches += 1 (NULL = not
 * counted), save fp32 [2*C] = { mean | rstd } for the backward, scratch fp32 [dgtd_batchnorm_scratch(C)].  training == 0: the affine
 * map of the running statistics (save, scratch, num_batches unused).  C a power of two in [8, 128] (16-bit) / [4, 128] (fp32).
 * Backward (training statistics): dx [N,C], dgamma [C], dbeta [C] (either may be NULL) overwritten; nothing to zero.            */
int dgtd_batchnorm_supported(int64_t N, int C, dgtd_dtype dt);
int64_t dgtd_batchnorm_scratch(int C);
int dgtd_batchnorm_fwd(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                       long long* num_batches, void* y, float* save, float* scratch, int64_t N, int C, float eps, float momentum,
                       int training, dgtd_dtype dt, dgtd_stream s);
int dgtd_batchnorm_bwd(const void* dy, const void* x, const float* gamma, const float* save, void* dx, float* dgamma, float* dbeta,
                       float* scratch, int64_t N, int C, dgtd_dtype dt, dgtd_stream s);

/* Bilinear resize of an NHWC map, x [B,Hi,Wi,C] -> y [B,Ho,Wo,C]: F.interpolate(mode="bilinear") with either align_corners
 * convention - the x2 / x4 / x0.5 align_corners=True resizes between the Hitnet decoder levels (cod.py:757-789).  The backward
 * gathers, for every input pixel, the output pixels that read it (dx overwritten, no atomics).                               */
int dgtd_bilinear_fwd(const void* x, void* y, int B, int Hi, int Wi, int Ho, int Wo, int C, int align_corners,
                      dgtd_dtype dt, dgtd_stream s);
int dgtd_bilinear_bwd(const void* dy, void* dx, int B, int Hi, int Wi, int Ho, int Wo, int C, int align_corners,
                      dgtd_dtype dt, dgtd_stream s);

/* ---- Dense 3x3 convolution, stride 1, zero padding 1, NHWC bf16, small channel counts ---------------------
 * replaces the conv3x3(24->24)+ReLU pairs of the 16 prompt decoders (ShapePropDecoder, twig/model/cod.py:1216-1226, called at
 * cod.py:1316-1323) and the conv3x3 C->C bodies of the Hitnet CABs (cod.py:441-446).  Z independent convolutions per launch:
 * x [Z,B,H,W,Ci] (or [B,H,W,Ci] shared by all Z when shared_x != 0), w [Z,Co,3,3,Ci] (Conv2d weight in O,H,W,I order),
 * bias [Z,Co] or NULL, y [Z,B,H,W,Co]; all of dtype dt (DGTD_BF16 or DGTD_F16), fp32 accumulation.  Ci, Co in {24, 32, 64, 96}; W a multiple of 16.
 * mask (NULL or the shape of x): x is taken as zero where mask <= 0 - the ReLU backward fused into the tile load.
 * Backward w.r.t. x = dgtd_conv3x3_fwd on dy with the kernel from dgtd_conv3x3_flip (Ci and Co swapped, mask = forward output). */
int dgtd_conv3x3_supported(int Ci, int Co, int H, int W);
int dgtd_conv3x3_fwd(const void* x, const void* mask, const void* w, const void* bias, void* y, int Z, int B, int H, int W,
                     int Ci, int Co, int relu, int shared_x, dgtd_dtype dt, dgtd_stream s);
/* The same convolution with the epilogues a Hitnet CAB needs around its two convolutions (cod.py:441-451), so that the PReLU, its
 * backward and the skip-connection gradient are not separate launches.  y2 / add / ref have y's layout [Z,B,H,W,Co]; any may be NULL.
 *   act 0 none, 1 ReLU;  act 2 PReLU forward: y2 (optional) = pre-activation, y = pre > 0 ? pre : slope[0] * pre  (act = body[1], cod.py:444);
 *   act 3 PReLU backward: the convolution output v is the gradient w.r.t. the PReLU OUTPUT, ref = the saved pre-activation:
 *         y = ref > 0 ? v : slope[0] * v and sum(v * ref | ref <= 0) is added atomically to slope_grad[0] (one add per workgroup; optional);
 *   add: y += add, applied last (the gradient of the skip connection `+ x` of cod.py:451 that forks off the convolution's input).     */
int dgtd_conv3x3_fwd_ex(const void* x, const void* mask, const void* w, const void* bias, void* y, void* y2, const void* add,
                        const void* ref, const float* slope, float* slope_grad, int Z, int B, int H, int W, int Ci, int Co, int act,
                        int shared_x, dgtd_dtype dt, dgtd_stream s);
/* wt [Z,Ci,3,3,Co] = w [Z,Co,3,3,Ci] transposed, taps flipped (any 2-byte dtype).                                                                */
int dgtd_conv3x3_flip(const void* w, void* wt, int Z, int Co, int Ci, dgtd_stream s);
/* dw [Z,Co,3,3,Ci] and db [Z,Co] (or NULL) of dtype dt, OVERWRITTEN; dy is masked like x above (mask = forward output, or NULL).
 * workspace: dgtd_conv3x3_wgrad_workspace(...) bytes of per-workgroup partial sums, reduced in a fixed order.                   */
int64_t dgtd_conv3x3_wgrad_workspace(int Z, int B, int H, int W, int Ci, int Co);
/* Deferred weight-gradient phase: n convolutions of SEPARATE calls (same geometry) in one launch, feeding nslots weights.  x, dy, mask
 * (mask: NULL, or n pointers each of which may be NULL), slot: HOST arrays of n; slot[i] in [0, nslots) names the weight
 * entry i contributes to, and every weight receives exactly n / nslots entries (a module called that many times per step): dw[slot] =
 * sum over its entries, formed directly from the partial sums - the per-call gradients never exist.  dw, db (NULL or nslots pointers):
 * HOST arrays.  n <= 32.                                                                                                          */
int64_t dgtd_conv3x3_wgrad_batched_workspace(int n, int B, int H, int W, int Ci, int Co);
int dgtd_conv3x3_wgrad_batched(const void* const* x, const void* const* dy, const void* const* mask, const int* slot, int n, void* const* dw,
                               void* const* db, int nslots, void* workspace, int B, int H, int W, int Ci, int Co, dgtd_dtype dt, dgtd_stream s);
int dgtd_conv3x3_wgrad(const void* x, const void* dy, const void* mask, void* dw, void* db, void* workspace, int Z, int B,
                       int H, int W, int Ci, int Co, int shared_x, dgtd_dtype dt, dgtd_stream s);

/* ---- AdamW over one contiguous run of a flat fp32 parameter bucket (+ the bf16 working copy in the same pass) ----
 * replaces torch.optim.AdamW(fused) per the reference's optim_wrapper (config/sod.yml:56-76: AdamW, weight_decay 0.1, per-prefix
 * lr multipliers: one call per run of equal lr).  p, g, m, v fp32 [n] at the same 16-byte phase; w_bf16 [n] or NULL.
 * p *= 1-lr*wd; m += (1-b1)(g-m); v = b2 v + (1-b2) g*g; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps); w = bf16(p).               */
int dgtd_adamw_flat(float* p, const float* g, float* m, float* v, void* w_bf16, int64_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, float bias_correction1, float bias_correction2, dgtd_stream s);

/* The same with the working copy in w_dt (DGTD_BF16 or DGTD_F16) and the reference's AMP recipe (AmpOptimWrapper = GradScaler,
 * config/sod.yml:57) folded in.  amp_state (device, fp32 [5] = { scale, growth_tracker, 1/scale, found_inf, steps taken }, or NULL):
 * gradients are multiplied by 1/scale, the whole launch is a no-op when found_inf != 0 (GradScaler.unscale_ + the skipped step of
 * an overflowed iteration) and the bias corrections are 1 - beta^(steps taken + 1) computed on the device (skipped steps do not
 * count; bias_correction1/2 are ignored).  lr_dev (device scalar or NULL): when given, the learning rate is read from it instead of
 * `lr` - together with amp_state this leaves no per-step host value in the launch, so a captured hipGraph of the step replays
 * correctly while the schedule and the step count advance.                                                                        */
int dgtd_adamw_flat_amp(float* p, const float* g, float* m, float* v, void* w, dgtd_dtype w_dt, int64_t n, float lr, float beta1,
                        float beta2, float eps, float weight_decay, float bias_correction1, float bias_correction2,
                        const float* amp_state, const float* lr_dev, dgtd_stream s);
/* The same update with the gradient read from the 16-bit all-reduce payload g16 [n] (dtype w_dt, at the phase of w) instead of an fp32
 * bucket: the N > 1 step does not widen the payload into fp32 first (dist.GradReducer buckets, config/sod.yml:11's DDP reduction).   */
int dgtd_adamw_flat_g16(float* p, const void* g16, float* m, float* v, void* w, dgtd_dtype w_dt, int64_t n, float lr, float beta1,
                        float beta2, float eps, float weight_decay, float bias_correction1, float bias_correction2,
                        const float* amp_state, const float* lr_dev, dgtd_stream s);
/* found[0] = 1 when g[0,n) (fp32, the still scaled gradients) holds an inf or a NaN; untouched otherwise.                        */
int dgtd_found_inf(const float* g, int64_t n, float* found, dgtd_stream s);
/* GradScaler.update() on the device: state fp32 [5] (layout above); found_inf != 0: scale *= backoff, tracker = 0; else steps += 1,
 * tracker += 1 and every growth_interval clean steps scale *= growth (kept when the product overflows); found_inf is cleared.
 * No host synchronisation anywhere in the scaled step.                                                      */
int dgtd_loss_scale_update(float* state, float growth_factor, float backoff_factor, int growth_interval, dgtd_stream s);

/* ---- Dense KxK strided convolution as patch gather + library GEMM -----------------------------------------------------------
 * replaces the convolutions that are neither 3x3 stride-1 16-bit (dgtd_conv3x3_*) nor k == stride patchify views:
 * OverlapPatchEmbed.proj (twig/model/cod.py:974-975, :1000: k7 s4 p3 / k3 s2 p1), the prompt-decoder tails folded into 4x4 stride-s
 * convolutions (cod.py:1220 + :1471), Hitnet.compress_out (k8 s4 p2, cod.py:739) and, in fp32 parity mode, the 3x3 convolutions of
 * the Hitnet decoder (cod.py:441-446, :713).
 * im2col: x is read through element strides (sb, sy, sx, sc) - an NCHW image, an NHWC map or an offset view alike - in dtype x_dt;
 *         col [B*Ho*Wo, K*K*C] (column order ky, kx, c = the O,H,W,I kernel flattened) is written in col_dt; zero padding `pad`.
 * col2im: dx [B,H,W,C] (NHWC, contiguous, overwritten) = the gradient of im2col: every input pixel GATHERS the windows that cover
 *         it (no atomics).                                                                                                      */
int dgtd_im2col(const void* x, void* col, int B, int H, int W, int C, int64_t sb, int64_t sy, int64_t sx, int64_t sc, int K,
                int stride, int pad, int Ho, int Wo, dgtd_dtype x_dt, dgtd_dtype col_dt, dgtd_stream s);
int dgtd_col2im(const void* dcol, void* dx, int B, int H, int W, int C, int K, int stride, int pad, int Ho, int Wo, dgtd_dtype dt,
                dgtd_stream s);

/* ---- Many small tensors <-> one flat buffer, with dtype conversion, one launch per 128 tensors ---------------------------------
 * replaces torch.cat / torch.stack on the training path: the gradient-bucket gather of the data-parallel reducer (what DDP's bucket
 * copies do for config/sod.yml's MMDistributedDataParallel), the per-step stack of the 16 prompt-decoder kernels and of the five
 * deep-supervision maps.  tensors / offsets / counts are HOST arrays of n_tensors entries: device pointer of tensor i (contiguous,
 * dtype tensor_dt), its element offset in `flat` (dtype flat_dt) and its element count.  to_tensors = 0: flat[offset_i + k] =
 * tensor_i[k];  to_tensors = 1: tensor_i[k] = flat[offset_i + k].  The table travels by value in the kernel arguments, so the launch
 * can be captured in a hipGraph (torch.cat on ROCm stages its table through recycled pinned host memory and cannot).               */
int dgtd_multi_copy(const void* const* tensors, const int64_t* offsets, const int64_t* counts, int n_tensors, dgtd_dtype tensor_dt,
                    void* flat, dgtd_dtype flat_dt, int to_tensors, dgtd_stream s);

/* ---- Multi-scale deformable attention sampling: the reference's own native op (twig/ops) ------------------------
 * replaces MSDA.ms_deform_attn_forward / ms_deform_attn_backward (twig/ops/src/ms_deform_attn.h:20-60, bound at
 * twig/ops/functions/ms_deform_attn_func.py:24-46); semantics = ms_deform_attn_core_pytorch (ms_deform_attn_func.py:49-71).
 * value [N,S,M,D]; spatial_shapes int64 [L,2] = (H_l, W_l); level_start_index int64 [L]; sampling_loc [N,Lq,M,L,P,2] = (x, y) in
 * [0,1]; attn_weight [N,Lq,M,L,P]; out [N,Lq,M*D].  dt = DGTD_F32 or DGTD_F64 (the reference casts half inputs to float32,
 * ms_deform_attn_func.py:21).  The reference's im2col_step only batches its host loop and has no counterpart here.             */
int dgtd_ms_deform_attn_fwd(const void* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                            const void* sampling_loc, const void* attn_weight, void* out, int N, int S, int M, int D, int L,
                            int Lq, int P, dgtd_dtype dt, dgtd_stream s);
/* grad_value [N,S,M,D] must be ZEROED by the caller (atomics); grad_sampling_loc and grad_attn_weight are overwritten.           */
int dgtd_ms_deform_attn_bwd(const void* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                            const void* sampling_loc, const void* attn_weight, const void* grad_out, void* grad_value,
                            void* grad_sampling_loc, void* grad_attn_weight, int N, int S, int M, int D, int L, int Lq, int P,
                            dgtd_dtype dt, dgtd_stream s);

/* ---- Input pipeline of one sample on the device --------------------------------------------------------
 * replaces the per-sample transforms of twig/dataset/sod_train.py:31-54 (applied at :65-83): RandomHorizontalFlip (the caller
 * draws the coin) -> Resize((S,S)) = Pillow BILINEAR with antialiasing (requirements.txt:92, via torchvision Resize :148) ->
 * ToTensor -> optional Normalize.  img_u8 [Hin,Win,C] uint8 on the device (C = 3 RGB or 1 for GT / depth); out [C,S,S] in out_dt.
 * mean_host / std_host: HOST arrays of C floats or both NULL (no normalisation).  Bit-exact against Pillow + float32 ToTensor /
 * Normalize.  workspace: dgtd_preprocess_workspace(...) bytes on the device (uint8 intermediate + coefficient tables).           */
int64_t dgtd_preprocess_workspace(int Hin, int Win, int C, int S);
int dgtd_preprocess(const void* img_u8, void* out, const float* mean_host, const float* std_host, void* workspace, int Hin,
                    int Win, int C, int S, int flip, dgtd_dtype out_dt, dgtd_stream s);

#ifdef __cplusplus
}
#endif
#endif /* DGTD_H */
