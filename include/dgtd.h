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
 * fp32 accumulation (throughput mode, bf16 MFMA).  Statistics (lse, mean, rstd) and all
 * parameter gradients are always fp32.
 */
#ifndef DGTD_H
#define DGTD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { DGTD_F32 = 0, DGTD_BF16 = 1 } dgtd_dtype;
typedef void* dgtd_stream; /* hipStream_t */

int dgtd_version(void);
const char* dgtd_last_error(void);

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

/* ---- Spatial-reduction multi-head attention core: softmax(Q K^T * scale) V, head_dim = 64 -----
 * replaces twig/model/cod.py:913-917 (the q/kv/proj Linears stay GEMM calls).
 * q   [B, N,   heads*64]      = output of Attention.q   (cod.py:902) — head h in cols [64h, 64h+64)
 * kv  [B, Nkv, 2*heads*64]    = output of Attention.kv  (cod.py:908) — K then V, same head split
 * out [B, N,   heads*64]      = (attn @ v).transpose(1,2).reshape(B,N,C) (cod.py:917)
 * lse [B, heads, N] fp32      = log-sum-exp of the scaled scores (saved for the backward)       */
int dgtd_sra_attn_fwd(const void* q, const void* kv, void* out, float* lse,
                      int B, int N, int Nkv, int heads, float scale, dgtd_dtype dt, dgtd_stream s);
/* dq [B,N,C] (dt);  dkv_f32 [B,Nkv,2C] fp32, must be ZEROED by the caller (atomically accumulated).
 * workspace: dgtd_sra_attn_bwd_workspace(B,N,heads) bytes (holds delta = rowsum(dO*O)).          */
int64_t dgtd_sra_attn_bwd_workspace(int B, int N, int heads);
int dgtd_sra_attn_bwd(const void* q, const void* kv, const void* out, const void* dout,
                      const float* lse, void* dq, float* dkv_f32, void* workspace,
                      int B, int N, int Nkv, int heads, float scale, dgtd_dtype dt, dgtd_stream s);

#ifdef __cplusplus
}
#endif
#endif /* DGTD_H */
