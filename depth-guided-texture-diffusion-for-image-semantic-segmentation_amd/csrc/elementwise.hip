// elementwise.hip — fused residual epilogues and column sums on token-major [rows, C] tensors.
//   scale_residual: out = x + s[b] * gamma[c] * y      (convnext_Block tail: gamma * x, DropPath, residual, twig/model/cod.py:1112-1116;
//                                                       Block residuals x + DropPath(.) cod.py:958-959; s = per-sample stochastic-depth scale)
//   its backward:   dy = s[b] * gamma[c] * g ,  dgamma[c] = sum_r s[b] * g * y      (dx = g is the identity, no kernel)
//   colsum:         out[c] = sum_r x[r, c]             (bias gradients of the Linear layers: one pass instead of a generic reduce)
// HBM-bound: fwd 3e*rows*C, bwd 3e*rows*C, colsum e*rows*C.  Lanes run along C in 16-byte chunks; column partials stay in
// registers across the row loop, meet in LDS per workgroup, and a deterministic second stage sums the per-workgroup rows.
#include "common.h"

namespace {

constexpr int EW_MAX_BLOCKS = 256;        // partial rows per column strip (second stage runs inside the same launch)
constexpr int EW_STRIP = 8;               // 16-byte chunks per workgroup column strip = 128 bytes of every row
constexpr int EW_MAX_STRIPS = 1024;
constexpr int EW_COUNTER_BYTES = EW_MAX_STRIPS * 4;

template <typename T>
__global__ __launch_bounds__(256) void scale_residual_fwd_kernel(const T* __restrict__ x, const T* __restrict__ y,
                                                                 const float* __restrict__ s, const float* __restrict__ gamma,
                                                                 T* __restrict__ out, int64_t rows, int C, int64_t rows_per_sample) {
  typedef typename Vec16<T>::type VT;
  constexpr int V = Vec16<T>::N;
  const int CV = C / V;
  const int64_t total = rows * CV;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cv = (int)(i % CV);
    const int64_t r = i / CV;
    const float sb = s ? s[r / rows_per_sample] : 1.f;
    VT xv = *reinterpret_cast<const VT*>(x + i * V);
    VT yv = *reinterpret_cast<const VT*>(y + i * V);
    VT o;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float g = gamma ? gamma[cv * V + j] : 1.f;
      o[j] = (T)((float)xv[j] + sb * g * (float)yv[j]);
    }
    *reinterpret_cast<VT*>(out + i * V) = o;
  }
}

// MODE 0: colsum of g.   MODE 1: dy = s*gamma*g and partial dgamma = sum s*g*y.   MODE 2: dy = s*g only (no gamma, no sums).
template <typename T, int MODE>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ g, const T* __restrict__ y, const float* __restrict__ s,
                                                     const float* __restrict__ gamma, T* __restrict__ dy, float* __restrict__ ws,
                                                     unsigned* __restrict__ counters, void* __restrict__ out, int out_bf16,
                                                     int64_t rows, int C, int64_t rows_per_sample) {
  typedef typename Vec16<T>::type VT;
  constexpr int V = Vec16<T>::N;
  __shared__ float red[256 * V];                 // [RG][CVB * V]
  const int CV = C / V;
  const int CVB = min(CV - (int)blockIdx.y * EW_STRIP, EW_STRIP);   // 16-byte column chunks handled by this workgroup (a 128-byte strip)
  const int RG = 256 / CVB;                                         // row lanes inside the workgroup
  const int tid = threadIdx.x;
  const int cvl = tid % CVB, rg = tid / CVB;
  const int cv = blockIdx.y * EW_STRIP + cvl;
  float acc[V];
#pragma unroll
  for (int j = 0; j < V; ++j) acc[j] = 0.f;
  float gm[V];
#pragma unroll
  for (int j = 0; j < V; ++j) gm[j] = (MODE == 1 && gamma) ? gamma[cv * V + j] : 1.f;
  if (rg < RG) {
    for (int64_t r = (int64_t)blockIdx.x * RG + rg; r < rows; r += (int64_t)gridDim.x * RG) {
      const size_t off = (size_t)r * C + (size_t)cv * V;
      VT gv = *reinterpret_cast<const VT*>(g + off);
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] += (float)gv[j];
      } else {
        const float sb = s ? s[r / rows_per_sample] : 1.f;
        VT o;
        if (MODE == 1) {
          VT yv = *reinterpret_cast<const VT*>(y + off);
#pragma unroll
          for (int j = 0; j < V; ++j) {
            const float gg = sb * (float)gv[j];
            acc[j] += gg * (float)yv[j];
            o[j] = (T)(gg * gm[j]);
          }
        } else {
#pragma unroll
          for (int j = 0; j < V; ++j) o[j] = (T)(sb * (float)gv[j]);
        }
        *reinterpret_cast<VT*>(dy + off) = o;
      }
    }
  }
  if (MODE == 2) return;
  if (rg < RG)
#pragma unroll
    for (int j = 0; j < V; ++j) red[(rg * CVB + cvl) * V + j] = acc[j];
  __syncthreads();
  const int ncol = CVB * V, col0 = blockIdx.y * EW_STRIP * V;
  for (int i = tid; i < ncol; i += 256) {
    float t = 0.f;
    for (int k = 0; k < RG; ++k) t += red[k * ncol + i];
    ws[(size_t)blockIdx.x * C + col0 + i] = t;
  }
  // second stage in the same launch: the last workgroup of this column strip sums the gridDim.x partial rows in a fixed order
  if (!last_group_done(counters + blockIdx.y, gridDim.x)) return;
  const int nb = gridDim.x, i = tid % ncol, q = tid / ncol, Q = 256 / ncol;   // ncol <= 64: Q >= 4 interleaved partial sums per column
  float t = 0.f;
  if (q < Q)
    for (int k = q; k < nb; k += Q) t += ws[(size_t)k * C + col0 + i];
  __syncthreads();
  red[tid] = t;
  __syncthreads();
  if (tid < ncol) {
    float v = 0.f;
    for (int k = 0; k < Q; ++k) v += red[k * ncol + tid];
    if (out_bf16) ((bf16_t*)out)[col0 + tid] = (bf16_t)v;
    else ((float*)out)[col0 + tid] = v;
  }
}

template <typename T>
int check_c(int C, const char* who) {
  constexpr int V = Vec16<T>::N;
  DGTD_REQUIRE(C % V == 0, "%s: C=%d must be a multiple of %d", who, C, V);
  return 0;
}

template <typename T, int MODE>
int colsum_launch(const void* g, const void* y, const float* s, const float* gamma, void* dy, void* out, void* ws, int64_t rows,
                  int C, int64_t rps, hipStream_t st, const char* who, int out_bf16 = 0) {
  if (int rc = check_c<T>(C, who)) return rc;
  constexpr int V = Vec16<T>::N;
  const int CV = C / V, strips = (int)cdiv(CV, EW_STRIP);
  DGTD_REQUIRE(strips <= EW_MAX_STRIPS, "%s: C=%d is wider than %d columns", who, C, EW_MAX_STRIPS * EW_STRIP * V);
  const int rg = 256 / std::min(CV, EW_STRIP);
  // ~1024 workgroups in total, at least 2 rows per row lane, at most EW_MAX_BLOCKS partial rows for the in-launch second stage
  const int64_t want = std::max<int64_t>(1, 1024 / strips);
  const int gx = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(want, cdiv(rows, (int64_t)rg * 2)), EW_MAX_BLOCKS));
  unsigned* counters = (unsigned*)ws;
  float* partial = (float*)((char*)ws + EW_COUNTER_BYTES);
  hipLaunchKernelGGL((colsum_kernel<T, MODE>), dim3(gx, strips), dim3(256), 0, st, (const T*)g, (const T*)y, s, gamma, (T*)dy, partial,
                     counters, out, out_bf16, rows, C, rps);
  DGTD_CHECK_LAUNCH(who);
  return 0;
}

}  // namespace

extern "C" int dgtd_scale_residual_fwd(const void* x, const void* y, const float* s, const float* gamma, void* out, int64_t rows,
                                       int C, int64_t rows_per_sample, dgtd_dtype dt, dgtd_stream st) {
  DGTD_REQUIRE(rows > 0 && C > 0 && rows_per_sample > 0, "scale_residual_fwd: bad sizes");
  const int V = dt == DGTD_BF16 ? 8 : 4;
  DGTD_REQUIRE(C % V == 0, "scale_residual_fwd: C=%d must be a multiple of %d", C, V);
  const int grid = (int)std::min<int64_t>(cdiv(rows * (C / V), 256), 256 * 16);
  if (dt == DGTD_BF16) hipLaunchKernelGGL(scale_residual_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)st, (const bf16_t*)x, (const bf16_t*)y, s, gamma, (bf16_t*)out, rows, C, rows_per_sample);
  else if (dt == DGTD_F32) hipLaunchKernelGGL(scale_residual_fwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)st, (const float*)x, (const float*)y, s, gamma, (float*)out, rows, C, rows_per_sample);
  else DGTD_FAIL(2, "scale_residual_fwd: bad dtype %d", (int)dt);
  DGTD_CHECK_LAUNCH("scale_residual_fwd");
  return 0;
}

extern "C" int64_t dgtd_colsum_workspace(int C) { return EW_COUNTER_BYTES + (int64_t)EW_MAX_BLOCKS * C * sizeof(float); }

extern "C" int dgtd_scale_residual_bwd(const void* g, const void* y, const float* s, const float* gamma, void* dy, float* dgamma,
                                       void* workspace, int64_t rows, int C, int64_t rows_per_sample, dgtd_dtype dt, dgtd_stream st) {
  DGTD_REQUIRE(rows > 0 && C > 0 && rows_per_sample > 0, "scale_residual_bwd: bad sizes");
  DGTD_REQUIRE((gamma == nullptr) == (dgamma == nullptr), "scale_residual_bwd: gamma and dgamma go together");
  hipStream_t h = (hipStream_t)st;
  if (dt == DGTD_BF16) return gamma ? colsum_launch<bf16_t, 1>(g, y, s, gamma, dy, dgamma, workspace, rows, C, rows_per_sample, h, "scale_residual_bwd")
                                    : colsum_launch<bf16_t, 2>(g, y, s, gamma, dy, dgamma, workspace, rows, C, rows_per_sample, h, "scale_residual_bwd");
  if (dt == DGTD_F32) return gamma ? colsum_launch<float, 1>(g, y, s, gamma, dy, dgamma, workspace, rows, C, rows_per_sample, h, "scale_residual_bwd")
                                   : colsum_launch<float, 2>(g, y, s, gamma, dy, dgamma, workspace, rows, C, rows_per_sample, h, "scale_residual_bwd");
  DGTD_FAIL(2, "scale_residual_bwd: bad dtype %d", (int)dt);
}

extern "C" int dgtd_colsum(const void* x, void* out, dgtd_dtype out_dt, void* workspace, int64_t rows, int C, dgtd_dtype dt, dgtd_stream st) {
  DGTD_REQUIRE(rows > 0 && C > 0, "colsum: bad sizes");
  DGTD_REQUIRE(out_dt == DGTD_F32 || out_dt == DGTD_BF16, "colsum: bad output dtype %d", (int)out_dt);
  const int ob = out_dt == DGTD_BF16;
  if (dt == DGTD_BF16) return colsum_launch<bf16_t, 0>(x, nullptr, nullptr, nullptr, nullptr, out, workspace, rows, C, 1, (hipStream_t)st, "colsum", ob);
  if (dt == DGTD_F32) return colsum_launch<float, 0>(x, nullptr, nullptr, nullptr, nullptr, out, workspace, rows, C, 1, (hipStream_t)st, "colsum", ob);
  DGTD_FAIL(2, "colsum: bad dtype %d", (int)dt);
}
