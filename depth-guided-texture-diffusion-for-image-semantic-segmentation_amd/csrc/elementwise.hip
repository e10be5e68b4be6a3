// elementwise.hip — fused residual epilogues and column sums on token-major [rows, C] tensors.
//   scale_residual: out = x + s[b] * gamma[c] * y      (convnext_Block tail: gamma * x, DropPath, residual, twig/model/cod.py:1112-1116;
//                                                       Block residuals x + DropPath(.) cod.py:958-959; s = per-sample stochastic-depth scale)
//   its backward:   dy = s[b] * gamma[c] * g ,  dgamma[c] = sum_r s[b] * g * y      (dx = g is the identity, no kernel)
//   colsum:         out[c] = sum_r x[r, c]             (bias gradients of the Linear layers: one pass instead of a generic reduce)
// HBM-bound: fwd 3e*rows*C, bwd 3e*rows*C, colsum e*rows*C.  Lanes run along C in 16-byte chunks; column partials stay in
// registers across the row loop, meet in LDS per workgroup, and a deterministic second stage sums the per-workgroup rows.
#include "common.h"

namespace {

constexpr int EW_MAX_BLOCKS = 512;

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
                                                     int64_t rows, int C, int64_t rows_per_sample) {
  typedef typename Vec16<T>::type VT;
  constexpr int V = Vec16<T>::N;
  extern __shared__ float red[];                 // [RG][CVB * V]
  const int CV = C / V;
  const int CVB = min(CV - (int)blockIdx.y * 256, 256);       // chunk-columns handled by this workgroup
  const int RG = 256 / CVB;                                   // row groups inside the workgroup (CVB is a power of two or <= 256)
  const int tid = threadIdx.x;
  const int cvl = tid % CVB, rg = tid / CVB;
  const int cv = blockIdx.y * 256 + cvl;
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
  for (int i = tid; i < CVB * V; i += 256) {
    float t = 0.f;
    for (int k = 0; k < RG; ++k) t += red[k * CVB * V + i];
    ws[(size_t)blockIdx.x * C + (size_t)blockIdx.y * 256 * V + i] = t;
  }
}

// ---- multi-reduce: the second stage of every two-stage column reduction of the library ------------------------------------------
// One workgroup = 32 columns x 8 row groups of ONE entry (a narrow tile: the partial buffers are small and the work latency-bound);
// row group g sums rows g, g+8, ... with 8 loads in flight, the groups meet in LDS.  Fixed summation order.  The entry table travels
// by value in the kernel arguments (hipGraph-safe).
constexpr int MR_MAX = 48;
struct MrTable {
  const float* ws[MR_MAX];
  float* outA[MR_MAX];
  void* outB[MR_MAX];
  void* outC[MR_MAX];
  int nblocks[MR_MAX], ncols[MR_MAX], nA[MR_MAX], dtB[MR_MAX], tr_rows[MR_MAX], tr_cols[MR_MAX];
  int first_tile[MR_MAX + 1];
  int count;
};

__device__ __forceinline__ void mr_store(void* out, int dt, int i, float v) {
  if (dt == DGTD_BF16) ((bf16_t*)out)[i] = (bf16_t)v;
  else if (dt == DGTD_F16) ((f16_t*)out)[i] = (f16_t)v;
  else ((float*)out)[i] = v;
}

__global__ __launch_bounds__(256) void multi_reduce_kernel(MrTable t) {
  __shared__ float part[8][32];
  const int tile = blockIdx.x;
  int lo = 0, hi = t.count;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (t.first_tile[mid] <= tile) lo = mid; else hi = mid; }
  const int e = lo;
  const int lane = threadIdx.x & 31, rgp = threadIdx.x >> 5;
  const int col = (tile - t.first_tile[e]) * 32 + lane;
  const int ncols = t.ncols[e], nblocks = t.nblocks[e], nA = t.nA[e];
  const int trn = t.tr_rows[e] * t.tr_cols[e];          // columns [nA, nA + trn) are the transposed block, the rest goes to outC
  const bool live = col < ncols && (col >= nA || t.outA[e] != nullptr) && (trn == 0 || col < nA + trn || t.outC[e] != nullptr);
  float s0 = 0.f, s1 = 0.f;
  if (live) {
    const float* p = t.ws[e] + col;
    const size_t rs = (size_t)ncols;
    int b = rgp;
    for (; b + 56 < nblocks; b += 64) {
      const float v0 = p[(size_t)b * rs], v1 = p[(size_t)(b + 8) * rs], v2 = p[(size_t)(b + 16) * rs], v3 = p[(size_t)(b + 24) * rs];
      const float v4 = p[(size_t)(b + 32) * rs], v5 = p[(size_t)(b + 40) * rs], v6 = p[(size_t)(b + 48) * rs], v7 = p[(size_t)(b + 56) * rs];
      s0 += ((v0 + v1) + (v2 + v3));
      s1 += ((v4 + v5) + (v6 + v7));
    }
    for (; b < nblocks; b += 8) s0 += p[(size_t)b * rs];
  }
  part[rgp][lane] = s0 + s1;
  __syncthreads();
  if (rgp == 0 && live) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) v += part[k][lane];
    if (col < nA) t.outA[e][col] = v;
    else if (trn == 0) mr_store(t.outB[e], t.dtB[e], col - nA, v);
    else if (col - nA < trn) { const int j = col - nA, tt = j / t.tr_cols[e], c = j % t.tr_cols[e]; mr_store(t.outB[e], t.dtB[e], c * t.tr_rows[e] + tt, v); }
    else mr_store(t.outC[e], t.dtB[e], col - nA - trn, v);
  }
}

template <typename T>
int check_c(int C, const char* who) {
  constexpr int V = Vec16<T>::N;
  DGTD_REQUIRE(C % V == 0, "%s: C=%d must be a multiple of %d", who, C, V);
  return 0;
}

// first stage: ws [*nblocks][C] partial rows (MODE 2 writes none); `out` (dtype code out_dt) is reduced here unless nblocks != NULL
template <typename T, int MODE>
int colsum_launch(const void* g, const void* y, const float* s, const float* gamma, void* dy, void* out, void* ws, int64_t rows,
                  int C, int64_t rps, hipStream_t st, const char* who, int out_dt = 0, int* nblocks = nullptr) {
  if (int rc = check_c<T>(C, who)) return rc;
  constexpr int V = Vec16<T>::N;
  const int CV = C / V, ncb = (int)cdiv(CV, 256);
  const int cvb0 = std::min(CV, 256), rg = 256 / cvb0;
  int gx = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(rows, (int64_t)rg * 4), EW_MAX_BLOCKS));
  const size_t lds = (size_t)256 * V * sizeof(float);
  hipLaunchKernelGGL((colsum_kernel<T, MODE>), dim3(gx, ncb), dim3(256), lds, st, (const T*)g, (const T*)y, s, gamma, (T*)dy, (float*)ws, rows, C, rps);
  DGTD_CHECK_LAUNCH(who);
  if (nblocks) { *nblocks = gx; return 0; }
  if (MODE != 2) {
    const dgtd_reduce_entry e{(const float*)ws, gx, C, out_dt == DGTD_F32 ? (float*)out : nullptr, out_dt == DGTD_F32 ? C : 0,
                              out_dt == DGTD_F32 ? nullptr : out, out_dt, 0, 0, nullptr};
    return dgtd_multi_reduce_impl(&e, 1, st);
  }
  return 0;
}


// Two column sums in one pass, for the backward of a Linear fused with its consumer (the bias gradient of the Linear is the column
// sum of the gradient this kernel produces anyway):
//   MODE 3 (residual epilogue):  dy = s*gamma*g,  A = sum_r s*g*y (dgamma, only with gamma),  Bsum = sum_r dy   (bias gradient)
//   MODE 4 (GELU):               dy = g * gelu'(y)  (y = pre-activation),                     Bsum = sum_r dy
// ws [nblocks][2C] = { A | Bsum } per workgroup, reduced by colsum2_reduce_kernel.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void colsum2_kernel(const T* __restrict__ g, const T* __restrict__ y, const float* __restrict__ s,
                                                      const float* __restrict__ gamma, T* __restrict__ dy, float* __restrict__ ws,
                                                      int64_t rows, int C, int64_t rows_per_sample) {
  typedef typename Vec16<T>::type VT;
  constexpr int V = Vec16<T>::N;
  extern __shared__ float red[];                 // [RG][CVB * V]
  const int CV = C / V;
  const int CVB = min(CV - (int)blockIdx.y * 256, 256);
  const int RG = 256 / CVB;
  const int tid = threadIdx.x;
  const int cvl = tid % CVB, rg = tid / CVB;
  const int cv = blockIdx.y * 256 + cvl;
  float accA[V], accB[V], gm[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { accA[j] = 0.f; accB[j] = 0.f; gm[j] = (MODE == 3 && gamma) ? gamma[cv * V + j] : 1.f; }
  const bool need_y = MODE == 4 || gamma != nullptr;
  if (rg < RG) {
    // U rows in flight per thread (loads of all U rows issued before the first is consumed): one row per iteration leaves two
    // 16-byte loads outstanding per lane, far too little for HBM
    constexpr int U = 4;
    const int64_t stride = (int64_t)gridDim.x * RG;
    for (int64_t r0 = (int64_t)blockIdx.x * RG + rg; r0 < rows; r0 += stride * U) {
      VT gv[U], yv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t r = r0 + u * stride;
        if (r < rows) {
          const size_t off = (size_t)r * C + (size_t)cv * V;
          gv[u] = *reinterpret_cast<const VT*>(g + off);
          yv[u] = need_y ? *reinterpret_cast<const VT*>(y + off) : gv[u];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t r = r0 + u * stride;
        if (r >= rows) break;
        VT o;
        if (MODE == 3) {
          const float sb = s ? s[r / rows_per_sample] : 1.f;
#pragma unroll
          for (int j = 0; j < V; ++j) {
            const float gg = sb * (float)gv[u][j];
            accA[j] += gg * (float)yv[u][j];
            const float d = gg * gm[j];
            accB[j] += d;
            o[j] = (T)d;
          }
        } else {
#pragma unroll
          for (int j = 0; j < V; ++j) {
            const float d = (float)gv[u][j] * gelu_grad_fast((float)yv[u][j]);
            accB[j] += d;
            o[j] = (T)d;
          }
        }
        *reinterpret_cast<VT*>(dy + (size_t)r * C + (size_t)cv * V) = o;
      }
    }
  }
  float* wrow = ws + (size_t)blockIdx.x * 2 * C + (size_t)blockIdx.y * 256 * V;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 0 && !(MODE == 3 && gamma)) continue;      // uniform
    __syncthreads();
    if (rg < RG)
#pragma unroll
      for (int j = 0; j < V; ++j) red[(rg * CVB + cvl) * V + j] = pass == 0 ? accA[j] : accB[j];
    __syncthreads();
    for (int i = tid; i < CVB * V; i += 256) {
      float t = 0.f;
      for (int k = 0; k < RG; ++k) t += red[k * CVB * V + i];
      wrow[pass * C + i] = t;
    }
  }
}

template <typename T, int MODE>
int colsum2_launch(const void* g, const void* y, const float* s, const float* gamma, void* dy, float* outA, void* outB, int outB_bf16,
                   void* ws, int64_t rows, int C, int64_t rps, hipStream_t st, const char* who, int* nblocks = nullptr) {
  if (int rc = check_c<T>(C, who)) return rc;
  constexpr int V = Vec16<T>::N;
  const int CV = C / V, ncb = (int)cdiv(CV, 256);
  const int cvb0 = std::min(CV, 256), rg = 256 / cvb0;
  const int gx = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(rows, (int64_t)rg * 4), EW_MAX_BLOCKS));
  const size_t lds = (size_t)256 * V * sizeof(float);
  hipLaunchKernelGGL((colsum2_kernel<T, MODE>), dim3(gx, ncb), dim3(256), lds, st, (const T*)g, (const T*)y, s, gamma, (T*)dy, (float*)ws, rows, C, rps);
  DGTD_CHECK_LAUNCH(who);
  if (nblocks) { *nblocks = gx; return 0; }
  // columns [0, C) = dgamma partials (garbage when the kernel had no gamma: then outA is NULL and they are skipped), [C, 2C) = dbias
  const dgtd_reduce_entry e{(const float*)ws, gx, 2 * C, outA, C, outB, outB_bf16, 0, 0, nullptr};
  return dgtd_multi_reduce_impl(&e, 1, st);
}

}  // namespace

int dgtd_multi_reduce_impl(const dgtd_reduce_entry* entries, int n, hipStream_t st) {
  for (int b0 = 0; b0 < n; b0 += MR_MAX) {
    MrTable t;
    const int m = std::min(MR_MAX, n - b0);
    t.count = m;
    int tiles = 0;
    for (int i = 0; i < m; ++i) {
      const dgtd_reduce_entry& e = entries[b0 + i];
      DGTD_REQUIRE(e.ws && e.nblocks > 0 && e.ncols > 0 && e.nA >= 0 && e.nA <= e.ncols && (e.nA == e.ncols || e.outB), "multi_reduce: bad entry %d", b0 + i);
      DGTD_REQUIRE(e.tr_rows >= 0 && e.tr_cols >= 0 && e.nA + e.tr_rows * e.tr_cols <= e.ncols, "multi_reduce: bad transpose block in entry %d", b0 + i);
      t.ws[i] = e.ws; t.outA[i] = e.outA; t.outB[i] = e.outB; t.outC[i] = e.outC; t.nblocks[i] = e.nblocks; t.ncols[i] = e.ncols; t.nA[i] = e.nA;
      t.dtB[i] = e.dtB; t.tr_rows[i] = e.tr_rows; t.tr_cols[i] = e.tr_cols;
      t.first_tile[i] = tiles;
      tiles += (int)cdiv(e.ncols, 32);
    }
    for (int i = m; i <= MR_MAX; ++i) t.first_tile[i] = tiles;
    hipLaunchKernelGGL(multi_reduce_kernel, dim3(tiles), dim3(256), 0, st, t);
    DGTD_CHECK_LAUNCH("multi_reduce");
  }
  return 0;
}

extern "C" int dgtd_multi_reduce(const dgtd_reduce_entry* entries, int n, dgtd_stream s) {
  DGTD_REQUIRE(n >= 0 && (n == 0 || entries), "multi_reduce: bad arguments");
  if (n == 0) return 0;
  double bytes = 0;
  for (int i = 0; i < n; ++i) bytes += 4.0 * entries[i].nblocks * entries[i].ncols;
  DGTD_PROF(s, DGTD_HBM, bytes, "dgtd_multi_reduce[n=%d]", n);
  return dgtd_multi_reduce_impl(entries, n, (hipStream_t)s);
}

extern "C" int dgtd_scale_residual_bias_bwd_partial(const void* g, const void* y, const float* s, const float* gamma, void* dy, void* workspace,
                                                    int64_t rows, int C, int64_t rows_per_sample, dgtd_dtype dt, int* nblocks, dgtd_stream st) {
  DGTD_REQUIRE(rows > 0 && C > 0 && rows_per_sample > 0 && nblocks, "scale_residual_bias_bwd_partial: bad sizes");
  DGTD_PROF(st, DGTD_HBM, (gamma ? 3.0 : 2.0) * dgtd_esize(dt) * rows * C, "dgtd_scale_residual_bias_bwd[rows=%lld,C=%d]", (long long)rows, C);
  hipStream_t h = (hipStream_t)st;
  if (dt == DGTD_F16) return colsum2_launch<f16_t, 3>(g, y, s, gamma, dy, nullptr, nullptr, 0, workspace, rows, C, rows_per_sample, h, "scale_residual_bias_bwd", nblocks);
  if (dt == DGTD_BF16) return colsum2_launch<bf16_t, 3>(g, y, s, gamma, dy, nullptr, nullptr, 0, workspace, rows, C, rows_per_sample, h, "scale_residual_bias_bwd", nblocks);
  if (dt == DGTD_F32) return colsum2_launch<float, 3>(g, y, s, gamma, dy, nullptr, nullptr, 0, workspace, rows, C, rows_per_sample, h, "scale_residual_bias_bwd", nblocks);
  DGTD_FAIL(2, "scale_residual_bias_bwd_partial: bad dtype %d", (int)dt);
}

extern "C" int dgtd_gelu_bias_bwd_partial(const void* g, const void* pre, void* dpre, void* workspace, int64_t rows, int C, dgtd_dtype dt,
                                          int* nblocks, dgtd_stream st) {
  DGTD_REQUIRE(rows > 0 && C > 0 && nblocks, "gelu_bias_bwd_partial: bad sizes");
  DGTD_PROF(st, DGTD_HBM, 3.0 * dgtd_esize(dt) * rows * C, "dgtd_gelu_bias_bwd[rows=%lld,C=%d]", (long long)rows, C);
  hipStream_t h = (hipStream_t)st;
  if (dt == DGTD_F16) return colsum2_launch<f16_t, 4>(g, pre, nullptr, nullptr, dpre, nullptr, nullptr, 0, workspace, rows, C, 1, h, "gelu_bias_bwd", nblocks);
  if (dt == DGTD_BF16) return colsum2_launch<bf16_t, 4>(g, pre, nullptr, nullptr, dpre, nullptr, nullptr, 0, workspace, rows, C, 1, h, "gelu_bias_bwd", nblocks);
  if (dt == DGTD_F32) return colsum2_launch<float, 4>(g, pre, nullptr, nullptr, dpre, nullptr, nullptr, 0, workspace, rows, C, 1, h, "gelu_bias_bwd", nblocks);
  DGTD_FAIL(2, "gelu_bias_bwd_partial: bad dtype %d", (int)dt);
}

extern "C" int dgtd_colsum_partial(const void* x, void* workspace, int64_t rows, int C, dgtd_dtype dt, int* nblocks, dgtd_stream st) {
  DGTD_REQUIRE(rows > 0 && C > 0 && nblocks, "colsum_partial: bad sizes");
  DGTD_PROF(st, DGTD_HBM, 1.0 * dgtd_esize(dt) * rows * C, "dgtd_colsum[rows=%lld,C=%d]", (long long)rows, C);
  if (dt == DGTD_F16) return colsum_launch<f16_t, 0>(x, nullptr, nullptr, nullptr, nullptr, nullptr, workspace, rows, C, 1, (hipStream_t)st, "colsum", 0, nblocks);
  if (dt == DGTD_BF16) return colsum_launch<bf16_t, 0>(x, nullptr, nullptr, nullptr, nullptr, nullptr, workspace, rows, C, 1, (hipStream_t)st, "colsum", 0, nblocks);
  if (dt == DGTD_F32) return colsum_launch<float, 0>(x, nullptr, nullptr, nullptr, nullptr, nullptr, workspace, rows, C, 1, (hipStream_t)st, "colsum", 0, nblocks);
  DGTD_FAIL(2, "colsum_partial: bad dtype %d", (int)dt);
}

extern "C" int dgtd_scale_residual_fwd(const void* x, const void* y, const float* s, const float* gamma, void* out, int64_t rows,
                                       int C, int64_t rows_per_sample, dgtd_dtype dt, dgtd_stream st) {
  DGTD_PROF(st, DGTD_HBM, 3.0 * dgtd_esize(dt) * rows * C, "dgtd_scale_residual_fwd[rows=%lld,C=%d]", (long long)rows, C);
  DGTD_REQUIRE(rows > 0 && C > 0 && rows_per_sample > 0, "scale_residual_fwd: bad sizes");
  const int V = DGTD_IS_HALF(dt) ? 8 : 4;
  DGTD_REQUIRE(C % V == 0, "scale_residual_fwd: C=%d must be a multiple of %d", C, V);
  const int grid = (int)std::min<int64_t>(cdiv(rows * (C / V), 256), 256 * 16);
  if (dt == DGTD_BF16) hipLaunchKernelGGL(scale_residual_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)st, (const bf16_t*)x, (const bf16_t*)y, s, gamma, (bf16_t*)out, rows, C, rows_per_sample);
  else if (dt == DGTD_F16) hipLaunchKernelGGL(scale_residual_fwd_kernel<f16_t>, dim3(grid), dim3(256), 0, (hipStream_t)st, (const f16_t*)x, (const f16_t*)y, s, gamma, (f16_t*)out, rows, C, rows_per_sample);
  else if (dt == DGTD_F32) hipLaunchKernelGGL(scale_residual_fwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)st, (const float*)x, (const float*)y, s, gamma, (float*)out, rows, C, rows_per_sample);
  else DGTD_FAIL(2, "scale_residual_fwd: bad dtype %d", (int)dt);
  DGTD_CHECK_LAUNCH("scale_residual_fwd");
  return 0;
}

extern "C" int64_t dgtd_colsum_workspace(int C) { return (int64_t)EW_MAX_BLOCKS * C * sizeof(float); }

extern "C" int dgtd_scale_residual_bwd(const void* g, const void* y, const float* s, const float* gamma, void* dy, float* dgamma,
                                       void* workspace, int64_t rows, int C, int64_t rows_per_sample, dgtd_dtype dt, dgtd_stream st) {
  DGTD_PROF(st, DGTD_HBM, (gamma ? 3.0 : 2.0) * dgtd_esize(dt) * rows * C, "dgtd_scale_residual_bwd[rows=%lld,C=%d]", (long long)rows, C);
  DGTD_REQUIRE(rows > 0 && C > 0 && rows_per_sample > 0, "scale_residual_bwd: bad sizes");
  DGTD_REQUIRE((gamma == nullptr) == (dgamma == nullptr), "scale_residual_bwd: gamma and dgamma go together");
  hipStream_t h = (hipStream_t)st;
  if (dt == DGTD_BF16) return gamma ? colsum_launch<bf16_t, 1>(g, y, s, gamma, dy, dgamma, workspace, rows, C, rows_per_sample, h, "scale_residual_bwd")
                                    : colsum_launch<bf16_t, 2>(g, y, s, gamma, dy, dgamma, workspace, rows, C, rows_per_sample, h, "scale_residual_bwd");
  if (dt == DGTD_F16) return gamma ? colsum_launch<f16_t, 1>(g, y, s, gamma, dy, dgamma, workspace, rows, C, rows_per_sample, h, "scale_residual_bwd")
                                   : colsum_launch<f16_t, 2>(g, y, s, gamma, dy, dgamma, workspace, rows, C, rows_per_sample, h, "scale_residual_bwd");
  if (dt == DGTD_F32) return gamma ? colsum_launch<float, 1>(g, y, s, gamma, dy, dgamma, workspace, rows, C, rows_per_sample, h, "scale_residual_bwd")
                                   : colsum_launch<float, 2>(g, y, s, gamma, dy, dgamma, workspace, rows, C, rows_per_sample, h, "scale_residual_bwd");
  DGTD_FAIL(2, "scale_residual_bwd: bad dtype %d", (int)dt);
}

extern "C" int dgtd_colsum(const void* x, void* out, dgtd_dtype out_dt, void* workspace, int64_t rows, int C, dgtd_dtype dt, dgtd_stream st) {
  DGTD_PROF(st, DGTD_HBM, 1.0 * dgtd_esize(dt) * rows * C, "dgtd_colsum[rows=%lld,C=%d]", (long long)rows, C);
  DGTD_REQUIRE(rows > 0 && C > 0, "colsum: bad sizes");
  DGTD_REQUIRE(out_dt == DGTD_F32 || DGTD_IS_HALF(out_dt), "colsum: bad output dtype %d", (int)out_dt);
  const int ob = (int)out_dt;
  if (dt == DGTD_BF16) return colsum_launch<bf16_t, 0>(x, nullptr, nullptr, nullptr, nullptr, out, workspace, rows, C, 1, (hipStream_t)st, "colsum", ob);
  if (dt == DGTD_F16) return colsum_launch<f16_t, 0>(x, nullptr, nullptr, nullptr, nullptr, out, workspace, rows, C, 1, (hipStream_t)st, "colsum", ob);
  if (dt == DGTD_F32) return colsum_launch<float, 0>(x, nullptr, nullptr, nullptr, nullptr, out, workspace, rows, C, 1, (hipStream_t)st, "colsum", ob);
  DGTD_FAIL(2, "colsum: bad dtype %d", (int)dt);
}

extern "C" int64_t dgtd_colsum2_workspace(int C) { return (int64_t)EW_MAX_BLOCKS * 2 * C * sizeof(float); }

extern "C" int dgtd_scale_residual_bias_bwd(const void* g, const void* y, const float* s, const float* gamma, void* dy, float* dgamma,
                                            void* dbias, dgtd_dtype bias_dt, void* workspace, int64_t rows, int C,
                                            int64_t rows_per_sample, dgtd_dtype dt, dgtd_stream st) {
  DGTD_PROF(st, DGTD_HBM, (gamma ? 3.0 : 2.0) * dgtd_esize(dt) * rows * C, "dgtd_scale_residual_bias_bwd[rows=%lld,C=%d]", (long long)rows, C);
  DGTD_REQUIRE(rows > 0 && C > 0 && rows_per_sample > 0 && dbias, "scale_residual_bias_bwd: bad sizes");
  DGTD_REQUIRE((gamma == nullptr) == (dgamma == nullptr), "scale_residual_bias_bwd: gamma and dgamma go together");
  DGTD_REQUIRE(bias_dt == DGTD_F32 || DGTD_IS_HALF(bias_dt), "scale_residual_bias_bwd: bad bias dtype %d", (int)bias_dt);
  hipStream_t h = (hipStream_t)st;
  const int ob = (int)bias_dt;
  if (dt == DGTD_F16) return colsum2_launch<f16_t, 3>(g, y, s, gamma, dy, dgamma, dbias, ob, workspace, rows, C, rows_per_sample, h, "scale_residual_bias_bwd");
  if (dt == DGTD_BF16) return colsum2_launch<bf16_t, 3>(g, y, s, gamma, dy, dgamma, dbias, ob, workspace, rows, C, rows_per_sample, h, "scale_residual_bias_bwd");
  if (dt == DGTD_F32) return colsum2_launch<float, 3>(g, y, s, gamma, dy, dgamma, dbias, ob, workspace, rows, C, rows_per_sample, h, "scale_residual_bias_bwd");
  DGTD_FAIL(2, "scale_residual_bias_bwd: bad dtype %d", (int)dt);
}

extern "C" int dgtd_gelu_bias_bwd(const void* g, const void* pre, void* dpre, void* dbias, dgtd_dtype bias_dt, void* workspace,
                                  int64_t rows, int C, dgtd_dtype dt, dgtd_stream st) {
  DGTD_PROF(st, DGTD_HBM, 3.0 * dgtd_esize(dt) * rows * C, "dgtd_gelu_bias_bwd[rows=%lld,C=%d]", (long long)rows, C);
  DGTD_REQUIRE(rows > 0 && C > 0 && dbias, "gelu_bias_bwd: bad sizes");
  DGTD_REQUIRE(bias_dt == DGTD_F32 || DGTD_IS_HALF(bias_dt), "gelu_bias_bwd: bad bias dtype %d", (int)bias_dt);
  hipStream_t h = (hipStream_t)st;
  const int ob = (int)bias_dt;
  if (dt == DGTD_F16) return colsum2_launch<f16_t, 4>(g, pre, nullptr, nullptr, dpre, nullptr, dbias, ob, workspace, rows, C, 1, h, "gelu_bias_bwd");
  if (dt == DGTD_BF16) return colsum2_launch<bf16_t, 4>(g, pre, nullptr, nullptr, dpre, nullptr, dbias, ob, workspace, rows, C, 1, h, "gelu_bias_bwd");
  if (dt == DGTD_F32) return colsum2_launch<float, 4>(g, pre, nullptr, nullptr, dpre, nullptr, dbias, ob, workspace, rows, C, 1, h, "gelu_bias_bwd");
  DGTD_FAIL(2, "gelu_bias_bwd: bad dtype %d", (int)dt);
}
