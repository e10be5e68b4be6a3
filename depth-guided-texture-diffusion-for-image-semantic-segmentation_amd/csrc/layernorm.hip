// layernorm.hip — LayerNorm over the channel dim of a [rows, C] token matrix (fwd + bwd).
// Replaces nn.LayerNorm / F.layer_norm of twig/model/cod.py:979, 881, 929, 936, 1367-1391, 1043.
//
// HBM-bound: algorithmic bytes fwd = 2*e*rows*C, bwd = 3*e*rows*C (e = element size).
// Mapping: a row is cut into 16-byte chunks; a power-of-two group of G lanes owns one row
// (64/G rows per wave), every lane keeps its chunks in registers, moments are reduced with
// wave shuffles inside the group.  No LDS in the forward.
#include "common.h"

namespace {

constexpr int LN_MAX_NPL = 4;      // chunks per lane: covers C <= 1024 (fp32) / 2048 (bf16)
constexpr int LN_BWD_MAX_GRID = 1024;

template <typename T>
struct RowRegs {
  static constexpr int V = Vec16<T>::N;
  float v[LN_MAX_NPL][V];
};

template <typename T, int NPL>
__device__ __forceinline__ void load_row(const T* __restrict__ p, int cpr, int gl, int G, float (&v)[NPL][Vec16<T>::N]) {
  typedef typename Vec16<T>::type VT;
  constexpr int V = Vec16<T>::N;
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    int c = gl + i * G;
    if (c < cpr) {
      VT t = *reinterpret_cast<const VT*>(p + (size_t)c * V);
#pragma unroll
      for (int j = 0; j < V; ++j) v[i][j] = (float)t[j];
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j) v[i][j] = 0.f;
    }
  }
}

template <typename T, int NPL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd,
                                                     int64_t rows, int C, float eps, int G) {
  typedef typename Vec16<T>::type VT;
  constexpr int V = Vec16<T>::N;
  const int cpr = C / V;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gl = lane & (G - 1), gi = lane / G, rpw = 64 / G;
  const int64_t rows_per_block = (int64_t)rpw * 4;
  const float invC = 1.f / (float)C;
  // per-lane gamma/beta (column mapping is row-independent)
  float gm[NPL][V], bt[NPL][V];
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    int c = gl + i * G;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      gm[i][j] = (c < cpr) ? gamma[c * V + j] : 0.f;
      bt[i][j] = (c < cpr) ? beta[c * V + j] : 0.f;
    }
  }
  constexpr int U = (NPL <= 2) ? 4 : 2;   // rows in flight per lane group: all row loads are issued before the first reduction
  for (int64_t r0 = (int64_t)blockIdx.x * rows_per_block * U; r0 < rows; r0 += (int64_t)gridDim.x * rows_per_block * U) {
    float v[U][NPL][V];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = r0 + (int64_t)u * rows_per_block + wave * rpw + gi;
      ok[u] = row < rows;
      if (ok[u]) load_row<T, NPL>(x + row * C, cpr, gl, G, v[u]);
      else {
#pragma unroll
        for (int i = 0; i < NPL; ++i)
#pragma unroll
          for (int j = 0; j < V; ++j) v[u][i][j] = 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = r0 + (int64_t)u * rows_per_block + wave * rpw + gi;
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NPL; ++i)
#pragma unroll
        for (int j = 0; j < V; ++j) s += v[u][i][j];
      const float mu = group_sum(s, G) * invC;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        bool in = (gl + i * G) < cpr;
#pragma unroll
        for (int j = 0; j < V; ++j) { float d = in ? v[u][i][j] - mu : 0.f; q += d * d; }
      }
      const float rs = rsqrtf(group_sum(q, G) * invC + eps);
      if (ok[u]) {
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
          int c = gl + i * G;
          if (c < cpr) {
            VT o;
#pragma unroll
            for (int j = 0; j < V; ++j) o[j] = (T)((v[u][i][j] - mu) * rs * gm[i][j] + bt[i][j]);
            *reinterpret_cast<VT*>(y + row * C + (size_t)c * V) = o;
          }
        }
        if (gl == 0) { mean[row] = mu; rstd[row] = rs; }
      }
    }
  }
}

// Backward: dx per row; per-block partial dgamma/dbeta written to ws[block][2][C].
template <typename T, int NPL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const T* __restrict__ dres, T* __restrict__ dx,
                                                     float* __restrict__ ws, int64_t rows, int C, int G) {
  typedef typename Vec16<T>::type VT;
  constexpr int V = Vec16<T>::N;
  extern __shared__ float red[];  // [4 waves][2][C]
  const int cpr = C / V;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gl = lane & (G - 1), gi = lane / G, rpw = 64 / G;
  const int64_t rows_per_block = (int64_t)rpw * 4;
  const float invC = 1.f / (float)C;
  float gm[NPL][V], dg[NPL][V], db[NPL][V];
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    int c = gl + i * G;
#pragma unroll
    for (int j = 0; j < V; ++j) { gm[i][j] = (c < cpr) ? gamma[c * V + j] : 0.f; dg[i][j] = 0.f; db[i][j] = 0.f; }
  }
  constexpr int U = (NPL <= 2) ? 4 : 2;   // rows in flight per lane group: U independent row loads are issued before any reduction
  for (int64_t r0 = (int64_t)blockIdx.x * rows_per_block * U; r0 < rows; r0 += (int64_t)gridDim.x * rows_per_block * U) {
    float xv[U][NPL][V], gv[U][NPL][V], mu[U], rs[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = r0 + (int64_t)u * rows_per_block + wave * rpw + gi;
      ok[u] = row < rows;
      if (ok[u]) {
        load_row<T, NPL>(x + row * C, cpr, gl, G, xv[u]);
        load_row<T, NPL>(dy + row * C, cpr, gl, G, gv[u]);
        mu[u] = mean[row]; rs[u] = rstd[row];
      } else {
        mu[u] = 0.f; rs[u] = 0.f;
#pragma unroll
        for (int i = 0; i < NPL; ++i)
#pragma unroll
          for (int j = 0; j < V; ++j) { xv[u][i][j] = 0.f; gv[u][i][j] = 0.f; }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        bool in = (gl + i * G) < cpr;
#pragma unroll
        for (int j = 0; j < V; ++j) {
          float xh = in ? (xv[u][i][j] - mu[u]) * rs[u] : 0.f;
          float g = gv[u][i][j];
          dg[i][j] += g * xh;
          db[i][j] += g;
          g *= gm[i][j];
          a += g; b += g * xh;
          xv[u][i][j] = xh; gv[u][i][j] = g;
        }
      }
      a = group_sum(a, G) * invC;
      b = group_sum(b, G) * invC;
      if (ok[u]) {
        const int64_t row = r0 + (int64_t)u * rows_per_block + wave * rpw + gi;
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
          int c = gl + i * G;
          if (c < cpr) {
            VT o, r;
            if (dres) r = *reinterpret_cast<const VT*>(dres + row * C + (size_t)c * V);   // gradient of the residual branch that forks off x
#pragma unroll
            for (int j = 0; j < V; ++j) o[j] = (T)(rs[u] * (gv[u][i][j] - a - xv[u][i][j] * b) + (dres ? (float)r[j] : 0.f));
            *reinterpret_cast<VT*>(dx + row * C + (size_t)c * V) = o;
          }
        }
      }
    }
  }
  // combine the 64/G row-groups of the wave (same column set), then the 4 waves through LDS
#pragma unroll
  for (int i = 0; i < NPL; ++i)
#pragma unroll
    for (int j = 0; j < V; ++j)
      for (int o = G; o < 64; o <<= 1) { dg[i][j] += __shfl_xor(dg[i][j], o, 64); db[i][j] += __shfl_xor(db[i][j], o, 64); }
  if (gi == 0) {
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      int c = gl + i * G;
      if (c < cpr)
#pragma unroll
        for (int j = 0; j < V; ++j) { red[(wave * 2 + 0) * C + c * V + j] = dg[i][j]; red[(wave * 2 + 1) * C + c * V + j] = db[i][j]; }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * C; c += 256) {
    int which = c / C, col = c % C;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) s += red[(w * 2 + which) * C + col];
    ws[((size_t)blockIdx.x * 2 + which) * C + col] = s;
  }
}

struct LnGeom { int G, npl; int64_t rows_per_block; };
template <typename T> LnGeom ln_geom(int C) {
  int cpr = C / Vec16<T>::N;
  int G = 1;
  while (G < cpr && G < 64) G <<= 1;
  LnGeom g; g.G = G; g.npl = (cpr + G - 1) / G; g.rows_per_block = (64 / G) * 4;
  return g;
}

template <typename T>
int ln_fwd_launch(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                  int64_t rows, int C, float eps, hipStream_t s) {
  DGTD_REQUIRE(C % Vec16<T>::N == 0, "layernorm: C=%d must be a multiple of %d", C, Vec16<T>::N);
  LnGeom g = ln_geom<T>(C);
  DGTD_REQUIRE(g.npl <= LN_MAX_NPL, "layernorm: C=%d too large", C);
  if (rows == 0) return 0;
  const int U = g.npl <= 2 ? 4 : 2;     // must match ln_fwd_kernel
  int grid = (int)std::min<int64_t>(cdiv(rows, g.rows_per_block * U), 256 * 8);
#define LN_FWD(NPL) hipLaunchKernelGGL((ln_fwd_kernel<T, NPL>), dim3(grid), dim3(256), 0, s, (const T*)x, gamma, beta, (T*)y, mean, rstd, rows, C, eps, g.G)
  switch (g.npl) { case 1: LN_FWD(1); break; case 2: LN_FWD(2); break; case 3: LN_FWD(3); break; default: LN_FWD(4); }
#undef LN_FWD
  DGTD_CHECK_LAUNCH("layernorm_fwd");
  return 0;
}

template <typename T>
int ln_bwd_launch(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx,
                  float* dgamma, float* dbeta, void* ws, int64_t rows, int C, hipStream_t s, const void* dres = nullptr, int* nblocks = nullptr) {
  DGTD_REQUIRE(C % Vec16<T>::N == 0, "layernorm: C=%d must be a multiple of %d", C, Vec16<T>::N);
  LnGeom g = ln_geom<T>(C);
  DGTD_REQUIRE(g.npl <= LN_MAX_NPL, "layernorm: C=%d too large", C);
  DGTD_REQUIRE(rows > 0, "layernorm_bwd: rows must be > 0");
  int grid = (int)std::min<int64_t>(cdiv(rows, g.rows_per_block * 4), LN_BWD_MAX_GRID);
  size_t lds = (size_t)4 * 2 * C * sizeof(float);
#define LN_BWD(NPL) hipLaunchKernelGGL((ln_bwd_kernel<T, NPL>), dim3(grid), dim3(256), lds, s, (const T*)dy, (const T*)x, gamma, mean, rstd, (const T*)dres, (T*)dx, (float*)ws, rows, C, g.G)
  switch (g.npl) { case 1: LN_BWD(1); break; case 2: LN_BWD(2); break; case 3: LN_BWD(3); break; default: LN_BWD(4); }
#undef LN_BWD
  DGTD_CHECK_LAUNCH("layernorm_bwd");
  if (nblocks) { *nblocks = grid; return 0; }
  const dgtd_reduce_entry e{(const float*)ws, grid, 2 * C, dgamma, C, dbeta, DGTD_F32, 0, 0, nullptr};     // ws [grid][2C] = { dgamma | dbeta } partial rows
  return dgtd_multi_reduce_impl(&e, 1, s);
}

}  // namespace

extern "C" int dgtd_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                  int64_t rows, int C, float eps, dgtd_dtype dt, dgtd_stream s) {
  DGTD_PROF(s, DGTD_HBM, 2.0 * dgtd_esize(dt) * rows * C, "dgtd_layernorm_fwd[rows=%lld,C=%d]", (long long)rows, C);
  if (dt == DGTD_F32) return ln_fwd_launch<float>(x, gamma, beta, y, mean, rstd, rows, C, eps, (hipStream_t)s);
  if (dt == DGTD_BF16) return ln_fwd_launch<bf16_t>(x, gamma, beta, y, mean, rstd, rows, C, eps, (hipStream_t)s);
  if (dt == DGTD_F16) return ln_fwd_launch<f16_t>(x, gamma, beta, y, mean, rstd, rows, C, eps, (hipStream_t)s);
  DGTD_FAIL(2, "layernorm_fwd: bad dtype %d", (int)dt);
}

extern "C" int dgtd_layernorm_bwd_add(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                      const void* dx_add, void* dx, float* dgamma, float* dbeta, void* workspace, int64_t rows, int C,
                                      dgtd_dtype dt, dgtd_stream s) {
  DGTD_PROF(s, DGTD_HBM, (dx_add ? 4.0 : 3.0) * dgtd_esize(dt) * rows * C, "dgtd_layernorm_bwd[rows=%lld,C=%d]", (long long)rows, C);
  if (dt == DGTD_F32) return ln_bwd_launch<float>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, workspace, rows, C, (hipStream_t)s, dx_add);
  if (dt == DGTD_BF16) return ln_bwd_launch<bf16_t>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, workspace, rows, C, (hipStream_t)s, dx_add);
  if (dt == DGTD_F16) return ln_bwd_launch<f16_t>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, workspace, rows, C, (hipStream_t)s, dx_add);
  DGTD_FAIL(2, "layernorm_bwd_add: bad dtype %d", (int)dt);
}

extern "C" int dgtd_layernorm_bwd_partial(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                          const void* dx_add, void* dx, void* workspace, int64_t rows, int C, dgtd_dtype dt, int* nblocks,
                                          dgtd_stream s) {
  DGTD_REQUIRE(nblocks, "layernorm_bwd_partial: nblocks is NULL");
  DGTD_PROF(s, DGTD_HBM, (dx_add ? 4.0 : 3.0) * dgtd_esize(dt) * rows * C, "dgtd_layernorm_bwd[rows=%lld,C=%d]", (long long)rows, C);
  if (dt == DGTD_F32) return ln_bwd_launch<float>(dy, x, gamma, mean, rstd, dx, nullptr, nullptr, workspace, rows, C, (hipStream_t)s, dx_add, nblocks);
  if (dt == DGTD_BF16) return ln_bwd_launch<bf16_t>(dy, x, gamma, mean, rstd, dx, nullptr, nullptr, workspace, rows, C, (hipStream_t)s, dx_add, nblocks);
  if (dt == DGTD_F16) return ln_bwd_launch<f16_t>(dy, x, gamma, mean, rstd, dx, nullptr, nullptr, workspace, rows, C, (hipStream_t)s, dx_add, nblocks);
  DGTD_FAIL(2, "layernorm_bwd_partial: bad dtype %d", (int)dt);
}

extern "C" int64_t dgtd_layernorm_bwd_workspace(int C) { return (int64_t)LN_BWD_MAX_GRID * 2 * C * sizeof(float); }

extern "C" int dgtd_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                  void* dx, float* dgamma, float* dbeta, void* workspace, int64_t rows, int C,
                                  dgtd_dtype dt, dgtd_stream s) {
  DGTD_PROF(s, DGTD_HBM, 3.0 * dgtd_esize(dt) * rows * C, "dgtd_layernorm_bwd[rows=%lld,C=%d]", (long long)rows, C);
  if (dt == DGTD_F32) return ln_bwd_launch<float>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, workspace, rows, C, (hipStream_t)s);
  if (dt == DGTD_BF16) return ln_bwd_launch<bf16_t>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, workspace, rows, C, (hipStream_t)s);
  if (dt == DGTD_F16) return ln_bwd_launch<f16_t>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, workspace, rows, C, (hipStream_t)s);
  DGTD_FAIL(2, "layernorm_bwd: bad dtype %d", (int)dt);
}
