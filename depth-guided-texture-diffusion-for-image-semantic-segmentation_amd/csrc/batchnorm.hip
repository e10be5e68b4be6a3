// batchnorm.hip — nn.BatchNorm2d of BasicConv2d (twig/model/cod.py:359, :366) over NHWC maps, x [N = B*H*W rows, C]:
//   training: y = (x - mean_c) * rstd_c * gamma_c + beta_c with the batch mean / biased variance; running_mean / running_var move by
//             `momentum` (unbiased variance), num_batches_tracked += 1 - all inside the second launch
//   eval    : the same affine map from the running statistics (one launch)
//   backward: dx = gamma rstd (dy - mean(dy) - xhat mean(dy xhat)), dgamma = sum dy xhat, dbeta = sum dy
// Two launches each way (per-slice channel partials, then every workgroup of the apply pass combines the <= 64 slices itself in a fixed
// order - deterministic, nothing to zero) where the library path runs three; the maps are 2 - 8 MB, so the passes are latency-bound
// and the launch count is what there is to save.  HBM bytes: forward 3 e N C (x twice, y), backward 5 e N C (dy twice, x twice, dx).
#include "common.h"
#include <algorithm>

namespace {

constexpr int BN_MAXP = 64;        // statistic slices per call

// rows of slice p when slices take `rpp` consecutive rows round-robin
__host__ __device__ inline int64_t slice_rows(int p, int P, int rpp, int64_t N) {
  const int64_t cycle = (int64_t)P * rpp, full = N / cycle, rem = N % cycle;
  int64_t extra = rem - (int64_t)p * rpp;
  extra = extra < 0 ? 0 : (extra > rpp ? rpp : extra);
  return full * rpp + extra;
}

// partial[p][0][c] = sum_a over the slice; partial[p][1][c] = MODE 0: sum (a - slice mean)^2 (the slices combine by Chan's rule);
// MODE 1: sum a * xhat with xhat = (x - mean_c) rstd_c from `save`
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_partial_kernel(const T* __restrict__ a, const T* __restrict__ xin, const float* __restrict__ save,
                                                         float* __restrict__ partial, int64_t N, int C) {
  typedef typename Vec16<T>::type VT;
  constexpr int V = Vec16<T>::N;
  __shared__ float red_s[256 * V], red_q[256 * V];
  const int tid = threadIdx.x, CV = C / V, rpp = 256 / CV, cv = tid % CV, rl = tid / CV, P = gridDim.x;
  float s[V], q[V], mu[V], rs[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    s[j] = q[j] = 0.f;
    if (MODE == 1) { mu[j] = save[cv * V + j]; rs[j] = save[C + cv * V + j]; }
  }
  if (rl < rpp) {
    for (int64_t r = (int64_t)blockIdx.x * rpp + rl; r < N; r += (int64_t)P * rpp) {
      const VT v = *reinterpret_cast<const VT*>(a + r * C + cv * V);
      if (MODE == 1) {
        const VT xv = *reinterpret_cast<const VT*>(xin + r * C + cv * V);
#pragma unroll
        for (int j = 0; j < V; ++j) { s[j] += (float)v[j]; q[j] += (float)v[j] * (((float)xv[j] - mu[j]) * rs[j]); }
      } else {
#pragma unroll
        for (int j = 0; j < V; ++j) { s[j] += (float)v[j]; q[j] += (float)v[j] * (float)v[j]; }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < V; ++j) {
    red_s[(rl * CV + cv) * V + j] = rl < rpp ? s[j] : 0.f;
    red_q[(rl * CV + cv) * V + j] = rl < rpp ? q[j] : 0.f;
  }
  __syncthreads();
  if (tid < C) {
    float S = 0.f, Q = 0.f;
    for (int k = 0; k < rpp; ++k) { S += red_s[k * C + tid]; Q += red_q[k * C + tid]; }
    if (MODE == 0) {
      const float n = (float)slice_rows(blockIdx.x, P, rpp, N);
      Q = n > 0.f ? fmaxf(Q - S * S / n, 0.f) : 0.f;
    }
    partial[((size_t)blockIdx.x * 2 + 0) * C + tid] = S;
    partial[((size_t)blockIdx.x * 2 + 1) * C + tid] = Q;
  }
}

// y = x * scale_c + shift_c.  Training (P > 0): every workgroup combines the slice partials (same order everywhere); workgroup 0 keeps
// { mean | rstd } for the backward and moves the running statistics.  Eval (P == 0): scale / shift from the running statistics.
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ partial, int P, int rpp,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ running_mean, float* __restrict__ running_var,
                                                       long long* __restrict__ num_batches, T* __restrict__ y, float* __restrict__ save,
                                                       int64_t N, int C, float eps, float momentum) {
  typedef typename Vec16<T>::type VT;
  constexpr int V = Vec16<T>::N;
  __shared__ float red[256], mean_s[128], sc[128], sh[128];
  const int tid = threadIdx.x, c = tid % C, g = tid / C, G = 256 / C;
  if (P > 0) {
    float s = 0.f;
    for (int p = g; p < P; p += G) s += partial[((size_t)p * 2) * C + c];
    red[tid] = s;
    __syncthreads();
    if (tid < C) {
      float t = 0.f;
      for (int k = 0; k < G; ++k) t += red[k * C + tid];
      mean_s[tid] = t / (float)N;
    }
    __syncthreads();
    const float mu = mean_s[c];
    float m2 = 0.f;
    for (int p = g; p < P; p += G) {
      const float n = (float)slice_rows(p, P, rpp, N);
      if (n > 0.f) {
        const float d = partial[((size_t)p * 2) * C + c] / n - mu;
        m2 += partial[((size_t)p * 2 + 1) * C + c] + n * d * d;
      }
    }
    __syncthreads();
    red[tid] = m2;
    __syncthreads();
    if (tid < C) {
      float t = 0.f;
      for (int k = 0; k < G; ++k) t += red[k * C + tid];
      const float var = t / (float)N, rstd = 1.f / sqrtf(var + eps), m = mean_s[tid];
      const float scale = (gamma ? gamma[tid] : 1.f) * rstd;
      sc[tid] = scale;
      sh[tid] = (beta ? beta[tid] : 0.f) - m * scale;
      if (blockIdx.x == 0) {
        save[tid] = m;
        save[C + tid] = rstd;
        if (running_mean) running_mean[tid] = (1.f - momentum) * running_mean[tid] + momentum * m;
        if (running_var) running_var[tid] = (1.f - momentum) * running_var[tid] + momentum * (N > 1 ? t / (float)(N - 1) : var);
        if (tid == 0 && num_batches) *num_batches += 1;
      }
    }
  } else if (tid < C) {
    const float scale = (gamma ? gamma[tid] : 1.f) / sqrtf(running_var[tid] + eps);
    sc[tid] = scale;
    sh[tid] = (beta ? beta[tid] : 0.f) - running_mean[tid] * scale;
  }
  __syncthreads();
  const int CV = C / V;
  for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < N * CV; i += (int64_t)gridDim.x * 256) {
    const int cv = (int)(i % CV);
    const VT v = *reinterpret_cast<const VT*>(x + i * V);
    VT o;
#pragma unroll
    for (int j = 0; j < V; ++j) o[j] = (T)((float)v[j] * sc[cv * V + j] + sh[cv * V + j]);
    *reinterpret_cast<VT*>(y + i * V) = o;
  }
}

// dx = k dy + a x + b per channel with k = gamma rstd, a = -k rstd sum(dy xhat) / N, b = -k sum(dy) / N - a mean; workgroup 0 writes dgamma, dbeta
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ partial,
                                                           int P, const float* __restrict__ gamma, const float* __restrict__ save,
                                                           T* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           int64_t N, int C) {
  typedef typename Vec16<T>::type VT;
  constexpr int V = Vec16<T>::N;
  __shared__ float red_s[256], red_q[256], kk[128], aa[128], bb[128];
  const int tid = threadIdx.x, c = tid % C, g = tid / C, G = 256 / C;
  float s = 0.f, q = 0.f;
  for (int p = g; p < P; p += G) { s += partial[((size_t)p * 2) * C + c]; q += partial[((size_t)p * 2 + 1) * C + c]; }
  red_s[tid] = s;
  red_q[tid] = q;
  __syncthreads();
  if (tid < C) {
    float S = 0.f, Q = 0.f;
    for (int k = 0; k < G; ++k) { S += red_s[k * C + tid]; Q += red_q[k * C + tid]; }
    const float mu = save[tid], rstd = save[C + tid], k1 = (gamma ? gamma[tid] : 1.f) * rstd;
    const float a = -k1 * rstd * Q / (float)N;
    kk[tid] = k1;
    aa[tid] = a;
    bb[tid] = -k1 * S / (float)N - a * mu;
    if (blockIdx.x == 0) {
      if (dgamma) dgamma[tid] = Q;
      if (dbeta) dbeta[tid] = S;
    }
  }
  __syncthreads();
  const int CV = C / V;
  for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < N * CV; i += (int64_t)gridDim.x * 256) {
    const int cv = (int)(i % CV);
    const VT gv = *reinterpret_cast<const VT*>(dy + i * V), xv = *reinterpret_cast<const VT*>(x + i * V);
    VT o;
#pragma unroll
    for (int j = 0; j < V; ++j) o[j] = (T)((float)gv[j] * kk[cv * V + j] + (float)xv[j] * aa[cv * V + j] + bb[cv * V + j]);
    *reinterpret_cast<VT*>(dx + i * V) = o;
  }
}

inline int slices_for(int64_t N, int rpp) { return (int)std::max<int64_t>(1, std::min<int64_t>(BN_MAXP, N / ((int64_t)rpp * 2))); }
inline int apply_grid(int64_t vecs) { return (int)std::max<int64_t>(1, std::min<int64_t>((vecs + 1023) / 1024, 128)); }

bool geometry_ok(int64_t N, int C, dgtd_dtype dt) {
  const int V = DGTD_IS_HALF(dt) ? 8 : 4;
  return N > 0 && C >= V && C <= 128 && C % V == 0 && 256 % C == 0 && (DGTD_IS_HALF(dt) || dt == DGTD_F32);
}

}  // namespace

extern "C" int dgtd_batchnorm_supported(int64_t N, int C, dgtd_dtype dt) { return geometry_ok(N, C, dt) ? 1 : 0; }
extern "C" int64_t dgtd_batchnorm_scratch(int C) { return (int64_t)BN_MAXP * 2 * C; }

extern "C" int dgtd_batchnorm_fwd(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                  long long* num_batches, void* y, float* save, float* scratch, int64_t N, int C, float eps, float momentum,
                                  int training, dgtd_dtype dt, dgtd_stream s) {
  DGTD_PROF(s, DGTD_HBM, (training ? 3.0 : 2.0) * dgtd_esize(dt) * N * C, "dgtd_batchnorm_fwd[%s,N=%lld,C=%d]", training ? "train" : "eval", (long long)N, C);
  DGTD_REQUIRE(geometry_ok(N, C, dt), "batchnorm_fwd: unsupported geometry N=%lld C=%d dtype %d (C a power of two in [8, 128])", (long long)N, C, (int)dt);
  DGTD_REQUIRE(training ? (save && scratch) : (running_mean && running_var), "batchnorm_fwd: %s", training ? "training needs save and scratch" : "eval needs the running statistics");
  const hipStream_t st = (hipStream_t)s;
  const int V = DGTD_IS_HALF(dt) ? 8 : 4, rpp = 256 / (C / V), P = training ? slices_for(N, rpp) : 0, grid = apply_grid(N * (C / V));
  if (training) {
    if (dt == DGTD_F16) hipLaunchKernelGGL((bn_partial_kernel<f16_t, 0>), dim3(P), dim3(256), 0, st, (const f16_t*)x, (const f16_t*)nullptr, (const float*)nullptr, scratch, N, C);
    else if (dt == DGTD_BF16) hipLaunchKernelGGL((bn_partial_kernel<bf16_t, 0>), dim3(P), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)nullptr, (const float*)nullptr, scratch, N, C);
    else hipLaunchKernelGGL((bn_partial_kernel<float, 0>), dim3(P), dim3(256), 0, st, (const float*)x, (const float*)nullptr, (const float*)nullptr, scratch, N, C);
    DGTD_CHECK_LAUNCH("bn_partial");
  }
  if (dt == DGTD_F16) hipLaunchKernelGGL(bn_apply_kernel<f16_t>, dim3(grid), dim3(256), 0, st, (const f16_t*)x, (const float*)scratch, P, rpp, gamma, beta, running_mean, running_var, num_batches, (f16_t*)y, save, N, C, eps, momentum);
  else if (dt == DGTD_BF16) hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (const float*)scratch, P, rpp, gamma, beta, running_mean, running_var, num_batches, (bf16_t*)y, save, N, C, eps, momentum);
  else hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, (const float*)scratch, P, rpp, gamma, beta, running_mean, running_var, num_batches, (float*)y, save, N, C, eps, momentum);
  DGTD_CHECK_LAUNCH("bn_apply");
  return 0;
}

extern "C" int dgtd_batchnorm_bwd(const void* dy, const void* x, const float* gamma, const float* save, void* dx, float* dgamma, float* dbeta,
                                  float* scratch, int64_t N, int C, dgtd_dtype dt, dgtd_stream s) {
  DGTD_PROF(s, DGTD_HBM, 5.0 * dgtd_esize(dt) * N * C, "dgtd_batchnorm_bwd[N=%lld,C=%d]", (long long)N, C);
  DGTD_REQUIRE(geometry_ok(N, C, dt), "batchnorm_bwd: unsupported geometry N=%lld C=%d dtype %d (C a power of two in [8, 128])", (long long)N, C, (int)dt);
  DGTD_REQUIRE(save && scratch, "batchnorm_bwd: needs the saved statistics and scratch");
  const hipStream_t st = (hipStream_t)s;
  const int V = DGTD_IS_HALF(dt) ? 8 : 4, rpp = 256 / (C / V), P = slices_for(N, rpp), grid = apply_grid(N * (C / V));
  if (dt == DGTD_F16) hipLaunchKernelGGL((bn_partial_kernel<f16_t, 1>), dim3(P), dim3(256), 0, st, (const f16_t*)dy, (const f16_t*)x, save, scratch, N, C);
  else if (dt == DGTD_BF16) hipLaunchKernelGGL((bn_partial_kernel<bf16_t, 1>), dim3(P), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, save, scratch, N, C);
  else hipLaunchKernelGGL((bn_partial_kernel<float, 1>), dim3(P), dim3(256), 0, st, (const float*)dy, (const float*)x, save, scratch, N, C);
  DGTD_CHECK_LAUNCH("bn_bwd_partial");
  if (dt == DGTD_F16) hipLaunchKernelGGL(bn_bwd_apply_kernel<f16_t>, dim3(grid), dim3(256), 0, st, (const f16_t*)dy, (const f16_t*)x, (const float*)scratch, P, gamma, save, (f16_t*)dx, dgamma, dbeta, N, C);
  else if (dt == DGTD_BF16) hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, (const float*)scratch, P, gamma, save, (bf16_t*)dx, dgamma, dbeta, N, C);
  else hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)dy, (const float*)x, (const float*)scratch, P, gamma, save, (float*)dx, dgamma, dbeta, N, C);
  DGTD_CHECK_LAUNCH("bn_bwd_apply");
  return 0;
}
