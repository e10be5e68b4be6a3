// sam.hip — SAM (twig/model/cod.py:454-506) over NHWC maps x_h, x_l [B, HW, C]:
//   out = x_h * G(x_h) + x_l * G(x_l),  G(x)[b][c] = sigmoid(W2 relu(W1 y))[c] * sigmoid(V2 relu(V1 y)),  y = mean_hw(x)[b]
//   W1 [R,C], W2 [C,R] = SAM.fc (cod.py:459-464), V1 [R,C], V2 [1,R] = SAM.fc_wight (cod.py:465-470); the same weights gate both inputs.
// Two launches each way: per-slice pooled sums of both inputs (grid.z = input), then an apply pass in which every workgroup re-derives
// the two tiny MLPs of its sample from the slice partials (same arithmetic, same order everywhere: deterministic, nothing to zero).
// In the backward, workgroup (0, 0) also walks all samples once and sums the weight gradients in sample order.  The eager form of this
// module was ~25 launches forward and ~50 backward for 4 + 6 passes over 2 MB maps.  HBM bytes: forward 5 e n (both inputs twice, out),
// backward 7 e n (g three times, both inputs, both input gradients), n = B HW C.
#include "common.h"
#include <algorithm>

namespace {

constexpr int SAM_SLICES = 64;

// partial[z][slice][b][c] = sum over the slice's rows of x_z[b][hw][c] (* g[b][hw][c] when PRODUCT)
template <typename T, bool PRODUCT>
__global__ __launch_bounds__(256) void sam_pool_kernel(const T* __restrict__ xh, const T* __restrict__ xl, const T* __restrict__ g,
                                                       float* __restrict__ partial, int HW, int C) {
  typedef typename Vec16<T>::type VT;
  constexpr int V = Vec16<T>::N;
  __shared__ float red[256 * 8];
  const int b = blockIdx.y, z = blockIdx.z, tid = threadIdx.x, B = gridDim.y;
  const int CV = C / V, rpp = 256 / CV, cv = tid % CV, rl = tid / CV;
  float acc[V];
#pragma unroll
  for (int j = 0; j < V; ++j) acc[j] = 0.f;
  if (rl < rpp) {
    const T* ab = (z ? xl : xh) + (size_t)b * HW * C + (size_t)cv * V;
    const T* gb = PRODUCT ? g + (size_t)b * HW * C + (size_t)cv * V : nullptr;
    for (int r = blockIdx.x * rpp + rl; r < HW; r += gridDim.x * rpp) {
      const VT v = *reinterpret_cast<const VT*>(ab + (size_t)r * C);
      if (PRODUCT) {
        const VT w = *reinterpret_cast<const VT*>(gb + (size_t)r * C);
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] += (float)v[j] * (float)w[j];
      } else {
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] += (float)v[j];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < V; ++j) red[(rl * CV + cv) * V + j] = rl < rpp ? acc[j] : 0.f;
  __syncthreads();
  if (tid < C) {
    float s = 0.f;
    for (int k = 0; k < rpp; ++k) s += red[k * C + tid];
    partial[(((size_t)z * gridDim.x + blockIdx.x) * B + b) * C + tid] = s;
  }
}

// what the backward keeps, fp32: pooled [2][B][C] | gc [2][B][C] | hid_c [2][B][R] | hid_w [2][B][R] | gw [2][B]
struct SamStats { float *pooled, *gc, *hc, *hw, *gw; };
__host__ __device__ inline SamStats sam_stats(float* p, int B, int C, int R) {
  SamStats s;
  s.pooled = p;
  s.gc = s.pooled + (size_t)2 * B * C;
  s.hc = s.gc + (size_t)2 * B * C;
  s.hw = s.hc + (size_t)2 * B * R;
  s.gw = s.hw + (size_t)2 * B * R;
  return s;
}
inline size_t sam_stats_floats(int B, int C, int R) { return (size_t)4 * B * C + (size_t)4 * B * R + (size_t)2 * B; }

__device__ __forceinline__ float sigmoidf(float v) { return 1.f / (1.f + expf(-v)); }

template <typename T>
__global__ __launch_bounds__(256) void sam_apply_kernel(const T* __restrict__ xh, const T* __restrict__ xl, const float* __restrict__ partial,
                                                        int nparts, const float* __restrict__ w1, const float* __restrict__ w2,
                                                        const float* __restrict__ v1, const float* __restrict__ v2, float* __restrict__ stats,
                                                        T* __restrict__ out, int HW, int C, int R) {
  typedef typename Vec16<T>::type VT;
  constexpr int V = Vec16<T>::N;
  __shared__ float m[256], G[256], hid[2][2][32];
  const int b = blockIdx.y, B = gridDim.y, tid = threadIdx.x;
  const bool keeper = blockIdx.x == 0;
  const SamStats st = sam_stats(stats, B, C, R);
  const int z = tid / C, c = tid % C;
  if (tid < 2 * C) {
    float s0 = 0.f, s1 = 0.f;
    int k = 0;
    for (; k + 1 < nparts; k += 2) {
      s0 += partial[(((size_t)z * nparts + k) * B + b) * C + c];
      s1 += partial[(((size_t)z * nparts + k + 1) * B + b) * C + c];
    }
    if (k < nparts) s0 += partial[(((size_t)z * nparts + k) * B + b) * C + c];
    const float sum = s0 + s1;
    if (keeper) st.pooled[((size_t)z * B + b) * C + c] = sum;
    m[z * 128 + c] = sum / (float)HW;
  }
  __syncthreads();
  if (tid < 4 * R) {                               // (input, MLP, hidden unit)
    const int zz = tid / (2 * R), mlp = (tid / R) % 2, j = tid % R;
    const float* w = mlp ? v1 : w1;
    float a = 0.f;
    for (int cc = 0; cc < C; ++cc) a += w[j * C + cc] * m[zz * 128 + cc];
    a = fmaxf(a, 0.f);
    hid[zz][mlp][j] = a;
    if (keeper) (mlp ? st.hw : st.hc)[((size_t)zz * B + b) * R + j] = a;
  }
  __syncthreads();
  if (tid < 2 * C) {
    float a = 0.f, aw = 0.f;
    for (int j = 0; j < R; ++j) { a += w2[c * R + j] * hid[z][0][j]; aw += v2[j] * hid[z][1][j]; }
    const float gc = sigmoidf(a), gw = sigmoidf(aw);
    G[z * 128 + c] = gc * gw;
    if (keeper) {
      st.gc[((size_t)z * B + b) * C + c] = gc;
      if (c == 0) st.gw[(size_t)z * B + b] = gw;
    }
  }
  __syncthreads();
  const int CV = C / V;
  const size_t base = (size_t)b * HW * CV;
  for (int i = blockIdx.x * 256 + tid; i < HW * CV; i += gridDim.x * 256) {
    const int cv = i % CV;
    const VT a = *reinterpret_cast<const VT*>(xh + (base + i) * V), l = *reinterpret_cast<const VT*>(xl + (base + i) * V);
    VT o;
#pragma unroll
    for (int j = 0; j < V; ++j) o[j] = (T)((float)a[j] * G[cv * V + j] + (float)l[j] * G[128 + cv * V + j]);
    *reinterpret_cast<VT*>(out + (base + i) * V) = o;
  }
}

// d x_z = g * G_z[b][c] + dmean_z[b][c]; dw = { dW1 [R,C] | dW2 [C,R] | dV1 [R,C] | dV2 [R] } (3 R C + R <= 1024 entries)
template <typename T>
__global__ __launch_bounds__(256) void sam_bwd_kernel(const T* __restrict__ g, const float* __restrict__ partial, int nparts,
                                                      const float* __restrict__ stats, const float* __restrict__ w1, const float* __restrict__ w2,
                                                      const float* __restrict__ v1, const float* __restrict__ v2, T* __restrict__ dxh,
                                                      T* __restrict__ dxl, float* __restrict__ dw, int HW, int C, int R) {
  typedef typename Vec16<T>::type VT;
  constexpr int V = Vec16<T>::N;
  __shared__ float m[256], dzc[256], tw[256], G[256], dm[256], hc[2][32], hw[2][32], dhc[2][32], dhw[2][32], gws[2], dzw[2];
  const int b = blockIdx.y, B = gridDim.y, tid = threadIdx.x;
  const SamStats st = sam_stats(const_cast<float*>(stats), B, C, R);
  const int z = tid / C, c = tid % C;
  auto derive = [&](int bb) {          // all the per-sample quantities of sample bb into LDS (ends on a barrier)
    if (tid < 2 * C) {
      float d0 = 0.f, d1 = 0.f;
      int k = 0;
      for (; k + 1 < nparts; k += 2) {
        d0 += partial[(((size_t)z * nparts + k) * B + bb) * C + c];
        d1 += partial[(((size_t)z * nparts + k + 1) * B + bb) * C + c];
      }
      if (k < nparts) d0 += partial[(((size_t)z * nparts + k) * B + bb) * C + c];
      const float dG = d0 + d1, gc = st.gc[((size_t)z * B + bb) * C + c], gw = st.gw[(size_t)z * B + bb];
      m[z * 128 + c] = st.pooled[((size_t)z * B + bb) * C + c] / (float)HW;
      dzc[z * 128 + c] = dG * gw * gc * (1.f - gc);       // through the channel sigmoid
      tw[z * 128 + c] = dG * gc;                          // summed over c below: d loss / d gw
      G[z * 128 + c] = gc * gw;
      if (c == 0) gws[z] = gw;
    }
    if (tid < 2 * R) {
      const int zz = tid / R, j = tid % R;
      hc[zz][j] = st.hc[((size_t)zz * B + bb) * R + j];
      hw[zz][j] = st.hw[((size_t)zz * B + bb) * R + j];
    }
    __syncthreads();
    if (tid < 2) {
      float a = 0.f;
      for (int cc = 0; cc < C; ++cc) a += tw[tid * 128 + cc];
      dzw[tid] = a * gws[tid] * (1.f - gws[tid]);         // through the scalar sigmoid
    } else if (tid >= 64 && tid < 64 + 2 * R) {
      const int zz = (tid - 64) / R, j = (tid - 64) % R;
      float a = 0.f;
      for (int cc = 0; cc < C; ++cc) a += w2[cc * R + j] * dzc[zz * 128 + cc];
      dhc[zz][j] = hc[zz][j] > 0.f ? a : 0.f;             // through the ReLU
    }
    __syncthreads();
    if (tid < 2 * R) {
      const int zz = tid / R, j = tid % R;
      dhw[zz][j] = hw[zz][j] > 0.f ? v2[j] * dzw[zz] : 0.f;
    }
    __syncthreads();
    if (tid < 2 * C) {
      float a = 0.f;
      for (int j = 0; j < R; ++j) a += w1[j * C + c] * dhc[z][j] + v1[j * C + c] * dhw[z][j];
      dm[z * 128 + c] = a / (float)HW;                    // d loss / d x_z[b][hw][c] through the mean
    }
    __syncthreads();
  };
  if (blockIdx.x == 0 && b == 0) {                        // weight gradients: all samples in a fixed order; sample 0 (this workgroup's) last
    const int RC = R * C, total = 3 * RC + R;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int bb = B - 1; bb >= 0; --bb) {
      derive(bb);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = tid + 256 * u;
        if (i < total) {
          float a = 0.f;
          if (i < RC) { const int j = i / C, cc = i % C; a = dhc[0][j] * m[cc] + dhc[1][j] * m[128 + cc]; }
          else if (i < 2 * RC) { const int k = i - RC, cc = k / R, j = k % R; a = dzc[cc] * hc[0][j] + dzc[128 + cc] * hc[1][j]; }
          else if (i < 3 * RC) { const int k = i - 2 * RC, j = k / C, cc = k % C; a = dhw[0][j] * m[cc] + dhw[1][j] * m[128 + cc]; }
          else { const int j = i - 3 * RC; a = dzw[0] * hw[0][j] + dzw[1] * hw[1][j]; }
          acc[u] += a;
        }
      }
      if (bb > 0) __syncthreads();                        // the next derive() overwrites what was just read
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (tid + 256 * u < total) dw[tid + 256 * u] = acc[u];
  } else {
    derive(b);
  }
  const int CV = C / V;
  const size_t base = (size_t)b * HW * CV;
  for (int i = blockIdx.x * 256 + tid; i < HW * CV; i += gridDim.x * 256) {
    const int cv = i % CV;
    const VT gv = *reinterpret_cast<const VT*>(g + (base + i) * V);
    VT oh, ol;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      oh[j] = (T)((float)gv[j] * G[cv * V + j] + dm[cv * V + j]);
      ol[j] = (T)((float)gv[j] * G[128 + cv * V + j] + dm[128 + cv * V + j]);
    }
    *reinterpret_cast<VT*>(dxh + (base + i) * V) = oh;
    *reinterpret_cast<VT*>(dxl + (base + i) * V) = ol;
  }
}

bool sam_ok(int B, int HW, int C, int R, dgtd_dtype dt) {
  const int V = DGTD_IS_HALF(dt) ? 8 : 4;
  return B > 0 && B <= 65535 && HW > 0 && C >= V && C <= 128 && C % V == 0 && 256 % (C / V) == 0 && R > 0 && R <= 32 && 3 * R * C + R <= 1024 &&
         (DGTD_IS_HALF(dt) || dt == DGTD_F32);
}
inline int slices(int HW, int C) {
  const int cpr = std::max(1, 256 / C);
  return (int)std::max<int64_t>(1, std::min<int64_t>(((int64_t)HW + cpr * 8 - 1) / (cpr * 8), SAM_SLICES));
}
inline int apply_slices(int HW, int CV) { return (int)std::max<int64_t>(1, std::min<int64_t>(((int64_t)HW * CV + 511) / 512, 128)); }

}  // namespace

extern "C" int dgtd_sam_supported(int B, int HW, int C, int R, dgtd_dtype dt) { return sam_ok(B, HW, C, R, dt) ? 1 : 0; }
extern "C" int64_t dgtd_sam_stats_floats(int B, int C, int R) { return (int64_t)(sam_stats_floats(B, C, R) + (size_t)2 * SAM_SLICES * B * C); }
extern "C" int64_t dgtd_sam_scratch_floats(int B, int C) { return (int64_t)2 * SAM_SLICES * B * C; }

extern "C" int dgtd_sam_fwd(const void* xh, const void* xl, const float* w1, const float* w2, const float* v1, const float* v2, void* out,
                            float* stats, int B, int HW, int C, int R, dgtd_dtype dt, dgtd_stream s) {
  DGTD_PROF(s, DGTD_HBM, 5.0 * dgtd_esize(dt) * B * HW * C, "dgtd_sam_fwd[B=%d,HW=%d,C=%d]", B, HW, C);
  DGTD_REQUIRE(sam_ok(B, HW, C, R, dt), "sam_fwd: unsupported sizes B=%d HW=%d C=%d R=%d dtype %d", B, HW, C, R, (int)dt);
  const hipStream_t st = (hipStream_t)s;
  const int V = DGTD_IS_HALF(dt) ? 8 : 4, gx = slices(HW, C), ax = apply_slices(HW, C / V);
  float* partial = stats + sam_stats_floats(B, C, R);
  if (dt == DGTD_F16) hipLaunchKernelGGL((sam_pool_kernel<f16_t, false>), dim3(gx, B, 2), dim3(256), 0, st, (const f16_t*)xh, (const f16_t*)xl, (const f16_t*)nullptr, partial, HW, C);
  else if (dt == DGTD_BF16) hipLaunchKernelGGL((sam_pool_kernel<bf16_t, false>), dim3(gx, B, 2), dim3(256), 0, st, (const bf16_t*)xh, (const bf16_t*)xl, (const bf16_t*)nullptr, partial, HW, C);
  else hipLaunchKernelGGL((sam_pool_kernel<float, false>), dim3(gx, B, 2), dim3(256), 0, st, (const float*)xh, (const float*)xl, (const float*)nullptr, partial, HW, C);
  DGTD_CHECK_LAUNCH("sam_pool");
  if (dt == DGTD_F16) hipLaunchKernelGGL(sam_apply_kernel<f16_t>, dim3(ax, B), dim3(256), 0, st, (const f16_t*)xh, (const f16_t*)xl, (const float*)partial, gx, w1, w2, v1, v2, stats, (f16_t*)out, HW, C, R);
  else if (dt == DGTD_BF16) hipLaunchKernelGGL(sam_apply_kernel<bf16_t>, dim3(ax, B), dim3(256), 0, st, (const bf16_t*)xh, (const bf16_t*)xl, (const float*)partial, gx, w1, w2, v1, v2, stats, (bf16_t*)out, HW, C, R);
  else hipLaunchKernelGGL(sam_apply_kernel<float>, dim3(ax, B), dim3(256), 0, st, (const float*)xh, (const float*)xl, (const float*)partial, gx, w1, w2, v1, v2, stats, (float*)out, HW, C, R);
  DGTD_CHECK_LAUNCH("sam_apply");
  return 0;
}

extern "C" int dgtd_sam_bwd(const void* g, const void* xh, const void* xl, const float* w1, const float* w2, const float* v1, const float* v2,
                            const float* stats, void* dxh, void* dxl, float* dw, float* scratch, int B, int HW, int C, int R, dgtd_dtype dt,
                            dgtd_stream s) {
  DGTD_PROF(s, DGTD_HBM, 7.0 * dgtd_esize(dt) * B * HW * C, "dgtd_sam_bwd[B=%d,HW=%d,C=%d]", B, HW, C);
  DGTD_REQUIRE(sam_ok(B, HW, C, R, dt), "sam_bwd: unsupported sizes B=%d HW=%d C=%d R=%d dtype %d", B, HW, C, R, (int)dt);
  const hipStream_t st = (hipStream_t)s;
  const int V = DGTD_IS_HALF(dt) ? 8 : 4, gx = slices(HW, C), ax = apply_slices(HW, C / V);
  if (dt == DGTD_F16) hipLaunchKernelGGL((sam_pool_kernel<f16_t, true>), dim3(gx, B, 2), dim3(256), 0, st, (const f16_t*)xh, (const f16_t*)xl, (const f16_t*)g, scratch, HW, C);
  else if (dt == DGTD_BF16) hipLaunchKernelGGL((sam_pool_kernel<bf16_t, true>), dim3(gx, B, 2), dim3(256), 0, st, (const bf16_t*)xh, (const bf16_t*)xl, (const bf16_t*)g, scratch, HW, C);
  else hipLaunchKernelGGL((sam_pool_kernel<float, true>), dim3(gx, B, 2), dim3(256), 0, st, (const float*)xh, (const float*)xl, (const float*)g, scratch, HW, C);
  DGTD_CHECK_LAUNCH("sam_dgate_sum");
  if (dt == DGTD_F16) hipLaunchKernelGGL(sam_bwd_kernel<f16_t>, dim3(ax, B), dim3(256), 0, st, (const f16_t*)g, (const float*)scratch, gx, stats, w1, w2, v1, v2, (f16_t*)dxh, (f16_t*)dxl, dw, HW, C, R);
  else if (dt == DGTD_BF16) hipLaunchKernelGGL(sam_bwd_kernel<bf16_t>, dim3(ax, B), dim3(256), 0, st, (const bf16_t*)g, (const float*)scratch, gx, stats, w1, w2, v1, v2, (bf16_t*)dxh, (bf16_t*)dxl, dw, HW, C, R);
  else hipLaunchKernelGGL(sam_bwd_kernel<float>, dim3(ax, B), dim3(256), 0, st, (const float*)g, (const float*)scratch, gx, stats, w1, w2, v1, v2, (float*)dxh, (float*)dxl, dw, HW, C, R);
  DGTD_CHECK_LAUNCH("sam_bwd");
  return 0;
}
