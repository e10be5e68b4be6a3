// hitnet_glue.hip — elementwise / small-reduction glue of the Hitnet CAB decoder (twig/model/cod.py:415-451) on channels_last
// feature maps viewed as token matrices [B, HW, C]:
//   prelu      : y = x > 0 ? x : a*x with ONE shared slope (the default-argument nn.PReLU() of cod.py:686, used by all 8 CABs);
//                backward dx = g*(x>0 ? 1 : a), da = sum_{x<=0} g*x   (torch's generic PReLU backward materialises a full-size
//                per-element weight gradient and reduces it: 1.8 ms per step)
//   ca_gate    : out = res * sigmoid(W2 relu(W1 mean_hw(res))) + x     (CALayer cod.py:428-431 + the residual of CAB cod.py:449-451)
//                three launches forward (pooled sums, gate, apply), three backward; C <= 128, C/r <= 32
// HBM-bound: prelu 2e*n (fwd) / 3e*n (bwd); ca_gate fwd 4e*n (res twice, x, out), bwd 4e*n.
#include "common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void prelu_fwd_kernel(const T* __restrict__ x, const float* __restrict__ a, T* __restrict__ y, int64_t n) {
  typedef typename Vec16<T>::type VT;
  constexpr int V = Vec16<T>::N;
  const float slope = a[0];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n / V; i += (int64_t)gridDim.x * 256) {
    VT v = *reinterpret_cast<const VT*>(x + i * V), o;
#pragma unroll
    for (int j = 0; j < V; ++j) { const float f = (float)v[j]; o[j] = (T)(f > 0.f ? f : slope * f); }
    *reinterpret_cast<VT*>(y + i * V) = o;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void prelu_bwd_kernel(const T* __restrict__ x, const T* __restrict__ g, const float* __restrict__ a,
                                                        T* __restrict__ dx, float* __restrict__ da, int64_t n) {
  typedef typename Vec16<T>::type VT;
  constexpr int V = Vec16<T>::N;
  __shared__ float red[4];
  const float slope = a[0];
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n / V; i += (int64_t)gridDim.x * 256) {
    VT xv = *reinterpret_cast<const VT*>(x + i * V), gv = *reinterpret_cast<const VT*>(g + i * V), o;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float f = (float)xv[j], gg = (float)gv[j];
      o[j] = (T)(f > 0.f ? gg : slope * gg);
      acc += f > 0.f ? 0.f : gg * f;
    }
    *reinterpret_cast<VT*>(dx + i * V) = o;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(da, red[0] + red[1] + red[2] + red[3]);
}

// partial[x][b][c] = sum over slice x of the HW rows of a[b][hw][c] (* optional second factor bfac); the gridDim.x slices of a sample
// are summed in a fixed order by the per-sample MLP kernels (deterministic, nothing to zero)
template <typename T, bool PRODUCT>
__global__ __launch_bounds__(256) void pooled_sum_kernel(const T* __restrict__ a, const T* __restrict__ bfac, float* __restrict__ pooled,
                                                         int HW, int C) {
  // 16-byte chunks along C (C % V == 0): thread = (row slot, chunk); the row slots of a workgroup meet in LDS
  typedef typename Vec16<T>::type VT;
  constexpr int V = Vec16<T>::N;
  __shared__ float red[256 * 8];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int CV = C / V, rpp = 256 / CV;               // rows per pass (C <= 128 -> CV <= 32 -> at least 8)
  const int cv = tid % CV, rl = tid / CV;
  float acc[V];
#pragma unroll
  for (int j = 0; j < V; ++j) acc[j] = 0.f;
  if (rl < rpp) {
    const T* ab = a + (size_t)b * HW * C + (size_t)cv * V;
    const T* bb = PRODUCT ? bfac + (size_t)b * HW * C + (size_t)cv * V : nullptr;
    for (int r = blockIdx.x * rpp + rl; r < HW; r += gridDim.x * rpp) {
      const VT v = *reinterpret_cast<const VT*>(ab + (size_t)r * C);
      if (PRODUCT) {
        const VT w = *reinterpret_cast<const VT*>(bb + (size_t)r * C);
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] += (float)v[j] * (float)w[j];
      } else {
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] += (float)v[j];
      }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) red[(rl * CV + cv) * V + j] = acc[j];
  }
  __syncthreads();
  if (tid < C) {
    float s = 0.f;
    for (int k = 0; k < rpp; ++k) s += red[k * C + tid];
    pooled[((size_t)blockIdx.x * gridDim.y + b) * C + tid] = s;
  }
}

// gate[b][c] = sigmoid(W2 relu(W1 (pooled[b]/HW)));  hidden[b][j] kept for the backward.  One workgroup per sample.
__global__ __launch_bounds__(128) void ca_gate_mlp_kernel(const float* __restrict__ partial, int nparts, float* __restrict__ pooled,
                                                          const float* __restrict__ w1, const float* __restrict__ w2,
                                                          float* __restrict__ gate, float* __restrict__ hidden, int HW, int C, int R) {
  __shared__ float m[128], hdn[32];
  const int b = blockIdx.x, B = gridDim.x, tid = threadIdx.x;
  if (tid < C) {
    float s0 = 0.f, s1 = 0.f;
    int x = 0;
    for (; x + 1 < nparts; x += 2) { s0 += partial[((size_t)x * B + b) * C + tid]; s1 += partial[((size_t)(x + 1) * B + b) * C + tid]; }
    if (x < nparts) s0 += partial[((size_t)x * B + b) * C + tid];
    const float s = s0 + s1;
    pooled[b * C + tid] = s;
    m[tid] = s / (float)HW;
  }
  __syncthreads();
  if (tid < R) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += w1[tid * C + c] * m[c];
    s = fmaxf(s, 0.f);
    hdn[tid] = s;
    hidden[b * R + tid] = s;
  }
  __syncthreads();
  if (tid < C) {
    float s = 0.f;
    for (int j = 0; j < R; ++j) s += w2[tid * R + j] * hdn[j];
    gate[b * C + tid] = 1.f / (1.f + expf(-s));
  }
}

// out = res * gate[b][c] + x
template <typename T>
__global__ __launch_bounds__(256) void ca_apply_kernel(const T* __restrict__ res, const T* __restrict__ x, const float* __restrict__ gate,
                                                       T* __restrict__ out, int64_t rows, int HW, int C) {
  typedef typename Vec16<T>::type VT;
  constexpr int V = Vec16<T>::N;
  const int CV = C / V;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * CV; i += (int64_t)gridDim.x * 256) {
    const int cv = (int)(i % CV);
    const int b = (int)((i / CV) / HW);
    VT r = *reinterpret_cast<const VT*>(res + i * V), xv = *reinterpret_cast<const VT*>(x + i * V), o;
#pragma unroll
    for (int j = 0; j < V; ++j) o[j] = (T)((float)r[j] * gate[b * C + cv * V + j] + (float)xv[j]);
    *reinterpret_cast<VT*>(out + i * V) = o;
  }
}

// The per-sample MLP folded into the apply pass: grid (slices, B); every workgroup recomputes gate[b][:] from the pooled partial sums
// (C + R*C + C*R MACs, the same arithmetic in the same order as ca_gate_mlp_kernel, so all workgroups of a sample agree bit for bit)
// and applies it to its slice of the sample; workgroup 0 of the sample also writes pooled / hidden / gate for the backward.  The
// stand-alone MLP launch was 8.6 us of pure latency (one small workgroup per sample, three dependent phases) in front of a 5 us pass.
template <typename T>
__global__ __launch_bounds__(256) void ca_apply_fused_kernel(const T* __restrict__ res, const T* __restrict__ x, const float* __restrict__ partial,
                                                             int nparts, const float* __restrict__ w1, const float* __restrict__ w2,
                                                             float* __restrict__ pooled, float* __restrict__ hidden, float* __restrict__ gate,
                                                             T* __restrict__ out, int HW, int C, int R) {
  typedef typename Vec16<T>::type VT;
  constexpr int V = Vec16<T>::N;
  __shared__ float m[128], hdn[32], gt[128];
  const int b = blockIdx.y, B = gridDim.y, tid = threadIdx.x;
  const bool keeper = blockIdx.x == 0;
  if (tid < C) {
    float s0 = 0.f, s1 = 0.f;
    int k = 0;
    for (; k + 1 < nparts; k += 2) { s0 += partial[((size_t)k * B + b) * C + tid]; s1 += partial[((size_t)(k + 1) * B + b) * C + tid]; }
    if (k < nparts) s0 += partial[((size_t)k * B + b) * C + tid];
    const float sum = s0 + s1;
    if (keeper) pooled[b * C + tid] = sum;
    m[tid] = sum / (float)HW;
  }
  __syncthreads();
  if (tid < R) {
    float a = 0.f;
    for (int c = 0; c < C; ++c) a += w1[tid * C + c] * m[c];
    a = fmaxf(a, 0.f);
    hdn[tid] = a;
    if (keeper) hidden[b * R + tid] = a;
  }
  __syncthreads();
  if (tid < C) {
    float a = 0.f;
    for (int j = 0; j < R; ++j) a += w2[tid * R + j] * hdn[j];
    const float gv = 1.f / (1.f + expf(-a));
    gt[tid] = gv;
    if (keeper) gate[b * C + tid] = gv;
  }
  __syncthreads();
  const int CV = C / V;
  const size_t base = (size_t)b * HW * CV;
  for (int i = blockIdx.x * 256 + tid; i < HW * CV; i += gridDim.x * 256) {
    const int cv = i % CV;
    VT r = *reinterpret_cast<const VT*>(res + (base + i) * V), xv = *reinterpret_cast<const VT*>(x + (base + i) * V), o;
#pragma unroll
    for (int j = 0; j < V; ++j) o[j] = (T)((float)r[j] * gt[cv * V + j] + (float)xv[j]);
    *reinterpret_cast<VT*>(out + (base + i) * V) = o;
  }
}

// backward of the tiny MLP, one workgroup per sample: dgate_sum[b][c] = sum_hw g*res (per-slice partials, summed here in a fixed
// order) -> dmean[b][c]; the per-sample outer products dz x hidden and dh x mean go to dwp[b][2*R*C] and are summed over the batch
// in a fixed order by workgroup 0 of ca_apply_bwd_kernel (deterministic, nothing to zero, no extra launch)
__global__ __launch_bounds__(128) void ca_gate_mlp_bwd_kernel(const float* __restrict__ partial, int nparts, const float* __restrict__ gate,
                                                              const float* __restrict__ hidden, const float* __restrict__ pooled,
                                                              const float* __restrict__ w1, const float* __restrict__ w2,
                                                              float* __restrict__ dmean, float* __restrict__ dwp, int HW, int C, int R) {
  __shared__ float dz[128], dh[32], m[128], hdn[32];
  const int b = blockIdx.x, B = gridDim.x, tid = threadIdx.x;
  if (tid < C) {
    float dg0 = 0.f, dg1 = 0.f;
    int x = 0;
    for (; x + 1 < nparts; x += 2) { dg0 += partial[((size_t)x * B + b) * C + tid]; dg1 += partial[((size_t)(x + 1) * B + b) * C + tid]; }
    if (x < nparts) dg0 += partial[((size_t)x * B + b) * C + tid];
    const float gt = gate[b * C + tid];
    dz[tid] = (dg0 + dg1) * gt * (1.f - gt);      // through the sigmoid
    m[tid] = pooled[b * C + tid] / (float)HW;
  }
  if (tid < R) hdn[tid] = hidden[b * R + tid];
  __syncthreads();
  if (tid < R) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += w2[c * R + tid] * dz[c];
    dh[tid] = hdn[tid] > 0.f ? s : 0.f;                  // through the ReLU
  }
  __syncthreads();
  if (tid < C) {
    float s = 0.f;
    for (int j = 0; j < R; ++j) s += w1[j * C + tid] * dh[j];
    dmean[b * C + tid] = s / (float)HW;                  // d loss / d res[b][hw][c] through the mean
    float* o = dwp + (size_t)b * 2 * R * C;              // [dw1 (R x C) | dw2 (C x R)] of this sample
    for (int j = 0; j < R; ++j) {
      o[j * C + tid] = dh[j] * m[tid];
      o[R * C + tid * R + j] = dz[tid] * hdn[j];
    }
  }
}

// dres = g * gate[b][c] + dmean[b][c]
template <typename T>
__global__ __launch_bounds__(256) void ca_apply_bwd_kernel(const T* __restrict__ g, const float* __restrict__ gate, const float* __restrict__ dmean,
                                                           T* __restrict__ dres, int64_t rows, int HW, int C, const float* __restrict__ dwp,
                                                           float* __restrict__ dw1, float* __restrict__ dw2, int B, int R) {
  typedef typename Vec16<T>::type VT;
  constexpr int V = Vec16<T>::N;
  const int CV = C / V;
  if (blockIdx.x == 0) {      // dw1 [R,C] | dw2 [C,R] = sum over the samples of the per-sample outer products (fixed order)
    const int n = 2 * R * C;
    for (int i = threadIdx.x; i < n; i += 256) {
      float s = 0.f;
      for (int b = 0; b < B; ++b) s += dwp[(size_t)b * n + i];
      if (i < R * C) dw1[i] = s; else dw2[i - R * C] = s;
    }
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * CV; i += (int64_t)gridDim.x * 256) {
    const int cv = (int)(i % CV);
    const int b = (int)((i / CV) / HW);
    VT gv = *reinterpret_cast<const VT*>(g + i * V), o;
#pragma unroll
    for (int j = 0; j < V; ++j) o[j] = (T)((float)gv[j] * gate[b * C + cv * V + j] + dmean[b * C + cv * V + j]);
    *reinterpret_cast<VT*>(dres + i * V) = o;
  }
}

// The MLP backward folded into the apply pass (grid (slices, B), B <= CA_MAXB): every workgroup derives dz / dh / dmean of ITS sample
// (same arithmetic and order as ca_gate_mlp_bwd_kernel); workgroup (0, 0) additionally derives them for ALL samples at once and sums
// the per-sample outer products over the batch in a fixed order -> dw1, dw2 (the stand-alone kernel cost 11.4 us of latency).
constexpr int CA_MAXB = 32;
template <typename T>
__global__ __launch_bounds__(256) void ca_apply_bwd_fused_kernel(const T* __restrict__ g, const float* __restrict__ partial, int nparts,
                                                                 const float* __restrict__ gate, const float* __restrict__ hidden,
                                                                 const float* __restrict__ pooled, const float* __restrict__ w1,
                                                                 const float* __restrict__ w2, T* __restrict__ dres, float* __restrict__ dw1,
                                                                 float* __restrict__ dw2, int HW, int C, int R, int per_sample) {
  typedef typename Vec16<T>::type VT;
  constexpr int V = Vec16<T>::N;
  extern __shared__ float sh[];                 // own sample: dz[128] dh[32] gt[128] dm[128]; workgroup (0,0): + all samples dzA[B*128] dhA[B*32]
  float *dz = sh, *dh = sh + 128, *gt = sh + 160, *dm = sh + 288, *dzA = sh + 416, *dhA = dzA + (size_t)gridDim.y * 128;
  const int b = blockIdx.y, B = gridDim.y, tid = threadIdx.x;
  auto dgate_sum = [&](int bb, int c) {
    float d0 = 0.f, d1 = 0.f;
    int k = 0;
    for (; k + 1 < nparts; k += 2) { d0 += partial[((size_t)k * B + bb) * C + c]; d1 += partial[((size_t)(k + 1) * B + bb) * C + c]; }
    if (k < nparts) d0 += partial[((size_t)k * B + bb) * C + c];
    return d0 + d1;
  };
  if (tid < C) {
    const float gv = gate[b * C + tid];
    gt[tid] = gv;
    dz[tid] = dgate_sum(b, tid) * gv * (1.f - gv);           // through the sigmoid
  }
  __syncthreads();
  if (tid < R) {
    float a = 0.f;
    for (int c = 0; c < C; ++c) a += w2[c * R + tid] * dz[c];
    dh[tid] = hidden[b * R + tid] > 0.f ? a : 0.f;           // through the ReLU
  }
  __syncthreads();
  if (tid < C) {
    float a = 0.f;
    for (int j = 0; j < R; ++j) a += w1[j * C + tid] * dh[j];
    dm[tid] = a / (float)HW;                                  // d loss / d res[b][hw][c] through the mean
  }
  if (per_sample) {
    // rows mode: the first workgroup of every sample writes ITS sample's outer products { dh x mean | dz x hidden } as row b of dw1
    // (row stride 2 R C); whoever owns the rows sums them later (the deferred flush's multi_reduce) - no serial walk over the batch
    if (blockIdx.x == 0) {
      float* row = dw1 + (size_t)b * 2 * R * C;
      for (int i = tid; i < 2 * R * C; i += 256) {
        if (i < R * C) { const int j = i / C, c = i % C; row[i] = dh[j] * (pooled[b * C + c] / (float)HW); }
        else { const int k = i - R * C, c = k / R, j = k % R; row[i] = dz[c] * hidden[b * R + j]; }
      }
    }
  } else if (blockIdx.x == 0 && b == 0) {                     // weight gradients: all samples, summed over the batch in sample order
    for (int i = tid; i < B * C; i += 256) {
      const int bb = i / C, c = i % C;
      const float gv = gate[bb * C + c];
      dzA[bb * 128 + c] = dgate_sum(bb, c) * gv * (1.f - gv);
    }
    __syncthreads();
    for (int i = tid; i < B * R; i += 256) {
      const int bb = i / R, j = i % R;
      float a = 0.f;
      for (int c = 0; c < C; ++c) a += w2[c * R + j] * dzA[bb * 128 + c];
      dhA[bb * 32 + j] = hidden[bb * R + j] > 0.f ? a : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < 2 * R * C; i += 256) {
      float a = 0.f;
      if (i < R * C) {                                        // dw1[j][c] = sum_b dh_b[j] * mean_b[c]
        const int j = i / C, c = i % C;
        for (int bb = 0; bb < B; ++bb) a += dhA[bb * 32 + j] * (pooled[bb * C + c] / (float)HW);
        dw1[i] = a;
      } else {                                                // dw2[c][j] = sum_b dz_b[c] * hidden_b[j]
        const int k = i - R * C, c = k / R, j = k % R;
        for (int bb = 0; bb < B; ++bb) a += dzA[bb * 128 + c] * hidden[bb * R + j];
        dw2[k] = a;
      }
    }
  }
  __syncthreads();
  const int CV = C / V;
  const size_t base = (size_t)b * HW * CV;
  for (int i = blockIdx.x * 256 + tid; i < HW * CV; i += gridDim.x * 256) {
    const int cv = i % CV;
    VT gv = *reinterpret_cast<const VT*>(g + (base + i) * V), o;
#pragma unroll
    for (int j = 0; j < V; ++j) o[j] = (T)((float)gv[j] * gt[cv * V + j] + dm[cv * V + j]);
    *reinterpret_cast<VT*>(dres + (base + i) * V) = o;
  }
}


// ---- bilinear resize of NHWC maps (F.interpolate(mode="bilinear"), both align_corners conventions; cod.py:757-789 uses
// align_corners=True x2 / x4 / x0.5 between the decoder levels).  The backward is a GATHER over the output pixels that touch an
// input pixel (deterministic, no atomics): torch's NHWC bf16 backward scatters with atomics and costs 190 us per call here.
struct Axis { float scale; int align; };
__device__ __forceinline__ float src_index(int o, Axis a) {
  if (a.align) return a.scale * (float)o;
  const float s = a.scale * ((float)o + 0.5f) - 0.5f;
  return s < 0.f ? 0.f : s;
}
__device__ __forceinline__ void taps(int o, Axis a, int n_in, int& i0, int& i1, float& l1) {
  const float s = src_index(o, a);
  i0 = min((int)s, n_in - 1);
  i1 = min(i0 + 1, n_in - 1);
  l1 = s - (float)i0;
}

template <typename T>
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int Hi, int Wi, int Ho, int Wo,
                                                           int C, Axis ah, Axis aw) {
  typedef typename Vec16<T>::type VT;
  constexpr int V = Vec16<T>::N;
  const int CV = C / V;
  const int64_t n = (int64_t)B * Ho * Wo * CV;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int cv = (int)(i % CV);
    int64_t r = i / CV;
    const int ow = (int)(r % Wo);
    r /= Wo;
    const int oh = (int)(r % Ho), b = (int)(r / Ho);
    int h0, h1, w0, w1;
    float lh, lw;
    taps(oh, ah, Hi, h0, h1, lh);
    taps(ow, aw, Wi, w0, w1, lw);
    const T* xb = x + (size_t)b * Hi * Wi * C + cv * V;
    const VT v00 = *reinterpret_cast<const VT*>(xb + ((size_t)h0 * Wi + w0) * C), v01 = *reinterpret_cast<const VT*>(xb + ((size_t)h0 * Wi + w1) * C);
    const VT v10 = *reinterpret_cast<const VT*>(xb + ((size_t)h1 * Wi + w0) * C), v11 = *reinterpret_cast<const VT*>(xb + ((size_t)h1 * Wi + w1) * C);
    VT o;
#pragma unroll
    for (int e = 0; e < V; ++e)
      o[e] = (T)((1.f - lh) * ((1.f - lw) * (float)v00[e] + lw * (float)v01[e]) + lh * ((1.f - lw) * (float)v10[e] + lw * (float)v11[e]));
    *reinterpret_cast<VT*>(y + i * V) = o;
  }
}

// weight with which output index o reads input index i along one axis (0 when it does not)
__device__ __forceinline__ float tap_weight(int o, int i, Axis a, int n_in) {
  int i0, i1;
  float l1;
  taps(o, a, n_in, i0, i1, l1);
  return (i0 == i ? 1.f - l1 : 0.f) + (i1 == i ? l1 : 0.f);
}

// LPI neighbouring lanes share one (input pixel, 16-byte channel chunk): lane l takes the candidate output rows oh_lo + l, + LPI, ... and
// the partial sums meet through xor-shuffles in a fixed order.  With one lane per item an x8 up-sampling (x_hp path, 16x16 <- 128x128)
// left 6144 threads each walking a 20 x 20 window serially: 145 us for 0.8 MB.
template <typename T, int LPI>
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B, int Hi, int Wi, int Ho, int Wo,
                                                           int C, Axis ah, Axis aw, float inv_h, float inv_w) {
  typedef typename Vec16<T>::type VT;
  constexpr int V = Vec16<T>::N;
  const int CV = C / V;
  const int64_t n = (int64_t)B * Hi * Wi * CV;
  const int sub = threadIdx.x % LPI;
  const int64_t nround = (n + (256 / LPI) - 1) / (256 / LPI) * (256 / LPI);      // whole lane groups keep going together (shuffles)
  for (int64_t it = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LPI; it < nround; it += (int64_t)gridDim.x * (256 / LPI)) {
    const bool live = it < n;
    const int64_t i = live ? it : n - 1;
    const int cv = (int)(i % CV);
    int64_t r = i / CV;
    const int iw = (int)(r % Wi);
    r /= Wi;
    const int ih = (int)(r % Hi), b = (int)(r / Hi);
    // conservative candidate ranges: outputs whose source coordinate can lie within (i-1, i+1).  align_corners: src = scale*o, so
    // o in ((i-1)/scale, (i+1)/scale); otherwise src = scale*(o+0.5)-0.5, so o in ((i-0.5)/scale-0.5, (i+1.5)/scale-0.5).  One
    // range covers both (the exact tap weights below discard the extra candidates).
    const int oh_lo = max(0, (int)floorf(((float)ih - 1.f) * inv_h) - 2), oh_hi = min(Ho - 1, (int)ceilf(((float)ih + 1.5f) * inv_h) + 1);
    const int ow_lo = max(0, (int)floorf(((float)iw - 1.f) * inv_w) - 2), ow_hi = min(Wo - 1, (int)ceilf(((float)iw + 1.5f) * inv_w) + 1);
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
    const T* gb = dy + (size_t)b * Ho * Wo * C + cv * V;
    for (int oh = oh_lo + sub; oh <= oh_hi; oh += LPI) {
      const float wy = tap_weight(oh, ih, ah, Hi);
      if (wy == 0.f) continue;
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        const float wgt = wy * tap_weight(ow, iw, aw, Wi);
        if (wgt == 0.f) continue;
        const VT g = *reinterpret_cast<const VT*>(gb + ((size_t)oh * Wo + ow) * C);
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] += wgt * (float)g[e];
      }
    }
#pragma unroll
    for (int off = LPI >> 1; off > 0; off >>= 1)
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] += __shfl_xor(acc[e], off, 64);
    if (live && sub == 0) {
      VT o;
#pragma unroll
      for (int e = 0; e < V; ++e) o[e] = (T)acc[e];
      *reinterpret_cast<VT*>(dx + i * V) = o;
    }
  }
}

inline Axis make_axis(int n_in, int n_out, int align) {
  Axis a;
  a.align = align;
  a.scale = align ? (n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : 0.f) : (float)n_in / (float)n_out;
  return a;
}

inline int ew_grid(int64_t items) { return (int)std::min<int64_t>(cdiv(items, 256), 2048); }

}  // namespace

extern "C" int dgtd_prelu_fwd(const void* x, const float* a, void* y, int64_t n, dgtd_dtype dt, dgtd_stream s) {
  DGTD_PROF(s, DGTD_HBM, 2.0 * dgtd_esize(dt) * n, "dgtd_prelu_fwd[n=%lld]", (long long)n);
  const int V = DGTD_IS_HALF(dt) ? 8 : 4;
  DGTD_REQUIRE(n > 0 && n % V == 0, "prelu_fwd: n=%lld must be a positive multiple of %d", (long long)n, V);
  if (dt == DGTD_F16) hipLaunchKernelGGL(prelu_fwd_kernel<f16_t>, dim3(ew_grid(n / V)), dim3(256), 0, (hipStream_t)s, (const f16_t*)x, a, (f16_t*)y, n);
  else if (dt == DGTD_BF16) hipLaunchKernelGGL(prelu_fwd_kernel<bf16_t>, dim3(ew_grid(n / V)), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, a, (bf16_t*)y, n);
  else if (dt == DGTD_F32) hipLaunchKernelGGL(prelu_fwd_kernel<float>, dim3(ew_grid(n / V)), dim3(256), 0, (hipStream_t)s, (const float*)x, a, (float*)y, n);
  else DGTD_FAIL(2, "prelu_fwd: bad dtype %d", (int)dt);
  DGTD_CHECK_LAUNCH("prelu_fwd");
  return 0;
}

extern "C" int dgtd_prelu_bwd(const void* x, const void* g, const float* a, void* dx, float* da, int64_t n, dgtd_dtype dt, dgtd_stream s) {
  DGTD_PROF(s, DGTD_HBM, 3.0 * dgtd_esize(dt) * n, "dgtd_prelu_bwd[n=%lld]", (long long)n);
  const int V = DGTD_IS_HALF(dt) ? 8 : 4;
  DGTD_REQUIRE(n > 0 && n % V == 0, "prelu_bwd: n=%lld must be a positive multiple of %d", (long long)n, V);
  const int grid = std::min(ew_grid(n / V), 512);
  if (dt == DGTD_F16) hipLaunchKernelGGL(prelu_bwd_kernel<f16_t>, dim3(grid), dim3(256), 0, (hipStream_t)s, (const f16_t*)x, (const f16_t*)g, a, (f16_t*)dx, da, n);
  else if (dt == DGTD_BF16) hipLaunchKernelGGL(prelu_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, (const bf16_t*)g, a, (bf16_t*)dx, da, n);
  else if (dt == DGTD_F32) hipLaunchKernelGGL(prelu_bwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)s, (const float*)x, (const float*)g, a, (float*)dx, da, n);
  else DGTD_FAIL(2, "prelu_bwd: bad dtype %d", (int)dt);
  DGTD_CHECK_LAUNCH("prelu_bwd");
  return 0;
}

// stats fp32 [B*C (pooled sums) | B*C (gate) | B*R (hidden) | 64*B*C (per-slice partial sums)]: nothing to zero
extern "C" int dgtd_ca_gate_fwd(const void* res, const void* x, const float* w1, const float* w2, void* out, float* stats, int B, int HW,
                                int C, int R, dgtd_dtype dt, dgtd_stream s) {
  DGTD_PROF(s, DGTD_HBM, 3.0 * dgtd_esize(dt) * B * HW * C, "dgtd_ca_gate_fwd[B=%d,HW=%d,C=%d]", B, HW, C);
  const int V = DGTD_IS_HALF(dt) ? 8 : 4;
  DGTD_REQUIRE(B > 0 && HW > 0 && C > 0 && C <= 128 && R > 0 && R <= 32 && C % V == 0, "ca_gate_fwd: unsupported sizes C=%d R=%d", C, R);
  DGTD_REQUIRE(DGTD_IS_HALF(dt) || dt == DGTD_F32, "ca_gate_fwd: bad dtype %d", (int)dt);
  hipStream_t st = (hipStream_t)s;
  float *pooled = stats, *gate = stats + (size_t)B * C, *hidden = gate + (size_t)B * C, *partial = hidden + (size_t)B * R;
  const int cpr = std::max(1, 256 / C), gx = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(HW, cpr * 8), 64));   // <= 64 slices (the stats / scratch layout), summed per column by the per-sample MLP kernels
  if (dt == DGTD_F16) hipLaunchKernelGGL((pooled_sum_kernel<f16_t, false>), dim3(gx, B), dim3(256), 0, st, (const f16_t*)res, (const f16_t*)nullptr, partial, HW, C);
  else if (dt == DGTD_BF16) hipLaunchKernelGGL((pooled_sum_kernel<bf16_t, false>), dim3(gx, B), dim3(256), 0, st, (const bf16_t*)res, (const bf16_t*)nullptr, partial, HW, C);
  else hipLaunchKernelGGL((pooled_sum_kernel<float, false>), dim3(gx, B), dim3(256), 0, st, (const float*)res, (const float*)nullptr, partial, HW, C);
  DGTD_CHECK_LAUNCH("ca_pooled_sum");
  static const bool fused = !(getenv("DGTD_CA_FUSED") && getenv("DGTD_CA_FUSED")[0] == '0');
  if (fused && B <= 65535) {
    const int ax = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv((int64_t)HW * (C / V), 256 * 2), 128));
    if (dt == DGTD_F16) hipLaunchKernelGGL(ca_apply_fused_kernel<f16_t>, dim3(ax, B), dim3(256), 0, st, (const f16_t*)res, (const f16_t*)x, (const float*)partial, gx, w1, w2, pooled, hidden, gate, (f16_t*)out, HW, C, R);
    else if (dt == DGTD_BF16) hipLaunchKernelGGL(ca_apply_fused_kernel<bf16_t>, dim3(ax, B), dim3(256), 0, st, (const bf16_t*)res, (const bf16_t*)x, (const float*)partial, gx, w1, w2, pooled, hidden, gate, (bf16_t*)out, HW, C, R);
    else hipLaunchKernelGGL(ca_apply_fused_kernel<float>, dim3(ax, B), dim3(256), 0, st, (const float*)res, (const float*)x, (const float*)partial, gx, w1, w2, pooled, hidden, gate, (float*)out, HW, C, R);
    DGTD_CHECK_LAUNCH("ca_apply_fused");
    return 0;
  }
  hipLaunchKernelGGL(ca_gate_mlp_kernel, dim3(B), dim3(128), 0, st, (const float*)partial, gx, pooled, w1, w2, gate, hidden, HW, C, R);
  DGTD_CHECK_LAUNCH("ca_gate_mlp");
  const int64_t rows = (int64_t)B * HW;
  if (dt == DGTD_F16) hipLaunchKernelGGL(ca_apply_kernel<f16_t>, dim3(ew_grid(rows * (C / V))), dim3(256), 0, st, (const f16_t*)res, (const f16_t*)x, (const float*)gate, (f16_t*)out, rows, HW, C);
  else if (dt == DGTD_BF16) hipLaunchKernelGGL(ca_apply_kernel<bf16_t>, dim3(ew_grid(rows * (C / V))), dim3(256), 0, st, (const bf16_t*)res, (const bf16_t*)x, (const float*)gate, (bf16_t*)out, rows, HW, C);
  else hipLaunchKernelGGL(ca_apply_kernel<float>, dim3(ew_grid(rows * (C / V))), dim3(256), 0, st, (const float*)res, (const float*)x, (const float*)gate, (float*)out, rows, HW, C);
  DGTD_CHECK_LAUNCH("ca_apply");
  return 0;
}

// scratch fp32 [B*C (dmean) | 64*B*C (per-slice partial sums) | B*2*R*C (per-sample dw)]; dw1 [R,C], dw2 [C,R] overwritten; nothing to zero
static int ca_gate_bwd_impl(const void* g, const void* res, const float* w1, const float* w2, const float* stats, void* dres,
                            float* dw1, float* dw2, float* scratch, int B, int HW, int C, int R, dgtd_dtype dt, dgtd_stream s, int per_sample) {
  DGTD_PROF(s, DGTD_HBM, 3.0 * dgtd_esize(dt) * B * HW * C, "dgtd_ca_gate_bwd[B=%d,HW=%d,C=%d]", B, HW, C);
  const int V = DGTD_IS_HALF(dt) ? 8 : 4;
  DGTD_REQUIRE(B > 0 && HW > 0 && C > 0 && C <= 128 && R > 0 && R <= 32 && C % V == 0, "ca_gate_bwd: unsupported sizes C=%d R=%d", C, R);
  DGTD_REQUIRE(DGTD_IS_HALF(dt) || dt == DGTD_F32, "ca_gate_bwd: bad dtype %d", (int)dt);
  hipStream_t st = (hipStream_t)s;
  const float *pooled = stats, *gate = stats + (size_t)B * C, *hidden = gate + (size_t)B * C;
  float *dmean = scratch, *partial = scratch + (size_t)B * C, *dwp = partial + (size_t)64 * B * C;
  const int cpr = std::max(1, 256 / C), gx = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(HW, cpr * 8), 64));   // <= 64 slices (the stats / scratch layout), summed per column by the per-sample MLP kernels
  if (dt == DGTD_F16) hipLaunchKernelGGL((pooled_sum_kernel<f16_t, true>), dim3(gx, B), dim3(256), 0, st, (const f16_t*)g, (const f16_t*)res, partial, HW, C);
  else if (dt == DGTD_BF16) hipLaunchKernelGGL((pooled_sum_kernel<bf16_t, true>), dim3(gx, B), dim3(256), 0, st, (const bf16_t*)g, (const bf16_t*)res, partial, HW, C);
  else hipLaunchKernelGGL((pooled_sum_kernel<float, true>), dim3(gx, B), dim3(256), 0, st, (const float*)g, (const float*)res, partial, HW, C);
  DGTD_CHECK_LAUNCH("ca_dgate_sum");
  static const bool fused = !(getenv("DGTD_CA_FUSED") && getenv("DGTD_CA_FUSED")[0] == '0');
  if (per_sample || (fused && B <= CA_MAXB)) {
    const int ax = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv((int64_t)HW * (C / V), 256 * 2), 128));
    const size_t lds = (size_t)(416 + (per_sample ? 0 : B * 160)) * sizeof(float);
    if (dt == DGTD_F16) hipLaunchKernelGGL(ca_apply_bwd_fused_kernel<f16_t>, dim3(ax, B), dim3(256), lds, st, (const f16_t*)g, (const float*)partial, gx, gate, hidden, pooled, w1, w2, (f16_t*)dres, dw1, dw2, HW, C, R, per_sample);
    else if (dt == DGTD_BF16) hipLaunchKernelGGL(ca_apply_bwd_fused_kernel<bf16_t>, dim3(ax, B), dim3(256), lds, st, (const bf16_t*)g, (const float*)partial, gx, gate, hidden, pooled, w1, w2, (bf16_t*)dres, dw1, dw2, HW, C, R, per_sample);
    else hipLaunchKernelGGL(ca_apply_bwd_fused_kernel<float>, dim3(ax, B), dim3(256), lds, st, (const float*)g, (const float*)partial, gx, gate, hidden, pooled, w1, w2, (float*)dres, dw1, dw2, HW, C, R, per_sample);
    DGTD_CHECK_LAUNCH("ca_apply_bwd_fused");
    return 0;
  }
  hipLaunchKernelGGL(ca_gate_mlp_bwd_kernel, dim3(B), dim3(128), 0, st, (const float*)partial, gx, gate, hidden, pooled, w1, w2, dmean, dwp, HW, C, R);
  DGTD_CHECK_LAUNCH("ca_gate_mlp_bwd");
  const int64_t rows = (int64_t)B * HW;
  if (dt == DGTD_F16) hipLaunchKernelGGL(ca_apply_bwd_kernel<f16_t>, dim3(ew_grid(rows * (C / V))), dim3(256), 0, st, (const f16_t*)g, gate, (const float*)dmean, (f16_t*)dres, rows, HW, C, (const float*)dwp, dw1, dw2, B, R);
  else if (dt == DGTD_BF16) hipLaunchKernelGGL(ca_apply_bwd_kernel<bf16_t>, dim3(ew_grid(rows * (C / V))), dim3(256), 0, st, (const bf16_t*)g, gate, (const float*)dmean, (bf16_t*)dres, rows, HW, C, (const float*)dwp, dw1, dw2, B, R);
  else hipLaunchKernelGGL(ca_apply_bwd_kernel<float>, dim3(ew_grid(rows * (C / V))), dim3(256), 0, st, (const float*)g, gate, (const float*)dmean, (float*)dres, rows, HW, C, (const float*)dwp, dw1, dw2, B, R);
  DGTD_CHECK_LAUNCH("ca_apply_bwd");
  return 0;
}

extern "C" int dgtd_ca_gate_bwd(const void* g, const void* res, const float* w1, const float* w2, const float* stats, void* dres,
                                float* dw1, float* dw2, float* scratch, int B, int HW, int C, int R, dgtd_dtype dt, dgtd_stream s) {
  return ca_gate_bwd_impl(g, res, w1, w2, stats, dres, dw1, dw2, scratch, B, HW, C, R, dt, s, 0);
}

// the same with the weight gradients left as ONE ROW PER SAMPLE: dw_rows fp32 [B][2 R C] = { dh_b x mean_b | dz_b x hidden_b }; the caller
// sums the rows (csrc_torch: the deferred flush adds them up together with the rows of the module's other calls)
extern "C" int dgtd_ca_gate_bwd_rows(const void* g, const void* res, const float* w1, const float* w2, const float* stats, void* dres,
                                     float* dw_rows, float* scratch, int B, int HW, int C, int R, dgtd_dtype dt, dgtd_stream s) {
  DGTD_REQUIRE(B <= 65535, "ca_gate_bwd_rows: B=%d", B);
  return ca_gate_bwd_impl(g, res, w1, w2, stats, dres, dw_rows, nullptr, scratch, B, HW, C, R, dt, s, 1);
}

extern "C" int dgtd_bilinear_fwd(const void* x, void* y, int B, int Hi, int Wi, int Ho, int Wo, int C, int align_corners, dgtd_dtype dt,
                                 dgtd_stream s) {
  DGTD_PROF(s, DGTD_HBM, 1.0 * dgtd_esize(dt) * B * C * ((double)Hi * Wi + (double)Ho * Wo), "dgtd_bilinear_fwd[%dx%d->%dx%d,C=%d]", Hi, Wi, Ho, Wo, C);
  const int V = DGTD_IS_HALF(dt) ? 8 : 4;
  DGTD_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0 && C % V == 0, "bilinear_fwd: bad sizes (C=%d must be a multiple of %d)", C, V);
  const Axis ah = make_axis(Hi, Ho, align_corners), aw = make_axis(Wi, Wo, align_corners);
  const int grid = ew_grid((int64_t)B * Ho * Wo * (C / V));
  if (dt == DGTD_F16) hipLaunchKernelGGL(bilinear_fwd_kernel<f16_t>, dim3(grid), dim3(256), 0, (hipStream_t)s, (const f16_t*)x, (f16_t*)y, B, Hi, Wi, Ho, Wo, C, ah, aw);
  else if (dt == DGTD_BF16) hipLaunchKernelGGL(bilinear_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, (bf16_t*)y, B, Hi, Wi, Ho, Wo, C, ah, aw);
  else if (dt == DGTD_F32) hipLaunchKernelGGL(bilinear_fwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)s, (const float*)x, (float*)y, B, Hi, Wi, Ho, Wo, C, ah, aw);
  else DGTD_FAIL(2, "bilinear_fwd: bad dtype %d", (int)dt);
  DGTD_CHECK_LAUNCH("bilinear_fwd");
  return 0;
}

extern "C" int dgtd_bilinear_bwd(const void* dy, void* dx, int B, int Hi, int Wi, int Ho, int Wo, int C, int align_corners, dgtd_dtype dt,
                                 dgtd_stream s) {
  DGTD_PROF(s, DGTD_HBM, 1.0 * dgtd_esize(dt) * B * C * ((double)Hi * Wi + (double)Ho * Wo), "dgtd_bilinear_bwd[%dx%d<-%dx%d,C=%d]", Hi, Wi, Ho, Wo, C);
  const int V = DGTD_IS_HALF(dt) ? 8 : 4;
  DGTD_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0 && C % V == 0, "bilinear_bwd: bad sizes (C=%d must be a multiple of %d)", C, V);
  const Axis ah = make_axis(Hi, Ho, align_corners), aw = make_axis(Wi, Wo, align_corners);
  // d(src)/d(out index) = scale, so an input pixel i is touched by outputs around i / scale
  const float inv_h = ah.scale > 0.f ? 1.f / ah.scale : (float)Ho, inv_w = aw.scale > 0.f ? 1.f / aw.scale : (float)Wo;
  // lanes per item by the number of candidate output rows per input pixel (~ 2 x the vertical up-sampling factor + 3)
  const float up = ah.scale > 0.f ? 1.f / ah.scale : (float)Ho;
  const int lpi = up >= 6.f ? 16 : (up >= 3.f ? 8 : (up >= 1.5f ? 2 : 1));
  const int grid = ew_grid((int64_t)B * Hi * Wi * (C / V) * lpi);
#define DGTD_BIL_BWD(T_, L_) hipLaunchKernelGGL((bilinear_bwd_kernel<T_, L_>), dim3(grid), dim3(256), 0, (hipStream_t)s, (const T_*)dy, (T_*)dx, B, Hi, Wi, Ho, Wo, C, ah, aw, inv_h, inv_w)
#define DGTD_BIL_BWD_L(T_) do { if (lpi == 16) DGTD_BIL_BWD(T_, 16); else if (lpi == 8) DGTD_BIL_BWD(T_, 8); else if (lpi == 2) DGTD_BIL_BWD(T_, 2); else DGTD_BIL_BWD(T_, 1); } while (0)
  if (dt == DGTD_F16) DGTD_BIL_BWD_L(f16_t);
  else if (dt == DGTD_BF16) DGTD_BIL_BWD_L(bf16_t);
  else if (dt == DGTD_F32) DGTD_BIL_BWD_L(float);
  else DGTD_FAIL(2, "bilinear_bwd: bad dtype %d", (int)dt);
#undef DGTD_BIL_BWD_L
#undef DGTD_BIL_BWD
  DGTD_CHECK_LAUNCH("bilinear_bwd");
  return 0;
}
