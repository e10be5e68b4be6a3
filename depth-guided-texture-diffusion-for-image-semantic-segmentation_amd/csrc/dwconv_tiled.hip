// dwconv_tiled.hip — LDS-tiled depthwise KxK convolution on NHWC tensors (C % 128 == 0): forward / backward-data (same kernel,
// flipped filter) and the weight gradient.  Same math and entry points as dwconv.hip (which stays the generic fallback).
//
// Why: with lanes along channels there is no reuse between lanes, and the K-fold row reuse of the direct kernel had to come out
// of L1/L2 — measured (rocprofv3 PMC) it was bounded by the L1 path and by waiting, at 5-10 % of the HBM roofline although its
// HBM traffic was already minimal.  Here a workgroup stages the input tile WITH its halo into LDS once (coalesced 16-B copies),
// every lane keeps all K*K taps of its 2 channels in registers, and the inner loop is LDS reads + FMAs only.
//   tile   : TY = 4 output rows (one per wave) x TXW = 16 output columns x 128 channels (64 lanes x 2)
//   LDS    : (TY + K - 1) x (TXW + K - 1) x 128 bf16/fp32  (K = 7: 56 KB bf16)
//   bwd-w  : persistent workgroups keep the K*K x 2 accumulators across tiles; one partial row set per workgroup, fixed-order reduce
#include "common.h"

namespace {

constexpr int TY = 4, TXW = 16, TXS = 8;   // TXS = strip of outputs a lane computes at a time

__device__ __forceinline__ float gelu_t(float x) { return gelu_fast(x); }        // common.h: erfc by A&S 7.1.26, |err| <= 1.5e-7
__device__ __forceinline__ float gelu_grad_t(float x) { return gelu_grad_fast(x); }

template <typename T> struct Pair2;
template <> struct Pair2<float> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct Pair2<bf16_t> { typedef bf16_t type __attribute__((ext_vector_type(2))); };
template <> struct Pair2<f16_t> { typedef f16_t type __attribute__((ext_vector_type(2))); };

// stage rows [y0 - P, y0 + TY + P) x cols [x0 - P, x0 + TXW + P) x channels [cb*128, +128) of image b into LDS, zero padded.
// All of a thread's 16-byte loads are issued before the first LDS store (fully unrolled, registers), so the tile arrives with
// one round trip of latency instead of one per chunk.
template <typename T, int K>
__device__ __forceinline__ void stage_tile(T* __restrict__ tile, const T* __restrict__ x, int b, int y0, int x0, int cb, int H, int W,
                                           int C, int tid) {
  typedef typename Vec16<T>::type VT;
  constexpr int V = Vec16<T>::N, P = K / 2, RW = TXW + K - 1, RH = TY + K - 1, CPP = 128 / V;   // 16-B chunks per position
  constexpr int TOTAL = RH * RW * CPP, NCH = (TOTAL + 255) / 256;
  VT v[NCH];
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int i = tid + k * 256;
    const int ch = i % CPP, pos = i / CPP, xx = pos % RW, yy = pos / RW;
    const int gy = y0 + yy - P, gx = x0 + xx - P;
    if (i < TOTAL && gy >= 0 && gy < H && gx >= 0 && gx < W) v[k] = *reinterpret_cast<const VT*>(x + (((size_t)b * H + gy) * W + gx) * C + cb * 128 + ch * V);
    else
#pragma unroll
      for (int j = 0; j < V; ++j) v[k][j] = (T)0.f;
  }
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int i = tid + k * 256;
    if (i < TOTAL) *reinterpret_cast<VT*>(tile + (size_t)i * V) = v[k];   // pos * 128 + ch * V == i * V
  }
}

// MODE 0: y = conv + bias   MODE 1: y = gelu(conv + bias)   MODE 2: y = aux * gelu'(conv + bias)   MODE 3: y = conv + bias + aux
template <typename T, int K, int MODE>
__global__ __launch_bounds__(256) void dwconv_tiled_fwd_kernel(const T* __restrict__ x, const float* __restrict__ wt,
                                                               const float* __restrict__ bias, const T* __restrict__ aux,
                                                               T* __restrict__ y, int B, int H, int W, int C) {
  typedef typename Pair2<T>::type PT;
  constexpr int P = K / 2, RW = TXW + K - 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* tile = reinterpret_cast<T*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ncb = C / 128, txn = (W + TXW - 1) / TXW, tyn = (H + TY - 1) / TY;
  // XCD-aware remap (guide T1): logically adjacent tiles (next rows of the same column band) share an XCD's L2
  const int nb = gridDim.x, xcd = blockIdx.x & 7, qn = nb >> 3, rn = nb & 7;
  int t = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (blockIdx.x >> 3);
  const int cb = t % ncb; t /= ncb;
  const int ty = t % tyn; t /= tyn;
  const int tx = t % txn;
  const int b = t / txn;
  const int y0 = ty * TY, x0 = tx * TXW, c0 = cb * 128 + lane * 2;
  stage_tile<T, K>(tile, x, b, y0, x0, cb, H, W, C, tid);
  // all K*K taps of this lane's two channels, and the bias, stay in registers (loaded behind the tile: the staging registers are dead)
  // the lane's two channels travel as float2 values: the multiply-adds compile to v_pk_fma_f32 (two FMAs per VALU instruction)
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 w[K * K];
#pragma unroll
  for (int i = 0; i < K * K; ++i) { w[i][0] = wt[(size_t)i * C + c0]; w[i][1] = wt[(size_t)i * C + c0 + 1]; }
  f2 bv;
  bv[0] = bias ? bias[c0] : 0.f; bv[1] = bias ? bias[c0 + 1] : 0.f;
  __syncthreads();
  const int oy = y0 + wave;                       // this wave's output row
  if (oy >= H) return;
#pragma unroll 1
  for (int s = 0; s < TXW / TXS; ++s) {
    const int sx = s * TXS;
    if (x0 + sx >= W) break;
    f2 acc[TXS];
#pragma unroll
    for (int i = 0; i < TXS; ++i) acc[i] = bv;
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      f2 in[TXS + K - 1];
      const T* row = tile + ((size_t)(wave + ky) * RW + sx) * 128 + lane * 2;
#pragma unroll
      for (int i = 0; i < TXS + K - 1; ++i) {
        PT v = *reinterpret_cast<const PT*>(row + (size_t)i * 128);
        in[i][0] = (float)v[0]; in[i][1] = (float)v[1];
      }
#pragma unroll
      for (int kx = 0; kx < K; ++kx)
#pragma unroll
        for (int i = 0; i < TXS; ++i) acc[i] = __builtin_elementwise_fma(in[i + kx], w[ky * K + kx], acc[i]);
    }
#pragma unroll
    for (int i = 0; i < TXS; ++i) {
      const int ox = x0 + sx + i;
      if (ox < W) {
        const size_t o = (((size_t)b * H + oy) * W + ox) * C + c0;
        PT r;
        if (MODE == 2) {
          PT g = *reinterpret_cast<const PT*>(aux + o);
          r[0] = (T)((float)g[0] * gelu_grad_t(acc[i][0])); r[1] = (T)((float)g[1] * gelu_grad_t(acc[i][1]));
        } else if (MODE == 1) {
          r[0] = (T)gelu_t(acc[i][0]); r[1] = (T)gelu_t(acc[i][1]);
        } else if (MODE == 3) {
          PT g = *reinterpret_cast<const PT*>(aux + o);
          r[0] = (T)(acc[i][0] + (float)g[0]); r[1] = (T)(acc[i][1] + (float)g[1]);
        } else {
          r[0] = (T)acc[i][0]; r[1] = (T)acc[i][1];
        }
        *reinterpret_cast<PT*>(y + o) = r;
      }
    }
  }
}

// weight gradient: ws[blockIdx.x][K*K + 1][C] partial rows per (persistent) workgroup
template <typename T, int K>
__global__ __launch_bounds__(256) void dwconv_tiled_bww_kernel(const T* __restrict__ x, const T* __restrict__ du,
                                                               float* __restrict__ ws, int has_bias, int B, int H, int W, int C,
                                                               int tiles_per_cb) {
  typedef typename Pair2<T>::type PT;
  typedef typename Vec16<T>::type VT;
  constexpr int P = K / 2, RW = TXW + K - 1, RH = TY + K - 1, V = Vec16<T>::N, CPP = 128 / V;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* tile = reinterpret_cast<T*>(smem);
  T* gt = tile + (size_t)RH * RW * 128;           // du tile [TY][TXW][128]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cb = blockIdx.y, txn = (W + TXW - 1) / TXW, tyn = (H + TY - 1) / TY;
  float acc[K * K][2], accb[2] = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < K * K; ++i) { acc[i][0] = 0.f; acc[i][1] = 0.f; }
  for (int t = blockIdx.x; t < tiles_per_cb; t += gridDim.x) {
    int r = t;
    const int ty = r % tyn; r /= tyn;              // y fastest: consecutive tiles of a workgroup share halo rows in L2
    const int tx = r % txn;
    const int b = r / txn;
    const int y0 = ty * TY, x0 = tx * TXW;
    __syncthreads();                               // previous tile fully consumed
    stage_tile<T, K>(tile, x, b, y0, x0, cb, H, W, C, tid);
    for (int i = tid; i < TY * TXW * CPP; i += 256) {
      const int ch = i % CPP, pos = i / CPP, xx = pos % TXW, yy = pos / TXW;
      const int gy = y0 + yy, gx = x0 + xx;
      VT v;
      if (gy < H && gx < W) v = *reinterpret_cast<const VT*>(du + (((size_t)b * H + gy) * W + gx) * C + cb * 128 + ch * V);
      else
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] = (T)0.f;
      *reinterpret_cast<VT*>(gt + (size_t)pos * 128 + ch * V) = v;
    }
    __syncthreads();
#pragma unroll 1
    for (int s = 0; s < TXW / TXS; ++s) {
      const int sx = s * TXS;
      float g[TXS][2];
      const T* grow = gt + ((size_t)wave * TXW + sx) * 128 + lane * 2;
#pragma unroll
      for (int i = 0; i < TXS; ++i) {
        PT v = *reinterpret_cast<const PT*>(grow + (size_t)i * 128);
        g[i][0] = (float)v[0]; g[i][1] = (float)v[1];
        accb[0] += g[i][0]; accb[1] += g[i][1];
      }
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        float in[TXS + K - 1][2];
        const T* row = tile + ((size_t)(wave + ky) * RW + sx) * 128 + lane * 2;
#pragma unroll
        for (int i = 0; i < TXS + K - 1; ++i) {
          PT v = *reinterpret_cast<const PT*>(row + (size_t)i * 128);
          in[i][0] = (float)v[0]; in[i][1] = (float)v[1];
        }
#pragma unroll
        for (int kx = 0; kx < K; ++kx)
#pragma unroll
          for (int i = 0; i < TXS; ++i) {
            acc[ky * K + kx][0] += g[i][0] * in[i + kx][0];
            acc[ky * K + kx][1] += g[i][1] * in[i + kx][1];
          }
      }
    }
  }
  // merge the four waves (rows) through LDS, then one plain partial row set per workgroup
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);     // [K*K + 1][128]
  for (int i = tid; i < (K * K + 1) * 128; i += 256) red[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < K * K; ++i) { atomicAdd(&red[i * 128 + lane * 2], acc[i][0]); atomicAdd(&red[i * 128 + lane * 2 + 1], acc[i][1]); }
  atomicAdd(&red[K * K * 128 + lane * 2], accb[0]); atomicAdd(&red[K * K * 128 + lane * 2 + 1], accb[1]);
  __syncthreads();
  float* wsb = ws + (size_t)blockIdx.x * (K * K + 1) * C + cb * 128;
  for (int i = tid; i < (K * K + 1) * 128; i += 256) {
    const int t = i >> 7, c = i & 127;
    wsb[(size_t)t * C + c] = (t < K * K || has_bias) ? red[i] : 0.f;
  }
}

__global__ __launch_bounds__(256) void dwconv_tiled_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, int nblocks, int ncols) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  float s0 = 0.f, s1 = 0.f;
  if (col < ncols) {
    const float* p = ws + col;
    int b = wave;
    for (; b + 12 < nblocks; b += 16) {
      const float v0 = p[(size_t)b * ncols], v1 = p[(size_t)(b + 4) * ncols], v2 = p[(size_t)(b + 8) * ncols], v3 = p[(size_t)(b + 12) * ncols];
      s0 += v0 + v1; s1 += v2 + v3;
    }
    for (; b < nblocks; b += 4) s0 += p[(size_t)b * ncols];
  }
  part[wave][lane] = s0 + s1;
  __syncthreads();
  if (wave == 0 && col < ncols) out[col] = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
}

template <typename T, int K> constexpr size_t fwd_lds() { return (size_t)(TY + K - 1) * (TXW + K - 1) * 128 * sizeof(T); }
template <typename T, int K> constexpr size_t bww_lds() {
  return std::max(fwd_lds<T, K>() + (size_t)TY * TXW * 128 * sizeof(T), (size_t)(K * K + 1) * 128 * sizeof(float));
}

}  // namespace

// ---- entry points used by dwconv.hip's C ABI functions when C % 128 == 0 ------------------------------------------------------
int dgtd_dwconv_tiled_bww_groups(int B, int H, int W, int C) {
  const int64_t tiles = (int64_t)B * cdiv(H, TY) * cdiv(W, TXW);
  return (int)std::max<int64_t>(1, std::min<int64_t>(tiles, 512 / std::max(1, C / 128)));
}

template <typename T, int K>
static int tiled_fwd(const void* x, const float* wt, const float* bias, const void* aux, void* y, int B, int H, int W, int C, int mode,
                     hipStream_t s) {
  const int64_t tiles = (int64_t)B * cdiv(H, TY) * cdiv(W, TXW) * (C / 128);
  DGTD_REQUIRE(tiles < (1LL << 31), "dwconv_tiled_fwd: too many tiles");
  const size_t lds = fwd_lds<T, K>();
#define TILED(MODE) hipLaunchKernelGGL((dwconv_tiled_fwd_kernel<T, K, MODE>), dim3((unsigned)tiles), dim3(256), lds, s, (const T*)x, wt, bias, (const T*)aux, (T*)y, B, H, W, C)
  if (mode == 0) TILED(0); else if (mode == 1) TILED(1); else if (mode == 2) TILED(2); else TILED(3);
#undef TILED
  DGTD_CHECK_LAUNCH("dwconv_tiled_fwd");
  return 0;
}

int dgtd_dwconv_tiled_fwd(const void* x, const float* wt, const float* bias, const void* aux, void* y, int B, int H, int W, int C, int K,
                          int mode, dgtd_dtype dt, hipStream_t s) {
  if (dt == DGTD_BF16) return K == 7 ? tiled_fwd<bf16_t, 7>(x, wt, bias, aux, y, B, H, W, C, mode, s) : tiled_fwd<bf16_t, 3>(x, wt, bias, aux, y, B, H, W, C, mode, s);
  if (dt == DGTD_F16) return K == 7 ? tiled_fwd<f16_t, 7>(x, wt, bias, aux, y, B, H, W, C, mode, s) : tiled_fwd<f16_t, 3>(x, wt, bias, aux, y, B, H, W, C, mode, s);
  return K == 7 ? tiled_fwd<float, 7>(x, wt, bias, aux, y, B, H, W, C, mode, s) : tiled_fwd<float, 3>(x, wt, bias, aux, y, B, H, W, C, mode, s);
}

template <typename T, int K>
static int tiled_bww(const void* x, const void* du, float* grads, int has_bias, void* workspace, int B, int H, int W, int C, hipStream_t s) {
  const int ncb = C / 128, g = dgtd_dwconv_tiled_bww_groups(B, H, W, C);
  const int tiles = B * (int)cdiv(H, TY) * (int)cdiv(W, TXW);
  const size_t lds = bww_lds<T, K>();
  hipLaunchKernelGGL((dwconv_tiled_bww_kernel<T, K>), dim3(g, ncb), dim3(256), lds, s, (const T*)x, (const T*)du, (float*)workspace,
                     has_bias, B, H, W, C, tiles);
  DGTD_CHECK_LAUNCH("dwconv_tiled_bww");
  const int ncols = (K * K + 1) * C;
  hipLaunchKernelGGL(dwconv_tiled_reduce_kernel, dim3((int)cdiv(ncols, 64)), dim3(256), 0, s, (const float*)workspace, grads, g, ncols);
  DGTD_CHECK_LAUNCH("dwconv_tiled_reduce");
  return 0;
}

int dgtd_dwconv_tiled_bww(const void* x, const void* du, float* grads, int has_bias, void* workspace, int B, int H, int W, int C, int K,
                          dgtd_dtype dt, hipStream_t s) {
  if (dt == DGTD_BF16) return K == 7 ? tiled_bww<bf16_t, 7>(x, du, grads, has_bias, workspace, B, H, W, C, s) : tiled_bww<bf16_t, 3>(x, du, grads, has_bias, workspace, B, H, W, C, s);
  if (dt == DGTD_F16) return K == 7 ? tiled_bww<f16_t, 7>(x, du, grads, has_bias, workspace, B, H, W, C, s) : tiled_bww<f16_t, 3>(x, du, grads, has_bias, workspace, B, H, W, C, s);
  return K == 7 ? tiled_bww<float, 7>(x, du, grads, has_bias, workspace, B, H, W, C, s) : tiled_bww<float, 3>(x, du, grads, has_bias, workspace, B, H, W, C, s);
}
