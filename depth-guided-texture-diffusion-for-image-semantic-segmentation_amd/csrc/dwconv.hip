// dwconv.hip — depthwise KxK convolution (stride 1, "same" zero padding) on token-major / NHWC tensors.
// Replaces  convnext_Block.dwconv  7x7 (twig/model/cod.py:1095,1106; 36 blocks)   and
//           DWConv.dwconv 3x3 + bias, fused with the exact-erf GELU of Mlp (cod.py:1523-1531, :854-855; 16 blocks),
// both of which the reference runs in NCHW behind two permute copies.
//
// HBM-bound.  Algorithmic bytes: fwd 2*e*B*H*W*C;  bwd-data 2*e*B*H*W*C;  bwd-weight 2*e*B*H*W*C (+K*K*C*4).
// Mapping: channels are the contiguous dim, so lanes run along C (coalesced rows).
//   7x7 forward / input gradient: LDS-tiled kernels (dwconv_tiled.hip); direct strip kernel kept for C % 128 != 0.
//   3x3 forward / input gradient: sliding window (dwconv3_sw_kernel) - a thread owns 4 x-positions x 4 channels and walks down the
//     image with the three input rows in registers, every row loaded once; modes fuse bias, GELU, GELU' and the skip-gradient add.
//   weight gradient, one layer per launch: K waves per workgroup = the K filter rows of a strip column, per-workgroup partial rows,
//     fixed-order second stage (no atomics anywhere).
//   weight gradient, all same-shaped layers of the step in one launch (deferred phase): dwconv_bww_sw_kernel - one wave
//     owns all K filter rows of a strip and slides down the image, gradient rows in a register window.
//   weights are packed to fp32 tap-major (+ spatially flipped copy + bias) once per step for all layers (dwconv_pack_batched).
// bwd-data is the forward kernel with the spatially flipped filter.
#include "common.h"
#include <stdlib.h>

// LDS-tiled variants (dwconv_tiled.hip), used whenever C % 128 == 0; DGTD_DWCONV_TILED=0 selects the direct kernels for A/B runs
int dgtd_dwconv_tiled_fwd(const void* x, const float* wt, const float* bias, const void* aux, void* y, int B, int H, int W, int C, int K,
                          int mode, dgtd_dtype dt, hipStream_t s);
int dgtd_dwconv_tiled_bww(const void* x, const void* du, float* grads, int has_bias, void* workspace, int B, int H, int W, int C, int K,
                          dgtd_dtype dt, hipStream_t s);
int dgtd_dwconv_tiled_bww_groups(int B, int H, int W, int C);
static bool use_tiled() {
  static const bool on = [] { const char* e = getenv("DGTD_DWCONV_TILED"); return !(e && e[0] == '0'); }();
  return on;
}

namespace {

__device__ __forceinline__ float gelu_f(float x) { return gelu_fast(x); }        // common.h: erfc by A&S 7.1.26, |err| <= 1.5e-7
__device__ __forceinline__ float gelu_grad_f(float x) { return gelu_grad_fast(x); }

// MODE 0: y = conv(x) + bias          MODE 1: y = gelu(conv(x) + bias)
// MODE 2: y = aux * gelu'(conv(x) + bias)   (aux = upstream gradient; recomputes the pre-activation)
// MODE 3: y = conv(x) + bias + aux          (backward w.r.t. x of a residual block: aux = gradient of the skip branch)
template <typename T, int V> struct VecN { typedef T type __attribute__((ext_vector_type(V))); };

template <typename T, int V, int K, int TX, int MODE>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const T* __restrict__ x, const float* __restrict__ wt,
                                                         const float* __restrict__ bias, const T* __restrict__ aux,
                                                         T* __restrict__ y, int B, int H, int W, int C) {
  typedef typename VecN<T, V>::type VT;
  constexpr int P = K / 2;
  const int CV = C / V, XB = (W + TX - 1) / TX;
  const int64_t total = (int64_t)B * H * XB * CV;
  // XCD-aware block remap (guide T1): hardware deals consecutive block ids round-robin over the 8 XCDs, so logically adjacent
  // tiles (the next rows of the same band, which re-read K-1 of the same input rows) would never share an L2.  Give every XCD a
  // contiguous range of logical blocks instead (bijective for any grid size).
  const int nb = gridDim.x, xcd = blockIdx.x & 7, qn = nb >> 3, rn = nb & 7;
  const int lbid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (blockIdx.x >> 3);
  for (int64_t gid = (int64_t)lbid * 256 + threadIdx.x; gid < total; gid += (int64_t)gridDim.x * 256) {
    const int cv = (int)(gid % CV);
    int64_t strip = gid / CV;
    const int xb = (int)(strip % XB); strip /= XB;
    const int yy0 = (int)(strip % H);
    const int b = (int)(strip / H);
    const int x0 = xb * TX, c0 = cv * V;
    float acc[TX][V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float bv = bias ? bias[c0 + j] : 0.f;
#pragma unroll
      for (int t = 0; t < TX; ++t) acc[t][j] = bv;
    }
    // filter rows one at a time (keeps ~1 row of taps live); the raw 16-byte vectors of the NEXT row are requested before the
    // FMAs of the current one so that the L2 latency of a row is covered by the arithmetic of its predecessor
    VT rawc[TX + K - 1], rawn[TX + K - 1];
    auto fetch = [&](int ky, VT* dst) {
      const int yy = yy0 + ky - P;
      const bool rowok = yy >= 0 && yy < H;
      const T* row = x + (((size_t)b * H + (rowok ? yy : 0)) * W) * C + c0;
#pragma unroll
      for (int i = 0; i < TX + K - 1; ++i) {
        const int xx = x0 + i - P;
        VT z;
#pragma unroll
        for (int j = 0; j < V; ++j) z[j] = (T)0.f;
        dst[i] = (rowok && xx >= 0 && xx < W) ? *reinterpret_cast<const VT*>(row + (size_t)xx * C) : z;
      }
    };
    fetch(0, rawc);
#pragma unroll 1
    for (int ky = 0; ky < K; ++ky) {
      if (ky + 1 < K) fetch(ky + 1, rawn);
      float in[TX + K - 1][V];
#pragma unroll
      for (int i = 0; i < TX + K - 1; ++i)
#pragma unroll
        for (int j = 0; j < V; ++j) in[i][j] = (float)rawc[i][j];
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        float wv[V];
        const float* wp = wt + (size_t)(ky * K + kx) * C + c0;
#pragma unroll
        for (int j = 0; j < V; j += 4) {
          f32x4 w4 = *reinterpret_cast<const f32x4*>(wp + j);
          wv[j] = w4[0]; wv[j + 1] = w4[1]; wv[j + 2] = w4[2]; wv[j + 3] = w4[3];
        }
#pragma unroll
        for (int t = 0; t < TX; ++t)
#pragma unroll
          for (int j = 0; j < V; ++j) acc[t][j] += in[t + kx][j] * wv[j];
      }
#pragma unroll
      for (int i = 0; i < TX + K - 1; ++i) rawc[i] = rawn[i];
    }
    T* orow = y + (((size_t)b * H + yy0) * W) * C + c0;
#pragma unroll
    for (int t = 0; t < TX; ++t) {
      const int xx = x0 + t;
      if (xx < W) {
        VT o;
        if (MODE == 2) {
          VT g = *reinterpret_cast<const VT*>(aux + (((size_t)b * H + yy0) * W + xx) * C + c0);
#pragma unroll
          for (int j = 0; j < V; ++j) o[j] = (T)((float)g[j] * gelu_grad_f(acc[t][j]));
        } else if (MODE == 3) {
          VT g = *reinterpret_cast<const VT*>(aux + (((size_t)b * H + yy0) * W + xx) * C + c0);
#pragma unroll
          for (int j = 0; j < V; ++j) o[j] = (T)(acc[t][j] + (float)g[j]);
        } else {
#pragma unroll
          for (int j = 0; j < V; ++j) o[j] = (T)(MODE == 1 ? gelu_f(acc[t][j]) : acc[t][j]);
        }
        *reinterpret_cast<VT*>(orow + (size_t)xx * C) = o;
      }
    }
  }
}

// 3x3, sliding window down the image: a thread owns a strip of TX = 4 x-positions x V = 4 channels and walks RY output rows; the
// three input rows of the window live in registers as floats and every input row is loaded (and converted) ONCE per thread - the
// strip kernel above fetches it once per output row, i.e. three times, and re-reads the 9 x V weights per strip (here: registers).
// Same modes as dwconv_fwd_kernel.  Lanes run along C (64 lanes x 8 B = 512 B contiguous per pixel).
template <typename T, int MODE>
__global__ __launch_bounds__(256) void dwconv3_sw_kernel(const T* __restrict__ x, const float* __restrict__ wt,
                                                         const float* __restrict__ bias, const T* __restrict__ aux,
                                                         T* __restrict__ y, int B, int H, int W, int C, int RY) {
  constexpr int V = 4, TX = 4, K = 3, NI = TX + K - 1;
  typedef typename VecN<T, V>::type VT;
  typedef float f32v __attribute__((ext_vector_type(V)));
  const int CV = C / V, XB = (W + TX - 1) / TX, YC = (H + RY - 1) / RY;
  const int64_t total = (int64_t)B * YC * XB * CV;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int cv = (int)(gid % CV);
  int64_t r = gid / CV;
  const int xb = (int)(r % XB); r /= XB;
  const int yc = (int)(r % YC);
  const int b = (int)(r / YC);
  const int x0 = xb * TX, c0 = cv * V, y_begin = yc * RY, y_end = min(H, y_begin + RY);
  f32v w[K * K], bv = (f32v)0.f;
#pragma unroll
  for (int t = 0; t < K * K; ++t) w[t] = *reinterpret_cast<const f32v*>(wt + (size_t)t * C + c0);
  if (bias) bv = *reinterpret_cast<const f32v*>(bias + c0);
  const T* xb_ = x + (size_t)b * H * W * C + c0;
  auto load_row = [&](int yy, VT* dst) {                       // input row yy (zeros outside the image), columns x0-1 .. x0+TX
    const bool rowok = yy >= 0 && yy < H;
    const T* row = xb_ + (size_t)(rowok ? yy : 0) * W * C;
    VT z;
#pragma unroll
    for (int j = 0; j < V; ++j) z[j] = (T)0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int xx = x0 + i - 1;
      dst[i] = (rowok && xx >= 0 && xx < W) ? *reinterpret_cast<const VT*>(row + (size_t)xx * C) : z;
    }
  };
  auto cvt = [&](const VT* raw, f32v* dst) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < V; ++j) dst[i][j] = (float)raw[i][j];
  };
  f32v r0[NI], r1[NI], r2[NI];                                   // rows y-1, y, y+1 of the window
  VT raw[NI];
  load_row(y_begin - 1, raw); cvt(raw, r0);
  load_row(y_begin, raw); cvt(raw, r1);
  load_row(y_begin + 1, raw);
  for (int yy = y_begin; yy < y_end; ++yy) {
    cvt(raw, r2);
    if (yy + 1 < y_end) load_row(yy + 2, raw);                  // in flight during the arithmetic of this row
    VT g[TX];
    if (MODE >= 2) {
#pragma unroll
      for (int t = 0; t < TX; ++t)
        if (x0 + t < W) g[t] = *reinterpret_cast<const VT*>(aux + (((size_t)b * H + yy) * W + x0 + t) * C + c0);
    }
    T* orow = y + (((size_t)b * H + yy) * W) * C + c0;
#pragma unroll
    for (int t = 0; t < TX; ++t) {
      f32v acc = bv;
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        acc = __builtin_elementwise_fma(r0[t + kx], w[kx], acc);
        acc = __builtin_elementwise_fma(r1[t + kx], w[K + kx], acc);
        acc = __builtin_elementwise_fma(r2[t + kx], w[2 * K + kx], acc);
      }
      if (x0 + t < W) {
        VT o;
#pragma unroll
        for (int j = 0; j < V; ++j) {
          float v = acc[j];
          if (MODE == 1) v = gelu_f(v);
          else if (MODE == 2) v = (float)g[t][j] * gelu_grad_f(v);
          else if (MODE == 3) v = v + (float)g[t][j];
          o[j] = (T)v;
        }
        *reinterpret_cast<VT*>(orow + (size_t)(x0 + t) * C) = o;
      }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) { r0[i] = r1[i]; r1[i] = r2[i]; }
  }
}

template <typename T>
int fwd3_sw_launch(const void* x, const float* wt, const float* bias, const void* aux, void* y, int B, int H, int W, int C, int mode, hipStream_t s) {
  // rows per thread: long enough to amortise the two halo rows, short enough for >= ~16 waves per CU
  int RY = 32;
  while (RY > 8 && (int64_t)B * cdiv(H, RY) * cdiv(W, 4) * (C / 4) < (int64_t)256 * 16 * 64) RY /= 2;
  const int64_t total = (int64_t)B * cdiv(H, RY) * cdiv(W, 4) * (C / 4);
  const dim3 grid((unsigned)cdiv(total, 256));
#define DW3_LAUNCH(MODE) hipLaunchKernelGGL((dwconv3_sw_kernel<T, MODE>), grid, dim3(256), 0, s, (const T*)x, wt, bias, (const T*)aux, (T*)y, B, H, W, C, RY)
  if (mode == 0) DW3_LAUNCH(0); else if (mode == 1) DW3_LAUNCH(1); else if (mode == 2) DW3_LAUNCH(2); else DW3_LAUNCH(3);
#undef DW3_LAUNCH
  DGTD_CHECK_LAUNCH("dwconv3_sw");
  return 0;
}

// 2 channels per lane (one 4-byte bf16x2 / 8-byte float2 load); one wave = 128 channels of one strip.
template <typename T> struct Pair;
template <> struct Pair<float> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct Pair<bf16_t> { typedef bf16_t type __attribute__((ext_vector_type(2))); };
template <> struct Pair<f16_t> { typedef f16_t type __attribute__((ext_vector_type(2))); };

// ws[gridDim.x][K*K + 1][C] -> out[(K*K + 1) * C] (= { dw_t | db }); fixed summation order, no atomics
__global__ __launch_bounds__(256) void dwconv_bww_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, int nblocks,
                                                                int ncols) {
  __shared__ float part[8][32];     // 32 columns x 8 row groups per workgroup, 8 loads in flight per lane
  const int lane = threadIdx.x & 31, rgp = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + lane;
  float s0 = 0.f, s1 = 0.f;
  if (col < ncols) {
    const float* p = ws + col;
    int b = rgp;
    for (; b + 56 < nblocks; b += 64) {
      const float v0 = p[(size_t)b * ncols], v1 = p[(size_t)(b + 8) * ncols], v2 = p[(size_t)(b + 16) * ncols], v3 = p[(size_t)(b + 24) * ncols];
      const float v4 = p[(size_t)(b + 32) * ncols], v5 = p[(size_t)(b + 40) * ncols], v6 = p[(size_t)(b + 48) * ncols], v7 = p[(size_t)(b + 56) * ncols];
      s0 += (v0 + v1) + (v2 + v3); s1 += (v4 + v5) + (v6 + v7);
    }
    for (; b < nblocks; b += 8) s0 += p[(size_t)b * ncols];
  }
  part[rgp][lane] = s0 + s1;
  __syncthreads();
  if (rgp == 0 && col < ncols) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) v += part[k][lane];
    out[col] = v;
  }
}

// One workgroup = K waves; wave w owns filter row ky = w.  All K waves walk the SAME column of strips (fixed image, x-range and
// 128-channel block; y ascending), so the gradient strip and K-1 of the K input rows each wave needs were just fetched by a
// sibling wave on the same CU: the K-fold row reuse is served by that CU's L1 instead of L2/MALL.  A wave keeps K accumulators x
// 2 channels per lane and writes its own K x 128 partial rows at the end (no cross-wave reduction; the bias row comes from the
// centre wave); partials of all workgroups are summed by dwconv_bww_reduce_kernel in a fixed order.
template <typename T, int K, int TX>
__global__ __launch_bounds__(K * 64) void dwconv_bwd_weight_kernel(const T* __restrict__ x, const T* __restrict__ du,
                                                                   float* __restrict__ ws, int has_bias,
                                                                   int B, int H, int W, int C, int ysplit) {
  typedef typename Pair<T>::type PT;
  constexpr int P = K / 2;
  const int lane = threadIdx.x & 63, ky = threadIdx.x >> 6;
  const int c0 = blockIdx.y * 128 + lane * 2;
  const int XB = (W + TX - 1) / TX;
  // blockIdx.x enumerates (b, xb, ypart); XCD-aware remap keeps the parts of one column on one XCD
  const int nbx = gridDim.x, xcd = blockIdx.x & 7, qn = nbx >> 3, rn = nbx & 7;
  const int col = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (blockIdx.x >> 3);
  const int ypart = col % ysplit, xb = (col / ysplit) % XB, b = col / (ysplit * XB);
  const int rows_per = (H + ysplit - 1) / ysplit;
  const int y_begin = ypart * rows_per, y_end = min(H, y_begin + rows_per);
  const int x0 = xb * TX;
  float acc[K][2], accb[2] = {0.f, 0.f};
#pragma unroll
  for (int t = 0; t < K; ++t) { acc[t][0] = 0.f; acc[t][1] = 0.f; }
  // rows of this wave: yy = yy0 + ky - P must lie inside the image
  const int ya = max(y_begin, P - ky), yb = min(y_end, H + P - ky);
  const bool interior = x0 - P >= 0 && x0 + TX + P <= W;      // wave-uniform: straight-line loads
  // The packed (2 x bf16 / 2 x fp32) values of the NEXT row are fetched before the FMAs of the current one: a wave otherwise
  // alternates between 22 loads and ~150 VALU instructions with the full L2 latency exposed in between (6 waves per SIMD do
  // not cover it); the raw pairs cost one register each, the converted floats exist only for the row being processed.
  PT graw[TX], inraw[TX + K - 1], gnext[TX], innext[TX + K - 1];
  auto fetch = [&](int yy0, PT* gr, PT* ir) {
    const T* grow = du + (((size_t)b * H + yy0) * W + x0) * C + c0;
    const T* row = x + (((size_t)b * H + (yy0 + ky - P)) * W + x0 - P) * C + c0;
    if (interior) {
#pragma unroll
      for (int t = 0; t < TX; ++t) gr[t] = *reinterpret_cast<const PT*>(grow + t * C);
#pragma unroll
      for (int i = 0; i < TX + K - 1; ++i) ir[i] = *reinterpret_cast<const PT*>(row + i * C);
    } else {
#pragma unroll
      for (int t = 0; t < TX; ++t) {
        PT z; z[0] = (T)0.f; z[1] = (T)0.f;
        gr[t] = (x0 + t < W) ? *reinterpret_cast<const PT*>(grow + t * C) : z;
      }
#pragma unroll
      for (int i = 0; i < TX + K - 1; ++i) {
        const int xx = x0 + i - P;
        PT z; z[0] = (T)0.f; z[1] = (T)0.f;
        ir[i] = (xx >= 0 && xx < W) ? *reinterpret_cast<const PT*>(row + i * C) : z;
      }
    }
  };
  if (ya < yb) fetch(ya, graw, inraw);
  for (int yy0 = ya; yy0 < yb; ++yy0) {
    const bool more = yy0 + 1 < yb;
    if (more) fetch(yy0 + 1, gnext, innext);
    float g[TX][2], in[TX + K - 1][2];
#pragma unroll
    for (int t = 0; t < TX; ++t) { g[t][0] = (float)graw[t][0]; g[t][1] = (float)graw[t][1]; }
#pragma unroll
    for (int i = 0; i < TX + K - 1; ++i) { in[i][0] = (float)inraw[i][0]; in[i][1] = (float)inraw[i][1]; }
    if (ky == P) {
#pragma unroll
      for (int t = 0; t < TX; ++t) { accb[0] += g[t][0]; accb[1] += g[t][1]; }
    }
#pragma unroll
    for (int kx = 0; kx < K; ++kx)
#pragma unroll
      for (int t = 0; t < TX; ++t) {
        acc[kx][0] += g[t][0] * in[t + kx][0];
        acc[kx][1] += g[t][1] * in[t + kx][1];
      }
    if (more) {
#pragma unroll
      for (int t = 0; t < TX; ++t) graw[t] = gnext[t];
#pragma unroll
      for (int i = 0; i < TX + K - 1; ++i) inraw[i] = innext[i];
    }
  }
  // ws[blockIdx.x][K*K + 1][C]: this wave's K tap rows (+ the bias row from the centre wave), plain 8-byte stores
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  float* wsb = ws + (size_t)blockIdx.x * (K * K + 1) * C + c0;
#pragma unroll
  for (int t = 0; t < K; ++t) { f32x2 v; v[0] = acc[t][0]; v[1] = acc[t][1]; *reinterpret_cast<f32x2*>(wsb + (size_t)(ky * K + t) * C) = v; }
  if (ky == P) { f32x2 v; v[0] = has_bias ? accb[0] : 0.f; v[1] = has_bias ? accb[1] : 0.f; *reinterpret_cast<f32x2*>(wsb + (size_t)(K * K) * C) = v; }
}

// The same weight gradient for up to BW_MAX same-shaped layers in ONE launch (deferred weight-gradient phase: a layer's (input,
// output-gradient) pair stays in HBM until the backward pass is over, then every depthwise layer of one shape is served together).
// One layer at [8,32,32,512] is 16 MB: 1024 workgroups of 4 rows each, 4x its bytes in partial sums, and a launch that is over
// before the chip is busy.  27 such layers are 453 MB: here a workgroup (K waves = K filter rows, as above) walks a whole row range
// of one image x 128 channels with the accumulators in registers (x fastest, so the column halo re-reads hit L1/L2 at once) and
// writes ONE partial: partial traffic drops from 4x the input to a few per cent, and the launch is a plain HBM stream.
constexpr int BW_MAX = 32;
struct BwwTable { const void* x[BW_MAX]; const void* du[BW_MAX]; };

template <typename T, int K, int TX>
__global__ __launch_bounds__(K * 64) void dwconv_bww_batched_kernel(BwwTable tab, float* __restrict__ ws, int has_bias,
                                                                    int B, int H, int W, int C, int ysplit) {
  typedef typename Pair<T>::type PT;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  constexpr int P = K / 2;
  // The kernel is VALU-bound (56 packed FMAs + 44 conversions per 22 loads), so everything that can be wave-uniform is kept on the
  // scalar unit: the filter row (readfirstlane), the row base pointers and the boundary tests; the only per-lane address part is
  // the channel offset, which lets the loads take the scalar-base + VGPR-offset form.
  const int lane = threadIdx.x & 63, ky = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane2 = lane * 2;
  const int ypart = blockIdx.x % ysplit, b = (blockIdx.x / ysplit) % B, z = blockIdx.x / (ysplit * B);
  const size_t img = (size_t)b * H * W * C + (size_t)blockIdx.y * 128;
  const T* __restrict__ x = (const T*)tab.x[z] + img;
  const T* __restrict__ du = (const T*)tab.du[z] + img;
  const int rows_per = (H + ysplit - 1) / ysplit;
  const int y_begin = ypart * rows_per, y_end = min(H, y_begin + rows_per);
  f32x2 acc[K], accb = {0.f, 0.f};
#pragma unroll
  for (int t = 0; t < K; ++t) acc[t] = f32x2{0.f, 0.f};
  const int ya = max(y_begin, P - ky), yb = min(y_end, H + P - ky);
  const int XB = (W + TX - 1) / TX;
  const int nit = max(0, yb - ya) * XB;
  int fy = ya, fx = 0;                                            // coordinates of the NEXT fetch (x fastest)
  auto fetch = [&](PT* gr, PT* ir) {
    const T* grow = du + ((size_t)fy * W + fx) * C;
    const T* row = x + ((ptrdiff_t)(fy + ky - P) * W + (fx - P)) * (ptrdiff_t)C;
    if (fx - P >= 0 && fx + TX + P <= W) {                        // straight-line loads
#pragma unroll
      for (int t = 0; t < TX; ++t) gr[t] = *reinterpret_cast<const PT*>(grow + t * C + lane2);
#pragma unroll
      for (int i = 0; i < TX + K - 1; ++i) ir[i] = *reinterpret_cast<const PT*>(row + i * C + lane2);
    } else {
      PT zero; zero[0] = (T)0.f; zero[1] = (T)0.f;
#pragma unroll
      for (int t = 0; t < TX; ++t) gr[t] = (fx + t < W) ? *reinterpret_cast<const PT*>(grow + t * C + lane2) : zero;
#pragma unroll
      for (int i = 0; i < TX + K - 1; ++i) {
        const int xx = fx + i - P;
        ir[i] = (xx >= 0 && xx < W) ? *reinterpret_cast<const PT*>(row + i * C + lane2) : zero;
      }
    }
    fx += TX;
    if (fx >= W) { fx = 0; ++fy; }
  };
  auto compute = [&](const PT* gr, const PT* ir) {
    f32x2 g[TX], in[TX + K - 1];
#pragma unroll
    for (int t = 0; t < TX; ++t) g[t] = f32x2{(float)gr[t][0], (float)gr[t][1]};
#pragma unroll
    for (int i = 0; i < TX + K - 1; ++i) in[i] = f32x2{(float)ir[i][0], (float)ir[i][1]};
    if (ky == P) {
#pragma unroll
      for (int t = 0; t < TX; ++t) accb += g[t];
    }
#pragma unroll
    for (int kx = 0; kx < K; ++kx)
#pragma unroll
      for (int t = 0; t < TX; ++t) acc[kx] = __builtin_elementwise_fma(g[t], in[t + kx], acc[kx]);
  };
  // two register sets, alternating roles: the loads of tile it+1 are in flight while tile it is consumed, without register copies
  PT gA[TX], iA[TX + K - 1], gB[TX], iB[TX + K - 1];
  if (nit > 0) fetch(gA, iA);
  for (int it = 0; it < nit; it += 2) {
    if (it + 1 < nit) fetch(gB, iB);
    compute(gA, iA);
    if (it + 1 >= nit) break;
    if (it + 2 < nit) fetch(gA, iA);
    compute(gB, iB);
  }
  // ws[z][b * ysplit + ypart][K*K + 1][C]
  float* wsb = ws + (size_t)blockIdx.x * (K * K + 1) * C + blockIdx.y * 128 + lane2;
#pragma unroll
  for (int t = 0; t < K; ++t) *reinterpret_cast<f32x2*>(wsb + (size_t)(ky * K + t) * C) = acc[t];
  if (ky == P) *reinterpret_cast<f32x2*>(wsb + (size_t)(K * K) * C) = has_bias ? accb : f32x2{0.f, 0.f};
}

// Sliding-window variant of the batched weight gradient (the default): ONE wave owns all K filter rows of an 8-column strip x 128
// channels and walks down the image.  Each input row is loaded once per wave (the K-waves-per-workgroup kernel above loads it K times,
// and its 7x L1/L2 re-read volume, not HBM, bounds it: 1.3 TB/s at 27 x [8,32,32,512]); the K gradient rows it meets (output rows
// r-P..r+P) sit in a register window that shifts by one row per step.  K*K accumulators x 2 channels per lane; the 4 waves of a
// workgroup (4 neighbouring strips) add their partials through LDS in a fixed order, so a workgroup writes one partial row set.
template <typename T, int K, int TX, int NW>
__global__ __launch_bounds__(NW * 64) void dwconv_bww_sw_kernel(BwwTable tab, float* __restrict__ ws, int has_bias,
                                                                int B, int H, int W, int C, int ysplit, int xgroups) {
  typedef typename Pair<T>::type PT;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  constexpr int P = K / 2, KK = K * K;
  extern __shared__ float red[];                                  // [NW - 1][KK + 1][128]
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane2 = lane * 2;
  int rest = blockIdx.x;
  const int xg = rest % xgroups; rest /= xgroups;
  const int ypart = rest % ysplit; rest /= ysplit;
  const int b = rest % B, z = rest / B;
  const size_t img = (size_t)b * H * W * C + (size_t)blockIdx.y * 128;
  const T* __restrict__ x = (const T*)tab.x[z] + img;
  const T* __restrict__ du = (const T*)tab.du[z] + img;
  const int rows_per = (H + ysplit - 1) / ysplit;
  const int y_begin = ypart * rows_per, y_end = min(H, y_begin + rows_per);
  const int x0 = (xg * NW + wv) * TX;                             // this wave's strip (may lie outside the image: contributes zeros)
  f32x2 acc[K][K], accb = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < K; ++i)
#pragma unroll
    for (int j = 0; j < K; ++j) acc[i][j] = f32x2{0.f, 0.f};
  if (x0 < W && y_begin < y_end) {
    PT zero; zero[0] = (T)0.f; zero[1] = (T)0.f;
    const bool interior = x0 - P >= 0 && x0 + TX + P <= W;
    auto load_in = [&](int r, PT* ir) {                           // input row r (inside the image), columns x0-P .. x0+TX+P-1
      const T* row = x + ((ptrdiff_t)r * W + (x0 - P)) * (ptrdiff_t)C;
      if (interior) {
#pragma unroll
        for (int i = 0; i < TX + K - 1; ++i) ir[i] = *reinterpret_cast<const PT*>(row + i * C + lane2);
      } else {
#pragma unroll
        for (int i = 0; i < TX + K - 1; ++i) {
          const int xx = x0 + i - P;
          ir[i] = (xx >= 0 && xx < W) ? *reinterpret_cast<const PT*>(row + i * C + lane2) : zero;
        }
      }
    };
    auto load_g = [&](int y, PT* gr) {                            // gradient row y; zeros outside this workgroup's row range
      if (y >= y_begin && y < y_end) {
        const T* grow = du + ((size_t)y * W + x0) * C;
        if (x0 + TX <= W) {
#pragma unroll
          for (int t = 0; t < TX; ++t) gr[t] = *reinterpret_cast<const PT*>(grow + t * C + lane2);
        } else {
#pragma unroll
          for (int t = 0; t < TX; ++t) gr[t] = (x0 + t < W) ? *reinterpret_cast<const PT*>(grow + t * C + lane2) : zero;
        }
      } else {
#pragma unroll
        for (int t = 0; t < TX; ++t) gr[t] = zero;
      }
    };
    // input rows this range needs: r in [max(0, y_begin-P), min(H, y_end+P)); at step r the window holds gradient rows
    // gw[j] = row r + P - j ... i.e. filter row ky = j pairs input row r with output row y = r - ky + P
    const int r0 = max(0, y_begin - P), r1 = min(H, y_end + P);
    PT gw[K][TX], in_cur[TX + K - 1], in_nxt[TX + K - 1], g_nxt[TX];
#pragma unroll
    for (int j = 0; j < K; ++j) load_g(r0 + P - j, gw[j]);
    load_in(r0, in_cur);
    for (int r = r0; r < r1; ++r) {
      const bool more = r + 1 < r1;
      if (more) { load_in(r + 1, in_nxt); load_g(r + 1 + P, g_nxt); }
      f32x2 in[TX + K - 1];
#pragma unroll
      for (int i = 0; i < TX + K - 1; ++i) in[i] = f32x2{(float)in_cur[i][0], (float)in_cur[i][1]};
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const int y = r - ky + P;
        if (y >= y_begin && y < y_end) {                          // wave-uniform
#pragma unroll
          for (int t = 0; t < TX; ++t) {
            const f32x2 g = f32x2{(float)gw[ky][t][0], (float)gw[ky][t][1]};
            if (ky == P) accb += g;                               // y == r: every gradient row exactly once
#pragma unroll
            for (int kx = 0; kx < K; ++kx) acc[ky][kx] = __builtin_elementwise_fma(g, in[t + kx], acc[ky][kx]);
          }
        }
      }
      if (more) {
#pragma unroll
        for (int j = K - 1; j > 0; --j)
#pragma unroll
          for (int t = 0; t < TX; ++t) gw[j][t] = gw[j - 1][t];
#pragma unroll
        for (int t = 0; t < TX; ++t) gw[0][t] = g_nxt[t];
#pragma unroll
        for (int i = 0; i < TX + K - 1; ++i) in_cur[i] = in_nxt[i];
      }
    }
  }
  // waves 1..NW-1 park their partials in LDS; wave 0 adds them in wave order and writes ws[blockIdx.x][KK + 1][C]
  if (wv > 0) {
    float* mine = red + (size_t)(wv - 1) * (KK + 1) * 128 + lane2;
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
      for (int j = 0; j < K; ++j) *reinterpret_cast<f32x2*>(mine + (i * K + j) * 128) = acc[i][j];
    *reinterpret_cast<f32x2*>(mine + KK * 128) = accb;
  }
  __syncthreads();
  if (wv == 0) {
    float* wsb = ws + (size_t)blockIdx.x * (KK + 1) * C + blockIdx.y * 128 + lane2;
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
      for (int j = 0; j < K; ++j) {
        f32x2 v = acc[i][j];
        for (int w2 = 0; w2 < NW - 1; ++w2) v += *reinterpret_cast<const f32x2*>(red + ((size_t)w2 * (KK + 1) + i * K + j) * 128 + lane2);
        *reinterpret_cast<f32x2*>(wsb + (size_t)(i * K + j) * C) = v;
      }
    f32x2 vb = accb;
    for (int w2 = 0; w2 < NW - 1; ++w2) vb += *reinterpret_cast<const f32x2*>(red + ((size_t)w2 * (KK + 1) + KK) * 128 + lane2);
    *reinterpret_cast<f32x2*>(wsb + (size_t)KK * C) = has_bias ? vb : f32x2{0.f, 0.f};
  }
}

// weights [C, K*K] (Conv2d layout, dtype T) + bias [C] -> packed fp32 [ wt (K*K x C) | wt spatially flipped | bias ]
template <typename T>
__global__ __launch_bounds__(256) void dwconv_pack_kernel(const T* __restrict__ w, const T* __restrict__ bias,
                                                          float* __restrict__ packed, int C, int KK) {
  const int i = blockIdx.x * 256 + threadIdx.x;   // i = t * C + c
  if (i < KK * C) {
    const int t = i / C, c = i % C;
    const float v = (float)w[(size_t)c * KK + t];
    packed[i] = v;
    packed[(size_t)KK * C + (size_t)(KK - 1 - t) * C + c] = v;
  }
  if (i < C) packed[(size_t)2 * KK * C + i] = bias ? (float)bias[i] : 0.f;
}

// the same packing for up to PK_MAX layers in one launch (weights only change between steps: all layers are re-packed together at the
// start of a step instead of one 5 us launch in front of every depthwise convolution)
constexpr int PK_MAX = 64;
struct PackTable { const void* w[PK_MAX]; const void* b[PK_MAX]; float* out[PK_MAX]; int C[PK_MAX]; int KK[PK_MAX]; };
template <typename T>
__global__ __launch_bounds__(256) void dwconv_pack_batched_kernel(PackTable t) {
  const int e = blockIdx.y, C = t.C[e], KK = t.KK[e];
  const int i = blockIdx.x * 256 + threadIdx.x;   // i = tap * C + c
  const T* __restrict__ w = (const T*)t.w[e];
  const T* __restrict__ bias = (const T*)t.b[e];
  float* __restrict__ packed = t.out[e];
  if (i < KK * C) {
    const int tap = i / C, c = i % C;
    const float v = (float)w[(size_t)c * KK + tap];
    packed[i] = v;
    packed[(size_t)KK * C + (size_t)(KK - 1 - tap) * C + c] = v;
  }
  if (i < C) packed[(size_t)2 * KK * C + i] = bias ? (float)bias[i] : 0.f;
}

// grads fp32 [ dwt (K*K x C) | db (C) ] -> dweight [C, K*K] and dbias [C] in dtype T
template <typename T>
__global__ __launch_bounds__(256) void dwconv_unpack_kernel(const float* __restrict__ g, T* __restrict__ dw,
                                                            T* __restrict__ db, int C, int KK) {
  const int i = blockIdx.x * 256 + threadIdx.x;   // i = c * KK + t
  if (i < KK * C) {
    const int c = i / KK, t = i % KK;
    dw[i] = (T)g[(size_t)t * C + c];
  }
  if (db && i < C) db[i] = (T)g[(size_t)KK * C + i];
}

template <typename T, int V, int K, int TX>
int fwd_launch(const void* x, const float* wt, const float* bias, const void* aux, void* y, int B, int H, int W, int C,
               int mode, hipStream_t s) {
  DGTD_REQUIRE(C % V == 0, "dwconv: C=%d must be a multiple of %d", C, V);
  const int64_t total = (int64_t)B * H * cdiv(W, TX) * (C / V);
  const int grid = (int)std::min<int64_t>(cdiv(total, 256), 256 * 16);
#define DW_LAUNCH(MODE) hipLaunchKernelGGL((dwconv_fwd_kernel<T, V, K, TX, MODE>), dim3(grid), dim3(256), 0, s, (const T*)x, wt, bias, (const T*)aux, (T*)y, B, H, W, C)
  if (mode == 0) DW_LAUNCH(0); else if (mode == 1) DW_LAUNCH(1); else if (mode == 2) DW_LAUNCH(2); else DW_LAUNCH(3);
#undef DW_LAUNCH
  DGTD_CHECK_LAUNCH("dwconv_fwd");
  return 0;
}

// columns = B * ceil(W/TX) strip columns per channel block; split each column into `ysplit` row ranges until ~1024 workgroups exist
static int bww_ysplit(int B, int H, int W, int C, int TX, int K) {
  const int64_t cols = (int64_t)B * cdiv(W, TX) * (C / 128);
  static const int64_t t3 = getenv("DGTD_BWW_TARGET_K3") ? atol(getenv("DGTD_BWW_TARGET_K3")) : 4096;
  static const int64_t t7 = getenv("DGTD_BWW_TARGET_K7") ? atol(getenv("DGTD_BWW_TARGET_K7")) : 1024;
  const int64_t target = K == 3 ? t3 : t7;       // workgroups are K waves: keep ~8-12k waves in flight
  int ys = 1;
  while (ys < H && cols * ys < target && (H / (ys * 2)) >= 4) ys *= 2;
  return ys;
}

template <typename T, int K, int TX>
int bww_launch(const void* x, const void* du, float* grads, int has_bias, void* workspace, int B, int H, int W, int C, hipStream_t s,
               int* nblocks = nullptr) {
  DGTD_REQUIRE(C % 128 == 0, "dwconv_bwd_weight: C=%d must be a multiple of 128", C);
  const int ncb = C / 128, ys = bww_ysplit(B, H, W, C, TX, K);
  const int gx = B * (int)cdiv(W, TX) * ys;
  hipLaunchKernelGGL((dwconv_bwd_weight_kernel<T, K, TX>), dim3(gx, ncb), dim3(K * 64), 0, s, (const T*)x, (const T*)du, (float*)workspace,
                     has_bias, B, H, W, C, ys);
  DGTD_CHECK_LAUNCH("dwconv_bwd_weight");
  if (nblocks) { *nblocks = gx; return 0; }
  const int ncols = (K * K + 1) * C;
  hipLaunchKernelGGL(dwconv_bww_reduce_kernel, dim3((int)cdiv(ncols, 32)), dim3(256), 0, s, (const float*)workspace, grads, gx, ncols);
  DGTD_CHECK_LAUNCH("dwconv_bww_reduce");
  return 0;
}

// Launch geometry of the batched weight gradient.  Sliding-window kernel (default): a workgroup = 4 neighbouring 8-column strips x a
// row range x 128 channels; row ranges are halved until ~1024 workgroups exist but never below 16 rows (each range re-reads K-1 halo
// rows).  DGTD_BWW_SLIDING=0 selects the K-waves-per-workgroup kernel (whole-width row ranges).
struct BwwGeom { int ys, xg, blocks; bool sliding; };
static BwwGeom bww_batched_geom(int n, int B, int H, int W, int C, int K) {
  static const bool sliding = !(getenv("DGTD_BWW_SLIDING") && atoi(getenv("DGTD_BWW_SLIDING")) == 0);
  static const int64_t target = getenv("DGTD_BWW_BATCHED_WGS") ? atol(getenv("DGTD_BWW_BATCHED_WGS")) : (sliding ? 1024 : 1536);
  BwwGeom g;
  g.sliding = sliding;
  g.xg = sliding ? (int)cdiv(cdiv(W, 8), 4) : 1;
  g.ys = 1;
  const int64_t wgs1 = (int64_t)n * B * (C / 128) * g.xg;
  // measured (n = 27 x [8,32,32,512]): 1728 workgroups of 16 rows beat 864 of 32 although each re-reads K-1 halo rows
  const int min_rows = sliding ? (int)(getenv("DGTD_BWW_MIN_ROWS") ? atol(getenv("DGTD_BWW_MIN_ROWS")) : 8) : 4;
  while (wgs1 * g.ys < target && H / (g.ys * 2) >= min_rows) g.ys *= 2;
  g.blocks = B * g.ys * g.xg;
  return g;
}

template <typename T, int K, int TX>
int bww_batched_launch(const void* const* x, const void* const* du, int n, int has_bias, float* ws, int B, int H, int W, int C, const BwwGeom& g,
                       hipStream_t s) {
  constexpr int NW = 4;
  for (int z0 = 0; z0 < n; z0 += BW_MAX) {
    const int m = std::min(BW_MAX, n - z0);
    BwwTable tab;
    for (int i = 0; i < m; ++i) { tab.x[i] = x[z0 + i]; tab.du[i] = du[z0 + i]; }
    for (int i = m; i < BW_MAX; ++i) { tab.x[i] = nullptr; tab.du[i] = nullptr; }
    float* wsz = ws + (size_t)z0 * g.blocks * (K * K + 1) * C;
    if (g.sliding) {
      const size_t lds = (size_t)(NW - 1) * (K * K + 1) * 128 * sizeof(float);
      static bool once = [] { return hipFuncSetAttribute((const void*)dwconv_bww_sw_kernel<T, K, TX, NW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                         (int)((NW - 1) * (K * K + 1) * 128 * sizeof(float))) == hipSuccess; }();
      (void)once;
      hipLaunchKernelGGL((dwconv_bww_sw_kernel<T, K, TX, NW>), dim3(m * g.blocks, C / 128), dim3(NW * 64), lds, s, tab, wsz, has_bias, B, H, W, C, g.ys,
                         g.xg);
    } else {
      hipLaunchKernelGGL((dwconv_bww_batched_kernel<T, K, TX>), dim3(m * g.blocks, C / 128), dim3(K * 64), 0, s, tab, wsz, has_bias, B, H, W, C, g.ys);
    }
    DGTD_CHECK_LAUNCH("dwconv_bww_batched");
  }
  return 0;
}

}  // namespace

extern "C" int dgtd_dwconv_bwd_weight_batched_blocks(int n, int B, int H, int W, int C, int K) {
  (void)K;
  return C % 128 == 0 && n > 0 ? bww_batched_geom(n, B, H, W, C, K).blocks : 0;
}

extern "C" int dgtd_dwconv_bwd_weight_batched(const void* const* x, const void* const* du, int n, int has_bias, void* workspace, int B, int H,
                                              int W, int C, int K, dgtd_dtype dt, dgtd_stream s) {
  DGTD_REQUIRE(n > 0 && x && du && workspace, "dwconv_bwd_weight_batched: no layers");
  DGTD_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0, "dwconv_bwd_weight_batched: bad sizes");
  DGTD_REQUIRE(K == 3 || K == 7, "dwconv_bwd_weight_batched: K=%d (only 3 and 7 are on the path)", K);
  DGTD_REQUIRE(DGTD_IS_HALF(dt) || dt == DGTD_F32, "dwconv_bwd_weight_batched: bad dtype %d", (int)dt);
  DGTD_REQUIRE(C % 128 == 0, "dwconv_bwd_weight_batched: C=%d must be a multiple of 128", C);
  for (int i = 0; i < n; ++i) DGTD_REQUIRE(x[i] && du[i], "dwconv_bwd_weight_batched: null tensor in layer %d", i);
  DGTD_PROF(s, DGTD_HBM, 2.0 * dgtd_esize(dt) * n * B * H * W * C, "dgtd_dwconv_bwd_weight_batched[n%d,k%d,%dx%dx%d]", n, K, H, W, C);
  hipStream_t st = (hipStream_t)s;
  const BwwGeom ys = bww_batched_geom(n, B, H, W, C, K);
  float* ws = (float*)workspace;
  if (dt == DGTD_BF16) return K == 7 ? bww_batched_launch<bf16_t, 7, 8>(x, du, n, has_bias, ws, B, H, W, C, ys, st)
                                     : bww_batched_launch<bf16_t, 3, 8>(x, du, n, has_bias, ws, B, H, W, C, ys, st);
  if (dt == DGTD_F16) return K == 7 ? bww_batched_launch<f16_t, 7, 8>(x, du, n, has_bias, ws, B, H, W, C, ys, st)
                                    : bww_batched_launch<f16_t, 3, 8>(x, du, n, has_bias, ws, B, H, W, C, ys, st);
  return K == 7 ? bww_batched_launch<float, 7, 8>(x, du, n, has_bias, ws, B, H, W, C, ys, st)
                : bww_batched_launch<float, 3, 8>(x, du, n, has_bias, ws, B, H, W, C, ys, st);
}

extern "C" int dgtd_dwconv_fwd(const void* x, const float* w_t, const float* bias, const void* aux, void* y, int B, int H, int W,
                               int C, int K, int mode, dgtd_dtype dt, dgtd_stream s) {
  DGTD_PROF(s, DGTD_HBM, (mode >= 2 ? 3.0 : 2.0) * dgtd_esize(dt) * B * H * W * C, "dgtd_dwconv_fwd[k%d,mode%d,%dx%dx%d]", K, mode, H, W, C);
  DGTD_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0, "dwconv_fwd: bad sizes");
  DGTD_REQUIRE(K == 3 || K == 7, "dwconv_fwd: K=%d (only 3 and 7 are on the path)", K);
  DGTD_REQUIRE(mode >= 0 && mode <= 3 && (mode < 2 || aux), "dwconv_fwd: bad mode %d", mode);
  hipStream_t st = (hipStream_t)s;
  DGTD_REQUIRE(DGTD_IS_HALF(dt) || dt == DGTD_F32, "dwconv_fwd: bad dtype %d", (int)dt);
  // measured (profiles/r01_ops_device_times.txt): the LDS-tiled kernel wins 1.6-1.8x for 7x7 (49-tap halo reuse); for 3x3 the direct
  // kernel with 16-byte loads is as fast or faster
  if (K == 7 && C % 128 == 0 && use_tiled()) return dgtd_dwconv_tiled_fwd(x, w_t, bias, aux, y, B, H, W, C, K, mode, dt, st);
  static const bool sw3 = !(getenv("DGTD_DWCONV3_SLIDING") && getenv("DGTD_DWCONV3_SLIDING")[0] == '0');
  if (K == 3 && C % 4 == 0 && sw3 && H >= 8) {
    if (dt == DGTD_BF16) return fwd3_sw_launch<bf16_t>(x, w_t, bias, aux, y, B, H, W, C, mode, st);
    if (dt == DGTD_F16) return fwd3_sw_launch<f16_t>(x, w_t, bias, aux, y, B, H, W, C, mode, st);
    return fwd3_sw_launch<float>(x, w_t, bias, aux, y, B, H, W, C, mode, st);
  }
  if (dt == DGTD_BF16) return K == 7 ? fwd_launch<bf16_t, 4, 7, 4>(x, w_t, bias, aux, y, B, H, W, C, mode, st)
                                     : fwd_launch<bf16_t, 8, 3, 4>(x, w_t, bias, aux, y, B, H, W, C, mode, st);
  if (dt == DGTD_F16) return K == 7 ? fwd_launch<f16_t, 4, 7, 4>(x, w_t, bias, aux, y, B, H, W, C, mode, st)
                                    : fwd_launch<f16_t, 8, 3, 4>(x, w_t, bias, aux, y, B, H, W, C, mode, st);
  if (dt == DGTD_F32) return K == 7 ? fwd_launch<float, 4, 7, 4>(x, w_t, bias, aux, y, B, H, W, C, mode, st)
                                    : fwd_launch<float, 4, 3, 4>(x, w_t, bias, aux, y, B, H, W, C, mode, st);
  DGTD_FAIL(2, "dwconv_fwd: bad dtype %d", (int)dt);
}

extern "C" int64_t dgtd_dwconv_bwd_weight_workspace(int B, int H, int W, int C, int K) {
  const int ys = bww_ysplit(B, H, W, C, 8, K);
  const int64_t direct = (int64_t)B * cdiv(W, 8) * ys;
  const int64_t tiled = C % 128 == 0 ? dgtd_dwconv_tiled_bww_groups(B, H, W, C) : 0;
  return std::max(direct, tiled) * (K * K + 1) * C * sizeof(float);
}

extern "C" int dgtd_dwconv_bwd_weight(const void* x, const void* du, float* grads, int has_bias, void* workspace, int B, int H, int W,
                                      int C, int K, dgtd_dtype dt, dgtd_stream s) {
  DGTD_PROF(s, DGTD_HBM, 2.0 * dgtd_esize(dt) * B * H * W * C, "dgtd_dwconv_bwd_weight[k%d,%dx%dx%d]", K, H, W, C);
  DGTD_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0, "dwconv_bwd_weight: bad sizes");
  DGTD_REQUIRE(K == 3 || K == 7, "dwconv_bwd_weight: K=%d (only 3 and 7 are on the path)", K);
  hipStream_t st = (hipStream_t)s;
  DGTD_REQUIRE(DGTD_IS_HALF(dt) || dt == DGTD_F32, "dwconv_bwd_weight: bad dtype %d", (int)dt);
  DGTD_REQUIRE(C % 128 == 0, "dwconv_bwd_weight: C=%d must be a multiple of 128", C);
  // measured: the K-wave direct kernel below beats the LDS-tiled weight gradient on every shape of the model (its accumulators
  // never leave the wave); the tiled variant stays selectable for experiments
  if (use_tiled() && getenv("DGTD_DWCONV_TILED_BWW")) return dgtd_dwconv_tiled_bww(x, du, grads, has_bias, workspace, B, H, W, C, K, dt, st);
  if (dt == DGTD_BF16) return K == 7 ? bww_launch<bf16_t, 7, 8>(x, du, grads, has_bias, workspace, B, H, W, C, st)
                                     : bww_launch<bf16_t, 3, 8>(x, du, grads, has_bias, workspace, B, H, W, C, st);
  if (dt == DGTD_F16) return K == 7 ? bww_launch<f16_t, 7, 8>(x, du, grads, has_bias, workspace, B, H, W, C, st)
                                    : bww_launch<f16_t, 3, 8>(x, du, grads, has_bias, workspace, B, H, W, C, st);
  if (dt == DGTD_F32) return K == 7 ? bww_launch<float, 7, 8>(x, du, grads, has_bias, workspace, B, H, W, C, st)
                                    : bww_launch<float, 3, 8>(x, du, grads, has_bias, workspace, B, H, W, C, st);
  DGTD_FAIL(2, "dwconv_bwd_weight: bad dtype %d", (int)dt);
}

extern "C" int dgtd_dwconv_bwd_weight_partial(const void* x, const void* du, int has_bias, void* workspace, int B, int H, int W, int C, int K,
                                              dgtd_dtype dt, int* nblocks, dgtd_stream s) {
  DGTD_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && nblocks, "dwconv_bwd_weight_partial: bad sizes");
  DGTD_REQUIRE(K == 3 || K == 7, "dwconv_bwd_weight_partial: K=%d (only 3 and 7 are on the path)", K);
  DGTD_REQUIRE(DGTD_IS_HALF(dt) || dt == DGTD_F32, "dwconv_bwd_weight_partial: bad dtype %d", (int)dt);
  DGTD_REQUIRE(C % 128 == 0, "dwconv_bwd_weight_partial: C=%d must be a multiple of 128", C);
  DGTD_PROF(s, DGTD_HBM, 2.0 * dgtd_esize(dt) * B * H * W * C, "dgtd_dwconv_bwd_weight[k%d,%dx%dx%d]", K, H, W, C);
  hipStream_t st = (hipStream_t)s;
  if (dt == DGTD_BF16) return K == 7 ? bww_launch<bf16_t, 7, 8>(x, du, nullptr, has_bias, workspace, B, H, W, C, st, nblocks)
                                     : bww_launch<bf16_t, 3, 8>(x, du, nullptr, has_bias, workspace, B, H, W, C, st, nblocks);
  if (dt == DGTD_F16) return K == 7 ? bww_launch<f16_t, 7, 8>(x, du, nullptr, has_bias, workspace, B, H, W, C, st, nblocks)
                                    : bww_launch<f16_t, 3, 8>(x, du, nullptr, has_bias, workspace, B, H, W, C, st, nblocks);
  return K == 7 ? bww_launch<float, 7, 8>(x, du, nullptr, has_bias, workspace, B, H, W, C, st, nblocks)
                : bww_launch<float, 3, 8>(x, du, nullptr, has_bias, workspace, B, H, W, C, st, nblocks);
}

extern "C" int dgtd_dwconv_pack(const void* w, const void* bias, float* packed, int C, int K, dgtd_dtype wdt, dgtd_stream s) {
  DGTD_REQUIRE(C > 0 && (K == 3 || K == 7), "dwconv_pack: bad sizes C=%d K=%d", C, K);
  const int KK = K * K, grid = (int)cdiv((int64_t)KK * C, 256);
  if (wdt == DGTD_BF16) hipLaunchKernelGGL(dwconv_pack_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)s, (const bf16_t*)w, (const bf16_t*)bias, packed, C, KK);
  else if (wdt == DGTD_F16) hipLaunchKernelGGL(dwconv_pack_kernel<f16_t>, dim3(grid), dim3(256), 0, (hipStream_t)s, (const f16_t*)w, (const f16_t*)bias, packed, C, KK);
  else if (wdt == DGTD_F32) hipLaunchKernelGGL(dwconv_pack_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)s, (const float*)w, (const float*)bias, packed, C, KK);
  else DGTD_FAIL(2, "dwconv_pack: bad dtype %d", (int)wdt);
  DGTD_CHECK_LAUNCH("dwconv_pack");
  return 0;
}

extern "C" int dgtd_dwconv_pack_batched(const void* const* w, const void* const* bias, float* const* packed, const int* C, const int* K, int n,
                                        dgtd_dtype wdt, dgtd_stream s) {
  DGTD_REQUIRE(n > 0 && w && packed && C && K, "dwconv_pack_batched: no layers");
  DGTD_REQUIRE(DGTD_IS_HALF(wdt) || wdt == DGTD_F32, "dwconv_pack_batched: bad dtype %d", (int)wdt);
  for (int z0 = 0; z0 < n; z0 += PK_MAX) {
    const int m = std::min(PK_MAX, n - z0);
    PackTable t;
    int most = 0;
    for (int i = 0; i < PK_MAX; ++i) {
      const int j = z0 + (i < m ? i : 0);
      DGTD_REQUIRE(w[j] && packed[j] && C[j] > 0 && (K[j] == 3 || K[j] == 7), "dwconv_pack_batched: bad layer %d (C=%d K=%d)", j, C[j], K[j]);
      t.w[i] = w[j]; t.b[i] = bias ? bias[j] : nullptr; t.out[i] = packed[j]; t.C[i] = C[j]; t.KK[i] = K[j] * K[j];
      most = std::max(most, t.KK[i] * t.C[i]);
    }
    const dim3 grid((unsigned)cdiv(most, 256), (unsigned)m);
    if (wdt == DGTD_BF16) hipLaunchKernelGGL(dwconv_pack_batched_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)s, t);
    else if (wdt == DGTD_F16) hipLaunchKernelGGL(dwconv_pack_batched_kernel<f16_t>, grid, dim3(256), 0, (hipStream_t)s, t);
    else hipLaunchKernelGGL(dwconv_pack_batched_kernel<float>, grid, dim3(256), 0, (hipStream_t)s, t);
    DGTD_CHECK_LAUNCH("dwconv_pack_batched");
  }
  return 0;
}

extern "C" int dgtd_dwconv_unpack_grads(const float* grads, void* dw, void* db, int C, int K, dgtd_dtype wdt, dgtd_stream s) {
  DGTD_REQUIRE(C > 0 && (K == 3 || K == 7), "dwconv_unpack_grads: bad sizes C=%d K=%d", C, K);
  const int KK = K * K, grid = (int)cdiv((int64_t)KK * C, 256);
  if (wdt == DGTD_BF16) hipLaunchKernelGGL(dwconv_unpack_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)s, grads, (bf16_t*)dw, (bf16_t*)db, C, KK);
  else if (wdt == DGTD_F16) hipLaunchKernelGGL(dwconv_unpack_kernel<f16_t>, dim3(grid), dim3(256), 0, (hipStream_t)s, grads, (f16_t*)dw, (f16_t*)db, C, KK);
  else if (wdt == DGTD_F32) hipLaunchKernelGGL(dwconv_unpack_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)s, grads, (float*)dw, (float*)db, C, KK);
  else DGTD_FAIL(2, "dwconv_unpack_grads: bad dtype %d", (int)wdt);
  DGTD_CHECK_LAUNCH("dwconv_unpack_grads");
  return 0;
}
