// im2col.hip — patch gather / scatter-free patch gradient for dense KxK strided convolutions as GEMMs on token-major tensors.
// Serves every dense convolution of the path that is not a 3x3 stride-1 16-bit conv (those run on conv3x3.hip) and not a k == stride
// patchify (a free view): OverlapPatchEmbed.proj (k7 s4 p3 and k3 s2 p1, twig/model/cod.py:974-975, :1000), the prompt-decoder tails
// folded into 4x4 stride-s convolutions (cod.py:1220 + :1471), Hitnet.compress_out (k8 s4 p2, cod.py:739), the 1-channel heads and,
// in fp32 parity mode, the 3x3 convolutions of the Hitnet decoder.  col [B*Ho*Wo, K*K*C] then goes through the library GEMM
// (hipBLASLt, cached plans) with the O,H,W,I-stored kernel as its [O, K*K*C] matrix; the weight gradient is that GEMM's own.
//   im2col : col[(b,oy,ox)][(ky,kx,c)] = x[b, oy*s - p + ky, ox*s - p + kx, c]   (0 outside), any input strides (NCHW image, NHWC
//            map, offset views), optional dtype conversion (fp32 image -> 16-bit columns) in the same pass
//   col2im : dx[b,iy,ix,c] = sum over the (<= ceil(K/s)^2) windows covering the pixel - a GATHER per input pixel, no atomics
// HBM-bound copies: im2col moves e*(K*K/s^2 + 1)*B*H*W*C bytes, col2im the same.
#include "common.h"

namespace {

struct Geo { int B, H, W, C, K, S, P, Ho, Wo; long sb, sy, sx, sc; };

// vector path: 8 channels (16 B of a 16-bit type / 2 x 16 B of fp32) per thread, channels contiguous in x (sc == 1) and C % 8 == 0
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void im2col_vec_kernel(const TI* __restrict__ x, TO* __restrict__ col, Geo g, long total) {
  typedef TI i8 __attribute__((ext_vector_type(8)));
  typedef TO o8 __attribute__((ext_vector_type(8)));
  const int C8 = g.C / 8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    long r = i / C8;
    const int kx = (int)(r % g.K); r /= g.K;
    const int ky = (int)(r % g.K); r /= g.K;
    const int ox = (int)(r % g.Wo); r /= g.Wo;
    const int oy = (int)(r % g.Ho);
    const int b = (int)(r / g.Ho);
    const int iy = oy * g.S - g.P + ky, ix = ox * g.S - g.P + kx;
    o8 o;
    if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) {
      const i8 v = *reinterpret_cast<const i8*>(x + b * g.sb + iy * g.sy + ix * g.sx + c8 * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (TO)(float)v[j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (TO)0.f;
    }
    *reinterpret_cast<o8*>(col + i * 8) = o;
  }
}

// scalar path: any C / strides (the 3-channel NCHW image of patch_embed1)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void im2col_scalar_kernel(const TI* __restrict__ x, TO* __restrict__ col, Geo g, long total) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % g.C);
    long r = i / g.C;
    const int kx = (int)(r % g.K); r /= g.K;
    const int ky = (int)(r % g.K); r /= g.K;
    const int ox = (int)(r % g.Wo); r /= g.Wo;
    const int oy = (int)(r % g.Ho);
    const int b = (int)(r / g.Ho);
    const int iy = oy * g.S - g.P + ky, ix = ox * g.S - g.P + kx;
    float v = 0.f;
    if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) v = (float)x[b * g.sb + iy * g.sy + ix * g.sx + c * g.sc];
    col[i] = (TO)v;
  }
}

// dx NHWC contiguous [B,H,W,C]; dcol [B*Ho*Wo, K*K*C]; 8 channels per thread (C % 8 == 0) or scalar
template <typename T, int V>
__global__ __launch_bounds__(256) void col2im_kernel(const T* __restrict__ dcol, T* __restrict__ dx, Geo g, long total) {
  typedef T v8 __attribute__((ext_vector_type(V)));
  const int CV = g.C / V;
  const long rowlen = (long)g.K * g.K * g.C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int cv = (int)(i % CV);
    long r = i / CV;
    const int ix = (int)(r % g.W); r /= g.W;
    const int iy = (int)(r % g.H);
    const int b = (int)(r / g.H);
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    // windows (oy, ky) with oy*S - P + ky == iy: ky = iy + P - oy*S in [0, K)
    const int oy_hi = min(g.Ho - 1, (iy + g.P) / g.S), ox_hi = min(g.Wo - 1, (ix + g.P) / g.S);
    for (int oy = oy_hi; oy >= 0; --oy) {
      const int ky = iy + g.P - oy * g.S;
      if (ky >= g.K) break;
      for (int ox = ox_hi; ox >= 0; --ox) {
        const int kx = ix + g.P - ox * g.S;
        if (kx >= g.K) break;
        const T* p = dcol + (((long)b * g.Ho + oy) * g.Wo + ox) * rowlen + ((long)ky * g.K + kx) * g.C + cv * V;
        const v8 v = *reinterpret_cast<const v8*>(p);
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] += (float)v[j];
      }
    }
    v8 o;
#pragma unroll
    for (int j = 0; j < V; ++j) o[j] = (T)acc[j];
    *reinterpret_cast<v8*>(dx + i * V) = o;
  }
}

inline int grid_for(long total) { return (int)std::max<long>(1, std::min<long>(cdiv(total, 256), 256 * 32)); }

template <typename TI, typename TO>
int im2col_launch(const void* x, void* col, const Geo& g, hipStream_t st) {
  const bool vec = g.sc == 1 && g.C % 8 == 0 && g.sx % 8 == 0 && g.sy % 8 == 0 && g.sb % 8 == 0 && (uintptr_t)x % 16 == 0;
  if (vec) {
    const long total = (long)g.B * g.Ho * g.Wo * g.K * g.K * (g.C / 8);
    hipLaunchKernelGGL((im2col_vec_kernel<TI, TO>), dim3(grid_for(total)), dim3(256), 0, st, (const TI*)x, (TO*)col, g, total);
  } else {
    const long total = (long)g.B * g.Ho * g.Wo * g.K * g.K * g.C;
    hipLaunchKernelGGL((im2col_scalar_kernel<TI, TO>), dim3(grid_for(total)), dim3(256), 0, st, (const TI*)x, (TO*)col, g, total);
  }
  DGTD_CHECK_LAUNCH("im2col");
  return 0;
}

template <typename TI>
int im2col_out(const void* x, void* col, const Geo& g, dgtd_dtype col_dt, hipStream_t st) {
  if (col_dt == DGTD_F32) return im2col_launch<TI, float>(x, col, g, st);
  if (col_dt == DGTD_BF16) return im2col_launch<TI, bf16_t>(x, col, g, st);
  if (col_dt == DGTD_F16) return im2col_launch<TI, f16_t>(x, col, g, st);
  DGTD_FAIL(2, "im2col: bad column dtype %d", (int)col_dt);
}

template <typename T>
int col2im_launch(const void* dcol, void* dx, const Geo& g, hipStream_t st) {
  if (g.C % 8 == 0) {
    const long total = (long)g.B * g.H * g.W * (g.C / 8);
    hipLaunchKernelGGL((col2im_kernel<T, 8>), dim3(grid_for(total)), dim3(256), 0, st, (const T*)dcol, (T*)dx, g, total);
  } else {
    const long total = (long)g.B * g.H * g.W * g.C;
    hipLaunchKernelGGL((col2im_kernel<T, 1>), dim3(grid_for(total)), dim3(256), 0, st, (const T*)dcol, (T*)dx, g, total);
  }
  DGTD_CHECK_LAUNCH("col2im");
  return 0;
}

}  // namespace

extern "C" int dgtd_im2col(const void* x, void* col, int B, int H, int W, int C, int64_t sb, int64_t sy, int64_t sx, int64_t sc, int K,
                           int stride, int pad, int Ho, int Wo, dgtd_dtype x_dt, dgtd_dtype col_dt, dgtd_stream s) {
  DGTD_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && K > 0 && stride > 0 && pad >= 0 && Ho > 0 && Wo > 0, "im2col: bad sizes");
  DGTD_REQUIRE((Ho - 1) * stride - pad + K - 1 < H + pad && (Wo - 1) * stride - pad + K - 1 < W + pad, "im2col: output size does not fit the input");
  DGTD_PROF(s, DGTD_HBM, (double)B * Ho * Wo * K * K * C * (dgtd_esize(x_dt) * (double)std::min(1.0, (double)stride * stride / (K * K)) + dgtd_esize(col_dt)),
            "dgtd_im2col[%dx%dx%d,k%d,s%d->%dx%d]", H, W, C, K, stride, Ho, Wo);
  const Geo g{B, H, W, C, K, stride, pad, Ho, Wo, (long)sb, (long)sy, (long)sx, (long)sc};
  hipStream_t st = (hipStream_t)s;
  if (x_dt == DGTD_F32) return im2col_out<float>(x, col, g, col_dt, st);
  if (x_dt == DGTD_BF16) return im2col_out<bf16_t>(x, col, g, col_dt, st);
  if (x_dt == DGTD_F16) return im2col_out<f16_t>(x, col, g, col_dt, st);
  DGTD_FAIL(2, "im2col: bad input dtype %d", (int)x_dt);
}

extern "C" int dgtd_col2im(const void* dcol, void* dx, int B, int H, int W, int C, int K, int stride, int pad, int Ho, int Wo, dgtd_dtype dt,
                           dgtd_stream s) {
  DGTD_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && K > 0 && stride > 0 && pad >= 0 && Ho > 0 && Wo > 0, "col2im: bad sizes");
  DGTD_PROF(s, DGTD_HBM, (double)dgtd_esize(dt) * ((double)B * Ho * Wo * K * K * C + (double)B * H * W * C), "dgtd_col2im[%dx%dx%d,k%d,s%d<-%dx%d]", H, W, C, K,
            stride, Ho, Wo);
  const Geo g{B, H, W, C, K, stride, pad, Ho, Wo, 0, 0, 0, 1};
  hipStream_t st = (hipStream_t)s;
  if (dt == DGTD_F32) return col2im_launch<float>(dcol, dx, g, st);
  if (dt == DGTD_BF16) return col2im_launch<bf16_t>(dcol, dx, g, st);
  if (dt == DGTD_F16) return col2im_launch<f16_t>(dcol, dx, g, st);
  DGTD_FAIL(2, "col2im: bad dtype %d", (int)dt);
}
