// preprocess.hip — the reference's per-sample input pipeline on the device (twig/dataset/sod_train.py:31-54, :65-83):
// RandomHorizontalFlip -> Resize((S,S)) -> ToTensor (-> Normalize) for one uint8 HWC image already resident in HBM.
// Resize = Pillow's ImagingResample (BILINEAR, antialiased): triangle filter stretched by the down-scale factor, coefficients
// normalised in double and converted to 22-bit fixed point, horizontal pass then vertical pass with a rounded uint8 intermediate.
// Bit-exact against Pillow / the numpy restatement in oracle/preprocess_cpu.py: the coefficient arithmetic runs in IEEE double
// without FMA contraction, the passes in int32 fixed point, ToTensor/Normalize in IEEE float32 ((v/255 - mean)/std, in that order).
// Byte work, HBM/latency-bound: Hin*Win*C bytes in, C*S*S*4 bytes out; three small launches per image.
#include "common.h"

namespace {

constexpr int PBITS = 32 - 8 - 2;

// bounds[xx] = (xmin, count), kk[xx][ksize] fixed-point weights; one thread per output index
__global__ __launch_bounds__(256) void resize_coeffs_kernel(int in_size, int out_size, int ksize, int* __restrict__ bounds, int* __restrict__ kk) {
#pragma clang fp contract(off)
  const int xx = blockIdx.x * 256 + threadIdx.x;
  if (xx >= out_size) return;
  const double scale = (double)in_size / (double)out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale;
  const double ss = 1.0 / filterscale;
  const double center = ((double)xx + 0.5) * scale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) {
    double a = ((double)(x + xmin) - center + 0.5) * ss;
    a = a < 0.0 ? -a : a;
    ww += a < 1.0 ? 1.0 - a : 0.0;
  }
  int* k = kk + (size_t)xx * ksize;
  for (int x = 0; x < ksize; ++x) {
    int v = 0;
    if (x < xmax) {
      double a = ((double)(x + xmin) - center + 0.5) * ss;
      a = a < 0.0 ? -a : a;
      double w = a < 1.0 ? 1.0 - a : 0.0;
      if (ww != 0.0) w = w / ww;
      v = w < 0.0 ? (int)(-0.5 + w * (double)(1 << PBITS)) : (int)(0.5 + w * (double)(1 << PBITS));
    }
    k[x] = v;
  }
  bounds[2 * xx] = xmin;
  bounds[2 * xx + 1] = xmax;
}

__device__ __forceinline__ int clip8(int acc) {
  const int v = acc >> PBITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// horizontal pass: in [H][W][C] -> mid [H][S][C]; optional horizontal flip of the INPUT (flip happens before the resize)
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ mid, const int* __restrict__ bounds,
                                                       const int* __restrict__ kk, int H, int W, int C, int S, int ksize, int flip) {
  const int64_t total = (int64_t)H * S;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int xx = (int)(i % S), y = (int)(i / S);
    const int xmin = bounds[2 * xx], cnt = bounds[2 * xx + 1];
    const int* k = kk + (size_t)xx * ksize;
    const uint8_t* row = in + (size_t)y * W * C;
    for (int c = 0; c < C; ++c) {
      int acc = 1 << (PBITS - 1);
      for (int x = 0; x < cnt; ++x) {
        const int sx = flip ? W - 1 - (xmin + x) : xmin + x;
        acc += (int)row[(size_t)sx * C + c] * k[x];
      }
      mid[((size_t)y * S + xx) * C + c] = (uint8_t)clip8(acc);
    }
  }
}

struct NormArgs { float mean[4], stdv[4]; int normalize; };

// vertical pass + ToTensor (+ Normalize): mid [H][S][C] -> out [C][S][S]
template <typename T>
__global__ __launch_bounds__(256) void resize_v_norm_kernel(const uint8_t* __restrict__ mid, T* __restrict__ out, const int* __restrict__ bounds,
                                                            const int* __restrict__ kk, int H, int C, int S, int ksize, NormArgs na) {
#pragma clang fp contract(off)
  const int64_t total = (int64_t)S * S;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int xx = (int)(i % S), yy = (int)(i / S);
    const int ymin = bounds[2 * yy], cnt = bounds[2 * yy + 1];
    const int* k = kk + (size_t)yy * ksize;
    for (int c = 0; c < C; ++c) {
      int acc = 1 << (PBITS - 1);
      for (int y = 0; y < cnt; ++y) acc += (int)mid[((size_t)(ymin + y) * S + xx) * C + c] * k[y];
      float t = (float)clip8(acc) / 255.0f;
      if (na.normalize) t = (t - na.mean[c]) / na.stdv[c];
      out[((size_t)c * S + yy) * S + xx] = (T)t;
    }
  }
}

inline int ksize_for(int in_size, int out_size) {
  const double scale = (double)in_size / (double)out_size;
  const double support = scale < 1.0 ? 1.0 : scale;
  return (int)ceil(support) * 2 + 1;
}
inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

}  // namespace

extern "C" int64_t dgtd_preprocess_workspace(int Hin, int Win, int C, int S) {
  const size_t tabs = (size_t)S * (2 + ksize_for(Win, S)) * 4 + (size_t)S * (2 + ksize_for(Hin, S)) * 4;
  return (int64_t)(align16((size_t)Hin * S * C) + align16(tabs) + 64);
}

extern "C" int dgtd_preprocess(const void* img_u8, void* out, const float* mean_host, const float* std_host, void* workspace, int Hin,
                               int Win, int C, int S, int flip, dgtd_dtype out_dt, dgtd_stream s) {
  DGTD_PROF(s, DGTD_HBM, (double)Hin * Win * C + (double)dgtd_esize(out_dt) * C * S * S, "dgtd_preprocess[%dx%dx%d->%d]", Hin, Win, C, S);
  DGTD_REQUIRE(Hin > 0 && Win > 0 && S > 0 && C >= 1 && C <= 4, "preprocess: bad sizes H=%d W=%d C=%d S=%d", Hin, Win, C, S);
  DGTD_REQUIRE((mean_host == nullptr) == (std_host == nullptr), "preprocess: mean and std go together");
  DGTD_REQUIRE(out_dt == DGTD_F32 || DGTD_IS_HALF(out_dt), "preprocess: bad output dtype %d", (int)out_dt);
  hipStream_t st = (hipStream_t)s;
  const int kh = ksize_for(Win, S), kv = ksize_for(Hin, S);
  uint8_t* mid = (uint8_t*)workspace;
  int* bounds_h = (int*)((char*)workspace + align16((size_t)Hin * S * C));
  int* kk_h = bounds_h + 2 * S;
  int* bounds_v = kk_h + (size_t)S * kh;
  int* kk_v = bounds_v + 2 * S;
  NormArgs na;
  na.normalize = mean_host != nullptr;
  for (int c = 0; c < 4; ++c) { na.mean[c] = (na.normalize && c < C) ? mean_host[c] : 0.f; na.stdv[c] = (na.normalize && c < C) ? std_host[c] : 1.f; }
  hipLaunchKernelGGL(resize_coeffs_kernel, dim3((int)cdiv(S, 256)), dim3(256), 0, st, Win, S, kh, bounds_h, kk_h);
  hipLaunchKernelGGL(resize_coeffs_kernel, dim3((int)cdiv(S, 256)), dim3(256), 0, st, Hin, S, kv, bounds_v, kk_v);
  DGTD_CHECK_LAUNCH("preprocess_coeffs");
  hipLaunchKernelGGL(resize_h_kernel, dim3((int)std::min<int64_t>(cdiv((int64_t)Hin * S, 256), 4096)), dim3(256), 0, st, (const uint8_t*)img_u8, mid,
                     (const int*)bounds_h, (const int*)kk_h, Hin, Win, C, S, kh, flip);
  DGTD_CHECK_LAUNCH("preprocess_horizontal");
  const int gv = (int)std::min<int64_t>(cdiv((int64_t)S * S, 256), 4096);
  if (out_dt == DGTD_F32) hipLaunchKernelGGL(resize_v_norm_kernel<float>, dim3(gv), dim3(256), 0, st, (const uint8_t*)mid, (float*)out, (const int*)bounds_v, (const int*)kk_v, Hin, C, S, kv, na);
  else if (out_dt == DGTD_F16) hipLaunchKernelGGL(resize_v_norm_kernel<f16_t>, dim3(gv), dim3(256), 0, st, (const uint8_t*)mid, (f16_t*)out, (const int*)bounds_v, (const int*)kk_v, Hin, C, S, kv, na);
  else hipLaunchKernelGGL(resize_v_norm_kernel<bf16_t>, dim3(gv), dim3(256), 0, st, (const uint8_t*)mid, (bf16_t*)out, (const int*)bounds_v, (const int*)kk_v, Hin, C, S, kv, na);
  DGTD_CHECK_LAUNCH("preprocess_vertical");
  return 0;
}
