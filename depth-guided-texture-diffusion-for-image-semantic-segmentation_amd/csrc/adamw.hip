// adamw.hip — AdamW over one contiguous run of a flat parameter bucket (dist.GradReducer keeps fp32 masters, fp32 gradients and the
// bf16 working copies of a bucket in flat buffers at identical offsets).  Replaces torch.optim.AdamW(fused=True)'s multi-tensor
// launches (config/sod.yml:57-60: AdamW, lr 5e-4, weight_decay 0.1, per-prefix lr multipliers = one run per multiplier) and the
// master -> working-copy cast, in one pass:
//   p *= 1 - lr*wd;  m += (1-b1)(g - m);  v = b2 v + (1-b2) g^2;  p -= (lr/bc1) m / (sqrt(v)/sqrt(bc2) + eps);  w = bf16(p)
// (the update order of torch's _fused_adamw).  HBM-bound: 28 B/element (+2 with the working copy); 16-byte accesses on the
// aligned body, scalars on the unaligned head/tail of a run.
#include "common.h"

namespace {

struct AdamArgs { float lr, b1, b2, eps, wd, inv_bc1, inv_sqrt_bc2; };

__device__ __forceinline__ float adam_one(float p, float g, float& m, float& v, const AdamArgs& a) {
  p *= 1.f - a.lr * a.wd;
  m = m + (1.f - a.b1) * (g - m);
  v = a.b2 * v + (1.f - a.b2) * g * g;
  const float denom = sqrtf(v) * a.inv_sqrt_bc2 + a.eps;
  return p - (a.lr * a.inv_bc1) * (m / denom);
}

__global__ __launch_bounds__(256) void adamw_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, bf16_t* __restrict__ w, int64_t n, int64_t head,
                                                         AdamArgs a) {
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
  const int64_t body4 = (n - head) / 4;                      // float4 groups after the unaligned head
  for (int64_t i = tid; i < body4; i += nth) {
    const int64_t o = head + i * 4;
    f32x4 pv = *reinterpret_cast<f32x4*>(p + o), mv = *reinterpret_cast<f32x4*>(m + o), vv = *reinterpret_cast<f32x4*>(v + o);
    const f32x4 gv = *reinterpret_cast<const f32x4*>(g + o);
#pragma unroll
    for (int j = 0; j < 4; ++j) { float mj = mv[j], vj = vv[j]; pv[j] = adam_one(pv[j], gv[j], mj, vj, a); mv[j] = mj; vv[j] = vj; }
    *reinterpret_cast<f32x4*>(p + o) = pv;
    *reinterpret_cast<f32x4*>(m + o) = mv;
    *reinterpret_cast<f32x4*>(v + o) = vv;
    if (w) {
      bf16x4 wv;
#pragma unroll
      for (int j = 0; j < 4; ++j) wv[j] = (bf16_t)pv[j];
      *reinterpret_cast<bf16x4*>(w + o) = wv;
    }
  }
  // head [0, head) and tail [head + 4*body4, n): at most 6 scalars
  const int64_t tail0 = head + body4 * 4;
  const int64_t nscal = head + (n - tail0);
  if (tid < nscal) {
    const int64_t o = tid < head ? tid : tail0 + (tid - head);
    float mj = m[o], vj = v[o];
    const float pj = adam_one(p[o], g[o], mj, vj, a);
    p[o] = pj; m[o] = mj; v[o] = vj;
    if (w) w[o] = (bf16_t)pj;
  }
}

}  // namespace

extern "C" int dgtd_adamw_flat(float* p, const float* g, float* m, float* v, void* w_bf16, int64_t n, float lr, float beta1, float beta2,
                               float eps, float weight_decay, float bias_correction1, float bias_correction2, dgtd_stream s) {
  DGTD_REQUIRE(n > 0 && p && g && m && v, "adamw_flat: bad arguments");
  DGTD_REQUIRE(bias_correction1 > 0.f && bias_correction2 > 0.f, "adamw_flat: bias corrections must be positive");
  const uintptr_t ap = (uintptr_t)p;
  DGTD_REQUIRE(((uintptr_t)g - ap) % 16 == 0 && ((uintptr_t)m - ap) % 16 == 0 && ((uintptr_t)v - ap) % 16 == 0 && ap % 4 == 0,
               "adamw_flat: p, g, m, v must share their 16-byte phase");
  DGTD_REQUIRE(!w_bf16 || (((uintptr_t)w_bf16 % 8) * 2 == ap % 16), "adamw_flat: the working copy must share the phase of the masters");
  const int64_t head = std::min<int64_t>(n, ((16 - (int64_t)(ap % 16)) % 16) / 4);
  AdamArgs a{lr, beta1, beta2, eps, weight_decay, 1.f / bias_correction1, 1.f / sqrtf(bias_correction2)};
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv((n + 3) / 4 + 8, 256), 8192));
  hipLaunchKernelGGL(adamw_flat_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, p, g, m, v, (bf16_t*)w_bf16, n, head, a);
  DGTD_CHECK_LAUNCH("adamw_flat");
  return 0;
}
