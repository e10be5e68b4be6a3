// adamw.hip — AdamW over one contiguous run of a flat parameter bucket (dist.GradReducer keeps fp32 masters, fp32 gradients and the
// 16-bit working copies of a bucket in flat buffers at identical offsets).  Replaces torch.optim.AdamW(fused=True)'s multi-tensor
// launches (config/sod.yml:57-60: AdamW, lr 5e-4, weight_decay 0.1, per-prefix lr multipliers = one run per multiplier), the
// master -> working-copy cast and - in fp16 mode - GradScaler.unscale_/step of the reference's AmpOptimWrapper (config/sod.yml:57),
// in one pass:
//   g *= inv_scale;  p *= 1 - lr*wd;  m += (1-b1)(g - m);  v = b2 v + (1-b2) g^2;  p -= (lr/bc1) m / (sqrt(v)/sqrt(bc2) + eps);  w = half(p)
// (the update order of torch's _fused_adamw); the whole launch is a no-op when *found_inf != 0 (GradScaler skips the step).
// HBM-bound: 28 B/element (+2 with the working copy); 16-byte accesses on the aligned body, scalars on the unaligned head/tail.
#include "common.h"

namespace {

struct AdamArgs { float lr, b1, b2, eps, wd, inv_bc1, inv_sqrt_bc2, log_b1, log_b2; };

__device__ __forceinline__ float adam_one(float p, float g, float& m, float& v, const AdamArgs& a) {
  p *= 1.f - a.lr * a.wd;
  m = m + (1.f - a.b1) * (g - m);
  v = a.b2 * v + (1.f - a.b2) * g * g;
  const float denom = sqrtf(v) * a.inv_sqrt_bc2 + a.eps;
  return p - (a.lr * a.inv_bc1) * (m / denom);
}

template <typename WT>
__global__ __launch_bounds__(256) void adamw_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, WT* __restrict__ w, int64_t n, int64_t head,
                                                         AdamArgs a, const float* __restrict__ amp, const float* __restrict__ lr_dev,
                                                         const WT* __restrict__ g16) {
  // g16 != NULL: the gradient is the 16-bit all-reduce payload itself (same phase as w), g is not read
  // amp = the loss scaler's device state { scale, growth_tracker, 1/scale, found_inf, steps taken } or NULL
  // lr_dev = the learning rate in device memory (a captured hipGraph replays with whatever the schedule wrote there) or NULL
  float gs = 1.f;
  if (lr_dev) a.lr = *lr_dev;
  if (amp) {
    if (amp[3] != 0.f) return;                              // overflowed step: parameters, moments and working copies stay as they are
    gs = amp[2];
    const float t = amp[4] + 1.f;                           // skipped steps do not advance the bias corrections (GradScaler.step)
    a.inv_bc1 = -1.f / expm1f(t * a.log_b1);
    a.inv_sqrt_bc2 = rsqrtf(-expm1f(t * a.log_b2));
  }
  typedef typename Vec8<WT>::type W4;
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
  const int64_t body4 = (n - head) / 4;                      // float4 groups after the unaligned head
  for (int64_t i = tid; i < body4; i += nth) {
    const int64_t o = head + i * 4;
    f32x4 pv = *reinterpret_cast<f32x4*>(p + o), mv = *reinterpret_cast<f32x4*>(m + o), vv = *reinterpret_cast<f32x4*>(v + o);
    f32x4 gv;
    if (g16) {
      const W4 hv = *reinterpret_cast<const W4*>(g16 + o);
#pragma unroll
      for (int j = 0; j < 4; ++j) gv[j] = (float)hv[j];
    } else {
      gv = *reinterpret_cast<const f32x4*>(g + o);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { float mj = mv[j], vj = vv[j]; pv[j] = adam_one(pv[j], gv[j] * gs, mj, vj, a); mv[j] = mj; vv[j] = vj; }
    *reinterpret_cast<f32x4*>(p + o) = pv;
    *reinterpret_cast<f32x4*>(m + o) = mv;
    *reinterpret_cast<f32x4*>(v + o) = vv;
    if (w) {
      W4 wv;
#pragma unroll
      for (int j = 0; j < 4; ++j) wv[j] = (WT)pv[j];
      *reinterpret_cast<W4*>(w + o) = wv;
    }
  }
  // head [0, head) and tail [head + 4*body4, n): at most 6 scalars
  const int64_t tail0 = head + body4 * 4;
  const int64_t nscal = head + (n - tail0);
  if (tid < nscal) {
    const int64_t o = tid < head ? tid : tail0 + (tid - head);
    float mj = m[o], vj = v[o];
    const float pj = adam_one(p[o], (g16 ? (float)g16[o] : g[o]) * gs, mj, vj, a);
    p[o] = pj; m[o] = mj; v[o] = vj;
    if (w) w[o] = (WT)pj;
  }
}

// found[0] = 1 if any element of g[0, n) is inf or NaN (left untouched otherwise): the `found_inf` of GradScaler.unscale_.
__global__ __launch_bounds__(256) void found_inf_kernel(const float* __restrict__ g, int64_t n, int64_t head, float* __restrict__ found) {
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
  const int64_t body4 = (n - head) / 4;
  bool bad = false;
  for (int64_t i = tid; i < body4; i += nth) {
    const f32x4 gv = *reinterpret_cast<const f32x4*>(g + head + i * 4);
    // x - x is 0 for every finite x and NaN for inf / NaN
    const float t = (gv[0] - gv[0]) + (gv[1] - gv[1]) + (gv[2] - gv[2]) + (gv[3] - gv[3]);
    bad |= !(t == 0.f);
  }
  const int64_t tail0 = head + body4 * 4, nscal = head + (n - tail0);
  if (tid < nscal) { const float x = g[tid < head ? tid : tail0 + (tid - head)]; bad |= !((x - x) == 0.f); }
  if (__any(bad) && (threadIdx.x & 63) == 0) *found = 1.f;   // same value from every writer: no atomic needed
}

// GradScaler.update() (torch/amp/grad_scaler.py _amp_update_scale_): state = { scale, growth_tracker, inv_scale, found_inf, steps }.
__global__ void loss_scale_update_kernel(float* __restrict__ state, float growth, float backoff, int interval) {
  float scale = state[0], tracker = state[1];
  if (state[3] != 0.f) { scale *= backoff; tracker = 0.f; }
  else {
    state[4] += 1.f;                                        // optimizer steps actually taken
    tracker += 1.f;
    if (tracker >= (float)interval) { const float grown = scale * growth; if (grown - grown == 0.f) scale = grown; tracker = 0.f; }
  }
  state[0] = scale; state[1] = tracker; state[2] = 1.f / scale; state[3] = 0.f;
}

}  // namespace

static int adamw_impl(float* p, const float* g, const void* g16, float* m, float* v, void* w, dgtd_dtype w_dt, int64_t n, float lr, float beta1,
                      float beta2, float eps, float weight_decay, float bias_correction1, float bias_correction2,
                      const float* amp_state, const float* lr_dev, dgtd_stream s) {
  DGTD_PROF(s, DGTD_HBM, (w ? 30.0 : 28.0) * n - (g16 ? 2.0 * n : 0.0), "dgtd_adamw_flat[n=%lld%s]", (long long)n, g16 ? ",g16" : "");
  DGTD_REQUIRE(n > 0 && p && (g || g16) && m && v, "adamw_flat: bad arguments");
  if (!g) g = p;                                                             // never read; keeps the phase checks below trivially true
  DGTD_REQUIRE(!g16 || (DGTD_IS_HALF(w_dt) && ((uintptr_t)g16 % 8) * 2 == (uintptr_t)p % 16), "adamw_flat: the 16-bit gradient must share the phase of the masters");
  DGTD_REQUIRE(amp_state || (bias_correction1 > 0.f && bias_correction2 > 0.f), "adamw_flat: bias corrections must be positive");
  DGTD_REQUIRE(!w || DGTD_IS_HALF(w_dt), "adamw_flat: the working copy is bf16 or fp16, got dtype %d", (int)w_dt);
  const uintptr_t ap = (uintptr_t)p;
  DGTD_REQUIRE(((uintptr_t)g - ap) % 16 == 0 && ((uintptr_t)m - ap) % 16 == 0 && ((uintptr_t)v - ap) % 16 == 0 && ap % 4 == 0,
               "adamw_flat: p, g, m, v must share their 16-byte phase");
  DGTD_REQUIRE(!w || (((uintptr_t)w % 8) * 2 == ap % 16), "adamw_flat: the working copy must share the phase of the masters");
  const int64_t head = std::min<int64_t>(n, ((16 - (int64_t)(ap % 16)) % 16) / 4);
  AdamArgs a{lr, beta1, beta2, eps, weight_decay, 1.f / bias_correction1, 1.f / sqrtf(bias_correction2), (float)log((double)beta1), (float)log((double)beta2)};
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv((n + 3) / 4 + 8, 256), 8192));
  if (w_dt == DGTD_F16) hipLaunchKernelGGL(adamw_flat_kernel<f16_t>, dim3(grid), dim3(256), 0, (hipStream_t)s, p, g, m, v, (f16_t*)w, n, head, a, amp_state, lr_dev, (const f16_t*)g16);
  else hipLaunchKernelGGL(adamw_flat_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)s, p, g, m, v, (bf16_t*)w, n, head, a, amp_state, lr_dev, (const bf16_t*)g16);
  DGTD_CHECK_LAUNCH("adamw_flat");
  return 0;
}

extern "C" int dgtd_adamw_flat_amp(float* p, const float* g, float* m, float* v, void* w, dgtd_dtype w_dt, int64_t n, float lr, float beta1,
                                   float beta2, float eps, float weight_decay, float bias_correction1, float bias_correction2,
                                   const float* amp_state, const float* lr_dev, dgtd_stream s) {
  DGTD_REQUIRE(g, "adamw_flat: bad arguments");
  return adamw_impl(p, g, nullptr, m, v, w, w_dt, n, lr, beta1, beta2, eps, weight_decay, bias_correction1, bias_correction2, amp_state, lr_dev, s);
}

extern "C" int dgtd_adamw_flat_g16(float* p, const void* g16, float* m, float* v, void* w, dgtd_dtype w_dt, int64_t n, float lr, float beta1,
                                   float beta2, float eps, float weight_decay, float bias_correction1, float bias_correction2,
                                   const float* amp_state, const float* lr_dev, dgtd_stream s) {
  DGTD_REQUIRE(g16, "adamw_flat_g16: bad arguments");
  return adamw_impl(p, nullptr, g16, m, v, w, w_dt, n, lr, beta1, beta2, eps, weight_decay, bias_correction1, bias_correction2, amp_state, lr_dev, s);
}

extern "C" int dgtd_adamw_flat(float* p, const float* g, float* m, float* v, void* w_bf16, int64_t n, float lr, float beta1, float beta2,
                               float eps, float weight_decay, float bias_correction1, float bias_correction2, dgtd_stream s) {
  return dgtd_adamw_flat_amp(p, g, m, v, w_bf16, DGTD_BF16, n, lr, beta1, beta2, eps, weight_decay, bias_correction1, bias_correction2,
                             nullptr, nullptr, s);
}

extern "C" int dgtd_found_inf(const float* g, int64_t n, float* found, dgtd_stream s) {
  DGTD_PROF(s, DGTD_HBM, 4.0 * n, "dgtd_found_inf[n=%lld]", (long long)n);
  DGTD_REQUIRE(n > 0 && g && found, "found_inf: bad arguments");
  const uintptr_t ap = (uintptr_t)g;
  DGTD_REQUIRE(ap % 4 == 0, "found_inf: g must be 4-byte aligned");
  const int64_t head = std::min<int64_t>(n, ((16 - (int64_t)(ap % 16)) % 16) / 4);
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv((n + 3) / 4 + 8, 256), 4096));
  hipLaunchKernelGGL(found_inf_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, g, n, head, found);
  DGTD_CHECK_LAUNCH("found_inf");
  return 0;
}

extern "C" int dgtd_loss_scale_update(float* state, float growth_factor, float backoff_factor, int growth_interval, dgtd_stream s) {
  DGTD_REQUIRE(state && growth_factor >= 1.f && backoff_factor > 0.f && backoff_factor <= 1.f && growth_interval > 0, "loss_scale_update: bad arguments");
  hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(1), 0, (hipStream_t)s, state, growth_factor, backoff_factor, growth_interval);
  DGTD_CHECK_LAUNCH("loss_scale_update");
  return 0;
}
