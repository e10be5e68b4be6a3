// common.h — shared device/host helpers for libdgtd.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/dgtd.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- error plumbing (thread-local message, int status) ------------------------------------------
void dgtd_set_error(const char* fmt, ...);
#define DGTD_FAIL(code, ...) do { dgtd_set_error(__VA_ARGS__); return (code); } while (0)
#define DGTD_REQUIRE(cond, ...) do { if (!(cond)) DGTD_FAIL(2, __VA_ARGS__); } while (0)
#define DGTD_CHECK_LAUNCH(name) do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) DGTD_FAIL(3, "%s: launch failed: %s", name, hipGetErrorString(e_)); } while (0)

// ---- opt-in per-call device timing (bench.py's roofline leg; off by default: one relaxed atomic load per entry point) -------------
// DGTD_PROF(stream, bound, amount, "key[%d]", ...) at the top of an extern "C" entry point: when dgtd_profile_enable(1) is on, a HIP
// event pair on `stream` brackets everything the entry enqueues; `amount` = algorithmic HBM bytes (bound 0) or MFMA flops (bound 1)
// of the call.  Records are read back with dgtd_profile_dump().  Same key for both host binding layers.
struct DgtdProfScope {
  bool on;
  hipEvent_t a, b;
  hipStream_t st;
  int bound;
  double amount;
  char key[112];
  DgtdProfScope(hipStream_t st, int bound, double amount, const char* fmt, ...) __attribute__((format(printf, 5, 6)));
  ~DgtdProfScope();
};
#define DGTD_PROF(st, bound, amount, ...) DgtdProfScope dgtd_prof_scope_((hipStream_t)(st), (bound), (double)(amount), __VA_ARGS__)
enum { DGTD_HBM = 0, DGTD_MFMA = 1 };
static inline int dgtd_esize(dgtd_dtype dt) { return (dt == DGTD_BF16 || dt == DGTD_F16) ? 2 : (dt == DGTD_F64 ? 8 : 4); }

// ---- scalar load/store as float for both I/O dtypes ---------------------------------------------
template <typename T> __device__ __forceinline__ float to_f(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v) { return (T)v; }

// 16-byte vector of T: 4 floats or 8 bf16
template <typename T> struct Vec16;
template <> struct Vec16<float> { static constexpr int N = 4; typedef f32x4 type; };
template <> struct Vec16<bf16_t> { static constexpr int N = 8; typedef bf16x8 type; };
template <> struct Vec16<f16_t> { static constexpr int N = 8; typedef f16x8 type; };
// 8-byte vector of a 2-byte type
template <typename T> struct Vec8;
template <> struct Vec8<bf16_t> { typedef bf16x4 type; };
template <> struct Vec8<f16_t> { typedef f16x4 type; };

// the 16-bit MFMA of the I/O type: v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x16_f16 (same shape, same cycles)
__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// the two 16-bit I/O types (DGTD_BF16 / DGTD_F16)
#define DGTD_IS_HALF(dt) ((dt) == DGTD_BF16 || (dt) == DGTD_F16)
// run STMT with T_ bound to the element type of dtype code `dt` (float unless one of the 16-bit codes)
#define DGTD_DISPATCH(dt, STMT) do { if ((dt) == DGTD_BF16) { typedef bf16_t T_; STMT; } else if ((dt) == DGTD_F16) { typedef f16_t T_; STMT; } \
                                     else { typedef float T_; STMT; } } while (0)
// same for the 16-bit types only (callers have checked DGTD_IS_HALF)
#define DGTD_DISPATCH_HALF(dt, STMT) do { if ((dt) == DGTD_F16) { typedef f16_t T_; STMT; } else { typedef bf16_t T_; STMT; } } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// reduce across a power-of-two sub-group of `g` adjacent lanes
__device__ __forceinline__ float group_sum(float v, int g) {
  for (int o = g >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// row index of accumulator register `reg` of a 32x32 MFMA result in lane half `h` (guide §3)
__device__ __forceinline__ constexpr int mfma_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// A-operand fragment of M^T for v_mfma_f32_32x32x16_bf16 taken from a ROW-MAJOR bf16 tile in LDS with the gfx950 transposing
// read (ds_read_b64_tr_b16, CDNA4 guide T10): lane (r = lane&31, h = lane>>5), fragment element j holds
//   M[row0 + 8*(j>>2) + 4*h + (j&3)][col0 + r]
// i.e. exactly the k-permutation of an accumulator tile fed back as the other operand.  Per 16-lane group the instruction reads
// a 4-row x 16-column block: lane 4q+p supplies the address of row q, columns 4p..4p+3 and receives column (lane&15).
// Requirements: EXEC all ones, 8-byte aligned addresses (stride*2 and col0*2 multiples of 8).
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ typename Vec16<T>::type lds_tr_frag(const T* base, int stride, int row0, int col0, int lane) {
  const int i = lane & 15, g1 = (lane >> 4) & 1, h = lane >> 5;
  const T* p = base + (row0 + 4 * h + (i >> 2)) * stride + col0 + 16 * g1 + 4 * (i & 3);
  typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
  typedef typename Vec8<T>::type V4;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + 8 * stride));
  V4 l4 = __builtin_bit_cast(V4, lo), h4 = __builtin_bit_cast(V4, hi);
  typename Vec16<T>::type f;
#pragma unroll
  for (int j = 0; j < 4; ++j) { f[j] = l4[j]; f[4 + j] = h4[j]; }
  return f;
}

// erf-GELU (nn.GELU(), cod.py:854) on the VALU budget of an HBM-bound kernel.  Phi(x) = 0.5 erfc(-x/sqrt2) with erfc from
// Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32 rounding level) evaluated on |x| so the negative tail has no 1 + erf
// cancellation; the exponential exp(-x^2/2) is shared with the Gaussian term of the derivative.  ~14 VALU ops instead of ~50 for
// erff + expf.  Returns Phi(x); *pdf = exp(-x^2/2) / sqrt(2 pi).
__device__ __forceinline__ float gelu_phi(float x, float* pdf) {
  const float a = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.f));
  const float q = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-a * a * 1.4426950408889634f);
  const float hc = 0.5f * q * e;                     // 0.5 erfc(|x| / sqrt2)
  *pdf = 0.3989422804014327f * e;
  return x >= 0.f ? 1.f - hc : hc;
}
__device__ __forceinline__ float gelu_fast(float x) { float pdf; return x * gelu_phi(x, &pdf); }
__device__ __forceinline__ float gelu_grad_fast(float x) { float pdf; const float phi = gelu_phi(x, &pdf); return fmaf(x, pdf, phi); }

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// second stage of every two-stage column reduction (LayerNorm dgamma/dbeta, Linear bias gradients, layer-scale gradients): see
// dgtd_multi_reduce in elementwise.hip.  One entry = one partial buffer ws[nblocks][ncols] summed over its rows in a fixed order.
int dgtd_multi_reduce_impl(const dgtd_reduce_entry* entries, int n, hipStream_t st);
