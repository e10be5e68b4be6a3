// common.h — shared device/host helpers for libdgtd.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/dgtd.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- error plumbing (thread-local message, int status) ------------------------------------------
void dgtd_set_error(const char* fmt, ...);
#define DGTD_FAIL(code, ...) do { dgtd_set_error(__VA_ARGS__); return (code); } while (0)
#define DGTD_REQUIRE(cond, ...) do { if (!(cond)) DGTD_FAIL(2, __VA_ARGS__); } while (0)
#define DGTD_CHECK_LAUNCH(name) do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) DGTD_FAIL(3, "%s: launch failed: %s", name, hipGetErrorString(e_)); } while (0)

// ---- scalar load/store as float for both I/O dtypes ---------------------------------------------
template <typename T> __device__ __forceinline__ float to_f(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v) { return (T)v; }

// 16-byte vector of T: 4 floats or 8 bf16
template <typename T> struct Vec16;
template <> struct Vec16<float> { static constexpr int N = 4; typedef f32x4 type; };
template <> struct Vec16<bf16_t> { static constexpr int N = 8; typedef bf16x8 type; };

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

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
