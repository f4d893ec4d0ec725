// Shared device/host helpers for libvidtok_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/vidtok_amd.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

// ---- error plumbing ---------------------------------------------------------------------------
void vt_set_error(const char* fmt, ...);

#define VT_CHECK_ARG(cond, ...)      \
  do {                               \
    if (!(cond)) {                   \
      vt_set_error(__VA_ARGS__);     \
      return VT_ERR_ARG;             \
    }                                \
  } while (0)

#define VT_CHECK_HIP(expr)                                                              \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      vt_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return VT_ERR_HIP;                                                                \
    }                                                                                   \
  } while (0)

#define VT_CHECK_LAUNCH()                                                               \
  do {                                                                                  \
    hipError_t _e = hipGetLastError();                                                  \
    if (_e != hipSuccess) {                                                             \
      vt_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
      return VT_ERR_HIP;                                                                \
    }                                                                                   \
  } while (0)

// ---- scalar conversions ------------------------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t bits16) {
  return __uint_as_float(bits16 << 16);
}
// round-to-nearest-even fp32 -> bf16 bits (NaN kept quiet)
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}

template <typename T>
__device__ __forceinline__ float to_f32(T v);
template <>
__device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) {
  return bf16_bits_to_f32((uint32_t)__builtin_bit_cast(uint16_t, v));
}
template <typename T>
__device__ __forceinline__ T from_f32(float v);
template <>
__device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) {
  return __builtin_bit_cast(bf16_t, (uint16_t)f32_to_bf16_bits(v));
}

// x * sigmoid(x), the reference `nonlinearity` (model_3dcausal.py:26-27), in fp32.
__device__ __forceinline__ float silu_f32(float x) { return x / (1.0f + __expf(-x)); }

__device__ __forceinline__ float wave_sum(float v, int width) {
  for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v, int width) {
  for (int o = width >> 1; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
