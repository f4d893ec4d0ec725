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
// round-to-nearest-even fp32 -> bf16 bits: the native conversion (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
  return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)f);
}
// two values -> one word (lo in bits 0..15): ONE v_cvt_pk_bf16_f32, same rounding as f32_to_bf16_bits (the scalar form
// packs with two conversions, a shift and an OR -- four issue slots a pair)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
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

// Sum over aligned groups of `WIDTH` lanes (WIDTH = 4..64, power of two); every lane gets the total.
// The first four butterfly levels stay in the VALU (DPP quad_perm / row_half_mirror / row_mirror);
// only the 16- and 32-lane levels go through ds_bpermute.
template <int WIDTH>
__device__ __forceinline__ float group_sum_dpp(float v) {
#define VT_DPP_ADD(ctrl) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, false))
  if (WIDTH >= 2) VT_DPP_ADD(0xB1);    // quad_perm [1,0,3,2]
  if (WIDTH >= 4) VT_DPP_ADD(0x4E);    // quad_perm [2,3,0,1]
  if (WIDTH >= 8) VT_DPP_ADD(0x141);   // row_half_mirror
  if (WIDTH >= 16) VT_DPP_ADD(0x140);  // row_mirror
#undef VT_DPP_ADD
  if (WIDTH >= 32) v += __shfl_xor(v, 16, 64);
  if (WIDTH >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

// x * sigmoid(x) with the hardware reciprocal (1 ulp) instead of an IEEE division sequence
__device__ __forceinline__ float silu_fast(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

__device__ __forceinline__ float wave_sum(float v, int width) {
  for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v, int width) {
  for (int o = width >> 1; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
