// Shared device/host helpers for libvidtok_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/vidtok_amd.h"

typedef __bf16 bf16_t;
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
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

// The two 16-bit storage types (vt_dtype VT_BF16 / VT_F16): one word = two values (the lower address in bits 0..15), fp32 <-> storage
// conversions (round to nearest even both ways; fp16 results beyond 65 504 become +-inf, as the reference's autocast(float16) convolutions
// do), and the matrix instruction that takes the type -- v_mfma_f32_*_bf16 / _f16 run at the same rate on gfx950.  Kernels written for
// "a 16-bit type" take one of them as a template parameter and say h16<H>::...
template <typename H>
struct h16;
template <>
struct h16<bf16_t> {
  typedef bf16x8 vec8;
  static __device__ __forceinline__ float lo(uint32_t w) { return __uint_as_float(w << 16); }
  static __device__ __forceinline__ float hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
  static __device__ __forceinline__ uint32_t pack(float a, float b) { return pack_bf16x2(a, b); }
  static __device__ __forceinline__ f32x16 mfma32(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <>
struct h16<f16_t> {
  typedef f16x8 vec8;
  typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ float lo(uint32_t w) { return (float)__builtin_bit_cast(f16x2_t, w)[0]; }   // v_cvt_f32_f16
  static __device__ __forceinline__ float hi(uint32_t w) { return (float)__builtin_bit_cast(f16x2_t, w)[1]; }   // ... src0_sel:WORD_1
  static __device__ __forceinline__ uint32_t pack(float a, float b) {                                          // one v_cvt_pk_f16_f32
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
  }
  static __device__ __forceinline__ f32x16 mfma32(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};
template <typename T>
struct is_h16 {
  [[maybe_unused]] static constexpr bool value = false;
};
template <>
struct is_h16<bf16_t> {
  [[maybe_unused]] static constexpr bool value = true;
};
template <>
struct is_h16<f16_t> {
  [[maybe_unused]] static constexpr bool value = true;
};
// vt_dtype code of a storage type
template <typename T>
struct dtype_code;
template <>
struct dtype_code<float> {
  [[maybe_unused]] static constexpr int value = VT_F32;
};
template <>
struct dtype_code<bf16_t> {
  [[maybe_unused]] static constexpr int value = VT_BF16;
};
template <>
struct dtype_code<f16_t> {
  [[maybe_unused]] static constexpr int value = VT_F16;
};
inline bool vt_is_h16(int dtype) { return dtype == VT_BF16 || dtype == VT_F16; }

template <typename T>
__device__ __forceinline__ float to_f32(T v);
template <>
__device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) {
  return bf16_bits_to_f32((uint32_t)__builtin_bit_cast(uint16_t, v));
}
template <>
__device__ __forceinline__ float to_f32<f16_t>(f16_t v) { return (float)v; }
template <typename T>
__device__ __forceinline__ T from_f32(float v);
template <>
__device__ __forceinline__ f16_t from_f32<f16_t>(float v) { return (f16_t)v; }
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

// ---- LayerNorm(+SiLU) row arithmetic of the 16-bit kernels (bf16 / fp16 storage), "diet" form (round 6) --------------------------------------
// Every fused LayerNorm site is VALU time (scripts/valu_rate_bench.hip): the two-pass form costs 9.5 plain instructions + 2 transcendentals an
// element (unpack, sum, centre, square-sum, scale, affine, exp argument, exp2, + 1, rcp, multiply, half a pack).  This form costs 7.5 + 2:
//   * both moments in ONE pass (var = E[x^2] - mean^2 in fp32 over 128 / 256 values, clamped at 0): the two lane reductions are independent
//     (their DPP wait states fill each other) and the centring disappears into
//   * t = x rstd + (-mean rstd): one fma instead of a subtraction and a multiplication;
//   * SiLU with -log2(e) folded into the affine the CALLER passes (g' = -log2e g, b' = -log2e b, see ln_fold): a = t g' + b' = -log2e u, and
//     u sigmoid(u) = u / (1 + 2^a) = a / (-log2e (1 + 2^a)) = a * rcp(fma(2^a, -log2e, -log2e)): no separate exponent argument.
// The fp32-storage kernels (fp32 and split-bf16 modes, held to 1e-3 of the oracle with FSQ codes bit-exact) keep the two-pass form.
// Nothing here depends on FMA contraction (fused operations are spelled out, the rest are single operations).
[[maybe_unused]] constexpr float kNegLog2e = -1.4426950408889634f;
// the affine a site hands to ln_row8: scaled for the SiLU form, as is for a plain LayerNorm
__device__ __forceinline__ float ln_fold(float v, bool silu) { return silu ? v * kNegLog2e : v; }
// statistics of a row held as 8 values a lane on G lanes (C = 8 G channels): returns rstd, *nm = -mean rstd
template <int G>
__device__ __forceinline__ float ln_row_stats(float s, float q, float eps, float* nm) {
  constexpr float kInvC = 1.0f / (8.0f * G);
  const float mean = group_sum_dpp<G>(s) * kInvC;
  const float ex2 = group_sum_dpp<G>(q) * kInvC;
  const float var = fmaxf(__builtin_fmaf(-mean, mean, ex2), 0.0f);
  const float rstd = __builtin_amdgcn_rsqf(var + eps);
  *nm = -mean * rstd;
  return rstd;
}
__device__ __forceinline__ float ln_tail(float t, float g, float b, bool silu) {
  float a = __builtin_fmaf(t, g, b);
  if (silu) {
    const float den = __builtin_fmaf(__builtin_amdgcn_exp2f(a), kNegLog2e, kNegLog2e);
    a = a * __builtin_amdgcn_rcpf(den);
  }
  return a;
}
template <int G, bool SILU>
__device__ __forceinline__ void ln_row8(const float (&v)[8], const float (&g)[8], const float (&b)[8], float eps, float (&o)[8]) {
  float s = v[0], q = v[0] * v[0];
#pragma unroll
  for (int e = 1; e < 8; ++e) {
    s = s + v[e];
    q = __builtin_fmaf(v[e], v[e], q);
  }
  float nm;
  const float rstd = ln_row_stats<G>(s, q, eps, &nm);
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = ln_tail(__builtin_fmaf(v[e], rstd, nm), g[e], b[e], SILU);
}

__device__ __forceinline__ float wave_sum(float v, int width) {
  for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v, int width) {
  for (int o = width >> 1; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
