// Types and helpers of the convolution kernel (conv_igemm.hip).
#pragma once
#include "common.h"
#include "options.h"

namespace {

constexpr int kRowBytes = 128;     // bytes of K per tile row per step
constexpr int kMaxDevices = 64;    // per-device launch state (function attributes)

// Division of a 32-bit unsigned by a launch-invariant divisor (Granlund-Montgomery round-up form): the host computes
// (mul, shift), the device needs one mul_hi and three cheap ops instead of the ~25-instruction reciprocal sequence
// the compiler emits for `/` by a runtime value.  12 of them per lane decode the pixel coordinates of a tile.
struct FastDiv {
  unsigned mul, shift, d;
};
inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f;
  f.d = d;
  if (d <= 1) { f.mul = 0; f.shift = 0; return f; }
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  f.mul = (unsigned)((((1ull << 32) * ((1ull << l) - d)) / d) + 1);
  f.shift = l - 1;
  return f;
}
__device__ __forceinline__ unsigned fast_div(unsigned n, const FastDiv& f) {
  if (f.d <= 1) return n;                       // uniform
  const unsigned t = __umulhi(n, f.mul);
  return (t + ((n - t) >> 1)) >> f.shift;
}

struct ConvArgs;
__device__ __forceinline__ long long out_row(const ConvArgs& p, int m);

struct ConvArgs {
  const char* x;
  const char* w;
  const float* bias;
  char* y;
  const char* res;
  const char* cache;
  const float* mix_factor;
  const float* ln_gamma;
  const float* ln_beta;
  char* ln_out;
  int B, Ti, Hi, Wi, Cin;
  int To, Ho, Wo, Cout;
  int ldw, ldy;
  int KT, KH, KW;
  int st, sh, sw;
  int pt, ph, pw;
  int tmode, ncache;
  int ups_t, ups_s;
  int res_mode, res_tshift, Tr, ldr;
  int out_layout, t_trim;
  int M, K, ntaps, nsteps;
  int m_tiles, n_tiles;
  int ln_mode, ln_keep_y, ldn;  // fused LayerNorm of the result (only set when the lds128 epilogue will run)
  float ln_eps;
  FastDiv fd_wo, fd_ho, fd_to;   // pixel index -> (b, to, ho, wo)
  FastDiv fd_hw;                 // pixel index -> frame index (output frame interleave)
  int yt_mul;                    // output frame of computed frame f is f * yt_mul + yt_off (1, 0 = plain)
  long long yt_step, yt_base;    // (yt_mul - 1) * Ho*Wo and yt_off * Ho*Wo rows
  int ys_mul, ys_oh, ys_ow;      // 2: computed pixel (ho, wo) is output pixel (2 ho + ys_oh, 2 wo + ys_ow) of a 2Ho x 2Wo frame
  int lds_epi;                 // 128 x 128 tile: epilogue transposed through the LDS (coalesced rows)
  int hw_tiles;                // > 0: pixel tiles per frame, tile order (b, hw tile, t) -- see launch_variant
  unsigned x_bytes, w_bytes;   // BUF path: descriptor extents (0 = tensors too large, use pointers)
  unsigned c_bytes;            // BUF path in cache mode: extent of the cache tensor [B][ncache][Hi][Wi][Cin]
  long long xs_z, ws_z, ys_z, rs_z;
  int in8_rt, in8_ct, in8_ctl2, in8_segs;   // conv_in8_kernel: tile = in8_rt rows x in8_ct (= 2^in8_ctl2) columns; 64-pixel segments per patch row
  int tskip;                   // 1: zero-padded time taps in front of the clip are skipped per tile (tile inside one output frame, tmode ZERO; launch_variant)
  int ksplit;                  // split-K: blockIdx.z = tap plane, the walk covers that plane only; 1: planes = kt, 2: planes = kh (KT = 1)
  unsigned plane_bytes;        //      bytes of one tap plane in a weight row (KH * KW * Cin, or KW * Cin, elements)
  int nt_store;                // 1: the LDS epilogues write y / LayerNorm(y) with streaming (nt) stores (outputs of at least conv_nt_mb MiB)
  unsigned long long* prof;    // PROF instantiation only (vt_conv_profile): cycle stamps of workgroup 0
  int prof_mode;               // PROF instantiation of conv_ws2.hip only: option ws_prof_mode
};

// Split-bf16 arithmetic (vt_dtype VT_BF16X3, "bf16x3"): fp32 STORAGE on both sides of the convolution, bf16 MATRIX cores
// inside it.  An fp32 value v is carried as two bf16 planes, hi = bf16(v) (round to nearest even) and lo = bf16(v - hi):
// hi + lo holds 16-17 significant bits of v, and a product x * w is taken as x_lo * w_hi + x_hi * w_lo + x_hi * w_hi on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation (the x_lo * w_lo term, ~2^-18 relative, is dropped): three bf16 MFMAs
// per 16 k-values = 96 matrix-pipe cycles against 512 for the eight v_mfma_f32_32x32x2_f32 of the fp32 mode.  The weights
// are split once on the host (vidtok_amd/packing.py::pack_split3) and stored per row as blocks of 16 k-values,
// [hi 16 x bf16 | lo 16 x bf16] = 64 bytes = the bytes of 16 fp32 values, so the staging code (element size 4, K step =
// ROWB bytes of both operands) is the fp32 one; activations are split in registers right behind their fragment read.
struct split3_t {
  float v;
};
template <typename MT>
struct is_split3 {
  [[maybe_unused]] static constexpr bool value = false;
};
template <>
struct is_split3<split3_t> {
  [[maybe_unused]] static constexpr bool value = true;
};
// element type an MT tensor is stored in
template <typename MT>
struct storage_of {
  typedef MT type;
};
template <>
struct storage_of<split3_t> {
  typedef float type;
};

// 8 consecutive fp32 k-values (two 16-byte fragment reads) -> their bf16 hi / lo fragments (element j in bits 16 (j & 1)
// of word j >> 1: the order of the packed weight planes)
__device__ __forceinline__ void split3_x(const u32x4& r0, const u32x4& r1, u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float f0 = __uint_as_float(e < 2 ? r0[2 * e] : r1[2 * e - 4]);
    const float f1 = __uint_as_float(e < 2 ? r0[2 * e + 1] : r1[2 * e - 3]);
    const uint32_t h = pack_bf16x2(f0, f1);
    hi[e] = h;
    // the two subtractions as scalar v_sub_f32: left to the compiler they become one v_pk_add_f32, and a packed-fp32
    // instruction stalls the matrix pipe of its SIMD (profiles/r02_mfma_issue_microbench.txt) -- this runs beside MFMAs
    float d0, d1;
    asm("v_sub_f32_e32 %0, %1, %2" : "=v"(d0) : "v"(f0), "v"(h << 16));
    asm("v_sub_f32_e32 %0, %1, %2" : "=v"(d1) : "v"(f1), "v"(h & 0xffff0000u));
    lo[e] = pack_bf16x2(d0, d1);
  }
}
__device__ __forceinline__ void mma_bf16(const u32x4& wfrag, const u32x4& xfrag, f32x16& acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wfrag), __builtin_bit_cast(bf16x8, xfrag), acc, 0, 0, 0);
}

template <typename MT>
__device__ __forceinline__ void mma_step(const u32x4& wfrag, const u32x4& xfrag, f32x16& acc);

template <>
__device__ __forceinline__ void mma_step<bf16_t>(const u32x4& wfrag, const u32x4& xfrag, f32x16& acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wfrag),
                                                __builtin_bit_cast(bf16x8, xfrag), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_step<f16_t>(const u32x4& wfrag, const u32x4& xfrag, f32x16& acc) {
  acc = h16<f16_t>::mfma32(wfrag, xfrag, acc);
}
template <>
__device__ __forceinline__ void mma_step<float>(const u32x4& wfrag, const u32x4& xfrag, f32x16& acc) {
#pragma unroll
  for (int e = 0; e < 4; ++e)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(wfrag[e]), __uint_as_float(xfrag[e]),
                                               acc, 0, 0, 0);
}

// XCD-aware bijective remap of the linear block id (MI355X guide T1): block b runs on XCD b%8;
// give every XCD a contiguous chunk of the tile sequence.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int xcd = bid & 7, loc = bid >> 3;
  const int q = nblk >> 3, r = nblk & 7;
  const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + loc;
}

template <typename TOut>
struct Quad;   // 4 consecutive channels as stored
template <>
struct Quad<float> {
  f32x4 v;
  __device__ __forceinline__ float get(int e) const { return v[e]; }
};
template <typename H>
struct Quad16 {     // a 16-bit storage type: two words
  u32x2 v;
  __device__ __forceinline__ float get(int e) const {
    const uint32_t w = v[e >> 1];
    return (e & 1) ? h16<H>::hi(w) : h16<H>::lo(w);
  }
};
template <>
struct Quad<bf16_t> : Quad16<bf16_t> {};
template <>
struct Quad<f16_t> : Quad16<f16_t> {};
template <typename TOut>
__device__ __forceinline__ void store_quad(TOut* p, const float (&v)[4]);
template <>
__device__ __forceinline__ void store_quad<float>(float* p, const float (&v)[4]) {
  f32x4 t;
  t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3];
  *reinterpret_cast<f32x4*>(p) = t;
}
template <>
__device__ __forceinline__ void store_quad<bf16_t>(bf16_t* p, const float (&v)[4]) {
  u32x2 t;
  t[0] = h16<bf16_t>::pack(v[0], v[1]);
  t[1] = h16<bf16_t>::pack(v[2], v[3]);
  *reinterpret_cast<u32x2*>(p) = t;
}
template <>
__device__ __forceinline__ void store_quad<f16_t>(f16_t* p, const float (&v)[4]) {
  u32x2 t;
  t[0] = h16<f16_t>::pack(v[0], v[1]);
  t[1] = h16<f16_t>::pack(v[2], v[3]);
  *reinterpret_cast<u32x2*>(p) = t;
}

// 8 consecutive channels of one pixel row as stored (16 B bf16 / 32 B fp32)
template <typename TOut>
struct Oct;
// A 16-byte streaming store.  Inline assembly on purpose: written as `if (nt) __builtin_nontemporal_store(..) else plain store` the two
// stores to one address are merged by the optimiser into ONE plain store -- the hint vanishes (round 6: the first build of option
// conv_nt_mb compiled to 688 global_store_dwordx4 and not one "nt"; measured, it did nothing).  The compiler's wait-count bookkeeping
// does not see the store; nothing ever waits on it, and unseen stores can only make a counted wait for a later load stricter.
__device__ __forceinline__ void store16_nt(void* p, const u32x4& v) {
  asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
}

template <>
struct Oct<float> {
  f32x4 lo, hi;
  __device__ __forceinline__ void load(const float* p) {
    lo = *reinterpret_cast<const f32x4*>(p);
    hi = *reinterpret_cast<const f32x4*>(p + 4);
  }
  __device__ __forceinline__ float get(int e) const { return e < 4 ? lo[e] : hi[e - 4]; }
  static __device__ __forceinline__ void store(float* p, const float (&v)[8], bool nt = false) {
    f32x4 a, b;
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[e] = v[e]; b[e] = v[4 + e]; }
    if (nt) {
      store16_nt(p, __builtin_bit_cast(u32x4, a));
      store16_nt(p + 4, __builtin_bit_cast(u32x4, b));
    } else {
      *reinterpret_cast<f32x4*>(p) = a;
      *reinterpret_cast<f32x4*>(p + 4) = b;
    }
  }
};
template <typename H>
struct Oct16 {      // a 16-bit storage type: one 16-byte access
  u32x4 w;
  __device__ __forceinline__ void load(const H* p) { w = *reinterpret_cast<const u32x4*>(p); }
  __device__ __forceinline__ float get(int e) const {
    const uint32_t t = w[e >> 1];
    return (e & 1) ? h16<H>::hi(t) : h16<H>::lo(t);
  }
  // nt: streaming store (the launcher sets ConvArgs::nt_store for outputs far larger than the caches: the rows do not displace the weights and
  // halo rows the next tiles read again -- whole step + 0.6 ... 0.75 %, profiles/r06_nt_stores_ab.txt)
  static __device__ __forceinline__ void store(H* p, const float (&v)[8], bool nt = false) {
    u32x4 t;
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = h16<H>::pack(v[2 * e], v[2 * e + 1]);
    if (nt) store16_nt(p, t);
    else *reinterpret_cast<u32x4*>(p) = t;
  }
};
template <>
struct Oct<bf16_t> : Oct16<bf16_t> {};
template <>
struct Oct<f16_t> : Oct16<f16_t> {};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct TagTrue { [[maybe_unused]] static constexpr bool value = true; };
struct TagFalse { [[maybe_unused]] static constexpr bool value = false; };

// Output pixel row of computed pixel m.  With yt_mul > 1 the launch computes every yt_mul-th frame of the output
// tensor (the parity classes of a convolution over a x2 frame-repeated input, see vt_conv_desc.yt_mul).
__device__ __forceinline__ long long out_row(const ConvArgs& p, int m) {
  if (p.ys_mul == 2) {                                            // uniform: spatial parity class (see vt_conv_desc.ys_mul)
    const unsigned f = fast_div((unsigned)m, p.fd_hw);
    const unsigned hw = (unsigned)m - f * (unsigned)(p.Ho * p.Wo);
    const unsigned ho = fast_div(hw, p.fd_wo);
    const unsigned wo = hw - ho * (unsigned)p.Wo;
    return ((long long)f * (2 * p.Ho) + (2 * ho + p.ys_oh)) * (2 * p.Wo) + (2 * wo + p.ys_ow);
  }
  if (p.yt_mul == 1) return m;                                    // uniform
  return (long long)m + (long long)fast_div((unsigned)m, p.fd_hw) * p.yt_step + p.yt_base;
}

}  // namespace
