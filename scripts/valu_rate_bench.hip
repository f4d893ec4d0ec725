// What does a VALU instruction cost on gfx950 -- alone on its SIMD, beside a second VALU wave, beside a wave that issues MFMAs back to back?
// Calibration for the LayerNorm + SiLU row arithmetic: plain fp32, transcendental (v_exp_f32 / v_rcp_f32 /
// v_rsq_f32), packed fp32, packed f16, conversions; and the two row functions (8 channels of a 128-channel row on 16 lanes: the shipped
// two-pass form and the "diet" form) as compiled code.  One workgroup of 8 waves; waves 0-3 (one per SIMD) are timed with s_memtime,
// waves 4-7 idle / run the same stream / run MFMAs.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate_bench scripts/valu_rate_bench.hip && /tmp/valu_rate_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MFMA(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(wa), "v"(xb))

enum { OP_FMA, OP_EXP, OP_RCP, OP_RSQ, OP_PKFMA32, OP_PKFMA16, OP_EXP16, OP_CVTBF, OP_ROW_OLD, OP_ROW_NEW, OP_ROW_NEW_NOSILU, OP_ROW_OLD_NOSILU, OP_COUNT };
static const char* kNames[OP_COUNT] = {"v_fma_f32", "v_exp_f32", "v_rcp_f32", "v_rsq_f32", "v_pk_fma_f32", "v_pk_fma_f16", "v_exp_f16", "v_cvt_pk_bf16_f32",
                                       "row LN+SiLU shipped (8 el)", "row LN+SiLU diet (8 el)", "row LN diet (8 el)", "row LN shipped (8 el)"};

template <int N>
__device__ __forceinline__ float group_sum_dpp16(float v) {
  // row_shr-free butterfly over 16 lanes with DPP quad_perm / row_ror (same instruction count as the product's helper)
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));   // row_ror 4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));   // row_ror 8
  return v;
}

#pragma clang fp contract(off)
template <bool SILU>
__device__ __forceinline__ void row_old(const u32x4& xin, const float (&g)[8], const float (&b)[8], float eps, u32x4& out) {
  float v[8];
  for (int q = 0; q < 4; ++q) {
    v[2 * q] = __builtin_bit_cast(float, xin[q] << 16);
    v[2 * q + 1] = __builtin_bit_cast(float, xin[q] & 0xffff0000u);
  }
  float s = 0.f;
  for (int e = 0; e < 8; ++e) s = s + v[e];
  const float mean = group_sum_dpp16<16>(s) * (1.0f / 128.0f);
  float q = 0.f, d[8];
  for (int e = 0; e < 8; ++e) {
    d[e] = v[e] - mean;
    q = __builtin_fmaf(d[e], d[e], q);
  }
  const float rstd = __builtin_amdgcn_rsqf(__builtin_fmaf(group_sum_dpp16<16>(q), 1.0f / 128.0f, eps));
  float o[8];
  for (int e = 0; e < 8; ++e) {
    float u = __builtin_fmaf(d[e] * rstd, g[e], b[e]);
    if constexpr (SILU) {
      const float ex = __builtin_amdgcn_exp2f(u * -1.4426950408889634f);
      u = u * __builtin_amdgcn_rcpf(ex + 1.0f);
    }
    o[e] = u;
  }
  for (int q2 = 0; q2 < 4; ++q2) {
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(o[2 * q2]), "v"(o[2 * q2 + 1]));
    out[q2] = r;
  }
}
// diet: one-pass statistics (sum and sum of squares together), t = x * rstd - mean * rstd (one fma), SiLU with the -log2(e) folded into the affine:
// a = t * g' + b' (g' = -log2e g, b' = -log2e b), out = a / ((1 + 2^a) * -log2e) = a * rcp(fma(2^a, c, c)), c = -log2e
template <bool SILU>
__device__ __forceinline__ void row_new(const u32x4& xin, const float (&g)[8], const float (&b)[8], float eps, u32x4& out) {
  float v[8];
  for (int q = 0; q < 4; ++q) {
    v[2 * q] = __builtin_bit_cast(float, xin[q] << 16);
    v[2 * q + 1] = __builtin_bit_cast(float, xin[q] & 0xffff0000u);
  }
  float s = 0.f, q = 0.f;
  for (int e = 0; e < 8; ++e) {
    s = s + v[e];
    q = __builtin_fmaf(v[e], v[e], q);
  }
  const float mean = group_sum_dpp16<16>(s) * (1.0f / 128.0f);
  const float ex2 = group_sum_dpp16<16>(q) * (1.0f / 128.0f);
  const float var = __builtin_fmaf(-mean, mean, ex2);
  const float rstd = __builtin_amdgcn_rsqf(var + eps);
  const float nm = -mean * rstd;
  float o[8];
  for (int e = 0; e < 8; ++e) {
    const float t = __builtin_fmaf(v[e], rstd, nm);
    float a = __builtin_fmaf(t, g[e], b[e]);
    if constexpr (SILU) {
      const float ex = __builtin_amdgcn_exp2f(a);
      const float den = __builtin_fmaf(ex, -1.4426950408889634f, -1.4426950408889634f);
      a = a * __builtin_amdgcn_rcpf(den);
    }
    o[e] = a;
  }
  for (int q2 = 0; q2 < 4; ++q2) {
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(o[2 * q2]), "v"(o[2 * q2 + 1]));
    out[q2] = r;
  }
}

// MODE 0: waves 4-7 leave; 1: they run the same stream; 2: they run MFMAs back to back for as long as waves 0-3 work
template <int OP, int MODE>
__global__ __launch_bounds__(1024) void bench(unsigned long long* out, float* sink, int reps, volatile int* flag) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  u32x4 wa = {0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, xb = wa;
  __shared__ int done;
  if (threadIdx.x == 0) done = 0;
  __syncthreads();
  if (wave >= 4) {
    if constexpr (MODE == 0) return;
    if constexpr (MODE == 2) {
      f32x16 a0, a1;
      for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
      __builtin_amdgcn_s_setprio(1);
      int n = 0;
      while (__hip_atomic_load(&done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4 && n < (1 << 22)) {
        MFMA(a0); MFMA(a1); MFMA(a0); MFMA(a1); MFMA(a0); MFMA(a1); MFMA(a0); MFMA(a1);
        MFMA(a0); MFMA(a1); MFMA(a0); MFMA(a1); MFMA(a0); MFMA(a1); MFMA(a0); MFMA(a1);
        n += 16;
      }
      if (lane == 0) sink[threadIdx.x] = a0[0] + a1[3] + n;
      return;
    }
  }
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = 0.5f + 0.01f * (lane + i);
  f32x2 pk[8];
  for (int i = 0; i < 8; ++i) pk[i] = f32x2{0.5f + 0.01f * lane, 0.25f + i};
  unsigned h[8];
  for (int i = 0; i < 8; ++i) h[i] = 0x3c003c00u + lane + i;
  const float k = 0.999f;
  const f32x2 k2 = {0.999f, 0.998f};
  const unsigned kh = 0x3bff3bffu;
  float g[8], b[8];
  for (int i = 0; i < 8; ++i) { g[i] = 1.0f + 0.001f * (lane + i); b[i] = 0.01f * i; }
  u32x4 xr = {0x3f803f80u + lane * 65537u, 0x3f003f80u, 0xbf803f80u + lane, 0x3f80bf80u};
  // let the MFMA waves get going
  if constexpr (MODE == 2) __builtin_amdgcn_s_sleep(64);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    if constexpr (OP == OP_FMA) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(k));
    } else if constexpr (OP == OP_EXP) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i]));
    } else if constexpr (OP == OP_RCP) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
    } else if constexpr (OP == OP_RSQ) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_rsq_f32 %0, %0" : "+v"(f[i]));
    } else if constexpr (OP == OP_PKFMA32) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(pk[i]) : "v"(k2));
    } else if constexpr (OP == OP_PKFMA16) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f16 %0, %0, %1, %0" : "+v"(h[i]) : "v"(kh));
    } else if constexpr (OP == OP_EXP16) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_exp_f16 %0, %0" : "+v"(h[i]));
    } else if constexpr (OP == OP_CVTBF) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h[i]) : "v"(f[i]), "v"(f[(i + 1) & 7]));
    } else if constexpr (OP == OP_ROW_OLD || OP == OP_ROW_OLD_NOSILU) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        u32x4 o;
        row_old<OP == OP_ROW_OLD>(xr, g, b, 1e-6f, o);
        xr[0] ^= o[0] & 0x00010001u; xr[1] ^= o[1] & 0x00010001u; xr[2] ^= o[2] & 0x00010001u; xr[3] ^= o[3] & 0x00010001u;
      }
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        u32x4 o;
        row_new<OP == OP_ROW_NEW>(xr, g, b, 1e-6f, o);
        xr[0] ^= o[0] & 0x00010001u; xr[1] ^= o[1] & 0x00010001u; xr[2] ^= o[2] & 0x00010001u; xr[3] ^= o[3] & 0x00010001u;
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) {
    __hip_atomic_fetch_add(&done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    out[wave] = t1 - t0;
  }
  float acc = 0.f;
  for (int i = 0; i < 8; ++i) acc += f[i] + pk[i][0] + pk[i][1] + __builtin_bit_cast(float, h[i]);
  acc += __builtin_bit_cast(float, xr[0] ^ xr[1] ^ xr[2] ^ xr[3]);
  sink[threadIdx.x] = acc;
}

template <int OP, int MODE>
double run(unsigned long long* d_out, float* d_sink, int* d_flag, int reps, int threads = 512) {
  double best = 1e30;
  for (int it = 0; it < 3; ++it) {
    hipLaunchKernelGGL((bench<OP, MODE>), dim3(1), dim3(threads), 0, 0, d_out, d_sink, reps, d_flag);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0;
    for (int w = 0; w < 4; ++w) m += (double)h[w] / 4;
    if (m < best) best = m;
  }
  return best;
}

template <int OP>
void line(unsigned long long* d_out, float* d_sink, int* d_flag) {
  const int reps = 200;
  const bool row = OP >= OP_ROW_OLD;
  const double per = row ? 4.0 * reps : 64.0 * reps;     // instructions (or rows) per timed region
  const double a = run<OP, 0>(d_out, d_sink, d_flag, reps) / per;
  const double b = run<OP, 1>(d_out, d_sink, d_flag, reps) / per;
  const double c = run<OP, 2>(d_out, d_sink, d_flag, reps) / per;
  const double b3 = run<OP, 1>(d_out, d_sink, d_flag, reps, 768) / per;
  const double b4 = run<OP, 1>(d_out, d_sink, d_flag, reps, 1024) / per;
  const double c3 = run<OP, 2>(d_out, d_sink, d_flag, reps, 768) / per;      // waves 4-11: MFMAs (two MFMA waves a SIMD)
  printf("%-30s  alone %8.2f   2 waves/SIMD %8.2f   3 waves/SIMD %8.2f   4 waves/SIMD %8.2f   beside 1 MFMA wave %8.2f   beside 2 MFMA waves %8.2f   cycles per %s\n",
         kNames[OP], a, b, b3, b4, c, c3, row ? "row slice (8 elements a lane)" : "instruction");
}

int main() {
  unsigned long long* d_out;
  float* d_sink;
  int* d_flag;
  hipMalloc(&d_out, 64 * sizeof(unsigned long long));
  hipMalloc(&d_sink, 1024 * sizeof(float));
  hipMalloc(&d_flag, sizeof(int));
  hipMemset(d_flag, 0, sizeof(int));
  line<OP_FMA>(d_out, d_sink, d_flag);
  line<OP_EXP>(d_out, d_sink, d_flag);
  line<OP_RCP>(d_out, d_sink, d_flag);
  line<OP_RSQ>(d_out, d_sink, d_flag);
  line<OP_PKFMA32>(d_out, d_sink, d_flag);
  line<OP_PKFMA16>(d_out, d_sink, d_flag);
  line<OP_EXP16>(d_out, d_sink, d_flag);
  line<OP_CVTBF>(d_out, d_sink, d_flag);
  line<OP_ROW_OLD>(d_out, d_sink, d_flag);
  line<OP_ROW_NEW>(d_out, d_sink, d_flag);
  line<OP_ROW_OLD_NOSILU>(d_out, d_sink, d_flag);
  line<OP_ROW_NEW_NOSILU>(d_out, d_sink, d_flag);
  return 0;
}
