// What does one wave pay per instruction around its MFMAs?  Loops of 8 x v_mfma_f32_32x32x16_bf16 with different
// fillers behind every MFMA, timed with s_memtime on wave 0 of a single workgroup (1 or 2 waves per SIMD), printed as
// shader cycles per MFMA.  Calibration for DESIGN.md section 5: how much VALU / LDS / transcendental work rides in an
// MFMA's shadow on gfx950 when the SIMD has only one (or two) waves to issue from.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/build/mfma_issue_bench scripts/mfma_issue_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MFMA(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(wa), "v"(xb))
#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(k))
#define PKFMA(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(k2))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define DSR(d) asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(laddr))
#define LGKM0 asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <int V>
__global__ __launch_bounds__(512) void bench(unsigned long long* out, float* sink, int reps) {
  __shared__ __attribute__((aligned(16))) char lds[8192];
  const int lane = threadIdx.x & 63;
  u32x4 wa = {0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, xb = wa;
  f32x16 a0, a1, a2, a3;
  for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; a2[i] = 0.f; a3[i] = 0.f; }
  float f0 = lane, f1 = 1.f, f2 = 2.f, f3 = 3.f, f4 = 4.f, f5 = 5.f, f6 = 6.f, f7 = 7.f, k = 0.999f;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 p0 = {1.f, 2.f}, p1 = {3.f, 4.f}, p2 = {5.f, 6.f}, p3 = {7.f, 8.f}, k2 = {0.999f, 0.998f};
  u32x4 d0, d1;
  const unsigned laddr = (unsigned)(size_t)(lds) + lane * 16;
  ((u32x4*)lds)[threadIdx.x & 255] = wa;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int r = 0; r < reps; ++r) {
    if constexpr (V == 0) {            // 2 accumulators alternating, nothing else
      MFMA(a0); MFMA(a1); MFMA(a0); MFMA(a1); MFMA(a0); MFMA(a1); MFMA(a0); MFMA(a1);
    } else if constexpr (V == 1) {     // 4 accumulators rotating
      MFMA(a0); MFMA(a1); MFMA(a2); MFMA(a3); MFMA(a0); MFMA(a1); MFMA(a2); MFMA(a3);
    } else if constexpr (V == 2) {     // one accumulator
      MFMA(a0); MFMA(a0); MFMA(a0); MFMA(a0); MFMA(a0); MFMA(a0); MFMA(a0); MFMA(a0);
    } else if constexpr (V == 3) {     // 2 acc + one ds_read_b128 behind each
      MFMA(a0); DSR(d0); MFMA(a1); DSR(d1); MFMA(a0); DSR(d0); MFMA(a1); DSR(d1); MFMA(a0); DSR(d0); MFMA(a1); DSR(d1); MFMA(a0); DSR(d0); MFMA(a1); DSR(d1);
      LGKM0;
    } else if constexpr (V == 4) {     // 2 acc + 4 independent v_fma behind each
#define F4 FMA(f0); FMA(f1); FMA(f2); FMA(f3)
      MFMA(a0); F4; MFMA(a1); F4; MFMA(a0); F4; MFMA(a1); F4; MFMA(a0); F4; MFMA(a1); F4; MFMA(a0); F4; MFMA(a1); F4;
    } else if constexpr (V == 5) {     // 2 acc + 8 independent v_fma
#define F8 FMA(f0); FMA(f1); FMA(f2); FMA(f3); FMA(f4); FMA(f5); FMA(f6); FMA(f7)
      MFMA(a0); F8; MFMA(a1); F8; MFMA(a0); F8; MFMA(a1); F8; MFMA(a0); F8; MFMA(a1); F8; MFMA(a0); F8; MFMA(a1); F8;
    } else if constexpr (V == 6) {     // 2 acc + 16 v_fma
      MFMA(a0); F8; F8; MFMA(a1); F8; F8; MFMA(a0); F8; F8; MFMA(a1); F8; F8; MFMA(a0); F8; F8; MFMA(a1); F8; F8; MFMA(a0); F8; F8; MFMA(a1); F8; F8;
    } else if constexpr (V == 7) {     // 2 acc + 4 v_pk_fma
#define P4 PKFMA(p0); PKFMA(p1); PKFMA(p2); PKFMA(p3)
      MFMA(a0); P4; MFMA(a1); P4; MFMA(a0); P4; MFMA(a1); P4; MFMA(a0); P4; MFMA(a1); P4; MFMA(a0); P4; MFMA(a1); P4;
    } else if constexpr (V == 8) {     // 2 acc + 2 v_exp
#define E2 EXP(f0); EXP(f1)
      MFMA(a0); E2; MFMA(a1); E2; MFMA(a0); E2; MFMA(a1); E2; MFMA(a0); E2; MFMA(a1); E2; MFMA(a0); E2; MFMA(a1); E2;
    } else if constexpr (V == 9) {     // no MFMA: 64 v_fma (8 independent chains)
      F8; F8; F8; F8; F8; F8; F8; F8;
    } else if constexpr (V == 10) {    // no MFMA: 16 v_exp (2 chains)
      E2; E2; E2; E2; E2; E2; E2; E2;
    } else if constexpr (V == 11) {    // no MFMA: 32 v_pk_fma (4 chains)
      P4; P4; P4; P4; P4; P4; P4; P4;
    } else if constexpr (V == 12) {    // no MFMA: 8 ds_read_b128
      DSR(d0); DSR(d1); DSR(d0); DSR(d1); DSR(d0); DSR(d1); DSR(d0); DSR(d1); LGKM0;
    } else if constexpr (V == 13) {    // 4 acc + 8 v_fma
      MFMA(a0); F8; MFMA(a1); F8; MFMA(a2); F8; MFMA(a3); F8; MFMA(a0); F8; MFMA(a1); F8; MFMA(a2); F8; MFMA(a3); F8;
    } else if constexpr (V == 14) {    // 4 acc + ds_read
      MFMA(a0); DSR(d0); MFMA(a1); DSR(d1); MFMA(a2); DSR(d0); MFMA(a3); DSR(d1); MFMA(a0); DSR(d0); MFMA(a1); DSR(d1); MFMA(a2); DSR(d0); MFMA(a3); DSR(d1);
      LGKM0;
    } else if constexpr (V == 15) {    // 2 acc + ds_read + 4 fma + 2 exp (a row-phase slice)
#define S1 DSR(d0); F4; E2
      MFMA(a0); S1; MFMA(a1); S1; MFMA(a0); S1; MFMA(a1); S1; MFMA(a0); S1; MFMA(a1); S1; MFMA(a0); S1; MFMA(a1); S1;
      LGKM0;
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + p0[0] + p1[1] + p2[0] + p3[1] + a0[0] + a1[1] + a2[2] + a3[3] + __uint_as_float(d0[0] ^ d1[1]);
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if (s == 12345.678f) sink[threadIdx.x] = s;
}

template <int V>
void run(const char* what, int per_rep, const char* unit, unsigned long long* d_out, float* d_sink) {
  const int reps = 2000;
  for (int waves = 4; waves <= 8; waves += 4) {
    unsigned long long t = 0;
    for (int i = 0; i < 2; ++i) {
      hipLaunchKernelGGL(bench<V>, dim3(1), dim3(64 * waves), 0, 0, d_out, d_sink, reps);
      hipMemcpy(&t, d_out, sizeof(t), hipMemcpyDeviceToHost);
    }
    printf("%-58s %d wave/SIMD: %7.1f cycles per %s\n", what, waves / 4, (double)t / reps / per_rep, unit);
  }
}

int main() {
  unsigned long long* d_out; float* d_sink;
  hipMalloc(&d_out, 64); hipMalloc(&d_sink, 4096);
  run<0>("MFMA 32x32x16 bf16, 2 accumulators alternating", 8, "MFMA", d_out, d_sink);
  run<1>("MFMA, 4 accumulators rotating", 8, "MFMA", d_out, d_sink);
  run<2>("MFMA, 1 accumulator (dependent chain)", 8, "MFMA", d_out, d_sink);
  run<3>("MFMA (2 acc) + 1 ds_read_b128 each", 8, "MFMA", d_out, d_sink);
  run<14>("MFMA (4 acc) + 1 ds_read_b128 each", 8, "MFMA", d_out, d_sink);
  run<4>("MFMA (2 acc) + 4 v_fma_f32 each", 8, "MFMA", d_out, d_sink);
  run<5>("MFMA (2 acc) + 8 v_fma_f32 each", 8, "MFMA", d_out, d_sink);
  run<13>("MFMA (4 acc) + 8 v_fma_f32 each", 8, "MFMA", d_out, d_sink);
  run<6>("MFMA (2 acc) + 16 v_fma_f32 each", 8, "MFMA", d_out, d_sink);
  run<7>("MFMA (2 acc) + 4 v_pk_fma_f32 each", 8, "MFMA", d_out, d_sink);
  run<8>("MFMA (2 acc) + 2 v_exp_f32 each", 8, "MFMA", d_out, d_sink);
  run<15>("MFMA (2 acc) + ds_read + 4 v_fma + 2 v_exp each", 8, "MFMA", d_out, d_sink);
  run<9>("v_fma_f32 alone (8 chains)", 64, "v_fma", d_out, d_sink);
  run<11>("v_pk_fma_f32 alone (4 chains)", 32, "v_pk_fma", d_out, d_sink);
  run<10>("v_exp_f32 alone (2 chains)", 16, "v_exp", d_out, d_sink);
  run<12>("ds_read_b128 alone", 8, "ds_read", d_out, d_sink);
  return 0;
}
