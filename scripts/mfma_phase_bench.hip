// Why does an MFMA phase of conv3x3_ws2_kernel -- 72 x (v_mfma_f32_32x32x16_bf16 + the ds_read_b128 of a fragment used six
// MFMAs later), nothing else -- take 51 cycles per MFMA (profiles/r03_ws2_iteration_cycles.txt) when the issue
// micro-benchmark prices "MFMA + ds_read" at 38.5?  The same stream in isolation, one variable at a time; one workgroup,
// waves 0-3 run it (one per SIMD), s_memtime on wave 0, cycles per MFMA.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/build/mfma_phase_bench scripts/mfma_phase_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int I, int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, N>(f);
  }
}

// AGPR_A: weights in the accumulator half ("a") or the architectural half; NACC accumulators; FD fragment distance;
// REAL: patch addressing (272-B pixel rows, the kernel's lane -> pixel map) or one address for all reads;
// PRIO: s_setprio 1 around the phase; OTHER: what waves 4-7 do: 0 absent (256 threads), 1 parked at the barrier, 2 a
// plain-fp32 VALU loop (a row slot's arithmetic), 3 ds_read_b128 stream, 4 LDS-DMA requests (global -> LDS, 1 KiB per wave
// instruction), 5 global stores of 16 B per lane, 6 ds_write_b64 stream (the accumulator transposition)
template <bool AGPR_A, int NACC, int FD, bool REAL, bool PRIO, int OTHER>
__global__ __launch_bounds__(512, 1) void phase(unsigned long long* out, float* sink, int reps, char* big) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32x4 wreg[36];
#pragma unroll
  for (int i = 0; i < 36; ++i) wreg[i] = u32x4{0x3f803f80u + lane + i, 0x3f803f80u, 0x3f003f80u, 0x3f803f00u};
  for (int i = threadIdx.x; i < 49152 / 16; i += blockDim.x) ((u32x4*)lds)[i] = u32x4{0x3f803f80u, 0x3e803f80u + i, 0x3f803f80u, 0x3f803e80u};
  __syncthreads();
  f32x16 acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
  float f[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
  u32x4 dd = {0, 0, 0, 0};
  unsigned long long t0 = 0, t1 = 0;
  if (wave < 4) {
    const int m = lane & 31;
    const bool g0 = (m < 4) | ((m >= 12) & (m < 16)) | ((m >= 20) & (m < 28));
    const int rsel = g0 ? 0 : 1;
    const int col = g0 ? (m < 4 ? m : (m < 16 ? m - 8 : m - 12)) : (m < 12 ? m - 4 : (m < 20 ? m - 8 : m - 16));
    const char* pb = lds + (REAL ? (rsel * 18 + col) * 272 + (lane >> 5) * 16 : lane * 16);
    auto addr = [&](int mm) -> const u32x4* {
      if (!REAL) return reinterpret_cast<const u32x4*>(pb);
      const int gg = mm >> 1, jj = mm & 1;
      const int tap = gg >> 3, c = gg & 7, kh = tap / 3, kw = tap - 3 * kh;
      return reinterpret_cast<const u32x4*>(pb + ((2 * jj + kh) * 18 + kw) * 272 + c * 32);
    };
    t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
      u32x4 xf[FD + 1];
#pragma unroll
      for (int mm = 0; mm < FD; ++mm) xf[mm] = *addr(mm);
      if (PRIO) __builtin_amdgcn_s_setprio(1);
      sfor<0, 72>([&](auto mc) {
        constexpr int mm = decltype(mc)::value;
        if constexpr (AGPR_A) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[mm % NACC]) : "a"(wreg[(mm >> 1) % 32]), "v"(xf[mm % (FD + 1)]));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[mm % NACC]) : "v"(wreg[(mm >> 1) % 8]), "v"(xf[mm % (FD + 1)]));
        if constexpr (mm + FD < 72) xf[(mm + FD) % (FD + 1)] = *addr(mm + FD);
        __builtin_amdgcn_sched_barrier(0);
      });
      if (PRIO) __builtin_amdgcn_s_setprio(0);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    t1 = __builtin_amdgcn_s_memtime();
  } else if (OTHER == 2) {
    for (int r = 0; r < reps * 40; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(0.999f));
    }
  } else if (OTHER == 3) {
    const unsigned la = (unsigned)(size_t)lds + lane * 16;
    for (int r = 0; r < reps * 20; ++r) {
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(dd) : "v"(la));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  else if (OTHER == 4) {
    char* src = big + (size_t)blockIdx.x * (4u << 20);
    for (int r = 0; r < reps * 2; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const unsigned off = (unsigned)(((r * 8 + i) * 4 + (wave - 4)) & 4095) * 1024u + lane * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off),
                                         (__attribute__((address_space(3))) void*)(lds + 49152 + (wave - 4) * 8192 + i * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  } else if (OTHER == 5) {
    char* dst = big + (size_t)blockIdx.x * (4u << 20);
    for (int r = 0; r < reps * 2; ++r) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned off = (unsigned)(((r * 4 + i) * 4 + (wave - 4)) & 4095) * 1024u + lane * 16;
        *reinterpret_cast<u32x4*>(dst + off) = u32x4{(unsigned)r, 1u, 2u, 3u};
      }
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
  } else if (OTHER == 6) {
    const unsigned la = (unsigned)(size_t)lds + 49152 + (wave - 4) * 8192 + lane * 8;
    for (int r = 0; r < reps * 60; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(la), "v"(t0), "n"(i * 512));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  float s = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + f[0] + f[7] + __uint_as_float(dd[0]);
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  if (s == 12345.678f) sink[threadIdx.x] = s;
}

char* d_big;
template <bool AGPR_A, int NACC, int FD, bool REAL, bool PRIO, int OTHER>
void run(const char* what, unsigned long long* d_out, float* d_sink, int grid = 1) {
  const int reps = 400;
  auto k = phase<AGPR_A, NACC, FD, REAL, PRIO, OTHER>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  unsigned long long t = 0;
  for (int i = 0; i < 2; ++i) {
    hipLaunchKernelGGL(k, dim3(grid), dim3(OTHER == 0 ? 256 : 512), 98304, 0, d_out, d_sink, reps, d_big);
    hipError_t e = hipMemcpy(&t, d_out, sizeof(t), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { printf("%s: %s\n", what, hipGetErrorString(e)); fflush(stdout); return; }
  }
  printf("%-100s %7.1f cycles per MFMA\n", what, (double)t / reps / 72);
  fflush(stdout);
}

int main() {
  unsigned long long* d_out; float* d_sink;
  hipMalloc(&d_out, 64); hipMalloc(&d_sink, 4096); hipMalloc(&d_big, (size_t)1024 << 22);
  run<true, 2, 6, true, true, 1>("as in the kernel: A in AGPRs, 2 accumulators, distance 6, patch addressing, prio 1, partner parked", d_out, d_sink);
  run<true, 2, 6, true, true, 0>("... partner absent (one wave per SIMD)", d_out, d_sink);
  run<true, 2, 6, true, false, 1>("... no s_setprio", d_out, d_sink);
  run<false, 2, 6, true, true, 1>("... A in the architectural half", d_out, d_sink);
  run<true, 4, 6, true, true, 1>("... 4 accumulators", d_out, d_sink);
  run<true, 2, 3, true, true, 1>("... fragment distance 3", d_out, d_sink);
  run<true, 2, 12, true, true, 1>("... fragment distance 12", d_out, d_sink);
  run<true, 2, 6, false, true, 1>("... every read from one address pattern (lane * 16)", d_out, d_sink);
  run<true, 2, 6, true, true, 2>("... partner in a plain-fp32 VALU loop", d_out, d_sink);
  run<true, 2, 6, true, true, 3>("... partner streaming ds_read_b128", d_out, d_sink);
  run<true, 2, 6, true, true, 4>("... partner requesting LDS-DMA pieces (global -> LDS)", d_out, d_sink);
  run<true, 2, 6, true, true, 5>("... partner storing 16 B per lane to global memory", d_out, d_sink);
  run<true, 2, 6, true, true, 6>("... partner streaming ds_write_b64", d_out, d_sink);
  printf("256 workgroups (every CU busy):\n");
  run<true, 2, 6, true, true, 1>("as in the kernel, partner parked", d_out, d_sink, 256);
  run<true, 2, 6, true, true, 4>("... partner requesting LDS-DMA pieces", d_out, d_sink, 256);
  run<true, 2, 6, true, true, 5>("... partner storing to global memory", d_out, d_sink, 256);
  run<true, 2, 6, true, true, 6>("... partner streaming ds_write_b64", d_out, d_sink, 256);
  printf("1024 workgroups:\n");
  run<true, 2, 6, true, true, 1>("as in the kernel, partner parked", d_out, d_sink, 1024);
  run<true, 2, 6, true, true, 4>("... partner requesting LDS-DMA pieces", d_out, d_sink, 1024);
  return 0;
}
