// The K-loop skeleton of the 8-wave 256 x 256 tile in isolation: what does a K step (per wave 32 MFMAs 32x32x16 bf16, 24
// ds_read_b128 fragment reads from 128-B swizzled tile rows, one barrier) cost when NOTHING else is in it -- no DMA, no
// address arithmetic, no memory traffic?  (scripts/conv_profile.py with ws_prof_mode = 12 says 3 040 cycles against 2 048 of
// matrix work, every schedule.)  One workgroup on one CU, s_memtime on every wave, cycles per K step.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/build/mfma_tile_bench scripts/mfma_tile_bench.hip
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
#define MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

// BAR: barrier per K step; READS: 0 none, 1 real addressing, 2 one address for all; FD: read-ahead in MFMAs (stream form) or
// 0 = the batch form of schedule 1 (the six reads of sub-step k+1 in front of the eight MFMAs of sub-step k); PRIO: 0 none,
// 1 alternate s_setprio between the waves of a SIMD per sub-step (schedule 1), 2 setprio 1 always
// DMA: 0 none; 1 the stage refill of the real kernel: 8 LDS-DMA pieces of 1 KiB per wave and step (64 KiB per workgroup) from a
// per-workgroup region of `big`, issued behind MFMAs 1, 3, ... 15, awaited (vmcnt(0)) in front of the barrier; 2 the same
// requests against an extent of 0 (zero fills: LDS writes, no memory traffic)
template <bool BAR, int READS, int FD, int PRIO, int DMA = 0>
__global__ __launch_bounds__(512, 1) void kloop(unsigned long long* out, float* sink, int steps, const char* big) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave & 1, wn = wave >> 1, wgrp = wave >> 2;
  for (int i = threadIdx.x; i < 131072 / 16; i += blockDim.x) ((u32x4*)lds)[i] = u32x4{0x3f803f80u, 0x3e803f80u + (unsigned)i, 0x3f803f80u, 0x3f803e80u};
  __syncthreads();
  f32x16 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
  const int row = lane & 31, khalf = lane >> 5, swz = (row >> 1) & 7;
  const char* a_base = lds + (wm * 64) * 128 + row * 128;               // A tile: 256 rows x 128 B, wave rows [64 wm, +64)
  const char* b_base = lds + 32768 + (wn * 128) * 128 + row * 128;      // B tile: 256 rows, wave rows [128 wn... (4 waves along N)
  auto frag = [&](int stg, int k, int i) -> const u32x4* {              // i < 2: A fragment i, else B fragment i - 2
    const int slot = READS == 2 ? 0 : ((k * 2 + khalf) ^ swz) * 16;
    const char* base = (i < 2 ? a_base + i * 32 * 128 : b_base + (i - 2) * 32 * 128) + stg * 65536;
    return reinterpret_cast<const u32x4*>(READS == 2 ? lds + lane * 16 : base + slot);
  };
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(big) + (size_t)blockIdx.x * (8u << 20), 0, DMA == 2 ? 0u : (8u << 20), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(big) + ((size_t)1000 << 23), 0, 8u << 20, 0x00020000);
  auto piece = [&](int s, int stg, int q) {      // piece q of step s: wave w fills rows [32 q + 4 w ...) of the 64-KiB stage
    if constexpr (DMA != 0) {
      const unsigned win = DMA == 3 ? (64u << 10) : DMA == 4 ? (1u << 20) : (8u << 20);   // 3: 64 KiB per workgroup (16 MiB in all: L2 hits), 4: 1 MiB (256 MiB: the MALL)
      unsigned off = (unsigned)((((s * 8 + q) * 8 + wave) * 1024 + lane * 16) & (win - 1));
      if (DMA >= 5 && q >= 4) {
        // the layer's situation: pieces 0-3 = the workgroup's own pixel rows (unique, from memory), pieces 4-7 = the weight
        // slab (144 steps x 32 KiB = 4.5 MiB) that EVERY workgroup walks -- 5: all at the same step, 6: each at its own phase
        const unsigned ph = DMA == 6 ? (blockIdx.x * 37u) % 144u : 0u;
        off = ((s + ph) % 144u) * 32768u + ((q - 4) * 8 + wave) * 1024u + lane * 16;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(lds + stg * 65536 + (q * 8 + wave) * 1024), 16, off, 0, 0, 0);
        return;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(lds + stg * 65536 + (q * 8 + wave) * 1024), 16, off, 0, 0, 0);
    }
  };
  u32x4 fr[2][6];
#pragma unroll
  for (int i = 0; i < 6; ++i) fr[0][i] = fr[1][i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  __syncthreads();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int stage = 0;
  if constexpr (FD == 0) {
    if (READS) {
#pragma unroll
      for (int i = 0; i < 6; ++i) fr[0][i] = *frag(0, 0, i);
    }
    for (int s = 0; s < steps; ++s) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (PRIO == 1) {
          if (((k ^ wgrp) & 1) == 0) __builtin_amdgcn_s_setprio(2);
          else __builtin_amdgcn_s_setprio(0);
        }
        if (READS && k + 1 < 4) {
#pragma unroll
          for (int i = 0; i < 6; ++i) fr[(k + 1) & 1][i] = *frag(stage, k + 1, i);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          if (k == 3 && q == 4) {
            if (BAR) {
              if (DMA != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
              __builtin_amdgcn_s_barrier();
              asm volatile("" ::: "memory");
            }
            if (READS) {
#pragma unroll
              for (int i = 0; i < 6; ++i) fr[0][i] = *frag(stage ^ 1, 0, i);
            }
          }
          MFMA(acc[q >> 1][q & 1], fr[k & 1][2 + (q >> 1)], fr[k & 1][q & 1]);
          if (DMA != 0 && k < 2 && (q & 1)) piece(s, stage ^ 1, k * 4 + (q >> 1));
        }
      }
      stage ^= 1;
    }
  } else {
    // stream form: MFMA m of the step uses fragments of sub-step m / 8; the read issued behind MFMA m is the one needed
    // soonest: reads in use order, FD MFMAs ahead (a sub-step's six fragments are all needed by its first MFMAs, so
    // "ahead" is counted to the first MFMA of that sub-step)
    for (int s = 0; s < steps; ++s) {
      if (PRIO == 2) __builtin_amdgcn_s_setprio(1);
      sfor<0, 32>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        constexpr int k = m >> 3, q = m & 7;
        if constexpr (k == 3 && q == 4 && BAR) {
          if (DMA != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
        }
        MFMA(acc[q >> 1][q & 1], fr[k & 1][2 + (q >> 1)], fr[k & 1][q & 1]);
        // behind MFMAs 1..6 of sub-step k: the six fragments of sub-step k + 1 (other register set)
        if constexpr (READS != 0 && q >= 1 && q <= 6) {
          constexpr int kn = (k + 1) & 3;
          fr[(k + 1) & 1][q - 1] = *frag(k == 3 ? stage ^ 1 : stage, kn, q - 1);
        }
        if constexpr (DMA != 0 && k < 2 && (q & 1)) piece(s, stage ^ 1, k * 4 + (q >> 1));
        __builtin_amdgcn_sched_barrier(0);
      });
      stage ^= 1;
    }
    if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float sum = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a) sum += acc[a][0][0] + acc[a][1][3];
  if (lane == 0 && blockIdx.x == 0) {
    out[wave] = t1 - t0;
    out[8 + wave] = r1 - r0;
  }
  if (sum == 12345.678f) sink[threadIdx.x] = sum;
}

char* d_big;
// The same K loop on a 4-slot ring of HALF stages (64-B rows: 32 KiB per stage, 16 MFMAs + 12 fragment reads + 4 pieces per wave
// and stage, one barrier per stage), the pieces of stage s + DEPTH requested during stage s: is the 2-slot loop waiting for the
// LATENCY of its pieces (one step of cover) rather than for their bytes?  DMA modes as above (1 memory, 3 L2, 5 / 6 slab).
template <int DEPTH, int DMA>
__global__ __launch_bounds__(512, 1) void kloop4(unsigned long long* out, float* sink, int steps, const char* big) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  for (int i = threadIdx.x; i < 131072 / 16; i += blockDim.x) ((u32x4*)lds)[i] = u32x4{0x3f803f80u, 0x3e803f80u + (unsigned)i, 0x3f803f80u, 0x3f803e80u};
  __syncthreads();
  f32x16 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
  const int row = lane & 31, khalf = lane >> 5, swz = (row >> 2) & 3;
  const char* a_base = lds + (wm * 64) * 64 + row * 64;                 // A: 256 rows x 64 B
  const char* b_base = lds + 16384 + (wn * 128) * 64 + row * 64;        // B: 256 rows x 64 B
  auto frag = [&](int slot, int k, int i) -> const u32x4* {
    const char* base = (i < 2 ? a_base + i * 32 * 64 : b_base + (i - 2) * 32 * 64) + slot * 32768;
    return reinterpret_cast<const u32x4*>(base + ((k * 2 + khalf) ^ swz) * 16);
  };
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(big) + (size_t)blockIdx.x * (8u << 20), 0, 8u << 20, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(big) + ((size_t)1000 << 23), 0, 8u << 20, 0x00020000);
  auto piece = [&](int s, int q) {              // piece q < 4 of half-stage s: q < 2 pixel rows, q >= 2 weights
    const int slot = s & 3;
    unsigned off = (unsigned)((((s * 4 + q) * 8 + wave) * 1024 + lane * 16) & (DMA == 3 ? (64u << 10) - 1 : (8u << 20) - 1));
    if (DMA >= 5 && q >= 2) {
      const unsigned ph = DMA == 6 ? (blockIdx.x * 74u) % 288u : 0u;
      off = ((s + ph) % 288u) * 16384u + ((q - 2) * 8 + wave) * 1024u + lane * 16;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(lds + slot * 32768 + (q * 8 + wave) * 1024), 16, off, 0, 0, 0);
      return;
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(lds + slot * 32768 + (q * 8 + wave) * 1024), 16, off, 0, 0, 0);
  };
  u32x4 fr[2][6];
#pragma unroll
  for (int i = 0; i < 6; ++i) fr[0][i] = fr[1][i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  for (int s = 0; s < DEPTH; ++s)
#pragma unroll
    for (int q = 0; q < 4; ++q) piece(s, q);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < 2 * steps; ++s) {          // half stages
    const int slot = s & 3;
    sfor<0, 16>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      constexpr int k = m >> 3, q = m & 7;
      if constexpr (k == 1 && q == 4) {
        // pieces of stage s + 1 (requested DEPTH stages ago) have landed when at most the younger (DEPTH - 1) x 4 are out
        if constexpr (DEPTH == 3) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else if constexpr (DEPTH == 2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      MFMA(acc[q >> 1][q & 1], fr[k & 1][2 + (q >> 1)], fr[k & 1][q & 1]);
      if constexpr (q >= 1 && q <= 6) fr[(k + 1) & 1][q - 1] = *frag(k == 1 ? (slot + 1) & 3 : slot, (k + 1) & 1, q - 1);
      // the slot read DEPTH ... stages ago is free again behind the barrier of the previous stage: refill it for stage s + DEPTH
      if constexpr (k == 0 && (q & 1)) piece(s + DEPTH, q >> 1);
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float sum = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a) sum += acc[a][0][0] + acc[a][1][3];
  if (lane == 0 && blockIdx.x == 0) {
    out[wave] = t1 - t0;
    out[8 + wave] = r1 - r0;
  }
  if (sum == 12345.678f) sink[threadIdx.x] = sum;
}
template <int DEPTH, int DMA>
void run4(const char* what, unsigned long long* d_out, float* d_sink, char* big, int grid) {
  const int steps = 200;
  auto k = kloop4<DEPTH, DMA>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  unsigned long long t[16] = {0};
  for (int i = 0; i < 2; ++i) {
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), 131072, 0, d_out, d_sink, steps, big);
    hipError_t e = hipMemcpy(t, d_out, sizeof(t), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { printf("%s: %s\n", what, hipGetErrorString(e)); return; }
  }
  double lo = 1e30, hi = 0;
  for (int w = 0; w < 8; ++w) { lo = t[w] < lo ? t[w] : lo; hi = t[w] > hi ? t[w] : hi; }
  printf("%-100s %7.0f .. %7.0f s_memtime ticks per K step of 64 (matrix work: 2048); %.1f ns\n", what, lo / steps, hi / steps, (double)t[8] * 10.0 / steps);
  fflush(stdout);
}

template <bool BAR, int READS, int FD, int PRIO, int DMA = 0>
void run(const char* what, unsigned long long* d_out, float* d_sink, int waves = 8, int grid = 1) {
  const int steps = 200;
  auto k = kloop<BAR, READS, FD, PRIO, DMA>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  unsigned long long t[16] = {0};
  for (int i = 0; i < 2; ++i) {
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * waves), 131072, 0, d_out, d_sink, steps, d_big);
    hipError_t e = hipMemcpy(t, d_out, sizeof(t), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { printf("%s: %s\n", what, hipGetErrorString(e)); return; }
  }
  double lo = 1e30, hi = 0;
  for (int w = 0; w < waves; ++w) { lo = t[w] < lo ? t[w] : lo; hi = t[w] > hi ? t[w] : hi; }
  printf("%-100s %7.0f .. %7.0f s_memtime ticks per K step (matrix work: %d); %.1f ns per step by s_memrealtime (100 MHz) -> %.2f ticks per ns\n", what, lo / steps, hi / steps,
         waves == 8 ? 2048 : 1024, (double)t[8] * 10.0 / steps, (double)t[0] / ((double)t[8] * 10.0));
  fflush(stdout);
}

int main() {
  unsigned long long* d_out; float* d_sink;
  hipMalloc(&d_out, 256); hipMalloc(&d_sink, 4096); hipMalloc(&d_big, (size_t)1024 << 23); hipMemset(d_big, 0x3f, (size_t)1024 << 23);
  run<false, 0, 0, 0>("MFMAs only, 8 waves", d_out, d_sink);
  run<false, 0, 0, 0>("MFMAs only, 4 waves (one per SIMD)", d_out, d_sink, 4);
  run<true, 0, 0, 0>("MFMAs + barrier per step", d_out, d_sink);
  run<false, 1, 0, 0>("batch reads (schedule 1 form), no barrier", d_out, d_sink);
  run<true, 1, 0, 0>("batch reads + barrier", d_out, d_sink);
  run<true, 1, 0, 1>("batch reads + barrier + alternating priority (= schedule 1's skeleton)", d_out, d_sink);
  run<true, 2, 0, 1>("... all reads from one conflict-free address pattern", d_out, d_sink);
  run<true, 1, 0, 1>("... 4 waves (one per SIMD)", d_out, d_sink, 4);
  run<false, 1, 6, 0>("stream form: one read behind each of MFMAs 1-6 of a sub-step, no barrier", d_out, d_sink);
  run<true, 1, 6, 0>("stream form + barrier", d_out, d_sink);
  run<true, 1, 6, 2>("stream form + barrier + setprio 1", d_out, d_sink);
  run<true, 2, 6, 0>("stream form + barrier, one address pattern", d_out, d_sink);
  run<true, 1, 6, 0>("stream form + barrier, 4 waves", d_out, d_sink, 4);
  printf("the same on every CU (grid 256) and oversubscribed (grid 1024):\n");
  run<false, 0, 0, 0>("MFMAs only, 8 waves", d_out, d_sink, 8, 256);
  run<true, 1, 0, 1>("batch reads + barrier + alternating priority (= schedule 1's skeleton)", d_out, d_sink, 8, 256);
  run<true, 1, 6, 0>("stream form + barrier", d_out, d_sink, 8, 256);
  run<true, 1, 0, 1>("batch reads + barrier + alternating priority, grid 1024", d_out, d_sink, 8, 1024);
  printf("with the stage refill (8 LDS-DMA pieces per wave and step), grid 256:\n");
  run<true, 1, 0, 1, 1>("batch form + barrier + pieces from memory", d_out, d_sink, 8, 256);
  run<true, 1, 0, 1, 2>("batch form + barrier + pieces as zero fills", d_out, d_sink, 8, 256);
  run<true, 1, 6, 0, 1>("stream form + barrier + pieces from memory", d_out, d_sink, 8, 256);
  run<true, 1, 6, 0, 2>("stream form + barrier + pieces as zero fills", d_out, d_sink, 8, 256);
  run<true, 1, 6, 0, 1>("stream form + barrier + pieces from memory, one workgroup", d_out, d_sink, 8, 1);
  run<true, 0, 6, 0, 1>("MFMAs + barrier + pieces from memory (no fragment reads)", d_out, d_sink, 8, 256);
  run<true, 1, 6, 0, 3>("stream form + barrier + pieces that hit in the L2 (64 KiB window per workgroup)", d_out, d_sink, 8, 256);
  run<true, 1, 0, 1, 3>("batch form + barrier + pieces that hit in the L2", d_out, d_sink, 8, 256);
  run<true, 1, 6, 0, 4>("stream form + barrier + pieces from a 1 MiB window per workgroup (256 MiB in all)", d_out, d_sink, 8, 256);
  run<true, 1, 6, 0, 5>("stream form + barrier + own pixel rows from memory, shared weight slab walked in step", d_out, d_sink, 8, 256);
  run<true, 1, 6, 0, 6>("stream form + barrier + own pixel rows from memory, shared weight slab at 256 different phases", d_out, d_sink, 8, 256);
  printf("4-slot ring of half stages (64-B rows), grid 256:\n");
  run4<1, 3>("depth 1, pieces hit in the L2", d_out, d_sink, d_big, 256);
  run4<3, 3>("depth 3, pieces hit in the L2", d_out, d_sink, d_big, 256);
  run4<1, 1>("depth 1, pieces from memory", d_out, d_sink, d_big, 256);
  run4<2, 1>("depth 2, pieces from memory", d_out, d_sink, d_big, 256);
  run4<3, 1>("depth 3, pieces from memory", d_out, d_sink, d_big, 256);
  run4<1, 5>("depth 1, own pixel rows from memory + shared weight slab in step", d_out, d_sink, d_big, 256);
  run4<3, 5>("depth 3, own pixel rows from memory + shared weight slab in step", d_out, d_sink, d_big, 256);
  run4<3, 6>("depth 3, own pixel rows from memory + shared weight slab at different phases", d_out, d_sink, d_big, 256);
  return 0;
}
