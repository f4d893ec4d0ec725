// What does the memory system sustain for the access shapes of this library's kernels?  Read-only, write-only and
// read+write streams of 16 B per lane, grid = 256 CUs x k workgroups, 1 GiB per stream (far beyond the caches).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/build/hbm_bw_bench scripts/hbm_bw_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NT>   // 0 read, 1 write, 2 copy (1 read + 1 write), 3 one read + two writes (the fused blocks)
__global__ __launch_bounds__(256) void stream(const u32x4* __restrict__ a, u32x4* __restrict__ b, u32x4* __restrict__ c, size_t n, unsigned* sink) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  u32x4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    u32x4 v = {(unsigned)i, 1u, 2u, 3u};
    if (MODE != 1) v = NT ? __builtin_nontemporal_load(a + i) : a[i];
    if (MODE == 0) { acc[0] ^= v[0]; acc[1] += v[1]; acc[2] ^= v[2]; acc[3] += v[3]; }
    if (MODE >= 1) { if (NT) __builtin_nontemporal_store(v, b + i); else b[i] = v; }
    if (MODE == 3) { v[0] += 1; if (NT) __builtin_nontemporal_store(v, c + i); else c[i] = v; }
  }
  if (MODE == 0 && acc[0] + acc[1] + acc[2] + acc[3] == 0x12345u) sink[0] = 1;
}

template <int MODE, int NT>
void run(const char* what, u32x4* a, u32x4* b, u32x4* c, size_t n, unsigned* sink, int wgs) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < 4; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream<MODE, NT>), dim3(wgs), dim3(256), 0, 0, a, b, c, n, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (r > 0 && ms < best) best = ms;
  }
  const double bytes = (double)n * 16 * (MODE == 0 ? 1 : MODE == 1 ? 1 : MODE == 2 ? 2 : 3);
  printf("%-46s grid %5d: %7.3f ms  %7.1f GB/s total", what, wgs, best, bytes / best / 1e6);
  if (MODE == 2) printf("  (%.1f read + %.1f written)", bytes / 2 / best / 1e6, bytes / 2 / best / 1e6);
  if (MODE == 3) printf("  (%.1f read + %.1f written)", bytes / 3 / best / 1e6, bytes * 2 / 3 / best / 1e6);
  printf("\n");
  fflush(stdout);
}

int main() {
  const size_t n = (size_t)1 << 26;   // x 16 B = 1 GiB per stream
  u32x4 *a, *b, *c; unsigned* sink;
  hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&c, n * 16); hipMalloc(&sink, 64);
  hipMemset(a, 1, n * 16); hipMemset(b, 2, n * 16); hipMemset(c, 3, n * 16);
  for (int wgs : {256, 1024, 4096, 16384}) {
    run<0, 0>("read", a, b, c, n, sink, wgs);
    run<1, 0>("write", a, b, c, n, sink, wgs);
    run<2, 0>("copy", a, b, c, n, sink, wgs);
    run<3, 0>("one read, two writes", a, b, c, n, sink, wgs);
    run<1, 1>("write, non-temporal", a, b, c, n, sink, wgs);
    run<3, 1>("one read, two writes, non-temporal", a, b, c, n, sink, wgs);
  }
  return 0;
}
