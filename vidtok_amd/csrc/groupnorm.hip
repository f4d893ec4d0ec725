// GroupNorm(32 groups, eps, affine)(+SiLU) on NDHWC activations -- the `norm_type: groupnorm` branch of the reference's
// Normalize() (vidtok/modules/model_3dcausal.py:30-34).  No shipped config selects it (all 23 YAMLs use layernorm),
// so this is coverage, built for correctness first: HBM-bound, three passes.
//
// torch.nn.GroupNorm normalises over (C/G, *spatial) of whatever view the call site hands it, so the reduction
// domain depends on the site (`scope`):
//   VT_GN_FRAME  spatial ResnetBlock on "(b t) c h w"             : per (b, t, g) over (C/G, H, W)
//   VT_GN_PIXEL  temporal blocks on "(b h w) c t"                 : per (b, h, w, g) over (C/G, T)
//   VT_GN_CLIP   3-D blocks, attention norm, norm_out on b c t h w : per (b, g) over (C/G, T, H, W)
// Statistics: sum and sum of squares accumulated in fp64 (per-thread fp32 partials over <= 64 elements, fp64 block
// reduction, one fp64 atomic per workgroup and group) -- domains reach 5 M elements, an fp32 E[x^2]-E[x]^2 would not
// hold 1e-3.  PIXEL domains are tiny (T x C/G) and are reduced inside the apply kernel.
#include "common.h"

namespace {

constexpr int kBlock = 256;

template <typename T>
__device__ __forceinline__ float ldf(const T* p) { return to_f32<T>(*p); }

// instance id and pixel -> see vt_groupnorm_act: pixels of an instance are `P` consecutive pixel rows
// (FRAME: P = HW, CLIP: P = T*HW).  grid = (chunks, instances); thread = one (pixel, group) run of cg channels.
template <typename T>
__global__ __launch_bounds__(kBlock) void gn_stats_kernel(const T* __restrict__ x, long long ldx, double* __restrict__ st,
                                                          long long P, int G, int cg) {
  __shared__ double red[2][kBlock / 64][32];
  const long long inst = blockIdx.y;
  const int g = threadIdx.x % G;            // G = 32: a wave holds two pixels x 32 groups
  const int sub = threadIdx.x / G;          // pixel slot inside the block (kBlock / G of them)
  const int slots = kBlock / G;
  const T* xb = x + inst * P * ldx + g * cg;
  float s = 0.f, q = 0.f;
  double ds = 0.0, dq = 0.0;
  int run = 0;
  for (long long p = (long long)blockIdx.x * slots + sub; p < P; p += (long long)gridDim.x * slots) {
    const T* xp = xb + p * ldx;
    for (int c = 0; c < cg; ++c) {
      const float v = ldf<T>(xp + c);
      s += v;
      q += v * v;
    }
    if (++run == 8) {                        // flush the fp32 partials (<= 8 * cg <= 128 elements)
      ds += s; dq += q; s = q = 0.f; run = 0;
    }
  }
  ds += s; dq += q;
  // lanes l and l+32 of a wave hold the same group: combine, then the waves through LDS
  ds += __shfl_xor(ds, 32, 64);
  dq += __shfl_xor(dq, 32, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane < 32) { red[0][wave][lane] = ds; red[1][wave][lane] = dq; }
  __syncthreads();
  if (threadIdx.x < 32) {
    double a = 0.0, b = 0.0;
    for (int w = 0; w < kBlock / 64; ++w) { a += red[0][w][threadIdx.x]; b += red[1][w][threadIdx.x]; }
    atomicAdd(st + (inst * G + threadIdx.x) * 2, a);
    atomicAdd(st + (inst * G + threadIdx.x) * 2 + 1, b);
  }
}

// FRAME / CLIP apply: thread = 4 consecutive channels of one pixel
template <typename TI, typename TO, bool SILU>
__global__ __launch_bounds__(kBlock) void gn_apply_kernel(const TI* __restrict__ x, long long ldx, TO* __restrict__ y,
                                                          long long ldy, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const double* __restrict__ st,
                                                          long long M, long long P, int C, int G, int cg, float eps) {
  const int q4 = C / 4;
  const long long n = M * q4;
  const double inv_n = 1.0 / ((double)P * cg);
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
    const long long m = i / q4;
    const int c0 = (int)(i - m * q4) * 4;
    const long long inst = m / P;
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = c0 + e;
      const double* sg = st + (inst * G + c / cg) * 2;
      const double mean = sg[0] * inv_n;
      const double var = fmax(sg[1] * inv_n - mean * mean, 0.0);
      const float rstd = (float)(1.0 / sqrt(var + (double)eps));
      const float u = (ldf<TI>(x + m * ldx + c) - (float)mean) * rstd * gamma[c] + beta[c];
      o[e] = SILU ? silu_f32(u) : u;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) y[m * ldy + c0 + e] = from_f32<TO>(o[e]);
  }
}

// PIXEL: thread = one (b, hw, group): statistics over (T, cg) and the apply in one go (two passes + apply)
template <typename TI, typename TO, bool SILU>
__global__ __launch_bounds__(kBlock) void gn_pixel_kernel(const TI* __restrict__ x, long long ldx, TO* __restrict__ y,
                                                          long long ldy, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int B, int T, long long HW, int G,
                                                          int cg, float eps) {
  const long long n = (long long)B * HW * G;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
    const int g = (int)(i % G);
    const long long r = i / G;
    const long long hw = r % HW, b = r / HW;
    const long long pix0 = b * T * HW + hw;
    const float cnt = (float)(T * cg);
    float s = 0.f;
    for (int t = 0; t < T; ++t)
      for (int c = 0; c < cg; ++c) s += ldf<TI>(x + (pix0 + t * HW) * ldx + g * cg + c);
    const float mean = s / cnt;
    float q = 0.f;
    for (int t = 0; t < T; ++t)
      for (int c = 0; c < cg; ++c) {
        const float d = ldf<TI>(x + (pix0 + t * HW) * ldx + g * cg + c) - mean;
        q += d * d;
      }
    const float rstd = 1.0f / sqrtf(q / cnt + eps);
    for (int t = 0; t < T; ++t)
      for (int c = 0; c < cg; ++c) {
        const int ch = g * cg + c;
        const float u = (ldf<TI>(x + (pix0 + t * HW) * ldx + ch) - mean) * rstd * gamma[ch] + beta[ch];
        y[(pix0 + t * HW) * ldy + ch] = from_f32<TO>(SILU ? silu_f32(u) : u);
      }
  }
}

inline unsigned grid_for(long long n) {
  long long b = (n + kBlock - 1) / kBlock;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

template <typename TI, typename TO>
int launch_groupnorm(const void* x, long long ldx, void* y, long long ldy, const float* gamma, const float* beta, int B,
                     int T, long long HW, int C, int G, int scope, float eps, int silu, double* work, hipStream_t stream) {
  const int cg = C / G;
  if (scope == VT_GN_PIXEL) {
    const long long n = (long long)B * HW * G;
    if (silu)
      hipLaunchKernelGGL((gn_pixel_kernel<TI, TO, true>), dim3(grid_for(n)), dim3(kBlock), 0, stream, (const TI*)x, ldx, (TO*)y,
                         ldy, gamma, beta, B, T, HW, G, cg, eps);
    else
      hipLaunchKernelGGL((gn_pixel_kernel<TI, TO, false>), dim3(grid_for(n)), dim3(kBlock), 0, stream, (const TI*)x, ldx, (TO*)y,
                         ldy, gamma, beta, B, T, HW, G, cg, eps);
    VT_CHECK_LAUNCH();
    return VT_OK;
  }
  const long long ninst = scope == VT_GN_FRAME ? (long long)B * T : B;
  const long long P = scope == VT_GN_FRAME ? HW : (long long)T * HW;
  VT_CHECK_ARG(work != nullptr, "vt_groupnorm_act: frame / clip scope needs the work buffer");
  VT_CHECK_ARG(ninst <= 65535, "vt_groupnorm_act: too many instances (%lld)", ninst);
  VT_CHECK_HIP(hipMemsetAsync(work, 0, (size_t)ninst * G * 2 * sizeof(double), stream));
  const int slots = kBlock / G;
  long long chunks = (P + slots * 8 - 1) / (slots * 8);      // ~8 pixels per thread and flush
  if (chunks > 1024) chunks = 1024;
  if (chunks < 1) chunks = 1;
  hipLaunchKernelGGL(gn_stats_kernel<TI>, dim3((unsigned)chunks, (unsigned)ninst), dim3(kBlock), 0, stream, (const TI*)x, ldx,
                     work, P, G, cg);
  VT_CHECK_LAUNCH();
  const long long M = ninst * P;
  if (silu)
    hipLaunchKernelGGL((gn_apply_kernel<TI, TO, true>), dim3(grid_for(M * (C / 4))), dim3(kBlock), 0, stream, (const TI*)x, ldx,
                       (TO*)y, ldy, gamma, beta, (const double*)work, M, P, C, G, cg, eps);
  else
    hipLaunchKernelGGL((gn_apply_kernel<TI, TO, false>), dim3(grid_for(M * (C / 4))), dim3(kBlock), 0, stream, (const TI*)x, ldx,
                       (TO*)y, ldy, gamma, beta, (const double*)work, M, P, C, G, cg, eps);
  VT_CHECK_LAUNCH();
  return VT_OK;
}

}  // namespace

extern "C" int64_t vt_groupnorm_work_bytes(int32_t B, int32_t T, int32_t groups, int32_t scope) {
  if (scope == VT_GN_PIXEL) return 0;
  const int64_t ninst = scope == VT_GN_FRAME ? (int64_t)B * T : B;
  return ninst * groups * 2 * (int64_t)sizeof(double);
}

extern "C" int vt_groupnorm_act(const void* x, int in_dtype, int64_t ldx, void* y, int out_dtype, int64_t ldy,
                                const float* gamma, const float* beta, int32_t B, int32_t T, int64_t HW, int32_t C,
                                int32_t groups, int32_t scope, float eps, int32_t silu, void* work, vt_stream stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  VT_CHECK_ARG(x && y && gamma && beta, "vt_groupnorm_act: null pointer");
  VT_CHECK_ARG(B > 0 && T > 0 && HW > 0 && C > 0 && ldx >= C && ldy >= C, "vt_groupnorm_act: bad dims");
  VT_CHECK_ARG(groups == 32 && C % groups == 0 && C % 4 == 0, "vt_groupnorm_act: needs 32 groups dividing C (C=%d)", C);
  VT_CHECK_ARG(scope >= VT_GN_FRAME && scope <= VT_GN_CLIP, "vt_groupnorm_act: scope %d", scope);
  double* w = reinterpret_cast<double*>(work);
  if (in_dtype == VT_F32 && out_dtype == VT_F32)
    return launch_groupnorm<float, float>(x, ldx, y, ldy, gamma, beta, B, T, HW, C, groups, scope, eps, silu, w, stream);
  if (in_dtype == VT_BF16 && out_dtype == VT_BF16)
    return launch_groupnorm<bf16_t, bf16_t>(x, ldx, y, ldy, gamma, beta, B, T, HW, C, groups, scope, eps, silu, w, stream);
  if (in_dtype == VT_F32 && out_dtype == VT_BF16)
    return launch_groupnorm<float, bf16_t>(x, ldx, y, ldy, gamma, beta, B, T, HW, C, groups, scope, eps, silu, w, stream);
  if (in_dtype == VT_BF16 && out_dtype == VT_F32)
    return launch_groupnorm<bf16_t, float>(x, ldx, y, ldy, gamma, beta, B, T, HW, C, groups, scope, eps, silu, w, stream);
  if (in_dtype == VT_F16 && out_dtype == VT_F16)
    return launch_groupnorm<f16_t, f16_t>(x, ldx, y, ldy, gamma, beta, B, T, HW, C, groups, scope, eps, silu, w, stream);
  if (in_dtype == VT_F32 && out_dtype == VT_F16)
    return launch_groupnorm<float, f16_t>(x, ldx, y, ldy, gamma, beta, B, T, HW, C, groups, scope, eps, silu, w, stream);
  if (in_dtype == VT_F16 && out_dtype == VT_F32)
    return launch_groupnorm<f16_t, float>(x, ldx, y, ldy, gamma, beta, B, T, HW, C, groups, scope, eps, silu, w, stream);
  vt_set_error("vt_groupnorm_act: dtype combination %d -> %d", in_dtype, out_dtype);
  return VT_ERR_ARG;
}
