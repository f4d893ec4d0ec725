// Evaluation metrics of the reference's eval loop on the GPU: the
// post-processing output.clamp(-1,1), (x+1)/2 (scripts/inference_evaluate.py:175-176) fused with
// compute_psnr / compute_ssim (vidtok/modules/util.py:146-178, 181-222, 306-324).  Inputs are the
// NCTHW fp32 tensors of the model API; one value per frame comes back.
//
// SSIM: 11x11 Gaussian (sigma 1.5), valid convolution, k1 = 0.01, k2 = 0.03, data_range 1, optional
// f x f average pooling with f = max(1, round(min(H,W)/256)).  One workgroup = one 32x32 tile of the
// SSIM map of one (frame, channel): the 42x42 input patches of x and y go to LDS once, the five
// filtered moments (x, y, xx, yy, xy) are produced separably (row pass into LDS, column pass in
// registers), the SSIM values are reduced in the workgroup and added to the per-frame accumulator.
#include "common.h"

namespace {

constexpr int KS = 11;            // Gaussian taps
constexpr int TILE = 32;          // SSIM-map tile edge
constexpr int PATCH = TILE + KS - 1;

struct GaussTaps {
  float g[KS];                    // 1-D normalised taps; the reference's 2-D kernel is their outer product
};

// post-processed pixel of tensor p (NCTHW) at pooled coordinates: clamp (only the reconstruction), (v+1)/2, f x f mean
// raw = 1: the tensors are the model's input / output in [-1,1] and the eval loop's post-processing is applied here;
// raw = 0: they are already images in [0,1] (the reference's compute_psnr / compute_ssim argument convention)
__device__ __forceinline__ float post(float v, bool clamp, int raw) {
  if (!raw) return v;
  if (clamp) v = fminf(fmaxf(v, -1.0f), 1.0f);
  return (v + 1.0f) * 0.5f;
}

__device__ __forceinline__ float pooled(const float* __restrict__ p, long long base, int W, int py, int px, int f,
                                        bool clamp, int raw) {
  float s = 0.f;
  for (int dy = 0; dy < f; ++dy)
    for (int dx = 0; dx < f; ++dx) s += post(p[base + (long long)(py * f + dy) * W + (px * f + dx)], clamp, raw);
  return s / (float)(f * f);
}

__global__ __launch_bounds__(256) void ssim_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                   float* __restrict__ acc, GaussTaps gt, int B, int C, int T, int H,
                                                   int W, int f, int tiles_x, int tiles_y, int raw) {
  __shared__ float sx[PATCH][PATCH + 1], sy[PATCH][PATCH + 1];
  __shared__ float rows[5][PATCH][TILE + 1];
  __shared__ float red[4];
  const int Hp = H / f, Wp = W / f;               // pooled size (avg_pool2d floors)
  const int Ho = Hp - KS + 1, Wo = Wp - KS + 1;   // SSIM map size
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y; bid /= tiles_y;
  const int t = bid % T; bid /= T;
  const int c = bid % C;
  const int b = bid / C;
  const long long base = (((long long)b * C + c) * T + t) * (long long)H * W;
  const int y0 = ty * TILE, x0 = tx * TILE;
  for (int i = threadIdx.x; i < PATCH * PATCH; i += 256) {
    const int r = i / PATCH, q = i % PATCH;
    const int py = y0 + r, px = x0 + q;
    float vx = 0.f, vy = 0.f;
    if (py < Hp && px < Wp) {
      vx = pooled(x, base, W, py, px, f, false, raw);
      vy = pooled(y, base, W, py, px, f, true, raw);
    }
    sx[r][q] = vx;
    sy[r][q] = vy;
  }
  __syncthreads();
  // row pass: 5 moments for every patch row, TILE output columns
  for (int i = threadIdx.x; i < PATCH * TILE; i += 256) {
    const int r = i / TILE, q = i % TILE;
    float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const float a = sx[r][q + k], bb = sy[r][q + k], g = gt.g[k];
      m[0] += g * a; m[1] += g * bb; m[2] += g * a * a; m[3] += g * bb * bb; m[4] += g * a * bb;
    }
#pragma unroll
    for (int e = 0; e < 5; ++e) rows[e][r][q] = m[e];
  }
  __syncthreads();
  // column pass + SSIM
  const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
  float ssum = 0.f;
  for (int i = threadIdx.x; i < TILE * TILE; i += 256) {
    const int r = i / TILE, q = i % TILE;
    if (y0 + r < Ho && x0 + q < Wo) {
      float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        const float g = gt.g[k];
#pragma unroll
        for (int e = 0; e < 5; ++e) m[e] += g * rows[e][r + k][q];
      }
      const float mu_xx = m[0] * m[0], mu_yy = m[1] * m[1], mu_xy = m[0] * m[1];
      const float s_xx = m[2] - mu_xx, s_yy = m[3] - mu_yy, s_xy = m[4] - mu_xy;
      const float cs = (2.0f * s_xy + c2) / (s_xx + s_yy + c2);
      ssum += (2.0f * mu_xy + c1) / (mu_xx + mu_yy + c1) * cs;
    }
  }
  ssum = wave_sum(ssum, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ssum;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(acc + (long long)b * T + t, (red[0] + red[1]) + (red[2] + red[3]));
}

// squared error of the post-processed frames: acc[b*T+t] += sum over (c,h,w)
__global__ __launch_bounds__(256) void sqerr_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                    float* __restrict__ acc, int B, int C, int T, long long HW,
                                                    int chunks, int raw) {
  __shared__ float red[4];
  int bid = blockIdx.x;
  const int ch = bid % chunks; bid /= chunks;
  const int t = bid % T; bid /= T;
  const int c = bid % C;
  const int b = bid / C;
  const long long base = (((long long)b * C + c) * T + t) * HW;
  const long long per = (HW + chunks - 1) / chunks;
  const long long lo = ch * per, hi = min(HW, lo + per);
  float s = 0.f;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    const float a = post(x[base + i], false, raw);
    const float r = post(y[base + i], true, raw);
    s += (a - r) * (a - r);
  }
  s = wave_sum(s, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(acc + (long long)b * T + t, (red[0] + red[1]) + (red[2] + red[3]));
}

__global__ void metrics_finish_kernel(const float* __restrict__ sq, const float* __restrict__ ss, float* psnr,
                                      float* ssim, int n, float inv_mse_count, float inv_ssim_count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    psnr[i] = -10.0f * log10f(sq[i] * inv_mse_count + 1e-8f);   // util.py:152-154
    ssim[i] = ss[i] * inv_ssim_count;
  }
}

}  // namespace

extern "C" int64_t vt_eval_work_floats(int32_t B, int32_t T) { return 2ll * B * T; }

extern "C" int vt_eval_psnr_ssim(const float* x, const float* y, float* psnr, float* ssim, float* work, int32_t B,
                                 int32_t C, int32_t T, int32_t H, int32_t W, int32_t raw, vt_stream stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  VT_CHECK_ARG(x && y && psnr && ssim && work && B > 0 && C > 0 && T > 0 && H > 0 && W > 0,
               "vt_eval_psnr_ssim: bad arguments");
  int f = (int)lrintf((float)(H < W ? H : W) / 256.0f);   // python round(): half to even, like lrintf
  if (f < 1) f = 1;
  const int Hp = H / f, Wp = W / f;
  VT_CHECK_ARG(Hp >= KS && Wp >= KS, "vt_eval_psnr_ssim: Kernel size can't be greater than actual input size (%dx%d)", Hp, Wp);
  const int Ho = Hp - KS + 1, Wo = Wp - KS + 1;
  GaussTaps gt;
  {  // gaussian_filter (util.py:306-324): exp(-(i-5)^2 / (2 sigma^2)), normalised; 2-D kernel = outer product
    double s = 0.0, g[KS];
    for (int i = 0; i < KS; ++i) {
      const double d = i - (KS - 1) / 2.0;
      g[i] = exp(-(d * d) / (2.0 * 1.5 * 1.5));
      s += g[i];
    }
    for (int i = 0; i < KS; ++i) gt.g[i] = (float)(g[i] / s);
  }
  const int n = B * T;
  float* sq = work;
  float* ss = work + n;
  VT_CHECK_HIP(hipMemsetAsync(work, 0, 2ll * n * sizeof(float), stream));
  const long long HW = (long long)H * W;
  int chunks = (int)((HW + 65535) / 65536);
  if (chunks < 1) chunks = 1;
  hipLaunchKernelGGL(sqerr_kernel, dim3((unsigned)(B * C * T * chunks)), dim3(256), 0, stream, x, y, sq, B, C, T, HW,
                     chunks, (int)raw);
  VT_CHECK_LAUNCH();
  const int tiles_x = (Wo + TILE - 1) / TILE, tiles_y = (Ho + TILE - 1) / TILE;
  hipLaunchKernelGGL(ssim_kernel, dim3((unsigned)(B * C * T * tiles_x * tiles_y)), dim3(256), 0, stream, x, y, ss, gt, B,
                     C, T, H, W, f, tiles_x, tiles_y, (int)raw);
  VT_CHECK_LAUNCH();
  hipLaunchKernelGGL(metrics_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const float*)sq,
                     (const float*)ss, psnr, ssim, n, 1.0f / ((float)C * (float)HW), 1.0f / ((float)C * (float)Ho * (float)Wo));
  VT_CHECK_LAUNCH();
  return VT_OK;
}
