// Device halves of the video front / back end around model(x): what
// scripts/inference_reconstruct.py of the reference does between the codec and the model, minus the codec.
//   front: decoded uint8 frames [T][H0][W0][3] -> x fp32 NCTHW in [-1,1]:  /255, torchvision Resize(size, antialias=True)
//          (= ATen's separable anti-aliased bilinear filter), CenterCrop, Normalize(0.5, 0.5)
//          (inference_reconstruct.py:39-45,70-73; vidtok/data/vidtok.py:180-188)
//   back:  x_hat fp32 NCTHW -> uint8 frames [T][H][Wtot][3] (clamp, (x+1)/2, *255, truncation; optional side-by-side
//          with the input: inference_reconstruct.py:76-80,228-235)
//   chain: copy frames between NCTHW tensors with optional clamp (--pad_gen_frames: the last f-1 generated frames are
//          prepended to the next clip, inference_reconstruct.py:213-221)
// All three are HBM-bound byte / fp32 streams: one thread per output element, coalesced along W.
#include "common.h"

namespace {

constexpr int kBlock = 256;

// ATen _compute_indices_weights_aa (aten/src/ATen/native/cpu/UpSampleKernel.cpp), bilinear = triangle filter,
// align_corners = false, fp32: window [xmin, xmin + xsize) and the normalising total for output index i
struct AAxis {
  float scale, support, invscale;
  int in_size;
};
__device__ __forceinline__ float aa_filter(float x) {
  const float a = fabsf(x);
  return a < 1.0f ? 1.0f - a : 0.0f;
}
__device__ __forceinline__ void aa_window(const AAxis& a, int i, int& xmin, int& xsize, float& center, float& total) {
  center = __fmul_rn(a.scale, (float)i + 0.5f);
  xmin = max((int)(long long)(__fadd_rn(__fsub_rn(center, a.support), 0.5f)), 0);
  xsize = min((int)(long long)(__fadd_rn(__fadd_rn(center, a.support), 0.5f)), a.in_size) - xmin;
  total = 0.0f;
  for (int j = 0; j < xsize; ++j)
    total = __fadd_rn(total, aa_filter(__fmul_rn(__fadd_rn(__fsub_rn((float)(j + xmin), center), 0.5f), a.invscale)));
}
__device__ __forceinline__ float aa_weight(const AAxis& a, int xmin, int j, float center, float total) {
  const float w = aa_filter(__fmul_rn(__fadd_rn(__fsub_rn((float)(j + xmin), center), 0.5f), a.invscale));
  return total != 0.0f ? __fdiv_rn(w, total) : w;
}

// pass 1: horizontal filter of the columns the crop keeps: tmp[t][c][y][xo] = sum_j w_j * frames[t][y][xmin+j][c] / 255
__global__ __launch_bounds__(kBlock) void frames_resize_w_kernel(const uint8_t* __restrict__ frames, float* __restrict__ tmp,
                                                                 int T, int H0, int W0, int W, int left, AAxis ax) {
  const long long n = (long long)T * 3 * H0 * W;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
    const int xo = (int)(i % W);
    long long r = i / W;
    const int y = (int)(r % H0); r /= H0;
    const int c = (int)(r % 3);
    const int t = (int)(r / 3);
    int xmin, xsize;
    float center, total;
    aa_window(ax, xo + left, xmin, xsize, center, total);
    const uint8_t* src = frames + (((long long)t * H0 + y) * W0) * 3 + c;
    float acc = 0.0f;
    for (int j = 0; j < xsize; ++j) {
      const float v = __fdiv_rn((float)src[(long long)(xmin + j) * 3], 255.0f);
      acc = __fadd_rn(acc, __fmul_rn(v, aa_weight(ax, xmin, j, center, total)));
    }
    tmp[i] = acc;
  }
}

// pass 2: vertical filter of the rows the crop keeps, Normalize(0.5, 0.5), NCTHW store at frame offset t_off
__global__ __launch_bounds__(kBlock) void frames_resize_h_kernel(const float* __restrict__ tmp, float* __restrict__ x, int T,
                                                                 int H0, int H, int W, int top, int Tdst, int t_off, AAxis ay) {
  const long long n = (long long)T * 3 * H * W;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
    const int xo = (int)(i % W);
    long long r = i / W;
    const int yo = (int)(r % H); r /= H;
    const int c = (int)(r % 3);
    const int t = (int)(r / 3);
    int ymin, ysize;
    float center, total;
    aa_window(ay, yo + top, ymin, ysize, center, total);
    const float* src = tmp + (((long long)t * 3 + c) * H0) * W + xo;
    float acc = 0.0f;
    for (int j = 0; j < ysize; ++j) acc = __fadd_rn(acc, __fmul_rn(src[(long long)(ymin + j) * W], aa_weight(ay, ymin, j, center, total)));
    x[(((long long)c * Tdst + t_off + t) * H + yo) * W + xo] = __fdiv_rn(__fsub_rn(acc, 0.5f), 0.5f);
  }
}

// back end: x [C=3][Tsrc][H][W] frames [t0, t0+n) -> out[n][H][Wtot][3] at column offset w_off
__global__ __launch_bounds__(kBlock) void ncthw_to_frames_u8_kernel(const float* __restrict__ x, uint8_t* __restrict__ out, int Tsrc,
                                                                    int t0, int n, int H, int W, int Wtot, int w_off) {
  const long long total = (long long)n * H * W * 3;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
    const int c = (int)(i % 3);
    long long r = i / 3;
    const int xo = (int)(r % W); r /= W;
    const int y = (int)(r % H);
    const int t = (int)(r / H);
    float v = x[(((long long)c * Tsrc + t0 + t) * H + y) * W + xo];
    v = fminf(fmaxf(v, -1.0f), 1.0f);
    v = __fdiv_rn(__fadd_rn(v, 1.0f), 2.0f);
    out[(((long long)t * H + y) * Wtot + w_off + xo) * 3 + c] = (uint8_t)(int)__fmul_rn(v, 255.0f);   // truncation, as numpy astype
  }
}

// chain: dst[c][td0 + k][..] = (clamp) src[c][ts0 + k][..], k < n, for every channel c of NCTHW tensors with B = 1
__global__ __launch_bounds__(kBlock) void ncthw_copy_frames_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int Ts,
                                                                   int Td, int ts0, int td0, int n, long long HW, int clamp) {
  const long long total = (long long)C * n * HW;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
    const long long p = i % HW;
    long long r = i / HW;
    const int k = (int)(r % n);
    const int c = (int)(r / n);
    float v = src[((long long)c * Ts + ts0 + k) * HW + p];
    if (clamp) v = fminf(fmaxf(v, -1.0f), 1.0f);
    dst[((long long)c * Td + td0 + k) * HW + p] = v;
  }
}

inline unsigned grid_for(long long n) {
  long long b = (n + kBlock - 1) / kBlock;
  if (b > 65536) b = 65536;
  if (b < 1) b = 1;
  return (unsigned)b;
}

inline AAxis make_axis(int in_size, int out_size) {
  AAxis a;
  a.in_size = in_size;
  a.scale = (float)in_size / (float)out_size;
  a.support = a.scale >= 1.0f ? a.scale : 1.0f;
  a.invscale = a.scale >= 1.0f ? 1.0f / a.scale : 1.0f;
  return a;
}

}  // namespace

extern "C" int64_t vt_frames_work_floats(int32_t T, int32_t H0, int32_t W) { return (int64_t)T * 3 * H0 * W; }

extern "C" int vt_frames_u8_to_ncthw(const uint8_t* frames, int32_t T, int32_t H0, int32_t W0, int32_t Hr, int32_t Wr, int32_t top,
                                     int32_t left, float* x, int32_t Tdst, int32_t t_off, int32_t H, int32_t W, float* work,
                                     vt_stream stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  VT_CHECK_ARG(frames && x && work, "vt_frames_u8_to_ncthw: null pointer");
  VT_CHECK_ARG(T > 0 && H0 > 0 && W0 > 0 && Hr > 0 && Wr > 0 && H > 0 && W > 0, "vt_frames_u8_to_ncthw: bad dims");
  VT_CHECK_ARG(top >= 0 && left >= 0 && top + H <= Hr && left + W <= Wr, "vt_frames_u8_to_ncthw: crop window outside the resized frame");
  VT_CHECK_ARG(t_off >= 0 && t_off + T <= Tdst, "vt_frames_u8_to_ncthw: frame range outside the destination");
  hipLaunchKernelGGL(frames_resize_w_kernel, dim3(grid_for((long long)T * 3 * H0 * W)), dim3(kBlock), 0, stream, frames, work, T, H0,
                     W0, W, left, make_axis(W0, Wr));
  VT_CHECK_LAUNCH();
  hipLaunchKernelGGL(frames_resize_h_kernel, dim3(grid_for((long long)T * 3 * H * W)), dim3(kBlock), 0, stream, (const float*)work, x, T,
                     H0, H, W, top, Tdst, t_off, make_axis(H0, Hr));
  VT_CHECK_LAUNCH();
  return VT_OK;
}

extern "C" int vt_ncthw_to_frames_u8(const float* x, int32_t Tsrc, int32_t t0, int32_t n, int32_t H, int32_t W, uint8_t* out,
                                     int32_t Wtot, int32_t w_off, vt_stream stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  VT_CHECK_ARG(x && out && n > 0 && H > 0 && W > 0, "vt_ncthw_to_frames_u8: bad arguments");
  VT_CHECK_ARG(t0 >= 0 && t0 + n <= Tsrc && w_off >= 0 && w_off + W <= Wtot, "vt_ncthw_to_frames_u8: frame / column range");
  hipLaunchKernelGGL(ncthw_to_frames_u8_kernel, dim3(grid_for((long long)n * H * W * 3)), dim3(kBlock), 0, stream, x, out, Tsrc, t0, n,
                     H, W, Wtot, w_off);
  VT_CHECK_LAUNCH();
  return VT_OK;
}

extern "C" int vt_ncthw_copy_frames(const float* src, float* dst, int32_t C, int32_t Ts, int32_t Td, int32_t ts0, int32_t td0, int32_t n,
                                    int64_t HW, int32_t clamp, vt_stream stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  VT_CHECK_ARG(src && dst && C > 0 && n > 0 && HW > 0, "vt_ncthw_copy_frames: bad arguments");
  VT_CHECK_ARG(ts0 >= 0 && ts0 + n <= Ts && td0 >= 0 && td0 + n <= Td, "vt_ncthw_copy_frames: frame range");
  hipLaunchKernelGGL(ncthw_copy_frames_kernel, dim3(grid_for((long long)C * n * HW)), dim3(kBlock), 0, stream, src, dst, C, Ts, Td, ts0,
                     td0, n, (long long)HW, clamp);
  VT_CHECK_LAUNCH();
  return VT_OK;
}
