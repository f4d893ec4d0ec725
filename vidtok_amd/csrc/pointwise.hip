// HBM-bound kernels of the VidTok path on gfx950: per-position LayerNorm(+SiLU), row softmax,
// layout conversion, time resamplers and frame gather.  Operator contracts and the reference
// call sites they replace are in include/vidtok_amd.h.
//
// Design notes (MI355X): every kernel reads and writes 8-16 B per lane with the channel dim
// innermost (NDHWC), so a wave touches whole 128-B lines; reductions over C never leave the
// wave (sub-wave __shfl_xor trees, no LDS); grids are capped at ~2048 workgroups and
// grid-stride the rest (guide G11).
#include "common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxGrid = 2048;

template <typename T>
__device__ __forceinline__ void load4(const T* p, float (&v)[4]);
template <>
__device__ __forceinline__ void load4<float>(const float* p, float (&v)[4]) {
  const f32x4 t = *reinterpret_cast<const f32x4*>(p);
  v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
template <>
__device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float (&v)[4]) {
  const u32x2 t = *reinterpret_cast<const u32x2*>(p);
  v[0] = bf16_bits_to_f32(t[0] & 0xffffu);
  v[1] = bf16_bits_to_f32(t[0] >> 16);
  v[2] = bf16_bits_to_f32(t[1] & 0xffffu);
  v[3] = bf16_bits_to_f32(t[1] >> 16);
}
template <>
__device__ __forceinline__ void load4<f16_t>(const f16_t* p, float (&v)[4]) {
  const u32x2 t = *reinterpret_cast<const u32x2*>(p);
  v[0] = h16<f16_t>::lo(t[0]); v[1] = h16<f16_t>::hi(t[0]);
  v[2] = h16<f16_t>::lo(t[1]); v[3] = h16<f16_t>::hi(t[1]);
}
template <typename T>
__device__ __forceinline__ void store4(T* p, const float (&v)[4]);
template <>
__device__ __forceinline__ void store4<f16_t>(f16_t* p, const float (&v)[4]) {
  u32x2 t;
  t[0] = h16<f16_t>::pack(v[0], v[1]);
  t[1] = h16<f16_t>::pack(v[2], v[3]);
  *reinterpret_cast<u32x2*>(p) = t;
}
template <>
__device__ __forceinline__ void store4<float>(float* p, const float (&v)[4]) {
  f32x4 t;
  t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3];
  *reinterpret_cast<f32x4*>(p) = t;
}
template <>
__device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float (&v)[4]) {
  u32x2 t;
  t[0] = f32_to_bf16_bits(v[0]) | (f32_to_bf16_bits(v[1]) << 16);
  t[1] = f32_to_bf16_bits(v[2]) | (f32_to_bf16_bits(v[3]) << 16);
  *reinterpret_cast<u32x2*>(p) = t;
}

// ---- LayerNorm over C (+SiLU) ---------------------------------------------------------------
// LP lanes cooperate on one position; each lane owns R chunks of 8 channels (chunk r covers
// channels (r*LP + l)*8 .. +7: one 16-B load in bf16, two in fp32).  Two-pass statistics in
// registers (mean, then centred sum of squares): same biased variance as torch.nn.LayerNorm, fp32.
template <typename T>
__device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <>
__device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p);
  const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
}
template <>
__device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&v)[8]) {
  const u32x4 t = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    v[2 * e] = bf16_bits_to_f32(t[e] & 0xffffu);
    v[2 * e + 1] = bf16_bits_to_f32(t[e] >> 16);
  }
}
template <>
__device__ __forceinline__ void load8<f16_t>(const f16_t* p, float (&v)[8]) {
  const u32x4 t = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    v[2 * e] = h16<f16_t>::lo(t[e]);
    v[2 * e + 1] = h16<f16_t>::hi(t[e]);
  }
}
template <typename T>
__device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <>
__device__ __forceinline__ void store8<f16_t>(f16_t* p, const float (&v)[8]) {
  u32x4 t;
#pragma unroll
  for (int e = 0; e < 4; ++e) t[e] = h16<f16_t>::pack(v[2 * e], v[2 * e + 1]);
  *reinterpret_cast<u32x4*>(p) = t;
}
template <>
__device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
  f32x4 a, b;
#pragma unroll
  for (int e = 0; e < 4; ++e) { a[e] = v[e]; b[e] = v[4 + e]; }
  *reinterpret_cast<f32x4*>(p) = a;
  *reinterpret_cast<f32x4*>(p + 4) = b;
}
template <>
__device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float (&v)[8]) {
  u32x4 t;
#pragma unroll
  for (int e = 0; e < 4; ++e) t[e] = f32_to_bf16_bits(v[2 * e]) | (f32_to_bf16_bits(v[2 * e + 1]) << 16);
  *reinterpret_cast<u32x4*>(p) = t;
}

template <typename TI, typename TO, int LP, int R, bool SILU>
__global__ __launch_bounds__(kBlock) void layernorm_act_kernel(const TI* __restrict__ x, long long ldx,
                                                               TO* __restrict__ y, long long ldy,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, long long M,
                                                               int C, float eps) {
  constexpr int GROUPS = kBlock / LP;
  const int l = threadIdx.x % LP;
  const int g = threadIdx.x / LP;
  float gm[R][8], bt[R][8];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    load8<float>(gamma + (r * LP + l) * 8, gm[r]);
    load8<float>(beta + (r * LP + l) * 8, bt[r]);
  }
  const float invC = 1.0f / (float)C;
  // U positions per lane group and iteration: their loads are issued back to back so every lane has
  // U*R 16-B requests in flight (a single request per lane leaves the kernel latency-bound at ~3.7 TB/s)
  constexpr int U = (R == 1) ? 4 : 2;
  const long long stride = (long long)gridDim.x * GROUPS * U;
  for (long long m0 = ((long long)blockIdx.x * GROUPS + g) * U; m0 < M; m0 += stride) {
    float v[U][R][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long m = (m0 + u < M) ? m0 + u : M - 1;   // tail: re-read the last row, never stored
#pragma unroll
      for (int r = 0; r < R; ++r) load8<TI>(x + m * ldx + (r * LP + l) * 8, v[u][r]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[u][r][e];
      const float mean = group_sum_dpp<LP>(s) * invC;
      float q = 0.f;
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = v[u][r][e] - mean;
          q += d * d;
        }
      const float var = group_sum_dpp<LP>(q) * invC;
      const float rstd = __builtin_amdgcn_rsqf(var + eps);
      if (m0 + u < M) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float t = (v[u][r][e] - mean) * rstd * gm[r][e] + bt[r][e];
            o[e] = SILU ? silu_fast(t) : t;
          }
          store8<TO>(y + (m0 + u) * ldy + (r * LP + l) * 8, o);
        }
      }
    }
  }
}

template <typename TI, typename TO>
int launch_layernorm(const void* x, long long ldx, void* y, long long ldy, const float* gamma,
                     const float* beta, long long M, int C, float eps, int silu, hipStream_t stream) {
  // choose lanes-per-position LP (power of two <= 64) and chunks-per-lane R with C == 8*LP*R
  int LP = 0, R = 0;
  for (int lp = 64; lp >= 4 && LP == 0; lp >>= 1)
    for (int r = 1; r <= 4; r <<= 1)
      if (C == 8 * lp * r) { LP = lp; R = r; break; }
  VT_CHECK_ARG(LP != 0, "vt_layernorm_act: unsupported channel count C=%d (need C = 8*LP*R, LP in {4..64}, R in {1,2,4})", C);
  const int groups = (kBlock / LP) * ((R == 1) ? 4 : 2);   // positions per workgroup and iteration
  long long blocks = (M + groups - 1) / groups;
  if (blocks > kMaxGrid) blocks = kMaxGrid;
  if (blocks < 1) blocks = 1;
#define VT_LN_CASE(lp, r)                                                                            \
  if (LP == lp && R == r) {                                                                          \
    if (silu)                                                                                        \
      hipLaunchKernelGGL((layernorm_act_kernel<TI, TO, lp, r, true>), dim3((unsigned)blocks), dim3(kBlock), 0,  \
                         stream, (const TI*)x, ldx, (TO*)y, ldy, gamma, beta, M, C, eps);             \
    else                                                                                             \
      hipLaunchKernelGGL((layernorm_act_kernel<TI, TO, lp, r, false>), dim3((unsigned)blocks), dim3(kBlock), 0, \
                         stream, (const TI*)x, ldx, (TO*)y, ldy, gamma, beta, M, C, eps);             \
    VT_CHECK_LAUNCH();                                                                               \
    return VT_OK;                                                                                    \
  }
  VT_LN_CASE(64, 1) VT_LN_CASE(64, 2) VT_LN_CASE(64, 4)
  VT_LN_CASE(32, 1) VT_LN_CASE(16, 1) VT_LN_CASE(8, 1) VT_LN_CASE(4, 1)
#undef VT_LN_CASE
  vt_set_error("vt_layernorm_act: no kernel for LP=%d R=%d", LP, R);
  return VT_ERR_UNSUPPORTED;
}

// ---- row softmax ------------------------------------------------------------------------------
// one wave per row; the row (<= a few K floats) is re-read from L2 for the three passes.
template <typename TO>
__global__ __launch_bounds__(kBlock) void softmax_rows_kernel(const float* __restrict__ s, TO* __restrict__ p,
                                                              long long rows, int cols, long long ldp,
                                                              float scale) {
  const int lane = threadIdx.x & 63;
  const long long wave0 = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * (kBlock / 64);
  for (long long r = wave0; r < rows; r += nwaves) {
    const float* sr = s + r * (long long)cols;
    float mx = -INFINITY;
    for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, sr[c] * scale);
    mx = wave_max(mx, 64);
    float sum = 0.f;
    for (int c = lane; c < cols; c += 64) sum += __expf(sr[c] * scale - mx);
    sum = wave_sum(sum, 64);
    const float inv = 1.0f / sum;
    TO* pr = p + r * ldp;
    for (int c = lane; c < cols; c += 64) pr[c] = from_f32<TO>(__expf(sr[c] * scale - mx) * inv);
  }
}

// ---- layout ----------------------------------------------------------------------------------
template <typename TO>
__global__ __launch_bounds__(kBlock) void ncthw_to_ndhwc_kernel(const float* __restrict__ x, TO* __restrict__ y,
                                                                int B, int C, int T, int H, int W, int ldy,
                                                                int tpad) {
  const long long HW = (long long)H * W;
  const long long npix = (long long)B * (T + tpad) * HW;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < npix; i += (long long)gridDim.x * kBlock) {
    const long long hw = i % HW;
    long long r = i / HW;
    const int tp = (int)(r % (T + tpad));
    const int b = (int)(r / (T + tpad));
    const int t = tp < tpad ? 0 : tp - tpad;
    TO* yp = y + i * ldy;
    for (int c = 0; c < ldy; ++c) {
      float v = 0.f;
      if (c < C) v = x[(((long long)b * C + c) * T + t) * HW + hw];
      yp[c] = from_f32<TO>(v);
    }
  }
}

template <typename TI>
__global__ __launch_bounds__(kBlock) void ndhwc_to_ncthw_kernel(const TI* __restrict__ x, float* __restrict__ y,
                                                                int B, int C, int T, int H, int W, int ldx,
                                                                int ttrim) {
  const long long HW = (long long)H * W;
  const int To = T - ttrim;
  const long long n = (long long)B * C * To * HW;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
    const long long hw = i % HW;
    long long r = i / HW;
    const int t = (int)(r % To);
    r /= To;
    const int c = (int)(r % C);
    const int b = (int)(r / C);
    y[i] = to_f32<TI>(x[(((long long)b * T + (t + ttrim)) * HW + hw) * ldx + c]);
  }
}

// ---- time resamplers --------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void time_avgpool3s2_kernel(const T* __restrict__ x, const T* __restrict__ cache,
                                                                 T* __restrict__ y, int B, int Ti, long long F4,
                                                                 int tmode) {
  // F4 = frame elements / 4 ; one thread = 4 contiguous elements of one output frame
  const int To = Ti / 2;
  const long long n = (long long)B * To * F4;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
    const long long f = i % F4;
    long long r = i / F4;
    const int to = (int)(r % To);
    const int b = (int)(r / To);
    float a[4], c[4], d[4], o[4];
    const T* xb = x + ((long long)b * Ti) * F4 * 4 + f * 4;
    const int t0 = (tmode == VT_TPAD_ZERO_BACK) ? 2 * to : 2 * to - 1;  // x index of the first tap (-1 = front pad frame)
    if (t0 >= 0) {
      load4<T>(xb + (long long)t0 * F4 * 4, a);
    } else if (tmode == VT_TPAD_REPLICATE) {
      load4<T>(xb, a);
    } else if (tmode == VT_TPAD_CACHE) {
      load4<T>(cache + (long long)b * F4 * 4 + f * 4, a);
    } else {
      a[0] = a[1] = a[2] = a[3] = 0.f;
    }
    load4<T>(xb + (long long)(t0 + 1) * F4 * 4, c);
    if (t0 + 2 < Ti) load4<T>(xb + (long long)(t0 + 2) * F4 * 4, d);
    else d[0] = d[1] = d[2] = d[3] = 0.f;    // back pad frame (VT_TPAD_ZERO_BACK only)
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = ((a[e] + c[e]) + d[e]) / 3.0f;
    store4<T>(y + i * 4, o);
  }
}

// The sequence that is interpolated is [h (nh frames) | x (Tx frames)] per clip -- nh = 0: x alone -- and the first `skip`
// of its 2 (nh + Tx) output frames are not produced (v1.1 chunks after the first: h = the cached frames of the previous
// chunk, whose part of the output the previous chunk already delivered; model_3dcausal_v1_1.py:327-341).
// One output frame per blockIdx.y (round 6): the source frames and weights are scalar work done once per workgroup, a thread moves 16 bytes of
// the 16-bit types (two quads through the same load4 / store4 arithmetic, so the bits are those of the element-indexed first version, whose 64-bit
// divisions per quad held it at 0.6 TB/s: 330 us for the 8 frames of a 256 x 256 x 128 chunk, 13 ms of a four-pass tiled run, profiles/r05_tiled_kernel_stats.md)
template <typename T>
__global__ __launch_bounds__(kBlock) void time_lerp2x_kernel(const T* __restrict__ h, int nh, const T* __restrict__ x, T* __restrict__ y,
                                                             int B, int Tx, int skip, long long F4) {
  constexpr int Q = sizeof(T) == 2 ? 2 : 1;                // quads per thread and iteration
  const int Ti = nh + Tx;
  const int To = 2 * Ti - skip;
  const int r = blockIdx.y;                                // output frame (b, j - skip)
  const int b = r / To;
  const int j = r - b * To + skip;
  // align_corners=False source coordinate, scale 1/2:  src = (j + 0.5) * 0.5 - 0.5, clamped at 0
  float src = ((float)j + 0.5f) * 0.5f - 0.5f;
  if (src < 0.f) src = 0.f;
  const int t0 = (int)src;
  const int t1 = t0 + (t0 < Ti - 1 ? 1 : 0);
  const float l1 = src - (float)t0;
  const float l0 = 1.0f - l1;
  const T* p0 = t0 < nh ? h + ((long long)b * nh + t0) * F4 * 4 : x + ((long long)b * Tx + (t0 - nh)) * F4 * 4;
  const T* p1 = t1 < nh ? h + ((long long)b * nh + t1) * F4 * 4 : x + ((long long)b * Tx + (t1 - nh)) * F4 * 4;
  T* yo = y + (long long)r * F4 * 4;
  for (long long f = ((long long)blockIdx.x * kBlock + threadIdx.x) * Q; f < F4; f += (long long)gridDim.x * kBlock * Q) {
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      if (f + q < F4) {
        float a[4], c[4], o[4];
        load4<T>(p0 + (f + q) * 4, a);
        load4<T>(p1 + (f + q) * 4, c);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = l0 * a[e] + l1 * c[e];
        store4<T>(yo + (f + q) * 4, o);
      }
    }
  }
}

struct GatherIdx {
  int idx[128];
};

// VT = u32x4 when frames and strides are multiples of 16 bytes (every activation frame), uint32_t otherwise (small
// index maps of odd sizes): the unit counts are in VT units
template <typename VT>
__global__ __launch_bounds__(kBlock) void gather_frames_kernel(const VT* __restrict__ src, VT* __restrict__ dst,
                                                               int B, long long F16, long long sbs16,
                                                               long long dbs16, GatherIdx gi, int n) {
  const long long total = (long long)B * n * F16;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
    const long long f = i % F16;
    long long r = i / F16;
    const int j = (int)(r % n);
    const int b = (int)(r / n);
    dst[(long long)b * dbs16 + (long long)j * F16 + f] = src[(long long)b * sbs16 + (long long)gi.idx[j] * F16 + f];
  }
}

inline unsigned grid_for(long long n) {
  long long b = (n + kBlock - 1) / kBlock;
  if (b > kMaxGrid) b = kMaxGrid;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" int vt_layernorm_act(const void* x, int in_dtype, int64_t ldx, void* y, int out_dtype, int64_t ldy,
                                const float* gamma, const float* beta, int64_t M, int32_t C, float eps,
                                int32_t silu, vt_stream stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  VT_CHECK_ARG(x && y && gamma && beta, "vt_layernorm_act: null pointer");
  VT_CHECK_ARG(M >= 0 && C > 0 && ldx >= C && ldy >= C, "vt_layernorm_act: bad dims");
  VT_CHECK_ARG(ldx % 8 == 0 && ldy % 8 == 0, "vt_layernorm_act: row strides must be multiples of 8");
  if (M == 0) return VT_OK;
  if (in_dtype == VT_F32 && out_dtype == VT_F32)
    return launch_layernorm<float, float>(x, ldx, y, ldy, gamma, beta, M, C, eps, silu, stream);
  if (in_dtype == VT_BF16 && out_dtype == VT_BF16)
    return launch_layernorm<bf16_t, bf16_t>(x, ldx, y, ldy, gamma, beta, M, C, eps, silu, stream);
  if (in_dtype == VT_F32 && out_dtype == VT_BF16)
    return launch_layernorm<float, bf16_t>(x, ldx, y, ldy, gamma, beta, M, C, eps, silu, stream);
  if (in_dtype == VT_BF16 && out_dtype == VT_F32)
    return launch_layernorm<bf16_t, float>(x, ldx, y, ldy, gamma, beta, M, C, eps, silu, stream);
  if (in_dtype == VT_F16 && out_dtype == VT_F16)
    return launch_layernorm<f16_t, f16_t>(x, ldx, y, ldy, gamma, beta, M, C, eps, silu, stream);
  if (in_dtype == VT_F32 && out_dtype == VT_F16)
    return launch_layernorm<float, f16_t>(x, ldx, y, ldy, gamma, beta, M, C, eps, silu, stream);
  if (in_dtype == VT_F16 && out_dtype == VT_F32)
    return launch_layernorm<f16_t, float>(x, ldx, y, ldy, gamma, beta, M, C, eps, silu, stream);
  vt_set_error("vt_layernorm_act: dtype combination %d -> %d", in_dtype, out_dtype);
  return VT_ERR_ARG;
}

// tanh in place on an fp32 tensor: the `tanh_out` option of the decoders (model_3dcausal.py:866-869)
__global__ __launch_bounds__(256) void tanh_inplace_kernel(float* __restrict__ x, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) x[i] = tanhf(x[i]);
}

extern "C" int vt_tanh_inplace(float* x, int64_t n, vt_stream stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  VT_CHECK_ARG(x != nullptr && n >= 0, "vt_tanh_inplace: bad arguments");
  if (n == 0) return VT_OK;
  long long blocks = (n + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(tanh_inplace_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, (long long)n);
  VT_CHECK_LAUNCH();
  return VT_OK;
}

extern "C" int vt_softmax_rows(const float* s, void* p, int out_dtype, int64_t rows, int32_t cols, int64_t ldp,
                               float scale, vt_stream stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  VT_CHECK_ARG(s && p && rows >= 0 && cols > 0 && ldp >= cols, "vt_softmax_rows: bad arguments");
  if (rows == 0) return VT_OK;
  long long blocks = (rows + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  if (out_dtype == VT_F32)
    hipLaunchKernelGGL(softmax_rows_kernel<float>, dim3((unsigned)blocks), dim3(kBlock), 0, stream, s, (float*)p,
                       (long long)rows, cols, (long long)ldp, scale);
  else if (out_dtype == VT_BF16)
    hipLaunchKernelGGL(softmax_rows_kernel<bf16_t>, dim3((unsigned)blocks), dim3(kBlock), 0, stream, s, (bf16_t*)p,
                       (long long)rows, cols, (long long)ldp, scale);
  else if (out_dtype == VT_F16)
    hipLaunchKernelGGL(softmax_rows_kernel<f16_t>, dim3((unsigned)blocks), dim3(kBlock), 0, stream, s, (f16_t*)p,
                       (long long)rows, cols, (long long)ldp, scale);
  else
    VT_CHECK_ARG(false, "vt_softmax_rows: out_dtype %d", out_dtype);
  VT_CHECK_LAUNCH();
  return VT_OK;
}

extern "C" int vt_ncthw_to_ndhwc(const float* x, void* y, int out_dtype, int32_t B, int32_t C, int32_t T, int32_t H,
                                 int32_t W, int32_t ldy, int32_t tpad, vt_stream stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  VT_CHECK_ARG(x && y && B > 0 && C > 0 && T > 0 && H > 0 && W > 0 && ldy >= C && tpad >= 0,
               "vt_ncthw_to_ndhwc: bad arguments");
  const long long npix = (long long)B * (T + tpad) * H * W;
  if (out_dtype == VT_F32)
    hipLaunchKernelGGL(ncthw_to_ndhwc_kernel<float>, dim3(grid_for(npix)), dim3(kBlock), 0, stream, x, (float*)y, B, C,
                       T, H, W, ldy, tpad);
  else if (out_dtype == VT_BF16)
    hipLaunchKernelGGL(ncthw_to_ndhwc_kernel<bf16_t>, dim3(grid_for(npix)), dim3(kBlock), 0, stream, x, (bf16_t*)y, B,
                       C, T, H, W, ldy, tpad);
  else if (out_dtype == VT_F16)
    hipLaunchKernelGGL(ncthw_to_ndhwc_kernel<f16_t>, dim3(grid_for(npix)), dim3(kBlock), 0, stream, x, (f16_t*)y, B,
                       C, T, H, W, ldy, tpad);
  else
    VT_CHECK_ARG(false, "vt_ncthw_to_ndhwc: out_dtype %d", out_dtype);
  VT_CHECK_LAUNCH();
  return VT_OK;
}

extern "C" int vt_ndhwc_to_ncthw(const void* x, int in_dtype, float* y, int32_t B, int32_t C, int32_t T, int32_t H,
                                 int32_t W, int32_t ldx, int32_t ttrim, vt_stream stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  VT_CHECK_ARG(x && y && B > 0 && C > 0 && T > 0 && H > 0 && W > 0 && ldx >= C && ttrim >= 0 && ttrim < T,
               "vt_ndhwc_to_ncthw: bad arguments");
  const long long n = (long long)B * C * (T - ttrim) * H * W;
  if (in_dtype == VT_F32)
    hipLaunchKernelGGL(ndhwc_to_ncthw_kernel<float>, dim3(grid_for(n)), dim3(kBlock), 0, stream, (const float*)x, y, B,
                       C, T, H, W, ldx, ttrim);
  else if (in_dtype == VT_BF16)
    hipLaunchKernelGGL(ndhwc_to_ncthw_kernel<bf16_t>, dim3(grid_for(n)), dim3(kBlock), 0, stream, (const bf16_t*)x, y,
                       B, C, T, H, W, ldx, ttrim);
  else if (in_dtype == VT_F16)
    hipLaunchKernelGGL(ndhwc_to_ncthw_kernel<f16_t>, dim3(grid_for(n)), dim3(kBlock), 0, stream, (const f16_t*)x, y,
                       B, C, T, H, W, ldx, ttrim);
  else
    VT_CHECK_ARG(false, "vt_ndhwc_to_ncthw: in_dtype %d", in_dtype);
  VT_CHECK_LAUNCH();
  return VT_OK;
}

extern "C" int vt_time_avgpool3s2(const void* x, const void* cache, void* y, int dtype, int32_t B, int32_t Ti,
                                  int64_t HW, int32_t C, int32_t tmode, vt_stream stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  // To = Ti / 2 (floor): an odd Ti drops its last frame, as avg_pool3d(kernel 3, stride 2) over Ti + 1 padded frames does
  VT_CHECK_ARG(x && y && B > 0 && Ti >= 2 && HW > 0 && C > 0, "vt_time_avgpool3s2: bad dims (Ti=%d)", Ti);
  VT_CHECK_ARG((HW * C) % 4 == 0, "vt_time_avgpool3s2: frame size must be a multiple of 4 elements");
  VT_CHECK_ARG(tmode >= VT_TPAD_ZERO && tmode <= VT_TPAD_ZERO_BACK, "vt_time_avgpool3s2: tmode %d", tmode);
  VT_CHECK_ARG(tmode != VT_TPAD_CACHE || cache != nullptr, "vt_time_avgpool3s2: cache mode without cache");
  const long long F4 = HW * C / 4;
  const long long n = (long long)B * (Ti / 2) * F4;
  if (dtype == VT_F32)
    hipLaunchKernelGGL(time_avgpool3s2_kernel<float>, dim3(grid_for(n)), dim3(kBlock), 0, stream, (const float*)x,
                       (const float*)cache, (float*)y, B, Ti, F4, tmode);
  else if (dtype == VT_BF16)
    hipLaunchKernelGGL(time_avgpool3s2_kernel<bf16_t>, dim3(grid_for(n)), dim3(kBlock), 0, stream, (const bf16_t*)x,
                       (const bf16_t*)cache, (bf16_t*)y, B, Ti, F4, tmode);
  else if (dtype == VT_F16)
    hipLaunchKernelGGL(time_avgpool3s2_kernel<f16_t>, dim3(grid_for(n)), dim3(kBlock), 0, stream, (const f16_t*)x,
                       (const f16_t*)cache, (f16_t*)y, B, Ti, F4, tmode);
  else
    VT_CHECK_ARG(false, "vt_time_avgpool3s2: dtype %d", dtype);
  VT_CHECK_LAUNCH();
  return VT_OK;
}

extern "C" int vt_time_lerp2x_cat(const void* head, int32_t nh, const void* x, void* y, int dtype, int32_t B, int32_t Tx, int32_t skip,
                                  int64_t HWC, vt_stream stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  VT_CHECK_ARG(x && y && B > 0 && Tx > 0 && HWC > 0 && HWC % 4 == 0, "vt_time_lerp2x: bad dims");
  VT_CHECK_ARG(nh >= 0 && (nh == 0 || head != nullptr) && skip >= 0 && skip < 2 * (nh + Tx), "vt_time_lerp2x_cat: head frames %d, skip %d of %d", nh, skip, 2 * (nh + Tx));
  const long long F4 = HWC / 4;
  const int To = 2 * (nh + Tx) - skip;
  VT_CHECK_ARG((long long)B * To <= 65535, "vt_time_lerp2x: B * output frames = %lld > 65 535", (long long)B * To);
  const long long per = (long long)kBlock * (dtype == VT_F32 ? 1 : 2);                  // quads a workgroup moves per sweep of a frame
  long long gx = (F4 + per - 1) / per;
  if (gx > 1024) gx = 1024;
  const dim3 grid((unsigned)gx, (unsigned)(B * To));
  if (dtype == VT_F32)
    hipLaunchKernelGGL(time_lerp2x_kernel<float>, grid, dim3(kBlock), 0, stream, (const float*)head, nh, (const float*)x, (float*)y, B, Tx, skip, F4);
  else if (dtype == VT_BF16)
    hipLaunchKernelGGL(time_lerp2x_kernel<bf16_t>, grid, dim3(kBlock), 0, stream, (const bf16_t*)head, nh, (const bf16_t*)x, (bf16_t*)y, B, Tx, skip, F4);
  else if (dtype == VT_F16)
    hipLaunchKernelGGL(time_lerp2x_kernel<f16_t>, grid, dim3(kBlock), 0, stream, (const f16_t*)head, nh, (const f16_t*)x, (f16_t*)y, B, Tx, skip, F4);
  else
    VT_CHECK_ARG(false, "vt_time_lerp2x: dtype %d", dtype);
  VT_CHECK_LAUNCH();
  return VT_OK;
}

extern "C" int vt_time_lerp2x(const void* x, void* y, int dtype, int32_t B, int32_t Ti, int64_t HWC,
                              vt_stream stream_) {
  return vt_time_lerp2x_cat(nullptr, 0, x, y, dtype, B, Ti, 0, HWC, stream_);
}

extern "C" int vt_gather_frames(const void* src, void* dst, int32_t esize, int32_t B, int64_t frame_elems,
                                int64_t src_bstride, int64_t dst_bstride, const int32_t* idx_host, int32_t n,
                                vt_stream stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  VT_CHECK_ARG(src && dst && idx_host && B > 0 && n > 0 && n <= 128 && frame_elems > 0,
               "vt_gather_frames: bad arguments (n=%d)", n);
  VT_CHECK_ARG(esize == 2 || esize == 4, "vt_gather_frames: esize %d", esize);
  const bool v16 = (frame_elems * esize) % 16 == 0 && (src_bstride * esize) % 16 == 0 && (dst_bstride * esize) % 16 == 0 &&
                   ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
  VT_CHECK_ARG(v16 || ((frame_elems * esize) % 4 == 0 && (src_bstride * esize) % 4 == 0 && (dst_bstride * esize) % 4 == 0 &&
                       ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 3) == 0),
               "vt_gather_frames: frames, strides and pointers must be multiples of 4 bytes");
  GatherIdx gi;
  for (int j = 0; j < n; ++j) {
    VT_CHECK_ARG(idx_host[j] >= 0, "vt_gather_frames: negative index");
    gi.idx[j] = idx_host[j];
  }
  const int unit = v16 ? 16 : 4;
  const long long F16 = frame_elems * esize / unit;
  const long long total = (long long)B * n * F16;
  if (v16)
    hipLaunchKernelGGL(gather_frames_kernel<u32x4>, dim3(grid_for(total)), dim3(kBlock), 0, stream, (const u32x4*)src,
                       (u32x4*)dst, B, F16, (long long)(src_bstride * esize / 16), (long long)(dst_bstride * esize / 16),
                       gi, n);
  else
    hipLaunchKernelGGL(gather_frames_kernel<uint32_t>, dim3(grid_for(total)), dim3(kBlock), 0, stream, (const uint32_t*)src,
                       (uint32_t*)dst, B, F16, (long long)(src_bstride * esize / 4), (long long)(dst_bstride * esize / 4),
                       gi, n);
  VT_CHECK_LAUNCH();
  return VT_OK;
}
