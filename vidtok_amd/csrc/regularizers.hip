// KL (diagonal Gaussian) and FSQ regularizer kernels -- contracts in include/vidtok_amd.h.
// All tensors here are the small NCTHW fp32 latents of the reference API ([B][C][S], S = T*H*W).
#include <math.h>

#include "common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxD = 8;
constexpr int kMaxL = 16;

// ---- KL --------------------------------------------------------------------------------------
// DiagonalGaussianDistribution (reference distributions.py:6-28) + DiagonalGaussianRegularizer
// (regularizers.py:82-92).  One workgroup, fixed summation order => deterministic kl.
__global__ __launch_bounds__(1024) void kl_sample_kernel(const float* __restrict__ h, const float* __restrict__ noise,
                                                         float* __restrict__ z, float* __restrict__ kl_out, int B,
                                                         int zc, long long S) {
  __shared__ float red[16];
  const long long per_b = (long long)zc * S;
  const long long n = (long long)B * per_b;
  float acc = 0.f;
  for (long long i = threadIdx.x; i < n; i += 1024) {
    const long long b = i / per_b;
    const long long r = i - b * per_b;  // c*S + s
    const float mean = h[b * 2 * per_b + r];
    float logvar = h[b * 2 * per_b + per_b + r];
    logvar = fminf(fmaxf(logvar, -30.0f), 20.0f);
    const float stdv = expf(0.5f * logvar);
    const float var = expf(logvar);
    z[i] = noise ? mean + stdv * noise[i] : mean;
    acc += mean * mean + var - 1.0f - logvar;
  }
  acc = wave_sum(acc, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x < 64) {
    float v = threadIdx.x < 16 ? red[threadIdx.x] : 0.f;
    v = wave_sum(v, 64);
    if (threadIdx.x == 0) kl_out[0] = 0.5f * v / (float)B;
  }
}

// ---- FSQ -------------------------------------------------------------------------------------
struct FsqConsts {
  int D;
  int levels[kMaxD];
  int basis[kMaxD];
  float half_l[kMaxD];   // (L-1)*(1+eps)/2          regularizers.py:155
  float offset[kMaxD];   // 0.5 for even L           regularizers.py:156
  float shift[kMaxD];    // atanh(offset/half_l)     regularizers.py:157
  float half_w[kMaxD];   // L//2                     regularizers.py:163
};

// tanh rounded from a double evaluation: within 0.5 ulp of the exact value, so it differs from
// the host libm / Sleef result the reference uses by at most one fp32 ulp.
__device__ __forceinline__ float tanh_cr(float x) { return (float)tanh((double)x); }

// bound -> round half to even -> integer level (the value `quantized` of regularizers.py:160-164)
__device__ __forceinline__ float fsq_round(float zv, float shift, float half_l, float offset) {
  const float t = tanh_cr(__fadd_rn(zv, shift));
  const float bounded = __fsub_rn(__fmul_rn(t, half_l), offset);  // separate roundings, like torch
  return rintf(bounded);
}

// `ncb` codebooks (FSQRegularizer num_codebooks, "b n (c d) -> b n c d", regularizers.py:227): on the NCTHW latent the d channels
// of codebook c of clip b are contiguous -- the kernels see B = clips x codebooks "clips" of D channels -- and the reference keeps
// the codebook axis LAST on the indices: index of (clip b, codebook c, position s) sits at [b][s][c].
__global__ __launch_bounds__(kBlock) void fsq_quantize_kernel(const float* __restrict__ h, float* __restrict__ z,
                                                              int* __restrict__ indices, FsqConsts k, int B,
                                                              long long S, int ncb) {
  const long long n = (long long)B * S;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
    const long long b = i / S, s = i - b * S;
    int idx = 0;
    for (int d = 0; d < k.D; ++d) {
      const float q = fsq_round(h[(b * k.D + d) * S + s], k.shift[d], k.half_l[d], k.offset[d]);
      z[(b * k.D + d) * S + s] = q / k.half_w[d];
      // codes_to_indices (regularizers.py:174-178): (code*half_w + half_w)*basis summed in fp32 and
      // truncated.  Every term of that sum is an exact small integer (q/hw*hw rounds back to q), so
      // integer arithmetic gives the reference's value without depending on FMA contraction.
      idx += ((int)q + k.levels[d] / 2) * k.basis[d];
    }
    indices[ncb == 1 ? i : ((b / ncb) * S + s) * ncb + (b % ncb)] = idx;
  }
}

__global__ __launch_bounds__(kBlock) void fsq_indices_to_codes_kernel(const int* __restrict__ indices,
                                                                      float* __restrict__ z, FsqConsts k, int B,
                                                                      long long S, int ncb) {
  const long long n = (long long)B * S;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
    const long long b = i / S, s = i - b * S;
    const int idx = indices[ncb == 1 ? i : ((b / ncb) * S + s) * ncb + (b % ncb)];
    for (int d = 0; d < k.D; ++d) {
      const int lv = (idx / k.basis[d]) % k.levels[d];                  // regularizers.py:186
      z[(b * k.D + d) * S + s] = ((float)lv - k.half_w[d]) / k.half_w[d];  // _scale_and_shift_inverse :170-172
    }
  }
}

// aux statistics, stage 1: per token the per-dimension softmax tables p[d][l] of
// softmax_j(2*inv_temp * <z, c_j>) -- the implicit codebook is a product grid, so the softmax
// over all prod(L) codes factorises exactly into per-dimension softmaxes -- and the commitment
// partial sum.  work layout: [Ntok][D][kMaxL] tables, then 3 accumulators.
__global__ __launch_bounds__(kBlock) void fsq_aux_tables_kernel(const float* __restrict__ h, FsqConsts k, int B,
                                                                long long S, float inv_temp,
                                                                float* __restrict__ tables, float* __restrict__ accum) {
  const long long n = (long long)B * S;
  float commit = 0.f;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
    const long long b = i / S, s = i - b * S;
    for (int d = 0; d < k.D; ++d) {
      const float zv = h[(b * k.D + d) * S + s];
      const float q = fsq_round(zv, k.shift[d], k.half_l[d], k.offset[d]);
      const float code = q / k.half_w[d];
      commit += (zv - code) * (zv - code);
      float* t = tables + (i * k.D + d) * kMaxL;
      const int L = k.levels[d];
      float mx = -INFINITY;
      for (int l = 0; l < L; ++l) {
        const float c = ((float)l - k.half_w[d]) / k.half_w[d];
        const float logit = 2.0f * inv_temp * zv * c;
        t[l] = logit;
        mx = fmaxf(mx, logit);
      }
      float sum = 0.f;
      for (int l = 0; l < L; ++l) {
        const float e = expf(t[l] - mx);
        t[l] = e;
        sum += e;
      }
      const float inv = 1.0f / sum;
      for (int l = 0; l < L; ++l) t[l] *= inv;
    }
  }
  commit = wave_sum(commit, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(accum + 2, commit);
}

// stage 2: the product-codebook probabilities p[t][j] = prod_d table[t][d][digit_d(j)] for all tokens x all codes
// (20 480 x 32 768 = 671 M pairs for BASELINE configs[2]) -- needed pair by pair because of the reference's log clamp
// (regularizers.py:40-45).  The code index splits into LOW digits (dims < dl, owned by threads: JL = prod <= 512) and
// HIGH digits (looped in chunks of kHiChunk): a thread keeps base = prod of its low-digit table entries per token and
// multiplies by the high-digit products staged once per token in LDS, so a pair costs one multiply, one add and the
// entropy term.  A workgroup owns (token tile, high chunk); the per-tile sums of p go to part[tile][j] (no atomics:
// deterministic), the entropy partials to one atomic per wave.  (The first version walked all tokens per code from
// L2: 20 ms per call at B = 4, a quarter of an FSQ forward.)
constexpr int kTokTiles = 64;     // token tiles (rows of `part`)
constexpr int kHiChunk = 16;      // high-digit combinations per workgroup
constexpr int kTokSub = 32;       // tokens staged in LDS at a time

struct FsqSplit {
  int dl;        // dims [0, dl) are thread digits, [dl, D) loop digits
  int JL, JH;
};
inline FsqSplit fsq_split(const FsqConsts& k) {
  FsqSplit s;
  s.dl = 0;
  s.JL = 1;
  while (s.dl < k.D - 1 && s.JL * k.levels[s.dl] <= 512) s.JL *= k.levels[s.dl++];
  if (s.dl == 0) s.JL = k.levels[s.dl++];      // a single huge first level still becomes the thread digit
  s.JH = 1;
  for (int d = s.dl; d < k.D; ++d) s.JH *= k.levels[d];
  return s;
}

__global__ __launch_bounds__(kBlock) void fsq_aux_pairs_kernel(const float* __restrict__ tables, FsqConsts k, FsqSplit sp,
                                                               long long ntok, long long tile_tokens, int J,
                                                               float* __restrict__ part, float* __restrict__ accum) {
  __shared__ float tab[kTokSub][kMaxD][kMaxL];
  __shared__ float hi[kTokSub][kHiChunk];
  const int tile = blockIdx.x;
  const int hc0 = blockIdx.y * kHiChunk;
  const long long t_begin = (long long)tile * tile_tokens;
  const long long t_end = min(t_begin + tile_tokens, ntok);
  constexpr int kLowPerThread = 2;              // JL <= 512 = 2 x kBlock
  int lo_off[kLowPerThread][kMaxD];
  bool lo_ok[kLowPerThread];
#pragma unroll
  for (int u = 0; u < kLowPerThread; ++u) {
    int jl = threadIdx.x + u * kBlock;
    lo_ok[u] = jl < sp.JL;
#pragma unroll
    for (int d = 0; d < kMaxD; ++d) {          // static indices: the array stays in registers
      const int L = d < sp.dl ? k.levels[d] : 1;
      lo_off[u][d] = lo_ok[u] ? jl % L : 0;
      jl /= L;
    }
  }
  float acc[kLowPerThread][kHiChunk];
#pragma unroll
  for (int u = 0; u < kLowPerThread; ++u)
#pragma unroll
    for (int h = 0; h < kHiChunk; ++h) acc[u][h] = 0.f;
  float ent = 0.f;
  for (long long t0 = t_begin; t0 < t_end; t0 += kTokSub) {
    const int nt = (int)min((long long)kTokSub, t_end - t0);
    __syncthreads();
    for (int i = threadIdx.x; i < nt * k.D * kMaxL; i += kBlock) {
      const int tt = i / (k.D * kMaxL), r = i - tt * (k.D * kMaxL);
      tab[tt][r / kMaxL][r % kMaxL] = tables[(t0 + tt) * k.D * kMaxL + r];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nt * kHiChunk; i += kBlock) {
      const int tt = i / kHiChunk, h = i - tt * kHiChunk;
      int jh = hc0 + h;
      float pr = jh < sp.JH ? 1.0f : 0.0f;
      for (int d = sp.dl; d < k.D; ++d) {
        pr *= tab[tt][d][jh % k.levels[d]];
        jh /= k.levels[d];
      }
      hi[tt][h] = pr;
    }
    __syncthreads();
    for (int tt = 0; tt < nt; ++tt) {
#pragma unroll
      for (int u = 0; u < kLowPerThread; ++u) {
        if (!lo_ok[u]) continue;
        float base = 1.0f;
#pragma unroll
        for (int d = 0; d < kMaxD; ++d)
          if (d < sp.dl) base *= tab[tt][d][lo_off[u][d]];
#pragma unroll
        for (int h = 0; h < kHiChunk; ++h) {
          const float pv = base * hi[tt][h];
          acc[u][h] += pv;
          ent -= pv * logf(fmaxf(pv, 1e-5f));
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < kLowPerThread; ++u) {
    if (!lo_ok[u]) continue;
    const int jl = threadIdx.x + u * kBlock;
#pragma unroll
    for (int h = 0; h < kHiChunk; ++h) {
      const int jh = hc0 + h;
      if (jh < sp.JH) part[(long long)tile * J + (long long)jh * sp.JL + jl] = acc[u][h];
    }
  }
  ent = wave_sum(ent, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(accum + 0, ent);
}

// stage 3: avg[j] = sum over token tiles / ntok, its entropy term, optional copy out
__global__ __launch_bounds__(kBlock) void fsq_aux_reduce_kernel(const float* __restrict__ part, int ntiles, int J, long long ntok,
                                                                float* __restrict__ accum, float* __restrict__ avg_out) {
  const int j = blockIdx.x * kBlock + threadIdx.x;
  float cbe = 0.f;
  if (j < J) {
    float a = 0.f;
    for (int t = 0; t < ntiles; ++t) a += part[(long long)t * J + j];
    a /= (float)ntok;
    if (avg_out) avg_out[j] = a;   // batch-mean code distribution: what the reference all-reduces across ranks
    cbe = -a * logf(fmaxf(a, 1e-5f));
  }
  cbe = wave_sum(cbe, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(accum + 1, cbe);
}

// entropy(avg) with the reference's log clamp (regularizers.py:40-45) of a J-entry distribution: the codebook entropy
// recomputed after the cross-rank mean of avg_prob (maybe_distributed_mean, regularizers.py:49-59,240)
__global__ __launch_bounds__(kBlock) void entropy_kernel(const float* __restrict__ avg, long long J, float* __restrict__ out) {
  __shared__ float part[kBlock / 64];
  float e = 0.f;
  for (long long j = threadIdx.x; j < J; j += kBlock) {
    const float a = avg[j];
    e -= a * logf(fmaxf(a, 1e-5f));
  }
  e = wave_sum(e, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = e;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < kBlock / 64; ++w) t += part[w];
    out[0] = t;
  }
}

__global__ void fsq_aux_finish_kernel(const float* __restrict__ accum, long long ntok, long long nelem,
                                      float* __restrict__ out3) {
  out3[0] = accum[0] / (float)ntok;   // per-sample entropy, mean over tokens (num_codebooks = 1)
  out3[1] = accum[1];                 // entropy of the batch-mean distribution
  out3[2] = accum[2] / (float)nelem;  // F.mse_loss(..., "none").mean()
}

int make_consts(const int32_t* levels, int D, FsqConsts* k) {
  VT_CHECK_ARG(levels != nullptr && D >= 1 && D <= kMaxD, "fsq: D=%d out of range [1,%d]", D, kMaxD);
  k->D = D;
  int basis = 1;
  for (int d = 0; d < D; ++d) {
    VT_CHECK_ARG(levels[d] >= 2 && levels[d] <= kMaxL, "fsq: level %d out of range [2,%d]", levels[d], kMaxL);
    k->levels[d] = levels[d];
    k->basis[d] = basis;
    basis *= levels[d];
    // fp32 operation order of FSQRegularizer.bound (regularizers.py:153-158)
    const float half_l = (float)(levels[d] - 1) * (float)(1.0 + 1e-3) / 2.0f;
    const float offset = (levels[d] % 2 == 0) ? 0.5f : 0.0f;
    k->half_l[d] = half_l;
    k->offset[d] = offset;
    k->shift[d] = atanhf(offset / half_l);
    k->half_w[d] = (float)(levels[d] / 2);
  }
  return VT_OK;
}

inline unsigned grid_for(long long n) {
  long long b = (n + kBlock - 1) / kBlock;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" int vt_fsq_consts(const int32_t* levels_host, int32_t D, float* out_host) {
  VT_CHECK_ARG(out_host != nullptr, "vt_fsq_consts: null output");
  FsqConsts k;
  int rc = make_consts(levels_host, D, &k);
  if (rc != VT_OK) return rc;
  for (int d = 0; d < D; ++d) {
    out_host[d] = k.half_l[d];
    out_host[D + d] = k.offset[d];
    out_host[2 * D + d] = k.shift[d];
    out_host[3 * D + d] = (float)k.basis[d];
  }
  return VT_OK;
}

extern "C" int vt_kl_sample(const float* h, const float* noise, float* z, float* kl_out, int32_t B, int32_t zc,
                            int64_t S, vt_stream stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  VT_CHECK_ARG(h && z && kl_out && B > 0 && zc > 0 && S > 0, "vt_kl_sample: bad arguments");
  hipLaunchKernelGGL(kl_sample_kernel, dim3(1), dim3(1024), 0, stream, h, noise, z, kl_out, B, zc, (long long)S);
  VT_CHECK_LAUNCH();
  return VT_OK;
}

extern "C" int vt_fsq_quantize_cb(const float* h, float* z, int32_t* indices, const int32_t* levels_host, int32_t D,
                                  int32_t B, int32_t ncb, int64_t S, vt_stream stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  VT_CHECK_ARG(h && z && indices && B > 0 && S > 0 && ncb >= 1, "vt_fsq_quantize: bad arguments");
  FsqConsts k;
  int rc = make_consts(levels_host, D, &k);
  if (rc != VT_OK) return rc;
  hipLaunchKernelGGL(fsq_quantize_kernel, dim3(grid_for((long long)B * ncb * S)), dim3(kBlock), 0, stream, h, z, indices, k,
                     B * ncb, (long long)S, ncb);
  VT_CHECK_LAUNCH();
  return VT_OK;
}

extern "C" int vt_fsq_quantize(const float* h, float* z, int32_t* indices, const int32_t* levels_host, int32_t D,
                               int32_t B, int64_t S, vt_stream stream_) {
  return vt_fsq_quantize_cb(h, z, indices, levels_host, D, B, 1, S, stream_);
}

// y[b][o][s] = bias[o] + sum_i w[o][i] * x[b][i][s]  on NCTHW-flattened [B][C][S] fp32 tensors: nn.Linear along the
// channel axis -- FSQ's project_in / project_out when dim != len(levels) (regularizers.py:137-139, 225, 255).
// A few thousand positions x <= 64 channels: one thread per position, weights through the scalar cache.
__global__ __launch_bounds__(kBlock) void channel_linear_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ y,
                                                               int B, int Cin, int Cout, long long S) {
  const long long n = (long long)B * S;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
    const long long b = i / S, sp = i - b * S;
    const float* xb = x + b * Cin * S + sp;
    float* yb = y + b * Cout * S + sp;
    for (int o = 0; o < Cout; ++o) {
      float acc = bias ? bias[o] : 0.0f;
      for (int c = 0; c < Cin; ++c) acc = fmaf(w[o * Cin + c], xb[(long long)c * S], acc);
      yb[(long long)o * S] = acc;
    }
  }
}

extern "C" int vt_channel_linear(const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t Cin,
                                 int32_t Cout, int64_t S, vt_stream stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  VT_CHECK_ARG(x && w && y && B > 0 && Cin > 0 && Cout > 0 && S > 0 && Cin <= 1024 && Cout <= 1024,
               "vt_channel_linear: bad arguments");
  hipLaunchKernelGGL(channel_linear_kernel, dim3(grid_for((long long)B * S)), dim3(kBlock), 0, stream, x, w, bias, y, B,
                     Cin, Cout, (long long)S);
  VT_CHECK_LAUNCH();
  return VT_OK;
}

extern "C" int vt_fsq_indices_to_codes_cb(const int32_t* indices, float* z, const int32_t* levels_host, int32_t D,
                                          int32_t B, int32_t ncb, int64_t S, vt_stream stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  VT_CHECK_ARG(indices && z && B > 0 && S > 0 && ncb >= 1, "vt_fsq_indices_to_codes: bad arguments");
  FsqConsts k;
  int rc = make_consts(levels_host, D, &k);
  if (rc != VT_OK) return rc;
  hipLaunchKernelGGL(fsq_indices_to_codes_kernel, dim3(grid_for((long long)B * ncb * S)), dim3(kBlock), 0, stream, indices,
                     z, k, B * ncb, (long long)S, ncb);
  VT_CHECK_LAUNCH();
  return VT_OK;
}

extern "C" int vt_fsq_indices_to_codes(const int32_t* indices, float* z, const int32_t* levels_host, int32_t D,
                                       int32_t B, int64_t S, vt_stream stream_) {
  return vt_fsq_indices_to_codes_cb(indices, z, levels_host, D, B, 1, S, stream_);
}

extern "C" int64_t vt_fsq_aux_work_floats(const int32_t* levels_host, int32_t D, int32_t B, int64_t S) {
  int64_t J = 1;
  for (int d = 0; levels_host && d < D; ++d) J *= levels_host[d];
  return (int64_t)B * S * D * kMaxL + 4 + (int64_t)kTokTiles * J;   // per-token tables, accumulators, per-tile code sums
}

extern "C" int vt_fsq_aux_stats(const float* h, const int32_t* levels_host, int32_t D, int32_t B, int64_t S,
                                float inv_temperature, float* work, float* out3, vt_stream stream_) {
  return vt_fsq_aux_stats_avg(h, levels_host, D, B, S, inv_temperature, work, out3, nullptr, stream_);
}

// aux = (st[0] - gamma * codebook_entropy) * w_entropy + st[2] * w_commit: the last lines of FSQRegularizer.forward
// (regularizers.py:241,264-266) on the three statistics of vt_fsq_aux_stats; codebook_entropy = st[1] unless the caller
// passes the entropy of a cross-rank averaged distribution.  Each product / sum rounded on its own, as the host statement's
// separate tensor operations are.
__global__ void fsq_aux_loss_kernel(const float* __restrict__ st, const float* __restrict__ cb, float gamma, float w_ent,
                                    float w_commit, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float ce = cb ? cb[0] : st[1];
    const float ent = __fsub_rn(st[0], __fmul_rn(gamma, ce));
    out[0] = __fadd_rn(__fmul_rn(ent, w_ent), __fmul_rn(st[2], w_commit));
  }
}

extern "C" int vt_fsq_aux_loss(const float* stats3, const float* codebook_entropy, float diversity_gamma, float entropy_weight,
                               float commitment_weight, float* out, vt_stream stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  VT_CHECK_ARG(stats3 && out, "vt_fsq_aux_loss: null argument");
  hipLaunchKernelGGL(fsq_aux_loss_kernel, dim3(1), dim3(64), 0, stream, stats3, codebook_entropy, diversity_gamma, entropy_weight,
                     commitment_weight, out);
  VT_CHECK_LAUNCH();
  return VT_OK;
}

extern "C" int vt_entropy(const float* avg, int64_t J, float* out, vt_stream stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  VT_CHECK_ARG(avg && out && J > 0, "vt_entropy: bad arguments");
  hipLaunchKernelGGL(entropy_kernel, dim3(1), dim3(kBlock), 0, stream, avg, (long long)J, out);
  VT_CHECK_LAUNCH();
  return VT_OK;
}

extern "C" int vt_fsq_aux_stats_avg(const float* h, const int32_t* levels_host, int32_t D, int32_t B, int64_t S,
                                    float inv_temperature, float* work, float* out3, float* avg_out, vt_stream stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  VT_CHECK_ARG(h && work && out3 && B > 0 && S > 0, "vt_fsq_aux_stats: bad arguments");
  FsqConsts k;
  int rc = make_consts(levels_host, D, &k);
  if (rc != VT_OK) return rc;
  long long J = 1;
  for (int d = 0; d < D; ++d) J *= levels_host[d];
  VT_CHECK_ARG(J < (1ll << 30), "vt_fsq_aux_stats: codebook too large");
  const long long ntok = (long long)B * S;
  float* tables = work;
  float* accum = work + ntok * D * kMaxL;
  VT_CHECK_HIP(hipMemsetAsync(accum, 0, 4 * sizeof(float), stream));
  hipLaunchKernelGGL(fsq_aux_tables_kernel, dim3(grid_for(ntok)), dim3(kBlock), 0, stream, h, k, B, (long long)S,
                     inv_temperature, tables, accum);
  VT_CHECK_LAUNCH();
  float* part = accum + 4;
  const FsqSplit sp = fsq_split(k);
  VT_CHECK_ARG(sp.JL <= 2 * kBlock, "vt_fsq_aux_stats: first level %d too large", levels_host[0]);
  const long long tile_tokens = (ntok + kTokTiles - 1) / kTokTiles;
  const int ntiles = (int)((ntok + tile_tokens - 1) / tile_tokens);
  hipLaunchKernelGGL(fsq_aux_pairs_kernel, dim3((unsigned)ntiles, (unsigned)((sp.JH + kHiChunk - 1) / kHiChunk)), dim3(kBlock), 0, stream,
                     (const float*)tables, k, sp, ntok, tile_tokens, (int)J, part, accum);
  VT_CHECK_LAUNCH();
  hipLaunchKernelGGL(fsq_aux_reduce_kernel, dim3((unsigned)((J + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, (const float*)part,
                     ntiles, (int)J, ntok, accum, avg_out);
  VT_CHECK_LAUNCH();
  hipLaunchKernelGGL(fsq_aux_finish_kernel, dim3(1), dim3(1), 0, stream, (const float*)accum, ntok,
                     ntok * (long long)D, out3);
  VT_CHECK_LAUNCH();
  return VT_OK;
}
