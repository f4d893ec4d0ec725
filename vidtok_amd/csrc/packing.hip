// vt_pack_conv_weight -- reference parameter layout -> the rows vt_conv reads, on the device (include/vidtok_amd.h).
//
// The reference stores nn.Conv3d / Conv2d / Conv1d weights as [Cout][Cin][taps] fp32 (taps = kT*kH*kW row-major,
// SURVEY.md section 8b).  vt_conv wants, per output channel, k = tap * cin_p + c contiguous in the arithmetic type, Cin
// zero-padded to the activation's stored channel count, and -- for the parity classes of the up-samplers -- taps that are SUMS
// of reference taps (vidtok_amd/packing.py::{time,space}_upsample_parity_weights).  One thread per output element; the sums are
// plain fp32 additions in a fixed tree order, (t0 + t1) + (t2 + t3) with absent terms left out, which is the order the host
// statements produce (rows first, then columns), so the packed bits equal theirs.  Data movement + roundings: one-time work per
// weight version, but done here the step's kernel table holds no foreign kernel.
#include "common.h"

namespace {

constexpr int kMaxTaps = 64;

struct PackArgs {
  const float* w;
  void* out;
  int Cout, Cin, cin_p, taps_in, taps_out;
  long long ldw;          // k-values per packed row (>= taps_out * cin_p; the tail is zero)
  int mode;               // VT_F32, VT_BF16, VT_F16, VT_BF16X3
  int mix[kMaxTaps][4];   // source taps of output tap j, -1 = absent
};

__global__ __launch_bounds__(256) void pack_conv_weight_kernel(const PackArgs p) {
  const long long n = (long long)p.Cout * p.ldw;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long co = i / p.ldw;
    const int k = (int)(i - co * p.ldw);
    const int j = k / p.cin_p, ci = k - j * p.cin_p;
    float v = 0.0f;
    if (j < p.taps_out && ci < p.Cin) {
      const float* src = p.w + (co * p.Cin + ci) * p.taps_in;
      const int t0 = p.mix[j][0], t1 = p.mix[j][1], t2 = p.mix[j][2], t3 = p.mix[j][3];
      float a = src[t0];
      if (t1 >= 0) a = __fadd_rn(a, src[t1]);
      if (t2 >= 0) {
        float b = src[t2];
        if (t3 >= 0) b = __fadd_rn(b, src[t3]);
        a = __fadd_rn(a, b);
      }
      v = a;
    }
    if (p.mode == VT_F32) {
      reinterpret_cast<float*>(p.out)[i] = v;
    } else if (p.mode == VT_BF16) {
      reinterpret_cast<uint16_t*>(p.out)[i] = (uint16_t)f32_to_bf16_bits(v);
    } else if (p.mode == VT_F16) {
      reinterpret_cast<f16_t*>(p.out)[i] = (f16_t)v;
    } else {   // split-bf16 planes: per group of 16 k, [hi x 16 | lo x 16] (packing.py::pack_split3)
      const uint32_t hi = f32_to_bf16_bits(v);
      const uint32_t lo = f32_to_bf16_bits(__fsub_rn(v, bf16_bits_to_f32(hi)));
      uint16_t* row = reinterpret_cast<uint16_t*>(p.out) + co * p.ldw * 2;
      const int g = k >> 4, e = k & 15;
      row[g * 32 + e] = (uint16_t)hi;
      row[g * 32 + 16 + e] = (uint16_t)lo;
    }
  }
}

}  // namespace

extern "C" int vt_pack_conv_weight(const float* w, void* out, int32_t out_dtype, int32_t Cout, int32_t Cin, int32_t cin_p, int32_t taps_in,
                                   int32_t taps_out, const int32_t* mix_host, int64_t ldw, vt_stream stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  VT_CHECK_ARG(w && out && Cout > 0 && Cin > 0 && cin_p >= Cin && taps_in > 0 && taps_out > 0, "vt_pack_conv_weight: bad arguments");
  VT_CHECK_ARG(out_dtype == VT_F32 || out_dtype == VT_BF16 || out_dtype == VT_F16 || out_dtype == VT_BF16X3, "vt_pack_conv_weight: out_dtype %d", out_dtype);
  VT_CHECK_ARG(taps_out <= kMaxTaps && (mix_host != nullptr || taps_out == taps_in), "vt_pack_conv_weight: at most %d taps; without a mix table taps_out = taps_in", kMaxTaps);
  VT_CHECK_ARG(ldw >= (int64_t)taps_out * cin_p && (out_dtype != VT_BF16X3 || ldw % 32 == 0), "vt_pack_conv_weight: ldw %lld (split-bf16 rows: a multiple of 32)", (long long)ldw);
  PackArgs p;
  p.w = w; p.out = out; p.Cout = Cout; p.Cin = Cin; p.cin_p = cin_p; p.taps_in = taps_in; p.taps_out = taps_out; p.ldw = ldw; p.mode = out_dtype;
  for (int j = 0; j < kMaxTaps; ++j)
    for (int q = 0; q < 4; ++q) p.mix[j][q] = -1;
  for (int j = 0; j < taps_out; ++j) {
    if (!mix_host) {
      p.mix[j][0] = j;
      continue;
    }
    for (int q = 0; q < 4; ++q) {
      const int t = mix_host[j * 4 + q];
      VT_CHECK_ARG(t >= -1 && t < taps_in, "vt_pack_conv_weight: mix[%d][%d] = %d", j, q, t);
      p.mix[j][q] = t;
    }
    VT_CHECK_ARG(p.mix[j][0] >= 0 && !(p.mix[j][3] >= 0 && p.mix[j][2] < 0), "vt_pack_conv_weight: mix[%d] needs its first term (and term 2 before term 3)", j);
  }
  const long long n = (long long)Cout * ldw;
  const unsigned grid = (unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(grid), dim3(256), 0, stream, p);
  VT_CHECK_LAUNCH();
  return VT_OK;
}
