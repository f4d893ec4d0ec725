// vt_conv: argument validation, kernel selection and the split-K path of the implicit-GEMM convolution (include/vidtok_amd.h).
// The kernel itself is conv_igemm_kernel.h, instantiated per arithmetic in conv_igemm_{f32,bf16,f16,x3}.hip; the special-shape kernels
// it hands over to live in conv_ws2.hip (3 x 3, 128 -> 128), conv_in8.hip (the encoder's conv_in) and conv_narrow.hip (Cout <= 4).
#include <algorithm>
#include <atomic>
#include <type_traits>

#include "conv_select.h"

namespace {

// ---- split-K over the time taps (include/vidtok_amd.h, vt_conv_work_bytes) ------------------------------------------------------------
// y[m][n] = H(((p0 + p1) + p2) + bias[n]  [+ res[m][n]]) (H = bf16 / fp16) from the KT fp32 partials [KT][M][Cout]; a thread = 8 channels of a pixel
template <typename H>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int planes, long long M, int Cout, const float* __restrict__ bias,
                                                            const H* __restrict__ res, H* __restrict__ y) {
  const long long n8 = M * (Cout / 8);
  const long long plane = M * (long long)Cout;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    const long long e = i * 8;
    const int c = (int)(e % Cout);
    float v[8];
    {
      const f32x4 a = *reinterpret_cast<const f32x4*>(part + e), b = *reinterpret_cast<const f32x4*>(part + e + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[k] = a[k]; v[4 + k] = b[k]; }
    }
    for (int pz = 1; pz < planes; ++pz) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(part + pz * plane + e), b = *reinterpret_cast<const f32x4*>(part + pz * plane + e + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[k] += a[k]; v[4 + k] += b[k]; }
    }
    if (bias) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(bias + c), b = *reinterpret_cast<const f32x4*>(bias + c + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[k] += a[k]; v[4 + k] += b[k]; }
    }
    if (res) {
      Oct<H> r;
      r.load(res + e);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = r.get(k) + v[k];
    }
    Oct<H>::store(y + e, v);
  }
}

// how many tap planes vt_conv would split `a` into (0 = no split): bf16, 3 taps in time at stride 1, K long, few pixels PER CLIP,
// plain NDHWC rows, residual add at most
inline int splitk_planes(const vt_conv_desc* d, const ConvArgs& a, int nbatch, bool ln_fused, bool use_ws) {
  if (vt_opt(OPT_CONV_SPLITK) == 0 || !conv_buf()) return 0;
  if (!vt_is_h16(d->dtype) || d->out_dtype != d->dtype || nbatch != 1 || use_ws || ln_fused || a.prof != nullptr) return 0;
  const bool by_kt = a.KT == 3 && a.st == 1, by_kh = a.KT == 1 && a.KH == 3;      // three planes: the time taps, or the rows of a 3 x 3
  if (!(by_kt || by_kh) || a.ups_t || a.ups_s || a.out_layout != VT_NDHWC || a.yt_mul != 1 || a.ys_mul == 2) return 0;
  if (a.Cin % 64 != 0 || a.Cout % 128 != 0 || a.ldy != a.Cout || a.KT * a.KH * a.KW * a.Cin < 4608) return 0;
  if (a.res_mode == VT_RES_MIX || (a.res_mode == VT_RES_ADD && (a.ldr != a.Cout || a.Tr != a.To || a.res_tshift != 0))) return 0;
  if (a.tmode == VT_TPAD_CACHE && ((long long)a.Ho * a.Wo) % 256 != 0) return 0;     // the descriptor form of the cache gather: a tile inside one frame
  const unsigned long long xb = (unsigned long long)a.B * a.Ti * a.Hi * a.Wi * a.Cin * 2, wb = (unsigned long long)a.Cout * a.ldw * 2;
  if (xb >= 0xFFFF0000ull || wb >= 0xFFFF0000ull) return 0;
  const TileKind tk = select_tile(a, 1);
  if (tk != TILE_256x256 && tk != TILE_128x128) return 0;
  // The decision is a function of ONE CLIP's geometry (To, Ho, Wo, Cout, K) and never of B: a split launch sums its tap planes in
  // another order than a whole one, so a rule that looked at the launch's pixel count (round 4: "no more tiles than CUs") tied a
  // clip's bits to the batch it was part of.  A clip whose pixels make no more 128 x 128 tiles than the device has CUs splits --
  // alone it would leave every workgroup by itself on a CU walking the whole K -- whatever the batch around it.
  const long long clip_tiles = (((long long)a.To * a.Ho * a.Wo + 127) / 128) * ((a.Cout + 127) / 128);
  return clip_tiles <= device_cus() ? 3 : 0;
}

int launch_splitk(const vt_conv_desc* d, const ConvArgs& a_in, int planes, hipStream_t stream) {
  ConvArgs a = a_in;
  a.ksplit = a.KT == 3 ? 1 : 2;
  a.plane_bytes = (unsigned)((a.KT == 3 ? a.KH * a.KW : a.KW) * a.Cin * 2);
  a.y = reinterpret_cast<char*>(d->work);
  a.bias = nullptr;
  a.res = nullptr; a.res_mode = VT_RES_NONE;
  a.ln_mode = 0;
  a.xs_z = 0; a.ws_z = 0; a.rs_z = 0;
  a.ys_z = (long long)a.M * a.ldy;
  int rc = d->dtype == VT_F16 ? vt_igemm_dispatch_f16(&a, planes, 1, stream) : vt_igemm_dispatch_bf16(&a, planes, 1, stream);
  if (rc != VT_OK) return rc;
  const long long n8 = (long long)a.M * (a.Cout / 8);
  const unsigned grid = (unsigned)std::min<long long>((n8 + 255) / 256, 4096);
  const void* res = a_in.res_mode == VT_RES_ADD ? a_in.res : nullptr;
  if (d->dtype == VT_F16)
    hipLaunchKernelGGL(splitk_reduce_kernel<f16_t>, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const float*>(d->work), planes, (long long)a.M, a.Cout, a_in.bias,
                       reinterpret_cast<const f16_t*>(res), reinterpret_cast<f16_t*>(a_in.y));
  else
    hipLaunchKernelGGL(splitk_reduce_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const float*>(d->work), planes, (long long)a.M, a.Cout, a_in.bias,
                       reinterpret_cast<const bf16_t*>(res), reinterpret_cast<bf16_t*>(a_in.y));
  VT_CHECK_LAUNCH();
  return VT_OK;
}

}  // namespace

extern "C" int vt_conv_max_lds_bytes(void) { return 163840; }   // the persistent kernels take nearly all of a CU's LDS

namespace {
// argument validation + the kernel's view of the descriptor; shared by vt_conv and vt_conv_plan
int conv_prepare(const vt_conv_desc* d, ConvArgs& a, bool& ln_fused, int& nbatch_out, bool& use_ws) {
  VT_CHECK_ARG(d != nullptr, "vt_conv: null descriptor");
  VT_CHECK_ARG(d->x && d->w && d->y, "vt_conv: null tensor pointer");
  VT_CHECK_ARG(d->dtype == VT_F32 || d->dtype == VT_BF16 || d->dtype == VT_F16 || d->dtype == VT_BF16X3, "vt_conv: dtype %d", d->dtype);
  VT_CHECK_ARG((d->out_dtype == d->dtype && d->dtype != VT_BF16X3) || d->out_dtype == VT_F32, "vt_conv: out_dtype %d with dtype %d",
               d->out_dtype, d->dtype);
  const int vec = vt_is_h16(d->dtype) ? 8 : 4;
  if (d->dtype == VT_BF16X3)    // split weight planes: [hi 16 x bf16 | lo 16 x bf16] per 16 k-values, K padded to the block
    VT_CHECK_ARG(d->ldw % 32 == 0 && d->ldw >= (d->KT * d->KH * d->KW * d->Cin + 31) / 32 * 32 && d->nbatch <= 1,
                 "vt_conv: VT_BF16X3 needs ldw = K rounded up to 32 (got %d) and nbatch 1", d->ldw);
  VT_CHECK_ARG(d->B > 0 && d->Ti > 0 && d->Hi > 0 && d->Wi > 0 && d->Cin > 0, "vt_conv: bad input dims");
  VT_CHECK_ARG(d->To > 0 && d->Ho > 0 && d->Wo > 0 && d->Cout > 0, "vt_conv: bad output dims");
  VT_CHECK_ARG(d->Cin % vec == 0, "vt_conv: Cin=%d must be a multiple of %d (pad the channel dim)", d->Cin, vec);
  VT_CHECK_ARG(d->KT > 0 && d->KH > 0 && d->KW > 0 && d->KT * d->KH * d->KW <= 64, "vt_conv: bad taps");
  VT_CHECK_ARG(d->st > 0 && d->sh > 0 && d->sw > 0, "vt_conv: bad strides");
  VT_CHECK_ARG(d->ldw >= d->KT * d->KH * d->KW * d->Cin && d->ldw % vec == 0, "vt_conv: ldw=%d", d->ldw);
  VT_CHECK_ARG((reinterpret_cast<uintptr_t>(d->x) & 15) == 0 && (reinterpret_cast<uintptr_t>(d->w) & 15) == 0,
               "vt_conv: x / w must be 16-byte aligned");
  VT_CHECK_ARG(d->ups_t == 0 || d->ups_t == 1, "vt_conv: ups_t");
  VT_CHECK_ARG(d->ups_s == 0 || d->ups_s == 1, "vt_conv: ups_s");
  VT_CHECK_ARG(d->tmode >= VT_TPAD_ZERO && d->tmode <= VT_TPAD_CACHE, "vt_conv: tmode %d", d->tmode);
  if (d->tmode == VT_TPAD_CACHE && d->pt > 0) {
    VT_CHECK_ARG(d->cache != nullptr && d->ncache >= d->pt, "vt_conv: cache mode needs cache with >= pt frames");
    VT_CHECK_ARG(d->ups_t == 0, "vt_conv: cache mode with ups_t");
    VT_CHECK_ARG((reinterpret_cast<uintptr_t>(d->cache) & 15) == 0, "vt_conv: cache must be 16-byte aligned");
  }
  VT_CHECK_ARG(d->res_mode >= VT_RES_NONE && d->res_mode <= VT_RES_MIX, "vt_conv: res_mode %d", d->res_mode);
  if (d->res_mode != VT_RES_NONE) {
    VT_CHECK_ARG(d->res != nullptr && d->Tr > 0 && d->ldr >= d->Cout, "vt_conv: residual operand");
    VT_CHECK_ARG(((d->To - 1) >> d->res_tshift) < d->Tr, "vt_conv: residual time extent");
  }
  if (d->res_mode == VT_RES_MIX) VT_CHECK_ARG(d->mix_factor != nullptr, "vt_conv: mix_factor is null");
  if (d->out_layout == VT_NCTHW) {
    VT_CHECK_ARG(d->out_dtype == VT_F32, "vt_conv: NCTHW output is fp32 only");
    VT_CHECK_ARG(d->t_trim >= 0 && d->t_trim < d->To, "vt_conv: t_trim");
  } else {
    VT_CHECK_ARG(d->out_layout == VT_NDHWC && d->ldy >= d->Cout, "vt_conv: ldy=%d < Cout=%d", d->ldy, d->Cout);
  }
  const long long M = (long long)d->B * d->To * d->Ho * d->Wo;
  VT_CHECK_ARG(M < (1ll << 31), "vt_conv: M too large");
  const int nbatch = d->nbatch > 0 ? d->nbatch : 1;
  const int yt_mul = d->yt_mul > 0 ? d->yt_mul : 1;
  if (yt_mul != 1)
    VT_CHECK_ARG(d->out_layout == VT_NDHWC && nbatch == 1 && d->yt_off >= 0 && d->yt_off < yt_mul,
                 "vt_conv: output frame interleave needs NDHWC, nbatch 1 and 0 <= yt_off < yt_mul");
  const int ys_mul = d->ys_mul == 2 ? 2 : 1;
  VT_CHECK_ARG(d->ys_mul == 0 || d->ys_mul == 1 || d->ys_mul == 2, "vt_conv: ys_mul %d", d->ys_mul);
  if (ys_mul == 2)
    VT_CHECK_ARG(d->out_layout == VT_NDHWC && nbatch == 1 && yt_mul == 1 && (d->ys_oh | d->ys_ow | 1) == 1,
                 "vt_conv: output pixel interleave needs NDHWC, nbatch 1, no frame interleave, offsets in {0,1}");
  if (d->ln_mode != 0) {
    VT_CHECK_ARG(d->ln_mode == 1 || d->ln_mode == 2, "vt_conv: ln_mode %d", d->ln_mode);
    VT_CHECK_ARG(d->ln_gamma && d->ln_beta && d->ln_out, "vt_conv: fused LayerNorm needs gamma, beta and ln_out");
    VT_CHECK_ARG(d->out_layout == VT_NDHWC && nbatch == 1 && d->ldn >= d->Cout, "vt_conv: fused LayerNorm: NDHWC, nbatch 1, ldn >= Cout");
  }

  memset(&a, 0, sizeof(a));
  a.x = (const char*)d->x; a.w = (const char*)d->w; a.bias = d->bias; a.y = (char*)d->y;
  a.res = (const char*)d->res; a.cache = (const char*)d->cache; a.mix_factor = d->mix_factor;
  a.B = d->B; a.Ti = d->Ti; a.Hi = d->Hi; a.Wi = d->Wi; a.Cin = d->Cin;
  a.To = d->To; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
  a.ldw = d->ldw; a.ldy = d->ldy;
  a.KT = d->KT; a.KH = d->KH; a.KW = d->KW;
  a.st = d->st; a.sh = d->sh; a.sw = d->sw;
  a.pt = d->pt; a.ph = d->ph; a.pw = d->pw;
  a.tmode = d->tmode; a.ncache = d->ncache;
  a.ups_t = d->ups_t; a.ups_s = d->ups_s;
  a.res_mode = d->res_mode; a.res_tshift = d->res_tshift;
  a.Tr = d->res_mode != VT_RES_NONE ? d->Tr : d->To;
  a.ldr = d->ldr;
  // streaming (nt) stores of the LDS epilogues' rows for outputs far larger than the caches (option conv_nt_mb, MiB; 0 = never)
  a.nt_store = (vt_opt(OPT_CONV_NT_MB) > 0 && M * d->Cout * (vt_is_h16(d->out_dtype) ? 2 : 4) >= ((long long)vt_opt(OPT_CONV_NT_MB) << 20)) ? 1 : 0;
  a.out_layout = d->out_layout; a.t_trim = d->t_trim;
  a.M = (int)M; a.ntaps = d->KT * d->KH * d->KW; a.K = a.ntaps * d->Cin;
  a.ys_mul = ys_mul; a.ys_oh = d->ys_oh; a.ys_ow = d->ys_ow;
  a.yt_mul = yt_mul;
  a.yt_step = (long long)(yt_mul - 1) * d->Ho * d->Wo;
  a.yt_base = (long long)(yt_mul != 1 ? d->yt_off : 0) * d->Ho * d->Wo;
  a.fd_hw = make_fastdiv((unsigned)(d->Ho * d->Wo));
  a.fd_wo = make_fastdiv((unsigned)d->Wo); a.fd_ho = make_fastdiv((unsigned)d->Ho); a.fd_to = make_fastdiv((unsigned)d->To);
  a.xs_z = d->xs_z; a.ws_z = d->ws_z; a.ys_z = d->ys_z; a.rs_z = d->rs_z;

  // LayerNorm inside the epilogue: the 128 x 128 tile with the LDS epilogue on full tiles spanning the channel row
  const bool h16_io = vt_is_h16(d->dtype) && d->out_dtype == d->dtype;
  const bool ws_ln_ok = d->ln_mode == 0 || (d->ldn == 128 && (reinterpret_cast<uintptr_t>(d->ln_out) & 15) == 0);
  use_ws = ws_ln_ok && ws_eligible(a, nbatch, h16_io);
  ln_fused = d->ln_mode != 0 && (use_ws || (d->Cout == 128 && M % 128 == 0 && (d->ldy & 7) == 0 && (d->ldn & 7) == 0 &&
                        (d->res_mode == VT_RES_NONE || (d->ldr & 7) == 0) && vt_opt(OPT_CONV_LDSEPI) != 0 &&
                        vt_opt(OPT_CONV_FUSE_LN) != 0));
  // ... or inside the 8-wave 256 x 256 tile's epilogue for Cout = 256 (conv_epilogue_lds256): full tiles, plain rows
  if (d->ln_mode != 0 && !ln_fused && d->Cout == 256 && M % 256 == 0 && (d->dtype == VT_BF16X3 ? VT_F32 : d->dtype) == d->out_dtype && nbatch == 1 &&
      d->Cin % (kRowBytes / (vt_is_h16(d->dtype) ? 2 : 4)) == 0 && (d->ldy & 7) == 0 && (d->ldn & 7) == 0 &&
      (d->res_mode == VT_RES_NONE || (d->res_mode == VT_RES_ADD && (d->ldr & 7) == 0 && d->Tr == d->To && d->res_tshift == 0) ||
       // alpha-mix + LayerNorm (the consumer's norm behind a time up-sampler's parity launches; option conv_tup_ln)
       (d->res_mode == VT_RES_MIX && (d->ldr & 7) == 0 && d->Tr == d->To && d->res_tshift == 0 && vt_opt(OPT_CONV_TUP_LN) != 0)) &&
      vt_opt(OPT_CONV_FUSE_LN256) != 0 && select_tile(a, nbatch) == TILE_256x256)
    ln_fused = true;
  if (ln_fused) {
    a.ln_gamma = d->ln_gamma; a.ln_beta = d->ln_beta; a.ln_out = (char*)d->ln_out;
    a.ln_mode = d->ln_mode; a.ln_keep_y = d->ln_keep_y; a.ldn = d->ldn; a.ln_eps = d->ln_eps;
  }
  if (d->ln_mode != 0 && !ln_fused)
    VT_CHECK_ARG(yt_mul == 1 && ys_mul == 1,
                 "vt_conv: LayerNorm of an interleaved output is only available fused (Cout = 128, full tiles)");
  nbatch_out = nbatch;
  return VT_OK;
}
}  // namespace

// What vt_conv(d) will do, without launching: out[0..1] = pixel x channel tile, out[2] = waves per workgroup,
// out[3] = workgroups (tiles for the persistent kernel), out[4] = 1 when LayerNorm is produced by the conv kernel's
// epilogue (0: second launch of vt_layernorm_act, or no LayerNorm requested), out[5] = kernel launches the call
// performs, out[6] = kernel: 0 = conv_igemm_glds_kernel, 1 = conv3x3_ws128_kernel (weight-stationary), 2 = conv3d_narrow_kernel, 3 = conv3x3_ws2_kernel, 4 = conv_in8_kernel, out[7] = epilogue / ring form (see the header).  Lets tests assert which
// instantiation a parity case exercises and lets bench.py separate conv kernel time from LayerNorm passes.
extern "C" int vt_conv_plan(const vt_conv_desc* d, int32_t* out8) {
  VT_CHECK_ARG(out8 != nullptr, "vt_conv_plan: null output");
  ConvArgs a;
  bool ln_fused = false, use_ws = false;
  int nbatch = 1;
  const int rc = conv_prepare(d, a, ln_fused, nbatch, use_ws);
  if (rc != VT_OK) return rc;
  out8[6] = use_ws ? 1 : 0;
  out8[7] = 0;
  if (narrow_eligible(a, nbatch, d->dtype, d->out_dtype, d->ln_mode)) {   // independent waves: 8 x 14 output pixels x all frames of a time segment
    vt_conv_narrow_plan(&a, out8);
    out8[4] = 0; out8[5] = d->dtype == VT_BF16X3 ? 2 : 1; out8[6] = 2;      // split-bf16: two passes (hi / lo weight plane)
    return VT_OK;
  }
  if (use_ws) {   // persistent, at most one workgroup per CU, all 128 channels per tile: conv_ws2.hip walks 4 x 16-pixel tiles on 8 waves
    out8[0] = 64; out8[1] = 128; out8[2] = 8;
    out8[3] = (a.Wo / 16) * (a.Ho / 4) * a.B * a.To;
    out8[4] = d->ln_mode != 0 ? 1 : 0;
    out8[5] = 1;
    out8[6] = 3;
    return VT_OK;
  }
  if (in8_eligible(a, nbatch, d->dtype, d->out_dtype, ln_fused, d->ln_mode)) {   // conv_in8_kernel: the 128 x 128 tile from a halo patch, register-stationary weights
    out8[0] = 128; out8[1] = 128; out8[2] = 4;
    out8[3] = a.M / 128;
    out8[4] = ln_fused ? 1 : 0; out8[5] = 1; out8[6] = 4; out8[7] = 1;
    return VT_OK;
  }
  static const int dims[4][3] = {{256, 32, 4}, {256, 64, 4}, {256, 256, 8}, {128, 128, 4}};
  const int k = (int)select_tile(a, nbatch);
  out8[0] = dims[k][0]; out8[1] = dims[k][1]; out8[2] = dims[k][2];
  out8[3] = (int32_t)((long long)((a.M + dims[k][0] - 1) / dims[k][0]) * ((a.Cout + dims[k][1] - 1) / dims[k][1]) * nbatch);
  out8[4] = ln_fused ? 1 : 0;
  out8[5] = (d->ln_mode != 0 && !ln_fused) ? 2 : 1;
  const bool split_k = d->work != nullptr && splitk_planes(d, a, nbatch, ln_fused, use_ws) > 0;
  if (split_k) out8[5] += 1;                                                                      // partial launch + reduction
  // epilogue through the LDS (rows of 16-byte accesses) instead of the MFMA-layout vector epilogue
  if (k == TILE_256x256) {
    const bool h16_io = vt_is_h16(d->dtype) && d->out_dtype == d->dtype;
    out8[7] = (ln_fused || lds256_plain_eligible(a, nbatch, h16_io)) ? 1 : 0;
  }
  if (k == TILE_128x128 && deep_ring_eligible(a, nbatch, vt_is_h16(d->dtype) ? 2 : 4)) out8[7] = 2;   // 4-slot ring
  return VT_OK;
}

// Measurement aid (scripts/conv_profile.py): vt_conv on the 8-wave 256 x 256 bf16 tile with shader-clock stamps at the
// phase boundaries of K steps 8..11 of workgroup 0; stamps_out (device, 8 waves x 4 steps x 8 uint64).
extern "C" int vt_conv_profile(const vt_conv_desc* d, uint64_t* stamps_out, vt_stream stream_) {
  VT_CHECK_ARG(stamps_out != nullptr, "vt_conv_profile: null output");
  ConvArgs a;
  bool ln_fused = false, use_ws = false;
  int nbatch = 1;
  const int rc = conv_prepare(d, a, ln_fused, nbatch, use_ws);
  if (rc != VT_OK) return rc;
  a.prof = reinterpret_cast<unsigned long long*>(stamps_out);
  a.prof_mode = vt_opt(OPT_WS_PROF_MODE);
  if (use_ws) return vt_ws2_launch(&a, d->dtype, stream_);       // stamps [wave][16]: conv_ws2.hip iterations 8, 9 of workgroup 0 (8 waves; bf16 only)
  VT_CHECK_ARG(((d->dtype == VT_BF16 && d->out_dtype == VT_BF16) || d->dtype == VT_BF16X3) && d->ln_mode == 0 && select_tile(a, nbatch) == TILE_256x256,
               "vt_conv_profile: bf16 / split-bf16 launches on the 256 x 256 tile without LayerNorm, or on the weight-stationary kernel");
  if (d->dtype == VT_BF16X3) return vt_igemm_dispatch_x3(&a, nbatch, stream_);
  return vt_igemm_dispatch_bf16(&a, nbatch, 0, stream_);
}

extern "C" int64_t vt_conv_work_bytes(const vt_conv_desc* d) {
  ConvArgs a;
  bool ln_fused = false, use_ws = false;
  int nbatch = 1;
  if (conv_prepare(d, a, ln_fused, nbatch, use_ws) != VT_OK) return 0;
  if (narrow_eligible(a, nbatch, d->dtype, d->out_dtype, d->ln_mode)) return 0;
  const int planes = splitk_planes(d, a, nbatch, ln_fused, use_ws);
  return (int64_t)planes * a.M * a.Cout * 4;
}

extern "C" int vt_conv(const vt_conv_desc* d, vt_stream stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ConvArgs a;
  bool ln_fused = false, use_ws = false;
  int nbatch = 1;
  int rc = conv_prepare(d, a, ln_fused, nbatch, use_ws);
  if (rc != VT_OK) return rc;
  if (use_ws) return vt_ws2_launch(&a, d->dtype, stream_);
  if (narrow_eligible(a, nbatch, d->dtype, d->out_dtype, d->ln_mode))
    return vt_conv_narrow_launch(&a, stream_, d->dtype == VT_BF16X3 ? 2 : (d->dtype == VT_F16 ? 1 : 0));
  if (in8_eligible(a, nbatch, d->dtype, d->out_dtype, ln_fused, d->ln_mode)) return vt_conv_in8_launch(&a, d->dtype, stream_);
  const long long M = a.M;
  const int planes = d->work != nullptr ? splitk_planes(d, a, nbatch, ln_fused, use_ws) : 0;
  if (planes > 0 && d->work_bytes >= (int64_t)planes * M * a.Cout * 4 && (reinterpret_cast<uintptr_t>(d->work) & 15) == 0)
    rc = launch_splitk(d, a, planes, stream);
  else if (d->dtype == VT_F32) rc = vt_igemm_dispatch_f32(&a, nbatch, stream_);
  else if (d->dtype == VT_BF16X3) rc = vt_igemm_dispatch_x3(&a, nbatch, stream_);
  else if (d->dtype == VT_F16) rc = vt_igemm_dispatch_f16(&a, nbatch, d->out_dtype == VT_F32 ? 1 : 0, stream_);
  else rc = vt_igemm_dispatch_bf16(&a, nbatch, d->out_dtype == VT_F32 ? 1 : 0, stream_);
  if (rc != VT_OK || d->ln_mode == 0 || ln_fused) return rc;
  // not fusable here: the same contract in two launches
  return vt_layernorm_act(d->y, d->out_dtype, d->ldy, d->ln_out, d->out_dtype, d->ldn, d->ln_gamma, d->ln_beta, M, d->Cout,
                          d->ln_eps, d->ln_mode == 2 ? 1 : 0, stream_);
}

