// Handle-level C-ABI (include/vidtok_amd.h, "model handle"): the stage graph of the causal v1.0 tokenizers -- what
// AutoencodingEngine.encode / decode run (reference vidtok/models/autoencoder.py:197-229 over EncoderCausal3DPadding /
// DecoderCausal3DPadding, vidtok/modules/model_3dcausal.py:502-885) -- driven from C++ over the operator entry points of
// this library, so that a host without Python runs the model:  vt_create(config) -> vt_load_weight(reference state_dict
// key, fp32 data) x N -> vt_encode / vt_regularize_* / vt_decode.
//
// It is the same graph as vidtok_amd/modules.py builds (same operators, same descriptors, same fusion decisions: which
// convolution emits which LayerNorm, the fused temporal block, the parity classes of the up-samplers), so its results
// equal the Python engine's bit for bit (tests/test_gpu_e2e.py::test_model_handle_matches_engine drives it through ctypes
// only).  Scope: `norm_type: layernorm`, `resamp_with_conv: true` -- every shipped causal config, v1.0 and v1.1 (first-frame
// replicate padding, nearest or trilinear time up-sampling, model_3dcausal_v1_1.py) as ONE pass over the clip, and the
// temporal TILING of v1.1 (vt_tile_encode / vt_tile_decode: the chunk schedule, the per-module causal caches, the decoder's
// look-ahead frame with its cache offsets -- AutoencodingEngine.tile_encode / tile_decode, autoencoder_v1_1.py:202-331); and
// (version 2) the non-causal family, Encoder3D / Decoder3D of model_3dnoncausal.py:314-651: the same stages with centred temporal
// windows and plain nn.Conv parameter keys.
//
// Memory: weights are packed on the host when first used and live in device allocations owned by the handle;
// activations come from a caller-provided workspace, cut into two arenas that alternate between stages (a stage reads the
// previous stage's output from one arena and builds its own output and temporaries in the other);
// vt_workspace_bytes() runs the graph dry to size it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include "../../include/vidtok_amd.h"

void vt_set_error(const char* fmt, ...);

namespace {

struct Fail {
  int code;
};
#define M_CHECK(cond, ...)        \
  do {                            \
    if (!(cond)) {                \
      vt_set_error(__VA_ARGS__);  \
      throw Fail{VT_ERR_ARG};     \
    }                             \
  } while (0)
#define M_CALL(expr)                 \
  do {                               \
    const int rc_ = (expr);          \
    if (rc_ != VT_OK) throw Fail{rc_}; \
  } while (0)
#define M_HIP(expr)                                                        \
  do {                                                                     \
    const hipError_t e_ = (expr);                                          \
    if (e_ != hipSuccess) {                                                \
      vt_set_error("%s: %s", #expr, hipGetErrorString(e_));                \
      throw Fail{VT_ERR_HIP};                                              \
    }                                                                      \
  } while (0)

int pad8(int c) { return (c + 7) / 8 * 8; }
size_t esize(int dt) { return dt == VT_F32 ? 4 : 2; }

// fp32 -> bf16, round to nearest even (what torch's .to(bfloat16) does)
uint16_t bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

struct Arena {
  char* base = nullptr;
  size_t cap = 0, off = 0, peak = 0;
  void reset() { off = 0; }
  char* alloc(size_t n, bool dry) {
    off = (off + 255) & ~(size_t)255;
    char* p = base + off;        // dry: base = nullptr, pointers are never dereferenced
    off += n;
    if (off > peak) peak = off;
    if (!dry) M_CHECK(off <= cap, "vt_model: workspace too small (%zu of %zu bytes in one arena; ask vt_workspace_bytes)", off, cap);
    return p;
  }
};

struct Tens {                      // NDHWC activation
  char* p = nullptr;
  int B = 0, T = 0, H = 0, W = 0, ld = 0, dt = VT_BF16;
  size_t bytes() const { return (size_t)B * T * H * W * ld * esize(dt); }
};

struct Norm;
struct Act {                       // an activation, optionally with the LayerNorm(+SiLU) its consumer starts with
  Tens y, n;
  const Norm* norm = nullptr;
  bool silu = false;
  bool has_n() const { return norm != nullptr; }
};
struct NormRef {
  const Norm* norm = nullptr;
  bool silu = false;
};

struct Param {
  std::vector<float> data;
  std::vector<int64_t> shape;
};

struct Model;
struct Ctx {
  Model* m;
  hipStream_t stream;
  bool dry;
  Arena* cur;                      // arena the running stage allocates from
  Tens alloc(int B, int T, int H, int W, int ld, int dt, int c_real) {
    Tens t;
    t.B = B; t.T = T; t.H = H; t.W = W; t.ld = ld; t.dt = dt;
    t.p = cur->alloc(t.bytes(), dry);
    if (ld != c_real && !dry) M_HIP(hipMemsetAsync(t.p, 0, t.bytes(), stream));   // keep the pad lanes defined (they meet zero weights)
    return t;
  }
};

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
};

// Chunk-to-chunk state of one module of a v1.1 tiled pass: the last frames of a causal convolution's input
// (CausalConv*.causal_cache, model_3dcausal_v1_1.py:155-178) or the frames a time resampler keeps (:286-300, 323-343).  The
// buffer is owned by the handle and rewritten in place; `offset` = the module's cache_offset of an overlapped decode.
struct CState {
  std::unique_ptr<DevBuf> buf;
  size_t cap = 0;
  int frames = 0;                  // frames held (0 = nothing cached yet)
  int offset = 0;
};

struct Model {
  vt_model_config cfg;
  bool tiled = false;              // a chunk of a tiled pass is running: causal convolutions read / leave caches
  bool first_chunk = true;
  std::vector<CState*> states;     // every CState of both graphs (vt_reset_cache)
  std::vector<std::unique_ptr<DevBuf>> retired;   // outgrown cache buffers: work queued on them may still run; freed at reset / destroy
  // the cache buffer of `st` with room for `bytes` (grown by a synchronous hipMalloc the first time a chunk kind needs it)
  char* persistent(CState& st, size_t bytes, bool dry) {
    if (dry) return nullptr;
    if (st.cap < bytes) {
      auto b = std::make_unique<DevBuf>();
      if (hipMalloc(&b->p, bytes) != hipSuccess) {
        vt_set_error("vt_model: hipMalloc of a %zu-byte chunk cache failed", bytes);
        throw Fail{VT_ERR_HIP};
      }
      if (st.buf) retired.push_back(std::move(st.buf));
      st.buf = std::move(b);
      st.cap = bytes;
    }
    return (char*)st.buf->p;
  }
  int dt;                          // storage type of the activations (VT_BF16 | VT_F16 | VT_F32)
  bool x3 = false;                 // VT_BF16X3: fp32 storage, convolutions on split-bf16 weight planes (three bf16 MFMAs per product)
  bool v11() const { return cfg.version == 1; }
  bool noncausal() const { return cfg.version == 2; }          // Encoder3D / Decoder3D of model_3dnoncausal.py: centred temporal windows
  // the parameter level a Causal* wrapper adds in front of its nn.Conv ("...conv1.conv.weight"); plain nn.Conv3d / Conv1d have none
  std::string cv() const { return noncausal() ? "" : ".conv"; }
  // time padding of a convolution with taps before the clip: zeros (v1.0) or the first frame repeated (v1.1, one pass)
  int tpad() const { return v11() ? VT_TPAD_REPLICATE : VT_TPAD_ZERO; }
  std::map<std::string, Param> params;
  std::map<std::string, std::unique_ptr<DevBuf>> packed;      // cache key -> the packed tensor on the device (owned)
  std::unique_ptr<DevBuf> reg_buf; // FSQ with projections: the projected latent and its codes (grown on first use, like the chunk caches)
  size_t reg_cap = 0;
  float* reg_scratch(size_t floats) {
    if (reg_cap < floats * 4) {
      auto b = std::make_unique<DevBuf>();
      if (hipMalloc(&b->p, floats * 4) != hipSuccess) {
        vt_set_error("vt_model: hipMalloc of the %zu-byte FSQ projection scratch failed", floats * 4);
        throw Fail{VT_ERR_HIP};
      }
      if (reg_buf) retired.push_back(std::move(reg_buf));
      reg_buf = std::move(b);
      reg_cap = floats * 4;
    }
    return (float*)reg_buf->p;
  }
  int fsq_ncb() const { return cfg.fsq_num_codebooks > 1 ? cfg.fsq_num_codebooks : 1; }
  int fsq_eff() const { return cfg.n_levels * fsq_ncb(); }                     // effective_codebook_dim
  bool fsq_proj() const { return cfg.regularizer == 1 && cfg.fsq_dim > 0 && cfg.fsq_dim != fsq_eff(); }
  bool prepare = false;            // vt_prepare: a dry walk of the graphs that packs and uploads every weight it meets
  Arena arena[2];
  std::vector<std::string> expected;        // state_dict keys the graph reads (filled by a dry run)

  const Param& param(const std::string& key) {
    auto it = params.find(key);
    M_CHECK(it != params.end(), "vt_model: weight '%s' was not loaded (vt_load_weight)", key.c_str());
    return it->second;
  }
  void* upload(const std::string& ckey, const void* host, size_t bytes) {
    auto b = std::make_unique<DevBuf>();
    M_HIP(hipMalloc(&b->p, bytes));
    M_HIP(hipMemcpy(b->p, host, bytes, hipMemcpyHostToDevice));
    void* p = b->p;
    packed[ckey] = std::move(b);
    return p;
  }
  // fp32 vector on the device as it is (biases, LayerNorm affines, mix factors)
  const float* f32(const std::string& key, bool dry) {
    if (dry && !prepare) {
      expected.push_back(key);
      return nullptr;
    }
    auto it = packed.find("f32:" + key);
    if (it != packed.end()) return (const float*)it->second->p;
    const Param& p = param(key);
    return (const float*)upload("f32:" + key, p.data.data(), p.data.size() * 4);
  }
  // convolution weight [Cout][Cin][k...] -> [Cout][taps * cin_p] in the arithmetic dtype (vidtok_amd/packing.py); `xf`
  // optionally rewrites the fp32 weight first (parity classes of the up-samplers: pre-summed taps)
  typedef std::vector<float> (*Xform)(const Param&, std::vector<int64_t>& shape, int a, int b);
  // `rows` = plain rows in the storage type even under VT_BF16X3 (the row operand of a GEMM against activations)
  const void* conv_w(const std::string& key, int cin_p, bool dry, Xform xf = nullptr, int xa = 0, int xb = 0, bool rows = false) {
    if (dry && !prepare) {
      expected.push_back(key);
      return nullptr;
    }
    const bool split = x3 && !rows;
    const std::string ckey = "w:" + key + ":" + std::to_string(cin_p) + ":" + std::to_string(dt) + (split ? "x3" : "") + ":" + std::to_string((xf ? 1 : 0) * 100 + xa * 10 + xb);
    auto it = packed.find(ckey);
    if (it != packed.end()) return it->second->p;
    const Param& p = param(key);
    std::vector<int64_t> shape = p.shape;
    std::vector<float> tmp;
    const float* src = p.data.data();
    if (xf) {
      tmp = xf(p, shape, xa, xb);
      src = tmp.data();
    }
    M_CHECK(shape.size() >= 3, "vt_model: '%s' is not a convolution weight", key.c_str());
    const int64_t cout = shape[0], cin = shape[1];
    int64_t taps = 1;
    for (size_t i = 2; i < shape.size(); ++i) taps *= shape[i];
    M_CHECK(cin <= cin_p, "vt_model: '%s' has %lld input channels, the activation stores %d", key.c_str(), (long long)cin, cin_p);
    std::vector<float> w((size_t)cout * taps * cin_p, 0.0f);
    for (int64_t o = 0; o < cout; ++o)
      for (int64_t c = 0; c < cin; ++c)
        for (int64_t t = 0; t < taps; ++t) w[((size_t)o * taps + t) * cin_p + c] = src[((size_t)o * cin + c) * taps + t];
    if (split) {
      // vidtok_amd/packing.py::pack_split3: per group of 16 k, [hi 16 x bf16 | lo 16 x bf16], hi = bf16(w), lo = bf16(w - hi);
      // rows zero-padded to whole 128-byte K steps (ldw = K rounded up to 32, in 4-byte units)
      const size_t K = (size_t)taps * cin_p, Kp = (K + 31) / 32 * 32;
      std::vector<uint16_t> pl((size_t)cout * Kp * 2, 0);
      for (int64_t o = 0; o < cout; ++o)
        for (size_t k = 0; k < K; ++k) {
          const float v = w[(size_t)o * K + k];
          const uint16_t hi = bf16_rne(v);
          uint32_t hb = (uint32_t)hi << 16;
          float hf;
          memcpy(&hf, &hb, 4);
          uint16_t* blk = &pl[((size_t)o * Kp + (k / 16) * 16) * 2];
          blk[k % 16] = hi;
          blk[16 + k % 16] = bf16_rne(v - hf);
        }
      return upload(ckey, pl.data(), pl.size() * 2);
    }
    if (dt == VT_F32) return upload(ckey, w.data(), w.size() * 4);
    std::vector<uint16_t> h(w.size());
    if (dt == VT_F16) {
      for (size_t i = 0; i < w.size(); ++i) {          // round to nearest even (the compiler's float -> _Float16 conversion)
        const _Float16 f = (_Float16)w[i];
        memcpy(&h[i], &f, 2);
      }
    } else {
      for (size_t i = 0; i < w.size(); ++i) h[i] = bf16_rne(w[i]);
    }
    return upload(ckey, h.data(), h.size() * 2);
  }
};

// ---- pre-summed taps of the up-samplers (vidtok_amd/packing.py) ------------------------------------------------------------
// time: [Co][Ci][3][kh][kw] -> [Co][Ci][2][kh][kw]; early: [W0 + W1, W2], else [W0, W1 + W2]
std::vector<float> xf_time_parity(const Param& p, std::vector<int64_t>& shape, int early, int) {
  const int64_t co = shape[0], ci = shape[1], hw = shape[3] * shape[4];
  std::vector<float> o((size_t)co * ci * 2 * hw);
  for (int64_t a = 0; a < co * ci; ++a)
    for (int64_t s = 0; s < hw; ++s) {
      const float w0 = p.data[(a * 3 + 0) * hw + s], w1 = p.data[(a * 3 + 1) * hw + s], w2 = p.data[(a * 3 + 2) * hw + s];
      o[(a * 2 + 0) * hw + s] = early ? w0 + w1 : w0;
      o[(a * 2 + 1) * hw + s] = early ? w2 : w1 + w2;
    }
  shape[2] = 2;
  return o;
}
// space: [Co][Ci][3][3] -> [Co][Ci][2][2]; rows [W0, W1+W2] for py = 0, [W0+W1, W2] for py = 1, likewise columns
std::vector<float> xf_space_parity(const Param& p, std::vector<int64_t>& shape, int py, int px) {
  const int64_t n = shape[0] * shape[1];
  std::vector<float> o((size_t)n * 4);
  for (int64_t a = 0; a < n; ++a) {
    const float* w = &p.data[a * 9];
    float rows[2][3];
    for (int c = 0; c < 3; ++c) {
      rows[0][c] = py == 0 ? w[0 * 3 + c] : w[0 * 3 + c] + w[1 * 3 + c];
      rows[1][c] = py == 0 ? w[1 * 3 + c] + w[2 * 3 + c] : w[2 * 3 + c];
    }
    for (int r = 0; r < 2; ++r) {
      o[a * 4 + r * 2 + 0] = px == 0 ? rows[r][0] : rows[r][0] + rows[r][1];
      o[a * 4 + r * 2 + 1] = px == 0 ? rows[r][1] + rows[r][2] : rows[r][2];
    }
  }
  shape[2] = 2;
  shape[3] = 2;
  return o;
}

// ---- operators ---------------------------------------------------------------------------------------------------------
struct Geom {
  int kt = 1, kh = 1, kw = 1, st = 1, sh = 1, sw = 1, pt = 0, ph = 0, pw = 0, ph_hi = 0, pw_hi = 0;
  int pt_hi = 0;                   // zero frames after the clip (non-causal windows; the kernel reads 0 beyond the end)
  void out_dims(int Ti, int Hi, int Wi, int& To, int& Ho, int& Wo) const {
    To = (Ti + pt + pt_hi - kt) / st + 1;
    Ho = (Hi + ph + ph_hi - kh) / sh + 1;
    Wo = (Wi + pw + pw_hi - kw) / sw + 1;
  }
};
Geom causal3d(int kt, int kh, int kw, int st = 1, int sh = 1, int sw = 1) {
  Geom g;
  g.kt = kt; g.kh = kh; g.kw = kw; g.st = st; g.sh = sh; g.sw = sw;
  g.pt = (kt - 1) + (1 - st);
  const int hp = (kh - 1) + (1 - sh), wp = (kw - 1) + (1 - sw);
  g.ph = hp / 2; g.pw = wp / 2; g.ph_hi = hp - hp / 2; g.pw_hi = wp - wp / 2;
  return g;
}
// the same window centred in time (nn.Conv3d / Conv1d with padding k / 2 of the non-causal family): `before` frames in front,
// the rest of the k - 1 behind
Geom centred(Geom g, int before) {
  g.pt_hi = g.pt - before;
  g.pt = before;
  return g;
}

struct Norm {
  std::string key;                 // "...norm1": LayerNorm wrapper, parameters at key + ".norm.weight" / ".norm.bias"; GroupNorm: key + ".weight" / ".bias"
  float eps = 1e-6f;
  bool group = false;              // norm_type "groupnorm": torch.nn.GroupNorm(32, C, eps 1e-6) (model_3dcausal.py:30-32)
  int site = VT_GN_FRAME;          // GroupNorm only: the view the reference's call site normalises; -1 = single positions (the causal temporal blocks)
  std::string wkey() const { return key + (group ? ".weight" : ".norm.weight"); }
  std::string bkey() const { return key + (group ? ".bias" : ".norm.bias"); }
  // the norm of a plain tensor (LayerNorm: vt_layernorm_act; GroupNorm: vt_groupnorm_act over the site's domain, as
  // vidtok_amd/ops.py::groupnorm_act drives it)
  Tens of(Ctx& c, const Tens& t, bool silu, int c_real) const {
    Tens o = c.alloc(t.B, t.T, t.H, t.W, t.ld, c.m->dt, c_real);
    const float* g = c.m->f32(wkey(), c.dry);
    const float* b = c.m->f32(bkey(), c.dry);
    if (!group) {
      if (!c.dry)
        M_CALL(vt_layernorm_act(t.p, t.dt, t.ld, o.p, o.dt, o.ld, g, b, (int64_t)t.B * t.T * t.H * t.W, c_real, eps, silu ? 1 : 0, c.stream));
      return o;
    }
    // single positions ("(b t) c s" with s = 1, model_3dcausal.py:476-487) = the PIXEL domain of a one-frame view of all positions
    const int scope = site < 0 ? VT_GN_PIXEL : site;
    const int B = site < 0 ? 1 : t.B, T = site < 0 ? 1 : t.T;
    const int64_t HW = site < 0 ? (int64_t)t.B * t.T * t.H * t.W : (int64_t)t.H * t.W;
    const int64_t wb = std::max<int64_t>(vt_groupnorm_work_bytes(B, T, 32, scope), 8);
    void* work = c.cur->alloc((size_t)wb, c.dry);
    if (!c.dry) M_CALL(vt_groupnorm_act(t.p, t.dt, t.ld, o.p, o.dt, o.ld, g, b, B, T, HW, c_real, 32, scope, eps, silu ? 1 : 0, work, c.stream));
    return o;
  }
  // LayerNorm(+SiLU) of x, unless x already carries exactly that
  Tens apply(Ctx& c, const Act& x, bool silu, int c_real) const {
    if (x.has_n() && x.norm == this && x.silu == silu) return x.n;
    return of(c, x.y, silu, c_real);
  }
  // what a stage asks its producer to emit: nothing for GroupNorm (its statistics span more than a position: never fused)
  NormRef ref(bool silu) const { return group ? NormRef() : NormRef{this, silu}; }
};

struct ConvOpts {
  const Tens* res = nullptr;
  int res_mode = VT_RES_NONE;
  const float* mix = nullptr;
  NormRef ln;                      // emit this norm of the result
  bool keep_y = true;
  Tens* out = nullptr;             // preallocated interleaved output (parity classes)
  Tens* ln_out = nullptr;          // ... and its normalised twin (the emitted LayerNorm of an interleaved output)
  bool ln_optional = false;        // emit `ln` only if this launch's epilogue takes it (vt_conv_plan), else run without (ops.conv ln_optional)
  bool plan_only = false;          // with ln_optional: launch nothing, only answer (Act::has_n) whether the epilogue would take the LayerNorm
  int yt_mul = 1, yt_off = 0, ys = 0, ys_oh = 0, ys_ow = 0;
  float* ncthw = nullptr;          // write fp32 NCTHW here instead
  int t_trim = 0;
  int tmode = VT_TPAD_ZERO;
  const void* cache = nullptr;     // tmode VT_TPAD_CACHE: [B][ncache][H][W][ld]
  int ncache = 0;
};

// mirror of vidtok_amd/ops.py::conv
Act conv(Ctx& c, const Tens& x, const void* w, int ldw, const float* bias, const Geom& g, int cout, const ConvOpts& o) {
  int To, Ho, Wo;
  g.out_dims(x.T, x.H, x.W, To, Ho, Wo);
  M_CHECK(To > 0 && Ho > 0 && Wo > 0, "vt_model: convolution output is empty (%d x %d x %d)", To, Ho, Wo);
  Act r;
  vt_conv_desc d;
  memset(&d, 0, sizeof(d));
  int ldy;
  if (o.out) {
    r.y = *o.out;
    ldy = o.out->ld;
  } else if (o.ncthw) {
    ldy = cout;
  } else {
    ldy = pad8(cout);
    r.y = c.alloc(x.B, To, Ho, Wo, ldy, c.m->dt, cout);
  }
  d.x = x.p; d.w = w; d.bias = bias; d.y = o.ncthw ? (void*)o.ncthw : (void*)r.y.p;
  d.B = x.B; d.Ti = x.T; d.Hi = x.H; d.Wi = x.W; d.Cin = x.ld;
  d.To = To; d.Ho = Ho; d.Wo = Wo; d.Cout = cout;
  d.ldw = c.m->x3 ? (ldw + 31) / 32 * 32 : ldw; d.ldy = ldy;
  d.KT = g.kt; d.KH = g.kh; d.KW = g.kw; d.st = g.st; d.sh = g.sh; d.sw = g.sw; d.pt = g.pt; d.ph = g.ph; d.pw = g.pw;
  d.tmode = g.pt > 0 ? o.tmode : VT_TPAD_ZERO;
  if (d.tmode == VT_TPAD_CACHE) { d.cache = o.cache; d.ncache = o.ncache; }
  d.res_mode = o.res_mode;
  if (o.res_mode != VT_RES_NONE) {
    d.res = o.res->p; d.res_tshift = 0; d.Tr = o.res->T; d.ldr = o.res->ld;
    d.mix_factor = o.mix;
  }
  d.out_layout = o.ncthw ? VT_NCTHW : VT_NDHWC;
  d.t_trim = o.t_trim;
  d.dtype = c.m->x3 ? VT_BF16X3 : x.dt; d.out_dtype = o.ncthw ? VT_F32 : c.m->dt;
  d.nbatch = 1;
  d.yt_mul = o.yt_mul; d.yt_off = o.yt_off;
  if (o.ys) { d.ys_mul = 2; d.ys_oh = o.ys_oh; d.ys_ow = o.ys_ow; }
  const bool ln_after = o.ln.norm && o.ln.norm->group;      // GroupNorm of the result: its own pass behind the convolution (GroupNorm32.after)
  if (o.ln.norm && !ln_after) {
    r.n = o.ln_out ? *o.ln_out : c.alloc(x.B, To, Ho, Wo, ldy, c.m->dt, cout);
    d.ln_gamma = c.m->f32(o.ln.norm->wkey(), c.dry);
    d.ln_beta = c.m->f32(o.ln.norm->bkey(), c.dry);
    d.ln_out = r.n.p;
    d.ln_mode = o.ln.silu ? 2 : 1; d.ln_keep_y = o.keep_y ? 1 : 0; d.ldn = ldy; d.ln_eps = o.ln.norm->eps;
    r.norm = o.ln.norm; r.silu = o.ln.silu;
  }
  // (a dry run has no pointers to validate: the library is asked with stand-ins, its decisions depend on the geometry only)
  auto standins = [&](vt_conv_desc& q) {
    if (!c.dry) return;
    q.x = q.w = q.y = (void*)16;
    if (q.res_mode != VT_RES_NONE) q.res = (void*)16;
    if (q.res_mode == VT_RES_MIX) q.mix_factor = (const float*)16;
    if (q.tmode == VT_TPAD_CACHE) q.cache = (void*)16;
    if (q.ln_mode != 0) { q.ln_gamma = q.ln_beta = (const float*)16; q.ln_out = (void*)16; }
  };
  // the LayerNorm of an interleaved output exists only inside an epilogue: where this launch's does not take it, run without and let
  // the consumer normalise y itself (vidtok_amd/ops.py::conv ln_optional; option conv_tup_ln)
  if (o.ln_optional && d.ln_mode != 0) {
    vt_conv_desc q = d;
    standins(q);
    int32_t plan[8];
    if (q.ln_out == nullptr) q.ln_out = (void*)16;             // (plan_only: the twin tensor is not allocated yet)
    if (!(vt_conv_plan(&q, plan) == VT_OK && plan[4] == 1)) {
      d.ln_gamma = d.ln_beta = nullptr; d.ln_out = nullptr; d.ln_mode = 0;
      r.n = Tens(); r.norm = nullptr;
    }
  }
  if (o.plan_only) return r;
  // split-K over the time taps (small-M launches): the library says how much scratch; it comes from the stage's arena.
  {
    vt_conv_desc q = d;
    standins(q);
    const int64_t wb = ((x.dt == VT_BF16 || x.dt == VT_F16) && (g.kt == 3 || g.kh == 3)) ? vt_conv_work_bytes(&q) : 0;
    if (wb > 0) {
      d.work = c.cur->alloc((size_t)wb, c.dry);
      d.work_bytes = wb;
    }
  }
  if (!c.dry) M_CALL(vt_conv(&d, c.stream));
  if (ln_after) {
    M_CHECK(!o.out && !o.ncthw, "vt_model: GroupNorm of an interleaved / NCTHW result");
    r.n = o.ln.norm->of(c, r.y, o.ln.silu, cout);
    r.norm = o.ln.norm; r.silu = o.ln.silu;
  }
  return r;
}

// frames idx[0..n) of src [B][Ts][frame] -> dst frames [t0, t0 + n) of [B][Td][frame] (vidtok_amd/ops.py::gather_frames)
void gather(Ctx& c, const char* src, int Ts, char* dst, int Td, int t0, const std::vector<int>& idx, int B, int64_t frame_elems, int es) {
  for (size_t j0 = 0; j0 < idx.size(); j0 += 128) {
    int32_t part[128];
    const int n = (int)std::min<size_t>(128, idx.size() - j0);
    for (int j = 0; j < n; ++j) part[j] = idx[j0 + j];
    if (!c.dry)
      M_CALL(vt_gather_frames(src, dst + (size_t)(t0 + (int)j0) * frame_elems * es, es, B, frame_elems, (int64_t)Ts * frame_elems, (int64_t)Td * frame_elems, part, n, c.stream));
  }
}

// _CausalState._update_cache (vidtok_amd/modules.py; reference model_3dcausal_v1_1.py:172-176): keep the last P frames of
// padded[: len - offset], padded = [P pad frames (x[0] repeated on the first chunk | the previous cache), x]
void update_cache(Ctx& c, CState& st, const Tens& x, int P) {
  if (P == 0) return;
  Model* m = c.m;
  const int T = x.T, off = st.offset, es = (int)esize(x.dt);
  const int64_t fr = (int64_t)x.H * x.W * x.ld;
  std::vector<int> sx, sc;
  int jx = -1, jc = -1;
  for (int j = 0; j < P; ++j) {
    const int q = T - off + j;                      // index into the padded sequence
    M_CHECK(q >= 0, "vt_model: chunk of %d frames is shorter than its cache offset %d", T, off);
    if (q >= P) { if (jx < 0) jx = j; sx.push_back(q - P); }
    else if (m->first_chunk) { if (jx < 0) jx = j; sx.push_back(0); }
    else { if (jc < 0) jc = j; sc.push_back(q); }
  }
  // frames kept from the old cache move inside the buffer they are read from: through a temporary (stream order)
  char* kept = nullptr;
  if (!sc.empty()) {
    M_CHECK(c.dry || st.frames >= P, "vt_model: causal cache missing (the first chunk must come first)");
    kept = c.cur->alloc((size_t)x.B * sc.size() * fr * es, c.dry);
    gather(c, c.dry ? nullptr : (const char*)st.buf->p, P, kept, (int)sc.size(), 0, sc, x.B, fr, es);
  }
  char* buf = m->persistent(st, (size_t)x.B * P * fr * es, c.dry);
  if (kept) {
    std::vector<int> id(sc.size());
    for (size_t i = 0; i < id.size(); ++i) id[i] = (int)i;
    gather(c, kept, (int)sc.size(), buf, P, jc, id, x.B, fr, es);
  }
  if (!sx.empty()) gather(c, x.p, T, buf, P, jx, sx, x.B, fr, es);
  if (!c.dry) st.frames = P;
}

// a causal convolution with chunk state: CausalConv3d.run / CausalConv1d.run of vidtok_amd/modules.py
Act conv_causal(Ctx& c, CState& st, const Tens& x, const void* w, int ldw, const float* bias, const Geom& g, int cout, ConvOpts o) {
  Model* m = c.m;
  const int P = g.pt;
  if (m->tiled && P > 0) {
    if (m->first_chunk) o.tmode = VT_TPAD_REPLICATE;
    else {
      M_CHECK(c.dry || st.frames >= P, "vt_model: causal cache missing (the first chunk must come first)");
      o.tmode = VT_TPAD_CACHE;
      o.cache = c.dry ? (const void*)16 : st.buf->p;
      o.ncache = c.dry ? P : st.frames;
    }
  }
  const Act r = conv(c, x, w, ldw, bias, g, cout, o);
  if (m->tiled) update_cache(c, st, x, P);
  return r;
}

// batched C[z] = A[z] B[z]^T on the convolution kernel (vidtok_amd/ops.py::gemm_nt)
char* gemm_nt(Ctx& c, const void* a, bool a_batched, const void* b, int Z, int M, int N, int K, int in_dt, int out_dt, int ldo, const float* bias) {
  const size_t bytes = (size_t)Z * M * ldo * esize(out_dt);
  char* y = c.cur->alloc(bytes, c.dry);
  if (ldo != N && !c.dry) M_HIP(hipMemsetAsync(y, 0, bytes, c.stream));
  vt_conv_desc d;
  memset(&d, 0, sizeof(d));
  d.x = a; d.w = b; d.y = y; d.bias = bias;
  d.B = 1; d.Ti = 1; d.Hi = 1; d.Wi = M; d.Cin = K;
  d.To = 1; d.Ho = 1; d.Wo = M; d.Cout = N;
  d.ldw = K; d.ldy = ldo;
  d.KT = d.KH = d.KW = 1; d.st = d.sh = d.sw = 1;
  d.dtype = in_dt; d.out_dtype = out_dt;
  d.nbatch = Z;
  d.xs_z = a_batched ? (int64_t)M * K : 0; d.ws_z = (int64_t)N * K; d.ys_z = (int64_t)M * ldo;
  if (!c.dry) M_CALL(vt_conv(&d, c.stream));
  return y;
}

// ---- stages ------------------------------------------------------------------------------------------------------------
struct Stage {
  virtual ~Stage() {}
  virtual NormRef first_norm(Ctx&) const { return NormRef(); }      // the norm this stage wants from its producer
  virtual Act run(Ctx& c, const Act& x, NormRef next) = 0;
  virtual void states(std::vector<CState*>&) {}                     // the chunk state this stage keeps in a tiled pass
  virtual bool time_up() const { return false; }                    // a temporal up-sampler: cache offsets double from here on
};

ConvOpts emit(NormRef next, int tmode = VT_TPAD_ZERO) {
  ConvOpts o;
  o.ln = next;                     // keep_y = true
  o.tmode = tmode;
  return o;
}

struct ConvParams {                // an nn.ConvNd parameter pair under `key` (".weight" / ".bias")
  std::string key;
  int cin = 0, cout = 0;
};

struct ResBlock : Stage {          // ResnetBlock (2-D per frame) or ResnetCausalBlock (3-D causal), model_3dcausal.py:276-424
  bool causal3d_ = false;
  int cin, cout;
  Norm n1, n2;
  ConvParams c1, c2, sc;
  CState s1, s2;                   // the 3-D causal block's convolutions keep two frames each in a tiled pass
  void states(std::vector<CState*>& v) override {
    if (causal3d_) { v.push_back(&s1); v.push_back(&s2); }
  }
  NormRef first_norm(Ctx&) const override { return n1.ref(true); }
  Act run(Ctx& c, const Act& x, NormRef next) override {
    Model* m = c.m;
    Geom g3 = causal3d_ ? causal3d(3, 3, 3) : causal3d(1, 3, 3);
    const Geom g1 = Geom();
    if (causal3d_ && m->noncausal()) g3 = centred(g3, 1);           // ResnetNoncausalBlock: Conv3d padding 1
    const Tens h = n1.apply(c, x, true, cin);
    ConvOpts o1;
    o1.ln = NormRef{&n2, true};
    o1.keep_y = false;             // conv1's result is only ever seen through norm2 + SiLU
    o1.tmode = m->tpad();
    const int taps = causal3d_ ? 27 : 9;
    const Act h2 = conv_causal(c, s1, h, m->conv_w(c1.key + ".weight", h.ld, c.dry), taps * h.ld, m->f32(c1.key + ".bias", c.dry), g3, cout, o1);
    Tens xs = x.y;
    if (cin != cout) xs = conv(c, x.y, m->conv_w(sc.key + ".weight", x.y.ld, c.dry), x.y.ld, m->f32(sc.key + ".bias", c.dry), g1, cout, ConvOpts()).y;
    ConvOpts o2 = emit(next, m->tpad());
    o2.res = &xs;
    o2.res_mode = VT_RES_ADD;
    return conv_causal(c, s2, h2.n, m->conv_w(c2.key + ".weight", h2.n.ld, c.dry), taps * h2.n.ld, m->f32(c2.key + ".bias", c.dry), g3, cout, o2);
  }
};

struct TBlock : Stage {            // ResnetCausalBlock1D, model_3dcausal.py:427-499
  int ch;
  Norm n1, n2;
  ConvParams c1, c2;               // CausalConv1d: parameters at key + ".conv.weight"
  CState s1, s2;                   // chunk state of the two convolutions: [B][2][H][W][C] each
  void states(std::vector<CState*>& v) override { v.push_back(&s1); v.push_back(&s2); }
  // ResnetCausalBlock1D._fusable (vidtok_amd/modules.py): one launch for bf16, C = 128; in a tiled pass both convolutions must
  // stand at the same point of the chunk schedule and, past the first chunk, both caches must be there
  bool fusable(const Ctx& c) const {
    if (c.m->noncausal()) return false;                // the fused launch is the causal block
    if (!((c.m->dt == VT_BF16 || c.m->dt == VT_F16) && ch == 128 && n1.eps == n2.eps) || n1.group) return false;
    if (!c.m->tiled) return true;
    if (s1.offset != s2.offset) return false;
    return c.m->first_chunk || c.dry || (s1.frames >= 2 && s2.frames >= 2);
  }
  NormRef first_norm(Ctx& c) const override { return fusable(c) ? NormRef() : n1.ref(true); }
  Act run(Ctx& c, const Act& x, NormRef next) override {
    Model* m = c.m;
    const Tens& xp = x.y;
    Geom g;
    g.kt = 3; g.pt = 2;
    if (m->noncausal()) g = centred(g, 1);             // ResnetBlock1D: Conv1d padding 1
    const std::string cv = m->cv();
    vt_tblock_desc d;
    memset(&d, 0, sizeof(d));
    d.dtype = xp.dt; d.C = ch; d.ld = xp.ld; d.B = xp.B; d.T = xp.T; d.HW = (int64_t)xp.H * xp.W; d.tmode = m->tpad();
    const bool fus = fusable(c);
    if (fus && m->tiled) {           // the launch keeps the chunk state itself: both caches rewritten in place
      const size_t cb = (size_t)xp.B * 2 * xp.H * xp.W * xp.ld * esize(xp.dt);
      d.tmode = m->first_chunk ? VT_TPAD_REPLICATE : VT_TPAD_CACHE;
      d.cache1 = m->persistent(s1, cb, c.dry);
      d.cache2 = m->persistent(s2, cb, c.dry);
      if (c.dry) d.cache1 = d.cache2 = (void*)16;
      d.cache_offset = s1.offset;
    }
    if (fus && vt_temporal_block_supported(&d)) {
      if (m->tiled && !c.dry) s1.frames = s2.frames = 2;
      Act r;
      r.y = c.alloc(xp.B, xp.T, xp.H, xp.W, xp.ld, xp.dt, ch);
      const bool nx = next.norm != nullptr && next.norm->eps == n1.eps;
      if (nx) r.n = c.alloc(xp.B, xp.T, xp.H, xp.W, xp.ld, xp.dt, ch);
      d.x = xp.p; d.y = r.y.p; d.n_out = nx ? r.n.p : nullptr;
      d.w1 = m->conv_w(c1.key + ".conv.weight", xp.ld, c.dry); d.b1 = m->f32(c1.key + ".conv.bias", c.dry);
      d.w2 = m->conv_w(c2.key + ".conv.weight", xp.ld, c.dry); d.b2 = m->f32(c2.key + ".conv.bias", c.dry);
      d.norm1_gamma = m->f32(n1.key + ".norm.weight", c.dry); d.norm1_beta = m->f32(n1.key + ".norm.bias", c.dry);
      d.norm2_gamma = m->f32(n2.key + ".norm.weight", c.dry); d.norm2_beta = m->f32(n2.key + ".norm.bias", c.dry);
      if (nx) {
        d.next_gamma = m->f32(next.norm->key + ".norm.weight", c.dry); d.next_beta = m->f32(next.norm->key + ".norm.bias", c.dry);
        d.ln_next_mode = next.silu ? 2 : 1;
        r.norm = next.norm; r.silu = next.silu;
      }
      d.keep_y = 1; d.eps = n1.eps;
      if (!c.dry) M_CALL(vt_temporal_block(&d, c.stream));
      return r;
    }
    const Tens h = n1.apply(c, x, true, ch);
    ConvOpts o1;
    o1.ln = NormRef{&n2, true};
    o1.keep_y = false;
    o1.tmode = m->tpad();
    const Act h2 = conv_causal(c, s1, h, m->conv_w(c1.key + cv + ".weight", h.ld, c.dry), 3 * h.ld, m->f32(c1.key + cv + ".bias", c.dry), g, ch, o1);
    ConvOpts o2 = emit(next, m->tpad());
    o2.res = &xp;
    o2.res_mode = VT_RES_ADD;
    return conv_causal(c, s2, h2.n, m->conv_w(c2.key + cv + ".weight", h2.n.ld, c.dry), 3 * h2.n.ld, m->f32(c2.key + cv + ".bias", c.dry), g, ch, o2);
  }
};

struct Attn : Stage {              // AttnBlockWrapper, model_3dcausal.py:83-141 ("heads" = frames)
  int ch;
  Norm n;
  std::string key;                 // q / k / v / proj_out: CausalConv3d 1x1x1 at key + ".q.conv.weight" ...
  NormRef first_norm(Ctx&) const override { return n.ref(false); }
  Act run(Ctx& c, const Act& x, NormRef next) override {
    Model* m = c.m;
    const Tens hn = n.apply(c, x, false, ch);
    const Tens& xp = x.y;
    const int S = xp.H * xp.W, Z = xp.B * xp.T, Cc = xp.ld, Sp = pad8(S), dt = m->dt;
    M_CHECK(Cc == ch, "vt_model: attention over %d channels stored as %d (channel counts must be multiples of 8)", ch, Cc);
    const Geom g1;
    const std::string cv = m->cv();
    const Tens q = conv(c, hn, m->conv_w(key + ".q" + cv + ".weight", Cc, c.dry), Cc, m->f32(key + ".q" + cv + ".bias", c.dry), g1, ch, ConvOpts()).y;
    const Tens k = conv(c, hn, m->conv_w(key + ".k" + cv + ".weight", Cc, c.dry), Cc, m->f32(key + ".k" + cv + ".bias", c.dry), g1, ch, ConvOpts()).y;
    const void* wv = m->conv_w(key + ".v" + cv + ".weight", Cc, c.dry, nullptr, 0, 0, true);
    const float* bv = m->f32(key + ".v" + cv + ".bias", c.dry);
    // V^T directly: the weight is the row operand; v's bias is added after P V (rows of P sum to 1)
    char* vT = gemm_nt(c, wv, false, hn.p, Z, Cc, S, Cc, dt, dt, Sp, nullptr);                // [Z][C][Sp]
    // scale = C^-0.5 as the Python host passes it: computed in double, rounded once to float
    const float scale = (float)std::pow((double)Cc, -0.5);
    Tens o = xp;
    if (vt_flash_attention_supported(dt, S, Cc, Sp)) {
      // one launch with the online softmax: no [Z][S][S] scores in memory (vidtok_amd/modules.py takes the same decision)
      o.p = c.cur->alloc((size_t)Z * S * Cc * esize(dt), c.dry);
      if (!c.dry) M_CALL(vt_flash_attention(q.p, k.p, vT, bv, o.p, dt, Z, S, Cc, Sp, scale, c.stream));
    } else {
      char* s = gemm_nt(c, q.p, true, k.p, Z, S, S, Cc, dt, VT_F32, S, nullptr);               // [Z][S][S] fp32
      const size_t pbytes = (size_t)Z * S * Sp * esize(dt);
      char* p = c.cur->alloc(pbytes, c.dry);
      if (!c.dry) {
        if (Sp != S) M_HIP(hipMemsetAsync(p, 0, pbytes, c.stream));
        M_CALL(vt_softmax_rows((const float*)s, p, dt, (int64_t)Z * S, S, Sp, scale, c.stream));
      }
      o.p = gemm_nt(c, p, true, vT, Z, S, Cc, Sp, dt, dt, Cc, bv);                             // [Z][S][C] = [B][T][H][W][C]
    }
    ConvOpts op = emit(next);
    op.res = &xp;
    op.res_mode = VT_RES_ADD;
    return conv(c, o, m->conv_w(key + ".proj_out" + cv + ".weight", Cc, c.dry), Cc, m->f32(key + ".proj_out" + cv + ".bias", c.dry), g1, ch, op);
  }
};

struct Down : Stage {              // Downsample: F.pad(0,1,0,1) + conv3x3 stride 2, model_3dcausal.py:215-230
  int ch;
  std::string key;                 // nn.Conv2d at key + ".conv.weight"
  Act run(Ctx& c, const Act& x, NormRef next) override {
    Geom g;
    g.kh = 3; g.kw = 3; g.sh = 2; g.sw = 2; g.ph_hi = 1; g.pw_hi = 1;
    return conv(c, x.y, c.m->conv_w(key + ".conv.weight", x.y.ld, c.dry), 9 * x.y.ld, c.m->f32(key + ".conv.bias", c.dry), g, ch, emit(next));
  }
};

struct TimeDown : Stage {          // TimeDownsampleResCausal2x, model_3dcausal.py:233-252
  int ch;
  std::string key;                 // CausalConv3d at key + ".conv.conv.weight", key + ".mix_factor"
  CState sc, sp;                   // the convolution's cache (one frame) and the pooling branch's (the chunk's last frame)
  void states(std::vector<CState*>& v) override { v.push_back(&sc); v.push_back(&sp); }
  Act run(Ctx& c, const Act& x, NormRef next) override {
    Model* m = c.m;
    const Tens& xp = x.y;
    Tens x1 = c.alloc(xp.B, xp.T / 2, xp.H, xp.W, xp.ld, xp.dt, xp.ld);
    int tm = m->noncausal() ? VT_TPAD_ZERO_BACK : m->tpad();        // non-causal: both branches see [x, 0] (model_3dnoncausal.py:86-89)
    const void* pc = nullptr;
    if (m->tiled) {                // TimeDownsampleResCausal2x.run of the Python host (model_3dcausal_v1_1.py:286-300)
      tm = m->first_chunk ? VT_TPAD_REPLICATE : VT_TPAD_CACHE;
      if (!m->first_chunk) {
        M_CHECK(c.dry || sp.frames >= 1, "vt_model: time down-sampler cache missing (the first chunk must come first)");
        pc = c.dry ? nullptr : sp.buf->p;
      }
    }
    if (!c.dry) M_CALL(vt_time_avgpool3s2(xp.p, pc, x1.p, xp.dt, xp.B, xp.T, (int64_t)xp.H * xp.W, xp.ld, tm, c.stream));
    if (m->tiled) {
      const int64_t fr = (int64_t)xp.H * xp.W * xp.ld;
      char* buf = m->persistent(sp, (size_t)xp.B * fr * esize(xp.dt), c.dry);
      gather(c, xp.p, xp.T, buf, 1, 0, std::vector<int>{xp.T - 1}, xp.B, fr, (int)esize(xp.dt));
      if (!c.dry) sp.frames = 1;
    }
    ConvOpts o = emit(next, m->tpad());
    o.res = &x1;
    o.res_mode = VT_RES_MIX;
    o.mix = m->f32(key + ".mix_factor", c.dry);
    Geom g = causal3d(3, 3, 3, 2, 1, 1);
    if (m->noncausal()) g = centred(g, 0);                            // stride-2 Conv3d with padding (0,1,1) over [x, 0]
    const std::string ck = key + ".conv" + m->cv();
    return conv_causal(c, sc, xp, m->conv_w(ck + ".weight", xp.ld, c.dry), 27 * xp.ld, m->f32(ck + ".bias", c.dry), g, ch, o);
  }
};

struct Up : Stage {                // Upsample: nearest x2 + conv3x3 as four parity classes, model_3dcausal.py:200-212
  int ch;
  std::string key;
  Act run(Ctx& c, const Act& x, NormRef) override {
    const Tens& xp = x.y;
    Act r;
    r.y = c.alloc(xp.B, xp.T, 2 * xp.H, 2 * xp.W, pad8(ch), c.m->dt, ch);
    for (int py = 0; py < 2; ++py)
      for (int px = 0; px < 2; ++px) {
        Geom g;
        g.kh = 2; g.kw = 2; g.ph = 1 - py; g.pw = 1 - px; g.ph_hi = py; g.pw_hi = px;
        ConvOpts o;
        o.out = &r.y; o.ys = 1; o.ys_oh = py; o.ys_ow = px;
        conv(c, xp, c.m->conv_w(key + ".conv.weight", xp.ld, c.dry, xf_space_parity, py, px), 4 * xp.ld, c.m->f32(key + ".conv.bias", c.dry), g, ch, o);
      }
    return r;
  }
};

struct TimeUp : Stage {            // TimeUpsampleResCausal2x: v1.0 nearest as two parity classes (model_3dcausal.py:255-273);
  int ch;                          // v1.1 nearest / trilinear up-sampling, then the 27-tap convolution (model_3dcausal_v1_1.py:305-343)
  int n_up = 1;                    // num_temp_upsample: 1, 2, 4 ... along the decoder (how many frames the trilinear head covers)
  std::string key;
  CState sc, su;                   // the convolution's cache (two frames) and the trilinear interpolation's (num_temp_upsample frames)
  void states(std::vector<CState*>& v) override { v.push_back(&sc); v.push_back(&su); }
  bool time_up() const override { return true; }
  // later chunks of a tiled pass with trilinear up-sampling: [cache | x] interpolated, the frames the previous chunk already
  // delivered left out (TimeUpsampleResCausal2x._interp_v11 of the Python host; model_3dcausal_v1_1.py:325-343)
  Tens interp_cached(Ctx& c, const Tens& xp) {
    Model* m = c.m;
    const int n = n_up, T = xp.T, es = (int)esize(xp.dt);
    const int64_t fr = (int64_t)xp.H * xp.W * xp.ld;
    const int nc = c.dry ? n : su.frames;
    M_CHECK(nc >= 1, "vt_model: time up-sampler cache missing (the first chunk must come first)");
    const int Tc = nc + T;
    const char* head = c.dry ? nullptr : (const char*)su.buf->p;
    if (Tc - 2 * n >= nc) {
      Tens up = c.alloc(xp.B, 2 * Tc - 2 * n, xp.H, xp.W, xp.ld, xp.dt, xp.ld);
      if (!c.dry) M_CALL(vt_time_lerp2x_cat(head, nc, xp.p, up.p, xp.dt, xp.B, T, 2 * n, fr, c.stream));
      std::vector<int> keep;
      for (int t = Tc - 2 * n - nc; t < Tc - n - nc; ++t) keep.push_back(t);
      char* buf = m->persistent(su, (size_t)xp.B * keep.size() * fr * es, c.dry);
      gather(c, xp.p, T, buf, (int)keep.size(), 0, keep, xp.B, fr, es);
      if (!c.dry) su.frames = (int)keep.size();
      return up;
    }
    Tens xc = c.alloc(xp.B, Tc, xp.H, xp.W, xp.ld, xp.dt, xp.ld);     // [cache | x]
    std::vector<int> all;
    for (int t = 0; t < nc; ++t) all.push_back(t);
    gather(c, head, nc, xc.p, Tc, 0, all, xp.B, fr, es);
    all.clear();
    for (int t = 0; t < T; ++t) all.push_back(t);
    gather(c, xp.p, T, xc.p, Tc, nc, all, xp.B, fr, es);
    std::vector<int> keep;
    for (int t = std::max(0, Tc - 2 * n); t < Tc - n; ++t) keep.push_back(t);
    char* buf = m->persistent(su, (size_t)xp.B * keep.size() * fr * es, c.dry);   // (xc is a copy: the cache buffer is free to be rewritten)
    gather(c, xc.p, Tc, buf, (int)keep.size(), 0, keep, xp.B, fr, es);
    if (!c.dry) su.frames = (int)keep.size();
    Tens full = c.alloc(xp.B, 2 * Tc, xp.H, xp.W, xp.ld, xp.dt, xp.ld);
    if (!c.dry) M_CALL(vt_time_lerp2x(xc.p, full.p, xp.dt, xp.B, Tc, fr, c.stream));
    Tens up = c.alloc(xp.B, 2 * Tc - 2 * n, xp.H, xp.W, xp.ld, xp.dt, xp.ld);
    all.clear();
    for (int t = 2 * n; t < 2 * Tc; ++t) all.push_back(t);
    gather(c, full.p, 2 * Tc, up.p, 2 * Tc - 2 * n, 0, all, xp.B, fr, es);
    return up;
  }
  Act conv_of(Ctx& c, const Tens& up, NormRef next) {
    ConvOpts o = emit(next, VT_TPAD_REPLICATE);
    o.res = &up;
    o.res_mode = VT_RES_MIX;
    o.mix = c.m->f32(key + ".mix_factor", c.dry);
    return conv_causal(c, sc, up, c.m->conv_w(key + ".conv.conv.weight", up.ld, c.dry), 27 * up.ld, c.m->f32(key + ".conv.conv.bias", c.dry), causal3d(3, 3, 3), ch, o);
  }
  Act run_v11(Ctx& c, const Act& x, NormRef next) {
    const Tens& xp = x.y;
    const int T = xp.T;
    const int64_t fr = (int64_t)xp.H * xp.W * xp.ld;
    if (c.m->tiled && !c.m->first_chunk && c.m->cfg.interpolation_mode == 1) return conv_of(c, interp_cached(c, xp), next);
    Tens up = c.alloc(xp.B, 2 * T, xp.H, xp.W, xp.ld, xp.dt, xp.ld);
    if (c.m->cfg.interpolation_mode == 0) {                            // nearest: up[t] = x[t / 2]
      for (int j0 = 0; j0 < 2 * T; j0 += 128) {
        int32_t idx[128];
        const int n = std::min(128, 2 * T - j0);
        for (int j = 0; j < n; ++j) idx[j] = (j0 + j) / 2;
        if (!c.dry) M_CALL(vt_gather_frames(xp.p, up.p + (size_t)j0 * fr * esize(xp.dt), (int)esize(xp.dt), xp.B, fr, (int64_t)T * fr, (int64_t)2 * T * fr, idx, n, c.stream));
      }
    } else {
      // trilinear, first chunk (the only one here): the first n_up frames are interpolated on their own, the rest likewise
      // (F.interpolate on x[:, :n] and x[:, n:] separately, model_3dcausal_v1_1.py:327-341); one launch per clip and part
      const int hn = std::min(n_up, T);
      auto lerp_part = [&](int t0, int n, int out_t0) {
        Tens part = c.alloc(xp.B, n, xp.H, xp.W, xp.ld, xp.dt, xp.ld);
        for (int j0 = 0; j0 < n; j0 += 128) {
          int32_t idx[128];
          const int m = std::min(128, n - j0);
          for (int j = 0; j < m; ++j) idx[j] = t0 + j0 + j;
          if (!c.dry) M_CALL(vt_gather_frames(xp.p, part.p + (size_t)j0 * fr * esize(xp.dt), (int)esize(xp.dt), xp.B, fr, (int64_t)T * fr, (int64_t)n * fr, idx, m, c.stream));
        }
        for (int b = 0; b < xp.B; ++b)
          if (!c.dry)
            M_CALL(vt_time_lerp2x(part.p + (size_t)b * n * fr * esize(xp.dt), up.p + ((size_t)b * 2 * T + out_t0) * fr * esize(xp.dt), xp.dt, 1, n, fr, c.stream));
      };
      if (c.m->tiled) {            // first chunk of a tiled pass: keep the last n_up frames for the next chunk's interpolation
        std::vector<int> keep;
        for (int t = std::max(0, T - n_up); t < T; ++t) keep.push_back(t);
        char* buf = c.m->persistent(su, (size_t)xp.B * keep.size() * fr * esize(xp.dt), c.dry);
        gather(c, xp.p, T, buf, (int)keep.size(), 0, keep, xp.B, fr, (int)esize(xp.dt));
        if (!c.dry) su.frames = (int)keep.size();
      }
      lerp_part(0, hn, 0);
      if (T > n_up) lerp_part(n_up, T - n_up, 2 * hn);
    }
    return conv_of(c, up, next);
  }
  Act run(Ctx& c, const Act& x, NormRef next) override {
    if (c.m->v11()) return run_v11(c, x, next);
    const Tens& xp = x.y;
    Act r;
    r.y = c.alloc(xp.B, 2 * xp.T, xp.H, xp.W, pad8(ch), c.m->dt, ch);
    // the consumer's LayerNorm from the two launches' epilogues where they can take it (TimeUpsampleResCausal2x.run of the Python host)
    bool emit_ln = next.norm != nullptr;
    Tens nbuf;
    Geom g;
    g.kt = 2; g.kh = 3; g.kw = 3; g.pt = 1; g.ph = 1; g.pw = 1; g.ph_hi = 1; g.pw_hi = 1;
    const float* mf = c.m->f32(key + ".mix_factor", c.dry);
    const bool nc = c.m->noncausal();
    const std::string ck = key + ".conv" + c.m->cv();
    if (emit_ln) {
      // ask before allocating the twin (a launch that refuses -- the 512-channel up-sampler always does -- left a full-size
      // buffer in the arena's peak): the decision depends on the geometry only, the same for both parity launches
      ConvOpts o;
      Tens probe = r.y;
      probe.p = nullptr;
      o.out = &r.y; o.yt_mul = 2; o.yt_off = 0; o.ln = next; o.ln_out = &probe; o.ln_optional = true; o.plan_only = true;
      o.res = &xp; o.res_mode = VT_RES_MIX; o.mix = mf;
      Geom gp = g;
      if (nc) gp = centred(g, 1);
      emit_ln = conv(c, xp, c.m->conv_w(ck + ".weight", xp.ld, c.dry, xf_time_parity, nc ? 0 : 1, 0), 18 * xp.ld, c.m->f32(ck + ".bias", c.dry), gp, ch, o).has_n();
      if (emit_ln) nbuf = c.alloc(xp.B, 2 * xp.T, xp.H, xp.W, pad8(ch), c.m->dt, ch);
    }
    for (int par = 0; par < 2; ++par) {
      ConvOpts o;
      o.out = &r.y; o.yt_mul = 2; o.yt_off = par;
      if (emit_ln) { o.ln = next; o.ln_out = &nbuf; o.ln_optional = true; }
      o.res = &xp; o.res_mode = VT_RES_MIX; o.mix = mf;
      // causal (window ends at the output frame):  o[2j] = (W0+W1) x[j-1] + W2 x[j],  o[2j+1] = W0 x[j-1] + (W1+W2) x[j]
      // centred (TimeUpsampleRes2x, model_3dnoncausal.py:93-115):  o[2j] = W0 x[j-1] + (W1+W2) x[j],  o[2j+1] = (W0+W1) x[j] + W2 x[j+1]
      const int early = nc ? par : (par == 0 ? 1 : 0);
      Geom gp = g;
      if (nc) gp = centred(g, par == 0 ? 1 : 0);
      const Act a = conv(c, xp, c.m->conv_w(ck + ".weight", xp.ld, c.dry, xf_time_parity, early, 0), 18 * xp.ld, c.m->f32(ck + ".bias", c.dry), gp, ch, o);
      if (emit_ln && !a.has_n()) emit_ln = false;
    }
    if (emit_ln) { r.n = nbuf; r.norm = next.norm; r.silu = next.silu; }
    return r;
  }
};

bool in_list(const int32_t* v, int n, int x) {
  for (int i = 0; i < n; ++i)
    if (v[i] == x) return true;
  return false;
}

typedef std::map<std::string, std::vector<int64_t>> Shapes;     // reference state_dict key -> parameter shape
void sh_conv(Shapes& sh, const std::string& key, int cout, int cin, std::vector<int64_t> k) {
  std::vector<int64_t> w = {cout, cin};
  w.insert(w.end(), k.begin(), k.end());
  sh[key + ".weight"] = w;
  sh[key + ".bias"] = {cout};
}
bool g_group_norm = false;         // set by the builders from vt_model_config.norm_type while a graph is being built (single-threaded: vt_create)
void sh_norm(Shapes& sh, const std::string& key, int c) {
  sh[key + (g_group_norm ? ".weight" : ".norm.weight")] = {c};
  sh[key + (g_group_norm ? ".bias" : ".norm.bias")] = {c};
}
// a norm of the graph under `key`; `site` = the view its call site hands to GroupNorm (vidtok_amd/modules.py SITE_*)
void set_norm(Norm& n, const std::string& key, int site) {
  n.key = key;
  n.group = g_group_norm;
  n.site = site;
}

struct Graph {
  std::vector<std::unique_ptr<Stage>> stages;
  std::string conv_in, conv_out;
  Norm norm_out;
  int c_first = 0, c_last = 0;
  CState st_in, st_out;            // chunk state of conv_in / conv_out (two frames each)
  void states(std::vector<CState*>& v) {
    v.push_back(&st_in);
    for (auto& st : stages) st->states(v);
    v.push_back(&st_out);
  }
  // cache offsets of an overlapped decode (AutoencodingEngineV11._overlap_offsets, autoencoder_v1_1.py:307-320): 1 at latent
  // rate, doubling at each temporal up-sampler (the up-sampler itself already works at the doubled rate); 0 = no look-ahead
  void set_offsets(bool overlap) {
    int off = overlap ? 1 : 0;
    st_in.offset = off;
    for (auto& st : stages) {
      if (st->time_up()) off *= 2;
      std::vector<CState*> v;
      st->states(v);
      for (CState* s : v) s->offset = off;
    }
    st_out.offset = off;
  }
};

// `wrap`: the 3-D / 1-D convolutions sit under a Causal* wrapper (a ".conv" level in their keys); false for the non-causal family
ResBlock* res_block(Shapes& sh, const std::string& key, int cin, int cout, bool causal, bool wrap = true, int site = VT_GN_FRAME) {
  const std::string sfx0 = (causal && wrap) ? ".conv" : "";
  const std::vector<int64_t> k3 = causal ? std::vector<int64_t>{3, 3, 3} : std::vector<int64_t>{3, 3};
  const std::vector<int64_t> k1 = causal ? std::vector<int64_t>{1, 1, 1} : std::vector<int64_t>{1, 1};
  sh_norm(sh, key + ".norm1", cin);
  sh_norm(sh, key + ".norm2", cout);
  sh_conv(sh, key + ".conv1" + sfx0, cout, cin, k3);
  sh_conv(sh, key + ".conv2" + sfx0, cout, cout, k3);
  if (cin != cout) sh_conv(sh, key + ".nin_shortcut" + sfx0, cout, cin, k1);
  auto* b = new ResBlock();
  b->causal3d_ = causal; b->cin = cin; b->cout = cout;
  set_norm(b->n1, key + ".norm1", site); set_norm(b->n2, key + ".norm2", site);
  const std::string sfx = sfx0;
  b->c1.key = key + ".conv1" + sfx; b->c2.key = key + ".conv2" + sfx; b->sc.key = key + ".nin_shortcut" + sfx;
  return b;
}
TBlock* t_block(Shapes& sh, const std::string& key, int ch, bool wrap = true) {
  sh_norm(sh, key + ".norm1", ch);
  sh_norm(sh, key + ".norm2", ch);
  sh_conv(sh, key + ".conv1" + (wrap ? ".conv" : ""), ch, ch, {3});
  sh_conv(sh, key + ".conv2" + (wrap ? ".conv" : ""), ch, ch, {3});
  auto* b = new TBlock();
  b->ch = ch;
  // causal temporal blocks normalise single positions (model_3dcausal.py:476-487), the non-causal ones pixels over time (model_3dnoncausal.py:228-236)
  set_norm(b->n1, key + ".norm1", wrap ? -1 : VT_GN_PIXEL); set_norm(b->n2, key + ".norm2", wrap ? -1 : VT_GN_PIXEL);
  b->c1.key = key + ".conv1"; b->c2.key = key + ".conv2";
  return b;
}
Attn* attn(Shapes& sh, const std::string& key, int ch, bool wrap = true) {
  sh_norm(sh, key + ".norm", ch);
  for (const char* n : {".q", ".k", ".v", ".proj_out"}) sh_conv(sh, key + n + (wrap ? ".conv" : ""), ch, ch, {1, 1, 1});
  auto* a = new Attn();
  a->ch = ch; a->key = key;
  set_norm(a->n, key + ".norm", wrap ? VT_GN_FRAME : VT_GN_CLIP);
  return a;
}

Graph build_encoder(const vt_model_config& cf, Shapes& sh) {
  Graph g;
  g_group_norm = cf.norm_type == 1;
  const int L = cf.num_resolutions;
  const bool wrap = cf.version != 2;
  const std::string cv = wrap ? ".conv" : "";
  int block_in = cf.ch;
  for (int i = 0; i < L; ++i) {
    block_in = cf.ch * (i == 0 ? 1 : cf.ch_mult[i - 1]);
    const int block_out = cf.ch * cf.ch_mult[i];
    const std::string d = "encoder.down." + std::to_string(i), dt = "encoder.down_temporal." + std::to_string(i);
    for (int b = 0; b < cf.num_res_blocks; ++b) {
      g.stages.emplace_back(res_block(sh, d + ".block." + std::to_string(b), block_in, block_out, false));
      g.stages.emplace_back(t_block(sh, dt + ".block." + std::to_string(b), block_out, wrap));
      block_in = block_out;
    }
    if (in_list(cf.spatial_ds, cf.n_spatial_ds, i)) {
      auto* s = new Down();
      s->ch = block_in; s->key = d + ".downsample";
      sh_conv(sh, d + ".downsample.conv", block_in, block_in, {3, 3});
      g.stages.emplace_back(s);
      if (in_list(cf.tempo_ds, cf.n_tempo_ds, i)) {
        auto* t = new TimeDown();
        t->ch = block_in; t->key = dt + ".downsample";
        sh_conv(sh, dt + ".downsample.conv" + cv, block_in, block_in, {3, 3, 3});
        sh[dt + ".downsample.mix_factor"] = {1};
        g.stages.emplace_back(t);
      }
    }
  }
  g.stages.emplace_back(res_block(sh, "encoder.mid.block_1", block_in, block_in, true, wrap, wrap ? VT_GN_FRAME : VT_GN_CLIP));
  g.stages.emplace_back(attn(sh, "encoder.mid.attn_1", block_in, wrap));
  g.stages.emplace_back(res_block(sh, "encoder.mid.block_2", block_in, block_in, true, wrap, wrap ? VT_GN_FRAME : VT_GN_CLIP));
  g.conv_in = "encoder.conv_in" + cv; g.conv_out = "encoder.conv_out" + cv;
  set_norm(g.norm_out, "encoder.norm_out", wrap ? VT_GN_FRAME : VT_GN_CLIP);
  g.c_first = cf.ch; g.c_last = block_in;
  sh_conv(sh, g.conv_in, cf.ch, cf.in_channels, {3, 3, 3});
  sh_conv(sh, g.conv_out, cf.double_z ? 2 * cf.z_channels : cf.z_channels, block_in, {3, 3, 3});
  sh_norm(sh, g.norm_out.key, block_in);
  return g;
}

Graph build_decoder(const vt_model_config& cf, Shapes& sh) {
  Graph g;
  g_group_norm = cf.norm_type == 1;
  const int L = cf.num_resolutions;
  const bool wrap = cf.version != 2;
  const std::string cv = wrap ? ".conv" : "";
  int block_in = cf.ch * cf.ch_mult[L - 1];
  int n_up = 1;
  g.c_first = block_in;
  g.stages.emplace_back(res_block(sh, "decoder.mid.block_1", block_in, block_in, true, wrap, wrap ? VT_GN_FRAME : VT_GN_CLIP));
  g.stages.emplace_back(attn(sh, "decoder.mid.attn_1", block_in, wrap));
  g.stages.emplace_back(res_block(sh, "decoder.mid.block_2", block_in, block_in, true, wrap, wrap ? VT_GN_FRAME : VT_GN_CLIP));
  for (int i = L - 1; i >= 0; --i) {
    const int block_out = cf.ch * cf.ch_mult[i];
    const std::string u = "decoder.up." + std::to_string(i), ut = "decoder.up_temporal." + std::to_string(i);
    for (int b = 0; b < cf.num_res_blocks + 1; ++b) {
      g.stages.emplace_back(res_block(sh, u + ".block." + std::to_string(b), block_in, block_out, false));
      g.stages.emplace_back(t_block(sh, ut + ".block." + std::to_string(b), block_out, wrap));
      block_in = block_out;
    }
    if (in_list(cf.spatial_us, cf.n_spatial_us, i)) {
      auto* s = new Up();
      s->ch = block_in; s->key = u + ".upsample";
      sh_conv(sh, u + ".upsample.conv", block_in, block_in, {3, 3});
      g.stages.emplace_back(s);
      if (in_list(cf.tempo_us, cf.n_tempo_us, i)) {
        auto* t = new TimeUp();
        t->ch = block_in; t->key = ut + ".upsample";
        t->n_up = n_up;
        n_up *= 2;
        sh_conv(sh, ut + ".upsample.conv" + cv, block_in, block_in, {3, 3, 3});
        sh[ut + ".upsample.mix_factor"] = {1};
        g.stages.emplace_back(t);
      }
    }
  }
  g.conv_in = "decoder.conv_in" + cv; g.conv_out = "decoder.conv_out" + cv;
  set_norm(g.norm_out, "decoder.norm_out", wrap ? VT_GN_FRAME : VT_GN_CLIP);
  g.c_last = block_in;
  sh_conv(sh, g.conv_in, g.c_first, cf.z_channels, {3, 3, 3});
  sh_conv(sh, g.conv_out, cf.out_ch, block_in, {3, 3, 3});
  sh_norm(sh, g.norm_out.key, block_in);
  return g;
}

// conv_in -> stages -> norm_out + SiLU -> conv_out (fp32 NCTHW); each stage is told which norm its consumer starts with
void run_graph(Ctx& c, Graph& g, const Tens& x_in, int cout_final, float* out_ncthw, int t_trim) {
  Model* m = c.m;
  int which = 1;                   // x_in lives in arena 0
  c.cur = &m->arena[which];
  c.cur->reset();
  const NormRef first = g.stages[0]->first_norm(c);
  const Geom g333 = m->noncausal() ? centred(causal3d(3, 3, 3), 1) : causal3d(3, 3, 3);       // conv_in / conv_out: causal, or Conv3d padding 1
  Act h = conv_causal(c, g.st_in, x_in, m->conv_w(g.conv_in + ".weight", x_in.ld, c.dry), 27 * x_in.ld, m->f32(g.conv_in + ".bias", c.dry), g333, g.c_first, emit(first, m->tpad()));
  for (size_t i = 0; i < g.stages.size(); ++i) {
    which ^= 1;
    c.cur = &m->arena[which];
    c.cur->reset();                // holds the input of the previous stage: no longer needed
    const NormRef next = i + 1 < g.stages.size() ? g.stages[i + 1]->first_norm(c) : g.norm_out.ref(true);
    h = g.stages[i]->run(c, h, next);
  }
  which ^= 1;
  c.cur = &m->arena[which];
  c.cur->reset();
  const Tens hn = g.norm_out.apply(c, h, true, g.c_last);
  ConvOpts o;
  o.ncthw = out_ncthw ? out_ncthw : (float*)16;      // dry runs pass no buffer
  o.t_trim = t_trim;
  o.tmode = m->tpad();
  conv_causal(c, g.st_out, hn, m->conv_w(g.conv_out + ".weight", hn.ld, c.dry), 27 * hn.ld, m->f32(g.conv_out + ".bias", c.dry), g333, cout_final, o);
}

int front_pad(const vt_model_config& cf, int T) {      // EncoderCausal3DPadding.forward: v1.0 pads f - 1 frames, v1.1 up to a multiple of f
  const int f = cf.time_downsample_factor;
  if (T % f == 0 || cf.version == 2) return 0;         // Encoder3D pads nothing (T must be a multiple of f: vt_encode checks)
  return cf.version == 1 ? f - T % f : f - 1;
}

void encode_impl(Model* m, Graph& g, const float* x, int B, int T, int H, int W, float* h_out, hipStream_t stream, bool dry) {
  Ctx c{m, stream, dry, &m->arena[0]};
  m->arena[0].reset();
  const int npad = front_pad(m->cfg, T);
  Tens xin = c.alloc(B, T + npad, H, W, pad8(m->cfg.in_channels), m->dt, pad8(m->cfg.in_channels));   // the kernel zero-fills the pad channels itself
  if (!dry) M_CALL(vt_ncthw_to_ndhwc(x, xin.p, m->dt, B, m->cfg.in_channels, T, H, W, xin.ld, npad, stream));
  const int cout = m->cfg.double_z ? 2 * m->cfg.z_channels : m->cfg.z_channels;
  run_graph(c, g, xin, cout, h_out, 0);
}

void decode_impl(Model* m, Graph& g, const float* z, int B, int T, int H, int W, float* x_out, hipStream_t stream, bool dry) {
  Ctx c{m, stream, dry, &m->arena[0]};
  m->arena[0].reset();
  Tens zin = c.alloc(B, T, H, W, pad8(m->cfg.z_channels), m->dt, pad8(m->cfg.z_channels));
  if (!dry) M_CALL(vt_ncthw_to_ndhwc(z, zin.p, m->dt, B, m->cfg.z_channels, T, H, W, zin.ld, 0, stream));
  run_graph(c, g, zin, m->cfg.out_ch, x_out, m->cfg.version != 0 ? 0 : m->cfg.time_downsample_factor - 1);   // v1.1 keeps every frame (the caller drops the padding's), the non-causal decoder drops nothing
}

// ---- temporal tiling of the v1.1 tokenizers (AutoencodingEngineV11.tile_encode / tile_decode, autoencoder_v1_1.py:218-331) -----------
// [[0, 1], [1, 1 + c], [1 + c, 1 + 2 c], ...]: the first chunk is the single leading frame (build_chunk_start_end, :218-228)
std::vector<std::pair<int, int>> chunk_list(int t, int step) {
  std::vector<std::pair<int, int>> v{{0, 1}};
  for (int start = 1; start < t;) {
    const int end = std::min(t, start + step);
    v.emplace_back(start, end);
    start = end;
  }
  return v;
}
int ceil_div(int a, int b) { return (a + b - 1) / b; }

struct ChunkBufs {                 // staging of one chunk in front of the arenas: its input frames and its result, contiguous NCTHW
  float* in = nullptr;
  float* out = nullptr;
  size_t bytes = 0;
};
ChunkBufs chunk_bufs(void* ws, size_t in_elems, size_t out_elems) {
  ChunkBufs b;
  const size_t ib = (in_elems * 4 + 255) & ~(size_t)255, ob = (out_elems * 4 + 255) & ~(size_t)255;
  b.in = (float*)ws;
  b.out = (float*)((char*)ws + ib);
  b.bytes = ib + ob;
  return b;
}
void reset_states(Model& m) {
  for (CState* st : m.states) st->frames = 0;
}

// The handle's tiled-pass state (tiled / first_chunk flags, the decoder's doubling cache offsets), restored on EVERY exit path of a
// tiled walk -- a walk that throws half way (an M_CHECK in Attn, update_cache, interp_cached, a workspace that is too small) must
// not leave a handle on which the next plain vt_encode / vt_decode runs as "a chunk" (extra allocations, cache reads, "causal cache
// missing")
struct TileScope {
  Model* m;
  Graph* g;
  TileScope(Model* m_, Graph& g_, bool overlap) : m(m_), g(&g_) {
    g->set_offsets(overlap);
    m->tiled = true;
  }
  ~TileScope() {
    m->tiled = false;
    m->first_chunk = true;
    g->set_offsets(false);
  }
  TileScope(const TileScope&) = delete;
  TileScope& operator=(const TileScope&) = delete;
};

// dry walks (workspace sizing, vt_prepare) run on empty arenas: the caller's arenas come back on every exit path
struct ArenaScope {
  Model* m;
  Arena save[2];
  explicit ArenaScope(Model* m_) : m(m_), save{m_->arena[0], m_->arena[1]} {
    m->arena[0] = Arena();
    m->arena[1] = Arena();
  }
  ~ArenaScope() {
    m->arena[0] = save[0];
    m->arena[1] = save[1];
  }
  ArenaScope(const ArenaScope&) = delete;
  ArenaScope& operator=(const ArenaScope&) = delete;
};

// every chunk through the encoder in order (the module caches carry the causal state); a chunk of n frames is front-padded
// to a multiple of f by the encoder and yields ceil(n / f) latent frames, written to h_out at their place
void tile_encode_impl(Model* m, Graph& g, const float* x, int B, int T, int H, int W, int t_chunk, float* h_out, char* ws, hipStream_t stream, bool dry) {
  const vt_model_config& cf = m->cfg;
  const int f = cf.time_downsample_factor, cin = cf.in_channels, cz = cf.double_z ? 2 * cf.z_channels : cf.z_channels;
  const int Hz = H >> cf.n_spatial_ds, Wz = W >> cf.n_spatial_ds;
  const auto chunks = chunk_list(T, t_chunk);
  int tz = 0, nmax = 1;
  for (auto& ch : chunks) {
    tz += ceil_div(ch.second - ch.first, f);
    nmax = std::max(nmax, ch.second - ch.first);
  }
  const ChunkBufs cb = chunk_bufs(ws, (size_t)B * cin * nmax * H * W, (size_t)B * cz * ceil_div(nmax, f) * Hz * Wz);
  TileScope scope(m, g, false);
  if (!dry) reset_states(*m);
  int done = 0;
  std::set<int> sized;                                     // sizing: the first chunk and one chunk of every other length (a short trailing
  for (size_t i = 0; i < chunks.size(); ++i) {             // chunk takes branches the largest one never sees)
    const int n = chunks[i].second - chunks[i].first, nz = ceil_div(n, f);
    if (dry && i != 0 && !sized.insert(n).second) continue;
    m->first_chunk = i == 0;
    if (!dry) M_CALL(vt_ncthw_copy_frames(x, cb.in, B * cin, T, n, chunks[i].first, 0, n, (int64_t)H * W, 0, stream));
    encode_impl(m, g, cb.in, B, n, H, W, cb.out, stream, dry);
    if (!dry) M_CALL(vt_ncthw_copy_frames(cb.out, h_out, B * cz, nz, tz, 0, done, nz, (int64_t)Hz * Wz, 0, stream));
    done += nz;
  }
}

// chunks of t_chunk_dec latent frames decoded in order, each with one look-ahead latent frame when `overlap` (its f trailing
// output frames are dropped, the module caches stop `cache_offset` frames early); f * Tz output frames in all
void tile_decode_impl(Model* m, Graph& g, const float* z, int B, int Tz, int Hz, int Wz, int t_chunk_dec, bool overlap, float* x_out, char* ws, hipStream_t stream, bool dry) {
  const vt_model_config& cf = m->cfg;
  const int f = cf.time_downsample_factor, zc = cf.z_channels, oc = cf.out_ch;
  const int H = Hz << cf.n_spatial_us, W = Wz << cf.n_spatial_us;
  const auto chunks = chunk_list(Tz, t_chunk_dec);
  int nmax = 1;
  for (auto& ch : chunks) nmax = std::max(nmax, ch.second - ch.first + ((overlap && ch.second + 1 <= Tz) ? 1 : 0));
  const ChunkBufs cb = chunk_bufs(ws, (size_t)B * zc * nmax * Hz * Wz, (size_t)B * oc * nmax * f * H * W);
  TileScope scope(m, g, overlap);
  if (!dry) reset_states(*m);
  int done = 0;
  std::set<int> sized;                                     // as in tile_encode_impl: every distinct chunk length is walked once
  for (size_t i = 0; i < chunks.size(); ++i) {
    const bool look = overlap && chunks[i].second + 1 <= Tz;
    const int nl = chunks[i].second - chunks[i].first + (look ? 1 : 0), n = nl * f - (look ? f : 0);
    if (dry && i != 0 && !sized.insert(nl).second) continue;
    m->first_chunk = i == 0;
    if (!dry) M_CALL(vt_ncthw_copy_frames(z, cb.in, B * zc, Tz, nl, chunks[i].first, 0, nl, (int64_t)Hz * Wz, 0, stream));
    decode_impl(m, g, cb.in, B, nl, Hz, Wz, cb.out, stream, dry);
    if (!dry) M_CALL(vt_ncthw_copy_frames(cb.out, x_out, B * oc, nl * f, Tz * f, 0, done, n, (int64_t)H * W, 0, stream));
    done += n;
  }
}

}  // namespace

struct vt_model {
  Model m;
  Graph enc, dec;
  Shapes shapes;                  // every parameter of the graph: the keys vt_weight_name lists, with their shapes
  std::vector<std::string> names;
};

extern "C" int vt_create(const vt_model_config* cfg, int32_t compute_dtype, vt_model** out) {
  try {
    M_CHECK(cfg != nullptr && out != nullptr, "vt_create: null argument");
    M_CHECK(compute_dtype == VT_BF16 || compute_dtype == VT_F16 || compute_dtype == VT_F32 || compute_dtype == VT_BF16X3,
            "vt_create: compute dtype must be VT_BF16, VT_F16, VT_F32 or VT_BF16X3");
    M_CHECK(cfg->version >= 0 && cfg->version <= 2, "vt_create: version 0 (v1.0 causal), 1 (v1.1 causal) or 2 (non-causal Encoder3D / Decoder3D)");
    M_CHECK(cfg->interpolation_mode == 0 || (cfg->interpolation_mode == 1 && cfg->version == 1), "vt_create: interpolation_mode 0 (nearest) or, for v1.1, 1 (trilinear)");
    M_CHECK(cfg->num_resolutions >= 1 && cfg->num_resolutions <= 8 && cfg->num_res_blocks >= 1 && cfg->ch > 0, "vt_create: bad level / block counts");
    M_CHECK(cfg->n_spatial_ds <= 8 && cfg->n_tempo_ds <= 8 && cfg->n_spatial_us <= 8 && cfg->n_tempo_us <= 8 && cfg->n_levels <= 8, "vt_create: list too long");
    M_CHECK(cfg->time_downsample_factor == 2 || cfg->time_downsample_factor == 4 || cfg->time_downsample_factor == 8, "vt_create: time_downsample_factor must be 2, 4 or 8");
    M_CHECK(cfg->norm_type == 0 || cfg->norm_type == 1, "vt_create: norm_type 0 (layernorm) or 1 (groupnorm)");
    M_CHECK(cfg->regularizer == 0 || cfg->regularizer == 1, "vt_create: regularizer 0 (KL) or 1 (FSQ)");
    if (cfg->regularizer == 1) {
      const int ncb = cfg->fsq_num_codebooks > 1 ? cfg->fsq_num_codebooks : 1, eff = cfg->n_levels * ncb;
      const int dim = cfg->fsq_dim > 0 ? cfg->fsq_dim : eff;
      M_CHECK(cfg->n_levels >= 1 && !cfg->double_z && cfg->z_channels == dim && cfg->fsq_num_codebooks >= 0 && cfg->fsq_dim >= 0 && eff <= 1024 && dim <= 1024,
              "vt_create: FSQ needs double_z = 0 and z_channels = dim (%d; len(levels) * num_codebooks = %d unless fsq_dim says otherwise)", dim, eff);
    }
    for (int i = 0; i < cfg->n_tempo_ds; ++i) M_CHECK(in_list(cfg->spatial_ds, cfg->n_spatial_ds, cfg->tempo_ds[i]), "vt_create: a temporal down-sampler sits behind a spatial one (tempo_ds must be a subset of spatial_ds)");
    for (int i = 0; i < cfg->n_tempo_us; ++i) M_CHECK(in_list(cfg->spatial_us, cfg->n_spatial_us, cfg->tempo_us[i]), "vt_create: tempo_us must be a subset of spatial_us");
    auto* h = new vt_model();
    h->m.cfg = *cfg;
    h->m.x3 = compute_dtype == VT_BF16X3;              // fp32 storage, split-bf16 convolutions (set_compute_dtype("bf16x3") of the Python host)
    h->m.dt = h->m.x3 ? VT_F32 : compute_dtype;
    h->enc = build_encoder(*cfg, h->shapes);
    h->dec = build_decoder(*cfg, h->shapes);
    h->enc.states(h->m.states);
    h->dec.states(h->m.states);
    if (h->m.fsq_proj()) {         // nn.Linear(dim, effective_codebook_dim) / back (regularizers.py:137-139)
      const int64_t eff = h->m.fsq_eff(), dim = cfg->fsq_dim;
      h->shapes["regularization.project_in.weight"] = {eff, dim};
      h->shapes["regularization.project_in.bias"] = {eff};
      h->shapes["regularization.project_out.weight"] = {dim, eff};
      h->shapes["regularization.project_out.bias"] = {dim};
    }
    for (const auto& kv : h->shapes) h->names.push_back(kv.first);
    *out = h;
    return VT_OK;
  } catch (const Fail& f) {
    return f.code;
  } catch (const std::exception& e) {
    vt_set_error("vt_create: %s", e.what());
    return VT_ERR_ARG;
  }
}

extern "C" int vt_model_config_size(void) { return (int)sizeof(vt_model_config); }

extern "C" int vt_destroy(vt_model* h) {
  delete h;
  return VT_OK;
}

extern "C" int vt_load_weight(vt_model* h, const char* ref_key, const float* data_host, const int64_t* shape, int32_t ndim) {
  try {
    M_CHECK(h && ref_key && data_host && shape && ndim >= 1 && ndim <= 5, "vt_load_weight: bad argument");
    const auto want = h->shapes.find(ref_key);
    M_CHECK(want != h->shapes.end(), "vt_load_weight: '%s' is not a parameter of this model (vt_weight_name lists them)", ref_key);
    bool same = (int)want->second.size() == ndim;
    for (int i = 0; same && i < ndim; ++i) same = want->second[(size_t)i] == shape[i];
    M_CHECK(same, "vt_load_weight: '%s' has the wrong shape (%d dims, first %lld; expected %zu dims, first %lld)", ref_key, ndim,
            (long long)shape[0], want->second.size(), (long long)want->second[0]);
    Param p;
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) {
      M_CHECK(shape[i] > 0, "vt_load_weight: '%s' has an empty dimension", ref_key);
      p.shape.push_back(shape[i]);
      n *= shape[i];
    }
    p.data.assign(data_host, data_host + n);
    h->m.params[ref_key] = std::move(p);
    // packed copies of an earlier version of THIS tensor are stale ("f32:<key>", "w:<key>:..."): freed here, after the
    // device has finished whatever was queued on them
    const std::string fk = "f32:" + std::string(ref_key), wk = "w:" + std::string(ref_key) + ":";
    bool synced = false;
    for (auto it = h->m.packed.begin(); it != h->m.packed.end();) {
      if (it->first == fk || it->first.compare(0, wk.size(), wk) == 0) {
        if (!synced) {
          M_HIP(hipDeviceSynchronize());
          synced = true;
        }
        it = h->m.packed.erase(it);
      } else {
        ++it;
      }
    }
    return VT_OK;
  } catch (const Fail& f) {
    return f.code;
  } catch (const std::exception& e) {
    vt_set_error("vt_load_weight: %s", e.what());
    return VT_ERR_ARG;
  }
}

namespace {
void check_loaded_all(vt_model* h) {
  for (const std::string& k : h->names) (void)h->m.param(k);
}
}  // namespace

extern "C" int vt_latent_dims(const vt_model* h, int32_t T, int32_t H, int32_t W, int32_t* out4) {
  if (!h || !out4) {
    vt_set_error("vt_latent_dims: null argument");
    return VT_ERR_ARG;
  }
  const vt_model_config& cf = h->m.cfg;
  const int Tp = T + front_pad(cf, T);
  out4[0] = cf.double_z ? 2 * cf.z_channels : cf.z_channels;
  out4[1] = Tp >> cf.n_tempo_ds;
  out4[2] = H >> cf.n_spatial_ds;
  out4[3] = W >> cf.n_spatial_ds;
  return VT_OK;
}

extern "C" int64_t vt_workspace_bytes(vt_model* h, int32_t B, int32_t T, int32_t H, int32_t W) {
  try {
    M_CHECK(h && B > 0 && T > 0 && H > 0 && W > 0, "vt_workspace_bytes: bad argument");
    Model& m = h->m;
    ArenaScope arenas(&m);
    m.expected.clear();
    encode_impl(&m, h->enc, nullptr, B, T, H, W, nullptr, nullptr, true);
    int32_t ld[4];
    vt_latent_dims(h, T, H, W, ld);
    decode_impl(&m, h->dec, nullptr, B, ld[1], ld[2], ld[3], nullptr, nullptr, true);
    const size_t peak = std::max(m.arena[0].peak, m.arena[1].peak);
    return (int64_t)(2 * ((peak + 255) & ~(size_t)255) + 512);
  } catch (const Fail& f) {
    return -1;
  } catch (const std::exception& e) {
    vt_set_error("vt_workspace_bytes: %s", e.what());
    return -1;
  }
}

// Pack and upload every weight now (host-side repacking + blocking copies), so that the first vt_encode / vt_decode is as
// asynchronous as the later ones -- e.g. before capturing a stream.  Walks both graphs dry on a minimal clip.
extern "C" int vt_prepare(vt_model* h) {
  try {
    M_CHECK(h != nullptr, "vt_prepare: null handle");
    check_loaded_all(h);
    Model& m = h->m;
    Arena save[2] = {m.arena[0], m.arena[1]};
    m.arena[0] = Arena();
    m.arena[1] = Arena();
    m.prepare = true;
    const int s = 1 << m.cfg.n_spatial_ds, T = m.cfg.time_downsample_factor;
    int32_t ld[4];
    vt_latent_dims(h, T, 8 * s, 8 * s, ld);
    try {
      encode_impl(&m, h->enc, nullptr, 1, T, 8 * s, 8 * s, nullptr, nullptr, true);
      decode_impl(&m, h->dec, nullptr, 1, ld[1], ld[2], ld[3], nullptr, nullptr, true);
    } catch (...) {
      m.prepare = false;
      m.arena[0] = save[0];
      m.arena[1] = save[1];
      throw;
    }
    m.prepare = false;
    m.arena[0] = save[0];
    m.arena[1] = save[1];
    if (m.fsq_proj())              // FSQ projections: uploaded with the rest
      for (const char* k : {"regularization.project_in.weight", "regularization.project_in.bias", "regularization.project_out.weight", "regularization.project_out.bias"})
        (void)m.f32(k, false);
    return VT_OK;
  } catch (const Fail& f) {
    return f.code;
  } catch (const std::exception& e) {
    vt_set_error("vt_prepare: %s", e.what());
    return VT_ERR_ARG;
  }
}

// the reference state_dict keys of the model's parameters (sorted) with their shapes: a loader walks them
extern "C" int vt_weight_count(vt_model* h) { return h ? (int)h->names.size() : 0; }
extern "C" const char* vt_weight_name(vt_model* h, int32_t i) {
  if (!h || i < 0 || i >= (int)h->names.size()) return nullptr;
  return h->names[(size_t)i].c_str();
}
extern "C" int vt_weight_shape(vt_model* h, int32_t i, int64_t* shape5, int32_t* ndim) {
  if (!h || !shape5 || !ndim || i < 0 || i >= (int)h->names.size()) {
    vt_set_error("vt_weight_shape: bad argument");
    return VT_ERR_ARG;
  }
  const std::vector<int64_t>& sh = h->shapes[h->names[(size_t)i]];
  *ndim = (int32_t)sh.size();
  for (size_t k = 0; k < sh.size(); ++k) shape5[k] = sh[k];
  return VT_OK;
}

namespace {
// every tensor the graph reads is there before anything is launched (prefix: "encoder." / "decoder.")
void check_loaded(vt_model* h, const char* prefix) {
  for (const std::string& k : h->names)
    if (k.compare(0, strlen(prefix), prefix) == 0) (void)h->m.param(k);
}
void bind_workspace(Model& m, void* ws, int64_t bytes, size_t skip = 0) {
  M_CHECK(ws != nullptr && bytes >= 1024 && (reinterpret_cast<uintptr_t>(ws) & 255) == 0, "vt_model: workspace must be a 256-byte aligned device buffer");
  M_CHECK((size_t)bytes > skip + 1024, "vt_model: workspace too small (%lld bytes; ask vt_workspace_bytes / vt_tile_workspace_bytes)", (long long)bytes);
  const size_t half = (((size_t)bytes - skip) / 2) & ~(size_t)255;
  m.arena[0].base = (char*)ws + skip; m.arena[0].cap = half; m.arena[0].peak = 0;
  m.arena[1].base = (char*)ws + skip + half; m.arena[1].cap = half; m.arena[1].peak = 0;
}
// bytes the chunk staging buffers of a tiled pass take at the front of the workspace
// (sized for the largest chunk the schedule can hold, whatever the clip length: t_chunk_enc frames in, t_chunk_dec + 1 latent frames)
size_t tile_staging_bytes(const vt_model_config& cf, int B, int H, int W, int t_chunk_enc) {
  const int f = cf.time_downsample_factor, cz = cf.double_z ? 2 * cf.z_channels : cf.z_channels;
  const int Hz = H >> cf.n_spatial_ds, Wz = W >> cf.n_spatial_ds;
  const int ne = std::max(1, t_chunk_enc);
  const size_t enc = chunk_bufs(nullptr, (size_t)B * cf.in_channels * ne * H * W, (size_t)B * cz * ceil_div(ne, f) * Hz * Wz).bytes;
  const int nd = std::max(1, t_chunk_enc / f) + 1;
  const size_t dec = chunk_bufs(nullptr, (size_t)B * cf.z_channels * nd * Hz * Wz, (size_t)B * cf.out_ch * nd * f * H * W).bytes;
  return std::max(enc, dec);
}
}  // namespace

extern "C" int vt_encode(vt_model* h, const float* x, int32_t B, int32_t T, int32_t H, int32_t W, float* h_out, void* workspace,
                         int64_t workspace_bytes, vt_stream stream) {
  try {
    M_CHECK(h && x && h_out && B > 0 && T > 0 && H > 0 && W > 0, "vt_encode: bad argument");
    const int ds = 1 << h->m.cfg.n_spatial_ds;
    M_CHECK(H % ds == 0 && W % ds == 0, "vt_encode: H and W must be multiples of %d", ds);
    M_CHECK(!h->m.noncausal() || T % h->m.cfg.time_downsample_factor == 0, "vt_encode: the non-causal encoder takes clips whose length is a multiple of %d",
            h->m.cfg.time_downsample_factor);
    check_loaded(h, "encoder.");
    bind_workspace(h->m, workspace, workspace_bytes);
    encode_impl(&h->m, h->enc, x, B, T, H, W, h_out, reinterpret_cast<hipStream_t>(stream), false);
    return VT_OK;
  } catch (const Fail& f) {
    return f.code;
  } catch (const std::exception& e) {
    vt_set_error("vt_encode: %s", e.what());
    return VT_ERR_ARG;
  }
}

extern "C" int vt_decode(vt_model* h, const float* z, int32_t B, int32_t Tz, int32_t Hz, int32_t Wz, float* x_out, void* workspace,
                         int64_t workspace_bytes, vt_stream stream) {
  try {
    M_CHECK(h && z && x_out && B > 0 && Tz > 0 && Hz > 0 && Wz > 0, "vt_decode: bad argument");
    M_CHECK(h->m.cfg.version != 0 || (Tz << h->m.cfg.n_tempo_us) > h->m.cfg.time_downsample_factor - 1, "vt_decode: too few latent frames");
    check_loaded(h, "decoder.");
    bind_workspace(h->m, workspace, workspace_bytes);
    decode_impl(&h->m, h->dec, z, B, Tz, Hz, Wz, x_out, reinterpret_cast<hipStream_t>(stream), false);
    return VT_OK;
  } catch (const Fail& f) {
    return f.code;
  } catch (const std::exception& e) {
    vt_set_error("vt_decode: %s", e.what());
    return VT_ERR_ARG;
  }
}

extern "C" int vt_regularize_kl(vt_model* h, const float* moments, const float* noise, float* z, float* kl_out, int32_t B, int32_t Tz,
                                int32_t Hz, int32_t Wz, vt_stream stream) {
  if (!h || h->m.cfg.regularizer != 0) {
    vt_set_error("vt_regularize_kl: the handle's regularizer is not the diagonal Gaussian");
    return VT_ERR_ARG;
  }
  return vt_kl_sample(moments, noise, z, kl_out, B, h->m.cfg.z_channels, (int64_t)Tz * Hz * Wz, stream);
}

// FSQRegularizer.forward without the auxiliary loss (regularizers.py:206-230,247-268): [project_in ->] bound / round / pack per
// codebook [-> project_out]; indices [B][T'][H'][W'] (num_codebooks c > 1: [B][T'][H'][W'][c])
extern "C" int vt_regularize_fsq(vt_model* h, const float* pre, float* z, int32_t* indices, int32_t B, int32_t Tz, int32_t Hz, int32_t Wz,
                                 vt_stream stream) {
  if (!h || h->m.cfg.regularizer != 1) {
    vt_set_error("vt_regularize_fsq: the handle's regularizer is not FSQ");
    return VT_ERR_ARG;
  }
  try {
    Model& m = h->m;
    const int64_t S = (int64_t)Tz * Hz * Wz;
    const int ncb = m.fsq_ncb();
    if (!m.fsq_proj()) return vt_fsq_quantize_cb(pre, z, indices, m.cfg.levels, m.cfg.n_levels, B, ncb, S, stream);
    const int eff = m.fsq_eff(), dim = m.cfg.fsq_dim;
    float* hp = m.reg_scratch((size_t)2 * B * eff * S);
    float* codes = hp + (size_t)B * eff * S;
    M_CALL(vt_channel_linear(pre, m.f32("regularization.project_in.weight", false), m.f32("regularization.project_in.bias", false), hp, B, dim, eff, S, stream));
    M_CALL(vt_fsq_quantize_cb(hp, codes, indices, m.cfg.levels, m.cfg.n_levels, B, ncb, S, stream));
    M_CALL(vt_channel_linear(codes, m.f32("regularization.project_out.weight", false), m.f32("regularization.project_out.bias", false), z, B, eff, dim, S, stream));
    return VT_OK;
  } catch (const Fail& f) {
    return f.code;
  } catch (const std::exception& e) {
    vt_set_error("vt_regularize_fsq: %s", e.what());
    return VT_ERR_ARG;
  }
}

// FSQ: token indices -> latent codes (AutoencodingEngine.indices_to_latent / decode(..., decode_from_indices=True),
// reference autoencoder.py:205-229; FSQRegularizer.indices_to_codes with project_out, regularizers.py:180-198): z then goes to vt_decode
extern "C" int vt_indices_to_latent(vt_model* h, const int32_t* indices, float* z, int32_t B, int32_t Tz, int32_t Hz, int32_t Wz, vt_stream stream) {
  if (!h || h->m.cfg.regularizer != 1) {
    vt_set_error("vt_indices_to_latent: the handle's regularizer is not FSQ");
    return VT_ERR_ARG;
  }
  try {
    Model& m = h->m;
    const int64_t S = (int64_t)Tz * Hz * Wz;
    const int ncb = m.fsq_ncb();
    if (!m.fsq_proj()) return vt_fsq_indices_to_codes_cb(indices, z, m.cfg.levels, m.cfg.n_levels, B, ncb, S, stream);
    const int eff = m.fsq_eff(), dim = m.cfg.fsq_dim;
    float* codes = m.reg_scratch((size_t)2 * B * eff * S);
    M_CALL(vt_fsq_indices_to_codes_cb(indices, codes, m.cfg.levels, m.cfg.n_levels, B, ncb, S, stream));
    M_CALL(vt_channel_linear(codes, m.f32("regularization.project_out.weight", false), m.f32("regularization.project_out.bias", false), z, B, eff, dim, S, stream));
    return VT_OK;
  } catch (const Fail& f) {
    return f.code;
  } catch (const std::exception& e) {
    vt_set_error("vt_indices_to_latent: %s", e.what());
    return VT_ERR_ARG;
  }
}

// forget the chunk state of a tiled pass (vt_tile_encode / vt_tile_decode do it themselves at the start of a clip) and give the
// cache buffers back: SYNCHRONISES the device first -- work queued on them may still be running
extern "C" int vt_reset_cache(vt_model* h) {
  if (!h) {
    vt_set_error("vt_reset_cache: null handle");
    return VT_ERR_ARG;
  }
  if (hipDeviceSynchronize() != hipSuccess) {
    vt_set_error("vt_reset_cache: hipDeviceSynchronize failed");
    return VT_ERR_HIP;
  }
  for (CState* st : h->m.states) {
    st->frames = 0;
    st->cap = 0;
    st->buf.reset();
  }
  h->m.retired.clear();
  return VT_OK;
}

extern "C" int32_t vt_tile_latent_frames(const vt_model* h, int32_t T, int32_t t_chunk_enc) {
  if (!h || T <= 0 || t_chunk_enc <= 0) return -1;
  int tz = 0;
  for (auto& ch : chunk_list(T, t_chunk_enc)) tz += ceil_div(ch.second - ch.first, h->m.cfg.time_downsample_factor);
  return tz;
}

extern "C" int64_t vt_tile_workspace_bytes(vt_model* h, int32_t B, int32_t T, int32_t H, int32_t W, int32_t t_chunk_enc, int32_t use_overlap) {
  try {
    M_CHECK(h && B > 0 && T > 0 && H > 0 && W > 0 && t_chunk_enc > 0, "vt_tile_workspace_bytes: bad argument");
    M_CHECK(h->m.v11(), "vt_tile_workspace_bytes: temporal tiling exists only in the v1.1 models (version 1)");
    Model& m = h->m;
    const int f = m.cfg.time_downsample_factor;
    M_CHECK(t_chunk_enc >= f, "vt_tile_workspace_bytes: t_chunk_enc must be at least the temporal factor %d", f);
    ArenaScope arenas(&m);
    const int Hz = H >> m.cfg.n_spatial_ds, Wz = W >> m.cfg.n_spatial_ds;
    const int tz = vt_tile_latent_frames(h, T, t_chunk_enc);
    tile_encode_impl(&m, h->enc, nullptr, B, T, H, W, t_chunk_enc, nullptr, nullptr, nullptr, true);
    tile_decode_impl(&m, h->dec, nullptr, B, tz, Hz, Wz, t_chunk_enc / f, use_overlap != 0, nullptr, nullptr, nullptr, true);
    const size_t peak = std::max(m.arena[0].peak, m.arena[1].peak);
    return (int64_t)(tile_staging_bytes(m.cfg, B, H, W, t_chunk_enc) + 2 * ((peak + 255) & ~(size_t)255) + 1024);
  } catch (const Fail& f) {
    return -1;
  } catch (const std::exception& e) {
    vt_set_error("vt_tile_workspace_bytes: %s", e.what());
    return -1;
  }
}

extern "C" int vt_tile_encode(vt_model* h, const float* x, int32_t B, int32_t T, int32_t H, int32_t W, int32_t t_chunk_enc, float* h_out,
                              void* workspace, int64_t workspace_bytes, vt_stream stream) {
  try {
    M_CHECK(h && x && h_out && B > 0 && T > 0 && H > 0 && W > 0, "vt_tile_encode: bad argument");
    M_CHECK(h->m.v11(), "vt_tile_encode: temporal tiling exists only in the v1.1 models (version 1)");
    M_CHECK(t_chunk_enc >= h->m.cfg.time_downsample_factor, "vt_tile_encode: t_chunk_enc must be at least the temporal factor");
    const int ds = 1 << h->m.cfg.n_spatial_ds;
    M_CHECK(H % ds == 0 && W % ds == 0, "vt_tile_encode: H and W must be multiples of %d", ds);
    check_loaded(h, "encoder.");
    const size_t skip = tile_staging_bytes(h->m.cfg, B, H, W, t_chunk_enc);
    bind_workspace(h->m, workspace, workspace_bytes, skip);
    tile_encode_impl(&h->m, h->enc, x, B, T, H, W, t_chunk_enc, h_out, (char*)workspace, reinterpret_cast<hipStream_t>(stream), false);
    return VT_OK;
  } catch (const Fail& f) {
    return f.code;                                  // TileScope has put the handle's tiled-pass state back
  } catch (const std::exception& e) {
    vt_set_error("vt_tile_encode: %s", e.what());
    return VT_ERR_ARG;
  }
}

extern "C" int vt_tile_decode(vt_model* h, const float* z, int32_t B, int32_t Tz, int32_t Hz, int32_t Wz, int32_t t_chunk_dec, int32_t use_overlap,
                              float* x_out, void* workspace, int64_t workspace_bytes, vt_stream stream) {
  try {
    M_CHECK(h && z && x_out && B > 0 && Tz > 0 && Hz > 0 && Wz > 0 && t_chunk_dec > 0, "vt_tile_decode: bad argument");
    M_CHECK(h->m.v11(), "vt_tile_decode: temporal tiling exists only in the v1.1 models (version 1)");
    check_loaded(h, "decoder.");
    const vt_model_config& cf = h->m.cfg;
    const int f = cf.time_downsample_factor;
    const size_t skip = tile_staging_bytes(cf, B, Hz << cf.n_spatial_us, Wz << cf.n_spatial_us, t_chunk_dec * f);
    bind_workspace(h->m, workspace, workspace_bytes, skip);
    tile_decode_impl(&h->m, h->dec, z, B, Tz, Hz, Wz, t_chunk_dec, use_overlap != 0, x_out, (char*)workspace, reinterpret_cast<hipStream_t>(stream), false);
    return VT_OK;
  } catch (const Fail& f) {
    return f.code;
  } catch (const std::exception& e) {
    vt_set_error("vt_tile_decode: %s", e.what());
    return VT_ERR_ARG;
  }
}

// FSQ: the three statistics of the auxiliary loss on the pre-quantisation latent (FSQRegularizer.forward, regularizers.py:229-262):
// out3 = {per-sample entropy, codebook entropy, commitment}; the caller forms (out3[0] - diversity_gamma * out3[1]) *
// entropy_loss_weight + out3[2] * commitment_loss_weight with its YAML's weights.  work: vt_fsq_aux_work_floats(levels, D, B, S) floats
extern "C" int vt_regularize_fsq_aux(vt_model* h, const float* pre, int32_t B, int32_t Tz, int32_t Hz, int32_t Wz, float inv_temperature,
                                     float* work, float* out3, vt_stream stream) {
  if (!h || h->m.cfg.regularizer != 1) {
    vt_set_error("vt_regularize_fsq_aux: the handle's regularizer is not FSQ");
    return VT_ERR_ARG;
  }
  try {
    Model& m = h->m;
    const int64_t S = (int64_t)Tz * Hz * Wz;
    // with the codebook axis kept the reference's own forward raises (its implicit codebook is one-dimensional, regularizers.py:143-146,234)
    M_CHECK(m.fsq_ncb() == 1, "vt_regularize_fsq_aux: the reference cannot compute the FSQ auxiliary loss with num_codebooks > 1 either");
    const float* hq = pre;
    if (m.fsq_proj()) {            // the statistics are those of the PROJECTED latent (regularizers.py:225-246)
      const int eff = m.fsq_eff();
      float* hp = m.reg_scratch((size_t)2 * B * eff * S);
      M_CALL(vt_channel_linear(pre, m.f32("regularization.project_in.weight", false), m.f32("regularization.project_in.bias", false), hp, B, m.cfg.fsq_dim, eff, S, stream));
      hq = hp;
    }
    return vt_fsq_aux_stats_avg(hq, m.cfg.levels, m.cfg.n_levels, B, S, inv_temperature, work, out3, nullptr, stream);
  } catch (const Fail& f) {
    return f.code;
  } catch (const std::exception& e) {
    vt_set_error("vt_regularize_fsq_aux: %s", e.what());
    return VT_ERR_ARG;
  }
}
