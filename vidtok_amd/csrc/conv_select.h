// Which kernel / tile / epilogue a vt_conv call gets: the decisions shared by the dispatcher (conv_igemm.hip: vt_conv, vt_conv_plan) and
// the per-type translation units that hold the kernel instantiations (conv_igemm_{f32,bf16,f16,x3}.hip).  Pure functions of the
// descriptor (ConvArgs) and of the option table.
#pragma once
#include <atomic>

#include "conv_common.h"

namespace {

// Test / A-B switches of the option table (options.h; vt_set_option, seeded once from VT_<NAME> -- no launch path reads the environment):
//   conv_buf = 0      gather through 64-bit pointers (global_load_lds) instead of buffer descriptors
//   conv_tinner = 0   plain pixel order for temporal convs
//   conv_tile = 256   force the 8-wave 256x256 tile wherever it is legal (Cout % 256 == 0, vector epilogue), however
//                     few tiles that gives; = 128 forbids it -- lets small parity cases reach either instantiation
inline bool conv_buf() { return vt_opt(OPT_CONV_BUF) != 0; }
inline bool conv_tinner() { return vt_opt(OPT_CONV_TINNER) != 0; }

// CUs of the current device (cached per device; 256 when it cannot be asked, e.g. vt_conv_plan on a host without a GPU)
inline int device_cus() {
  static std::atomic<int> cus[kMaxDevices];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return 256;
  }
  const bool dev_ok = dev >= 0 && dev < kMaxDevices;
  int n = dev_ok ? cus[dev].load(std::memory_order_acquire) : 0;
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
      (void)hipGetLastError();
      n = 256;
    }
    if (dev_ok) cus[dev].store(n, std::memory_order_release);
  }
  return n;
}

// conv_in8_kernel (conv_in8.hip): the encoder's conv_in in a 16-bit type
inline bool in8_eligible(const ConvArgs& a, int nbatch, int dtype, int out_dtype, bool ln_fused, int ln_mode_asked) {
  if (vt_opt(OPT_CONV_IN8) == 0 || !conv_buf() || nbatch != 1 || a.prof != nullptr) return false;
  if (!vt_is_h16(dtype) || out_dtype != dtype || a.Cin != 8 || a.Cout != 128 || a.ldw != 216) return false;
  if (a.KT != 3 || a.KH != 3 || a.KW != 3 || a.st != 1 || a.sh != 1 || a.sw != 1 || a.pt != 2 || a.ph != 1 || a.pw != 1) return false;
  if (a.To != a.Ti || a.Ho != a.Hi || a.Wo != a.Wi || a.ups_t || a.ups_s || a.tmode == VT_TPAD_CACHE) return false;
  if (a.out_layout != VT_NDHWC || a.res_mode != VT_RES_NONE || a.yt_mul != 1 || a.ys_mul == 2) return false;
  // a tile = a 128-pixel segment of one row, or 128 / Wo whole rows, inside one frame
  if (a.Wo < 8 || !((a.Wo % 128 == 0) || (128 % a.Wo == 0)) || ((long long)a.Ho * a.Wo) % 128 != 0) return false;
  if (ln_mode_asked != 0 && !ln_fused) return false;                                            // the LayerNorm belongs to the epilogue or to nobody
  if (a.ldy % 8 != 0 || (a.ln_mode != 0 && a.ldn % 8 != 0) || vt_opt(OPT_CONV_LDSEPI) == 0) return false;   // the epilogue's 16-byte rows
  const unsigned long long xb = (unsigned long long)a.B * a.Ti * a.Hi * a.Wi * 8 * 2;
  return xb < 0xFFFF0000ull;
}

// 128 x 128 tile on a 4-slot ring (128 KB of LDS, three K steps of DMA in flight instead of one).  Two workgroups per CU
// cover each other's DMA latency; a launch with no more tiles than CUs leaves every workgroup alone on its CU, and with
// one step of look-ahead its K step then lasts one fabric round trip (the deep layers of a v1.1 chunk, M = 4 096 / 5 120
// at 512 channels and K = 13 824: ~1 600 cycles per step against 512 of MFMA work).  Alone on the CU it can have the LDS.
inline bool deep_ring_eligible(const ConvArgs& a, int nbatch, int elem_bytes) {
  if (vt_opt(OPT_CONV_DEEP) == 0 || a.prof != nullptr) return false;
  const int bk = kRowBytes / elem_bytes;
  if (a.Cin % bk != 0 || a.ntaps * (a.Cin / bk) < 8) return false;              // descriptor-walk form only; a K worth the ring
  const long long tiles = (long long)((a.M + 127) / 128) * ((a.Cout + 127) / 128) * nbatch;
  return tiles <= device_cus();
}

// Tile selection.  Bytes staged per FLOP fall with the tile area (128x128: 15.6 KB/MFLOP bf16, 256x256: 7.8),
// so Cout % 256 == 0 layers with enough pixels take the 8-wave 256x256 tile (measured 988 vs 814 TFLOP/s on
// the 27-tap 256->256 conv when introduced); everything else keeps 128x128 with two independent workgroups
// per CU, which cover each other's prologue / epilogue / DMA stalls (256x128 tiles measured slower).
enum TileKind { TILE_256x32 = 0, TILE_256x64, TILE_256x256, TILE_128x128 };

// Weight-stationary persistent kernel (conv_ws2.hip): 3x3 stride-1 pad-1 convolutions in a 16-bit type with Cin = Cout = 128 on
// frames that tile by 8 x 16 pixels -- the nine ResnetBlock convolutions of the widest level.  Option conv_ws = 0 keeps them
// on the tile-per-workgroup kernel (A/B runs, and the parity tests run both).
inline bool ws_eligible(const ConvArgs& a, int nbatch, bool h16_io) {
  if (!h16_io || vt_opt(OPT_CONV_WS) == 0) return false;
  if (a.Cin != 128 || a.Cout != 128 || a.ldw != 1152 || a.ldy != 128) return false;
  if (a.KT != 1 || a.KH != 3 || a.KW != 3 || a.st != 1 || a.sh != 1 || a.sw != 1 || a.ph != 1 || a.pw != 1) return false;
  if (a.ups_t || a.ups_s || a.Ho != a.Hi || a.Wo != a.Wi || a.To != a.Ti) return false;
  if (a.Ho % 8 != 0 || a.Wo % 16 != 0 || (long long)a.Ho * a.Wo * 256 > (1ll << 30)) return false;
  if (a.out_layout != VT_NDHWC || a.yt_mul != 1 || a.ys_mul == 2 || nbatch != 1) return false;
  if (a.res_mode == VT_RES_MIX) return false;
  if (a.res_mode == VT_RES_ADD && (a.ldr != 128 || a.Tr != a.To || a.res_tshift != 0 || (reinterpret_cast<uintptr_t>(a.res) & 15))) return false;
  if ((reinterpret_cast<uintptr_t>(a.y) & 15) || (a.bias && (reinterpret_cast<uintptr_t>(a.bias) & 15))) return false;
  return true;
}

// Narrow-output 3x3x3 convolution (conv_narrow.hip): 16-bit (or split-bf16) in, fp32 NCTHW out, Cin = 128, Cout <= 4 -- the decoder's
// conv_out (reference model_3dcausal.py:862-870).  Option conv_narrow = 0 keeps it on the 256 x 32 implicit-GEMM tile.
inline bool narrow_eligible(const ConvArgs& a, int nbatch, int dtype, int out_dtype, int ln_mode) {
  if ((!vt_is_h16(dtype) && dtype != VT_BF16X3) || out_dtype != VT_F32 || a.out_layout != VT_NCTHW || vt_opt(OPT_CONV_NARROW) == 0) return false;
  if (a.Cin != 128 || a.Cout > 4 || a.KT != 3 || a.KH != 3 || a.KW != 3) return false;
  if (a.st != 1 || a.sh != 1 || a.sw != 1 || a.ph != 1 || a.pw != 1 || a.pt < 1 || a.pt > 2) return false;
  if (a.ups_t || a.ups_s || a.Ho != a.Hi || a.Wo != a.Wi || a.To != a.Ti) return false;
  if (a.res_mode != VT_RES_NONE || ln_mode != 0 || nbatch != 1 || a.yt_mul != 1 || a.ys_mul == 2) return false;
  if ((long long)a.Ho * a.Wo * 256 > (1ll << 30)) return false;
  if (a.tmode == VT_TPAD_CACHE && a.ncache < a.pt) return false;
  if (dtype == VT_BF16X3 && a.ldw != 27 * 128) return false;      // the planes are addressed as [K / 16][hi 16 | lo 16], K = 3 456
  return true;
}

inline TileKind select_tile(const ConvArgs& a, int nbatch) {
  auto blocks = [&](int bm, int bn) {
    return (long long)((a.M + bm - 1) / bm) * ((a.Cout + bn - 1) / bn) * nbatch;
  };
  const bool vec_epi = a.out_layout == VT_NDHWC && (a.ldy & 3) == 0 && (a.res_mode == VT_RES_NONE || (a.ldr & 3) == 0);
  if (a.Cout <= 32) return TILE_256x32;
  if (a.Cout <= 64) return TILE_256x64;
  const int force = vt_opt(OPT_CONV_TILE);
  // at least conv_tile_min (default 128) tiles: with the scheduled K loops even half-filled single rounds of the 8-wave
  // tile beat 1.25 rounds of 128 x 128 tiles (M = 20 480, Cout = 512: 0.143 -> 0.121 ms at K = 4 608, 0.38 -> 0.29 at 13 824)
  if (a.Cout % 256 == 0 && vec_epi && force != 128 && (blocks(256, 256) >= vt_opt(OPT_CONV_TILE_MIN) || force == 256)) return TILE_256x256;
  return TILE_128x128;
}

// 8-wave tile, no LayerNorm, but everything the LDS-transposed epilogue needs (conv_epilogue_lds256 with ln_mode = 0):
// a 16-bit type, full tiles, plain NDHWC rows, a residual / mix operand indexed like the output
inline bool lds256_plain_eligible(const ConvArgs& a, int nbatch, bool h16_io) {
  return h16_io && a.ln_mode == 0 && a.prof == nullptr && vt_opt(OPT_CONV_LDSEPI) != 0 &&
         a.M % 256 == 0 && a.Cout % 256 == 0 && nbatch == 1 && a.Cin % (kRowBytes / 2) == 0 && a.out_layout == VT_NDHWC &&
         (a.ldy & 7) == 0 && (a.res_mode == VT_RES_NONE || ((a.ldr & 7) == 0 && a.Tr == a.To && a.res_tshift == 0));
}

}  // namespace

// the per-type translation units (conv_igemm_*.hip): tile selection + launch of the implicit-GEMM kernel; `args` = ConvArgs
extern "C" int vt_igemm_dispatch_f32(const void* args, int nbatch, void* stream);                     // fp32 -> fp32
extern "C" int vt_igemm_dispatch_x3(const void* args, int nbatch, void* stream);                      // split-bf16 arithmetic, fp32 storage
extern "C" int vt_igemm_dispatch_bf16(const void* args, int nbatch, int out_f32, void* stream);       // bf16 -> bf16 | fp32
extern "C" int vt_igemm_dispatch_f16(const void* args, int nbatch, int out_f32, void* stream);        // fp16 -> fp16 | fp32
extern "C" int vt_ws2_launch(const void* conv_args, int dtype, void* stream);                         // conv_ws2.hip
extern "C" int vt_conv_in8_launch(const void* conv_args, int dtype, void* stream);                    // conv_in8.hip
extern "C" int vt_conv_narrow_launch(const void* conv_args, void* stream, int mode);                  // conv_narrow.hip (mode: 0 bf16, 1 fp16, 2 split-bf16: fp32 x, two passes)
extern "C" void vt_conv_narrow_plan(const void* conv_args, int32_t* plan4);
