// Weight-stationary persistent 3x3 convolution for the widest level of the pyramid (Cin = Cout = 128, bf16, stride 1,
// pad 1, NDHWC) -- the nine ResnetBlock convolutions at 256x256 resolution (reference model_3dcausal.py:317-337 at
// `ch` = 128) that the tile-per-workgroup kernel (conv_igemm.hip) runs at 670 TFLOP/s: with K = 1152 a 128 x 128 tile
// has 18 K steps of ~1 us against ~13 us of per-tile fixed cost (launch, set-up, first DMA latency, epilogue), and every
// K step re-stages 32 KiB of weights next to 32 KiB of pixels.
//
// Here the WEIGHTS never move: one workgroup per CU (4 waves, one per SIMD, the whole 512-register file each), wave w
// keeps output channels [32w, 32w+32) x K = 1152 as 72 MFMA A-fragments in registers (288 of them) for the lifetime of
// the kernel and walks a contiguous range of 8 x 16-pixel tiles:
//   * per tile only the 10 x 18 halo patch of the input (45 KiB) is fetched, once, by LDS-DMA into a double-buffered
//     LDS patch -- the nine taps are nine shifted windows of the same patch (the generic kernel gathers each tap again);
//     patch rows are padded to 272 B so every 16-lane ds_read_b128 group covers all 64 banks (no XOR, one base VGPR,
//     every fragment address is base + immediate);
//   * K loop = 288 MFMAs per wave, fully unrolled, no barrier, no DMA wait inside; one ds_read_b128 per MFMA
//     (LDS read traffic = half of the 256 B/clk the LDS delivers at the MFMA peak);
//   * epilogue = the 128 x 128 fp32 tile transposed through LDS (aliased over the spent patch), rows written as whole
//     256-B lines with + residual and LayerNorm(+SiLU) fused exactly as conv_epilogue_lds128 does;
//   * the DMA of tile i+1 flies during the K loop of tile i; three barriers per tile.
// The pipeline relies on no ordering between loads and stores: every wave drains its own vmcnt after its K loop (the
// patch of the next tile has had the whole loop to land) and the barriers publish it.
#include <atomic>
#include <type_traits>

#include "conv_common.h"

namespace {

[[maybe_unused]] constexpr int WS_TH = 8, WS_TW = 16;                   // output tile: 8 rows x 16 columns of one frame
[[maybe_unused]] constexpr int WS_PH = WS_TH + 2, WS_PW = WS_TW + 2;    // halo patch
[[maybe_unused]] constexpr int WS_NPIX = WS_PH * WS_PW;                 // 180 pixel rows
[[maybe_unused]] constexpr int WS_ROWP = 272;                           // bytes per patch pixel row: 256 (128 bf16) + 16 pad
[[maybe_unused]] constexpr int WS_PIECES = 48;                          // 1-KiB DMA pieces per patch (180 * 272 = 48 960 B)
[[maybe_unused]] constexpr int WS_PATCH = WS_PIECES * 1024;             // 49 152
[[maybe_unused]] constexpr int WS_EXTRA = 128 * 128 * 4 - WS_PATCH;     // 16 384: T = patch[cur] + extra (contiguous either way)
[[maybe_unused]] constexpr int WS_LDS = 2 * WS_PATCH + WS_EXTRA;        // 114 688: [patch0][extra][patch1]
[[maybe_unused]] constexpr int WS_QPW = WS_PIECES / 4;                  // pieces per wave

// The register file is split by hand: the compiler's allocator, left to choose, spills part of the 288 stationary weight
// registers (and reloads them behind vmcnt(0), which also drains the in-flight patch DMA).  The first WS_W_AGPR weight
// fragments fill the accumulator half of the file (MFMA reads its A operand from there directly); the accumulators
// (64) live in the architectural half with the remaining weights, the fragment pipeline and the address registers --
// so the epilogue reads them without 64 v_accvgpr_read copies per tile.
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
[[maybe_unused]] constexpr int WS_W_AGPR = 63;   // 4 * 63 = 252 of 256 AGPRs; 4 * 9 = 36 VGPRs of weights
template <bool W_IN_AGPR, bool FIRST>
__device__ __forceinline__ void ws_mfma(const u32x4& w, const u32x4& x, f32x16& acc) {
  if constexpr (FIRST) {
    if constexpr (W_IN_AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc) : "a"(w), "v"(x));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc) : "v"(w), "v"(x));
  } else {
    if constexpr (W_IN_AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(x));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(x));
  }
}

__global__ __launch_bounds__(256, 1) void conv3x3_ws128_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.Ho, W = p.Wo;
  const int tiles_w = W / WS_TW;
  const int tiles_pf = tiles_w * (H / WS_TH);
  const int ntiles = tiles_pf * p.B * p.To;
  // contiguous tile range of this workgroup; consecutive ranges stay on one XCD (shared halo rows in its L2)
  const int G = gridDim.x;
  const int slot = xcd_remap(blockIdx.x, G);
  const int tq = ntiles / G, tr = ntiles - tq * G;
  const int t_begin = slot * tq + min(slot, tr);
  const int t_end = t_begin + tq + (slot < tr ? 1 : 0);
  if (t_begin >= t_end) return;

  const bf16_t* __restrict__ xg = reinterpret_cast<const bf16_t*>(p.x);
  bf16_t* __restrict__ yg = reinterpret_cast<bf16_t*>(p.y);
  const bf16_t* __restrict__ rg = reinterpret_cast<const bf16_t*>(p.res);
  bf16_t* __restrict__ ng = reinterpret_cast<bf16_t*>(p.ln_out);
  constexpr unsigned kOob = 0xFFFF0000u;

  // ---- stationary weights: 72 A-fragments (rows n = 32 wave + lane%32, k = 16 c + 8 (lane/32) .. +8) -----------------
  u32x4 wreg[72];
  {
    const bf16_t* row = reinterpret_cast<const bf16_t*>(p.w) + (long long)(wave * 32 + (lane & 31)) * p.ldw + (lane >> 5) * 8;
#pragma unroll
    for (int c = 0; c < 72; ++c) wreg[c] = *reinterpret_cast<const u32x4*>(row + c * 16);
  }

  // ---- DMA geometry of this lane: piece q of this wave writes LDS bytes [(wave*12+q)*1024 + 16 lane, +16) = 16-B unit
  // `unit` of patch pixel (pr, pc).  The descriptor is rebased to the tile's FRAME, so rows above / below the image are
  // out of range by themselves (negative or >= frame bytes: hardware zero fill); only the left / right halo columns of
  // border tiles need a test.  One register per piece: bo = byte offset of the unit relative to the tile origin, low
  // bits = "is halo column 0" (1) / "is halo column 17" (2); pad units and the tail beyond the patch get 2^31, which
  // stays out of range for any tile (frame bytes <= 2^30 by the launcher).
  int bo[WS_QPW];
#pragma unroll
  for (int q = 0; q < WS_QPW; ++q) {
    const int b = (wave * WS_QPW + q) * 1024 + lane * 16;
    const int pp = b / WS_ROWP;
    const int unit = (b - pp * WS_ROWP) >> 4;
    const int pr = pp / WS_PW, pc = pp - pr * WS_PW;
    bo[q] = (pp < WS_NPIX && unit < 16) ? ((((pr - 1) * W + (pc - 1)) * 256 + unit * 16) | (pc == 0 ? 1 : 0) | (pc == WS_PW - 1 ? 2 : 0))
                                        : (int)0x80000000;
  }
  const unsigned frame_bytes = (unsigned)H * (unsigned)W * 256u;
  auto tile_coords = [&](int tile, int& f, int& h0, int& w0) {
    f = tile / tiles_pf;
    const int r = tile - f * tiles_pf;
    const int th = r / tiles_w;
    h0 = th * WS_TH;
    w0 = (r - th * tiles_w) * WS_TW;
  };
  auto issue_patch = [&](int tile, int bufoff) {
    int f, h0, w0;
    tile_coords(tile, f, h0, w0);
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(xg) + (long long)f * H * W * 128, 0, frame_bytes, 0x00020000);
    char* dst = smem + bufoff + wave * (WS_QPW * 1024);
    const int tmask = (w0 == 0 ? 1 : 0) | (w0 + WS_TW == W ? 2 : 0);
    const int toff = (h0 * W + w0) * 256;
#pragma unroll
    for (int q = 0; q < WS_QPW; ++q) {
      const unsigned off = (bo[q] & tmask) ? kOob : (unsigned)((bo[q] & ~3) + toff);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dst + q * 1024), 16, off, 0, 0, 0);
    }
  };

  // ---- per-lane constants of the fragment reads and the epilogue ---------------------------------------------------------
  const int frag_off = (((lane & 31) >> 4) * WS_PW + (lane & 15)) * WS_ROWP + (lane >> 5) * 16;
  const int oct_j = tid & 15, row0 = tid >> 4;   // epilogue read-back: 16 lanes per pixel row, 8 channels each
  const bool has_res = p.res_mode != VT_RES_NONE;

  issue_patch(t_begin, 0);
  wait_vmcnt<0>();
  int cur = 0;
  for (int tile = t_begin; tile < t_end; ++tile, cur ^= 1) {
    const int bufoff = cur ? (WS_PATCH + WS_EXTRA) : 0;
    // (A) patch[cur] has landed for every wave (each drained its vmcnt before its last barrier); nobody still reads the
    //     previous tile's T, which overlaps the buffer refilled next
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (tile + 1 < t_end) issue_patch(tile + 1, cur ? 0 : (WS_PATCH + WS_EXTRA));
    int f, h0, w0;
    tile_coords(tile, f, h0, w0);
    const long long pix0 = ((long long)f * H + h0) * W + w0;   // global pixel of tile pixel (0,0)
    f32x16 acc[4];
    const char* pb = smem + bufoff + frag_off;
    // K loop: 72 groups (tap, 16-channel chunk) x 4 pixel sub-tiles, an explicit software pipeline: the four
    // B-fragments of group g+2 are requested before the MFMAs of group g (one group = 128 matrix-pipe cycles, LDS
    // latency under load ~ 1.5 groups); sched_barrier pins that order -- left to itself the scheduler hoists ~60
    // fragment reads to the top of the tile and spills the stationary weights to scratch.
    auto frag_addr = [&](int g, int j) -> const u32x4* {
      const int tap = g >> 3, c = g & 7;
      const int kh = tap / 3, kw = tap - 3 * kh;
      return reinterpret_cast<const u32x4*>(pb + ((2 * j + kh) * WS_PW + kw) * WS_ROWP + c * 32);
    };
    u32x4 xf[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) xf[0][j] = *frag_addr(0, j);
#pragma unroll
    for (int j = 0; j < 4; ++j) xf[1][j] = *frag_addr(1, j);
    static_for<0, 72>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      if constexpr (g + 2 < 72) {
#pragma unroll
        for (int j = 0; j < 4; ++j) xf[(g + 2) % 3][j] = *frag_addr(g + 2, j);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) ws_mfma<(g < WS_W_AGPR), (g == 0)>(wreg[g], xf[g % 3][j], acc[j]);
      __builtin_amdgcn_sched_barrier(0);
    });
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");   // last MFMA -> first reader of its accumulator (hipcc pads nothing around asm)
    wait_vmcnt<0>();                 // own DMA pieces of the next patch (and the residual): landed long ago
    __builtin_amdgcn_s_barrier();    // (B) every wave is done reading patch[cur]: T may overwrite it
    asm volatile("" ::: "memory");

    // ---- epilogue: transpose through T, rows = 16 lanes x 8 channels, + residual, LayerNorm(+SiLU) ----
    // the residual rows are requested here: their latency rides under the transposition, and their 32 registers never
    // coexist with the fragment pipeline (the register file holds 288 weight + 64 accumulator registers throughout)
    Oct<bf16_t> rq[8];
    if (has_res) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = row0 + 16 * it;
        rq[it].load(rg + (pix0 + (long long)(row >> 4) * W + (row & 15)) * p.ldr + 8 * oct_j);
      }
    }
    float* T = reinterpret_cast<float*>(smem + (cur ? WS_PATCH : 0));
    {
      const int h = lane >> 5;
      f32x4 bq[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (p.bias) bq[g] = *reinterpret_cast<const f32x4*>(p.bias + wave * 32 + 8 * g + 4 * h);
        else bq[g][0] = bq[g][1] = bq[g][2] = bq[g][3] = 0.0f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {      // one pixel sub-tile at a time: 16 accumulator values leave the AGPRs per round
        const int prow = 32 * j + (lane & 31);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = wave * 32 + 8 * g + 4 * h;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[j][4 * g + e] + bq[g][e];
          *reinterpret_cast<f32x4*>(T + prow * 128 + (((c >> 2) ^ (prow & 31)) << 2)) = v;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();                 // (C)
    float lg[8], lb[8];
    if (p.ln_mode) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        lg[e] = p.ln_gamma[8 * oct_j + e];
        lb[e] = p.ln_beta[8 * oct_j + e];
      }
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = row0 + 16 * it;
      const int sw = row & 31;
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(T + row * 128 + (((2 * oct_j) ^ sw) << 2));
      const f32x4 t1 = *reinterpret_cast<const f32x4*>(T + row * 128 + (((2 * oct_j + 1) ^ sw) << 2));
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[e] = e < 4 ? t0[e] : t1[e - 4];
        if (has_res) v[e] = rq[it].get(e) + v[e];
      }
      const long long orow = pix0 + (long long)(row >> 4) * W + (row & 15);
      if (!p.ln_mode || p.ln_keep_y) Oct<bf16_t>::store(yg + orow * p.ldy + 8 * oct_j, v);
      if (p.ln_mode) {   // same two-pass statistics as layernorm_act_kernel, taken before the rounding to bf16
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[e];
        const float mean = group_sum_dpp<16>(s) * (1.0f / 128.0f);
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[e] -= mean;
          q += v[e] * v[e];
        }
        const float rstd = __builtin_amdgcn_rsqf(group_sum_dpp<16>(q) * (1.0f / 128.0f) + p.ln_eps);
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float u = v[e] * rstd * lg[e] + lb[e];
          o[e] = (p.ln_mode == 2) ? silu_fast(u) : u;
        }
        Oct<bf16_t>::store(ng + orow * p.ldn + 8 * oct_j, o);
      }
    }
  }
#endif
}

}  // namespace

// conv_igemm.hip's dispatcher hands over launches that qualify (ws128_eligible there); `args` is its ConvArgs
extern "C" __attribute__((visibility("hidden"))) int vt_ws128_launch(const void* args, void* stream_) {
  const ConvArgs& a = *reinterpret_cast<const ConvArgs*>(args);
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const void* kern = reinterpret_cast<const void*>(&conv3x3_ws128_kernel);
  static std::atomic<int> cus[kMaxDevices];         // 0 = attribute not set yet on that device; else its CU count
  int dev = 0;
  VT_CHECK_HIP(hipGetDevice(&dev));
  int ncu = (dev >= 0 && dev < kMaxDevices) ? cus[dev].load(std::memory_order_acquire) : 0;
  if (ncu == 0) {
    VT_CHECK_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS));
    VT_CHECK_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    if (ncu <= 0) ncu = 256;
    if (dev >= 0 && dev < kMaxDevices) cus[dev].store(ncu, std::memory_order_release);
  }
  const int ntiles = (a.Wo / WS_TW) * (a.Ho / WS_TH) * a.B * a.To;
  const int grid = ntiles < ncu ? ntiles : ncu;     // one persistent workgroup per CU (114 KiB of LDS, 512 registers)
  ConvArgs args_copy = a;
  void* kargs[] = {&args_copy};
  VT_CHECK_HIP(hipLaunchKernel(kern, dim3((unsigned)grid), dim3(256), kargs, WS_LDS, stream));
  return VT_OK;
}
