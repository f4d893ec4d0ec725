// Weight-stationary persistent 3x3 convolution, second generation ("two groups"): the same operator and the same shapes
// as conv_ws128.hip (Cin = Cout = 128, bf16, stride 1, pad 1, NDHWC frames tiling by 8 x 16 -- the nine ResnetBlock
// convolutions of the widest level, reference model_3dcausal.py:317-337 at `ch` = 128), selected by option conv_ws = 2.
//
// What the first generation costs (profiles/r02_ws128_tile_cycles.txt): one wave per SIMD holds 32 output channels x
// K = 1152 and does EVERYTHING for its tile -- 288 MFMAs, their 288 fragment reads, the patch DMA, and the row phase
// (+ bias, + residual, y, LayerNorm + SiLU: ~900 VALU instructions per tile) sliced into the MFMA shadows.  A single wave
// hides at most ~5 plain VALU instructions behind an MFMA, so a tile takes 16 430 cycles against 9 216 of matrix work,
// 22 500 when it also emits a LayerNorm.
//
// Here a SIMD runs TWO waves that split K: wave (w, kh) keeps output channels [32w, 32w+32) x K-half kh (36 MFMA
// A-fragments = 144 registers, 128 of them in the accumulator half of the file), w = wave % 4, kh = wave / 4.  A tile is processed
// as two half-tiles ("units") of 64 pixels; per unit
//     M0(u): group 0 (kh = 0): 72 MFMAs over K-half 0 from zero                  -> partial sums P(u)      (LDS, fp32)
//     M1(u): group 1 (kh = 1): accumulators <- P(u), 72 MFMAs over K-half 1     -> T(u), in place over P(u)
//     R(u):  rows of T(u) (+ bias, added by M1) + residual -> y, LayerNorm(+SiLU) -> n; rows [0,32) by group 1, [32,64) by group 0
// and the two groups alternate, one barrier per half-step, so that a SIMD always has one wave in an MFMA phase and one in
// a row phase (plain-fp32 VALU work of a wave overlaps the other wave's MFMAs):
//     iteration u, first half:    group 0: M0(u)                 |  group 1: R(u-1) rows [0,32)  (+ patch DMA pieces)
//     iteration u, second half:   group 1: M1(u)                 |  group 0: R(u-1) rows [32,64) (+ patch DMA pieces)
// The fp32 sum of an output element is the same chain as in the first generation (K groups 0..35 from zero, then 36..71
// on top), so the result without LayerNorm is bit-identical to conv_ws128.hip and to the tile-per-workgroup kernel.
//
// LDS: two halo patches (2 x 49 152 B, rows padded to 272 B, see conv_ws128.hip) + two P/T buffers of 64 rows x 512 B
// = 163 840 B, all of the CU.  Buffers are handed over by barriers only; every LDS write is retired (lgkmcnt(0)) before
// the barrier that publishes it.  Global loads (residual rows) are requested at the START of a wave's MFMA phase for the
// row phase that follows it; each MFMA phase ends with vmcnt(0) -- by then the stores and DMA pieces of the wave's
// previous row phase have had ~2 500 cycles -- so loads and stores of a wave are never in flight together and nothing
// depends on their relative order.
#include <atomic>
#include <type_traits>

#include "conv_common.h"

namespace {

[[maybe_unused]] constexpr int W2_TH = 8, W2_TW = 16;
[[maybe_unused]] constexpr int W2_PH = W2_TH + 2, W2_PW = W2_TW + 2;
[[maybe_unused]] constexpr int W2_NPIX = W2_PH * W2_PW;                // 180 pixel rows
[[maybe_unused]] constexpr int W2_ROWP = 272;                          // bytes per patch pixel row: 256 + 16 pad
[[maybe_unused]] constexpr int W2_PIECES = 48;                         // 1-KiB DMA pieces per patch
[[maybe_unused]] constexpr int W2_PATCH = W2_PIECES * 1024;            // 49 152
[[maybe_unused]] constexpr int W2_TBUF = 64 * 128 * 4;                 // one P / T buffer: 64 rows x 128 fp32
[[maybe_unused]] constexpr int W2_LDS = 2 * W2_PATCH + 2 * W2_TBUF;    // 163 840
[[maybe_unused]] constexpr int W2_QPW = W2_PIECES / 8;                 // DMA pieces per wave and patch

template <int I, int N, typename F>
__device__ __forceinline__ void w2_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    w2_static_for<I + 1, N>(f);
  }
}

// The register file is split by hand (see conv_ws128.hip): two waves per SIMD = 256 registers per lane, which the
// compiler halves into 128 architectural + 128 accumulator registers.  The first W2_WA weight fragments fill the
// accumulator half ("a": MFMA reads its A operand from there directly), the remaining four live in the architectural
// half with the accumulators, the fragment ring and the row phase.  (With all 36 constrained to "a" the allocator kept
// 128 / 128 and spilled 10-13 registers into the LayerNorm instantiations.)
[[maybe_unused]] constexpr int W2_WA = 32;
template <bool FIRST, bool W_IN_AGPR>
__device__ __forceinline__ void w2_mfma(const u32x4& w, const u32x4& x, f32x16& acc) {
  if constexpr (FIRST) {
    if constexpr (W_IN_AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc) : "a"(w), "v"(x));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc) : "v"(w), "v"(x));
  } else {
    if constexpr (W_IN_AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(x));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(x));
  }
}

template <int LN, bool KEEP>
__global__ __launch_bounds__(512, 1) void conv3x3_ws2_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cw = wave & 3;                                             // channel group: output channels [32 cw, +32)
  const int grp = wave >> 2;                                           // K half and ping-pong group
  const int H = p.Ho, W = p.Wo;
  const int tiles_w = W / W2_TW;
  const int tiles_pf = tiles_w * (H / W2_TH);
  const int ntiles = tiles_pf * p.B * p.To;
  const int G = gridDim.x;
  const int slot = xcd_remap(blockIdx.x, G);
  const int tq = ntiles / G, tr = ntiles - tq * G;
  const int t_begin = slot * tq + min(slot, tr);
  const int t_end = t_begin + tq + (slot < tr ? 1 : 0);
  if (t_begin >= t_end) return;
  const int U = 2 * (t_end - t_begin);                                 // units (half-tiles) of this workgroup

  const bf16_t* __restrict__ xg = reinterpret_cast<const bf16_t*>(p.x);
  bf16_t* __restrict__ yg = reinterpret_cast<bf16_t*>(p.y);
  const bf16_t* __restrict__ rg = reinterpret_cast<const bf16_t*>(p.res);
  bf16_t* __restrict__ ng = reinterpret_cast<bf16_t*>(p.ln_out);
  constexpr unsigned kOob = 0xFFFF0000u;
  float* Tb = reinterpret_cast<float*>(smem + 2 * W2_PATCH);           // [2][64][128]

  // ---- stationary weights: K groups [36 grp, 36 grp + 36) of the 72 (group g = tap * 8 + 16-channel chunk) -------------
  u32x4 wreg[36];
  {
    const bf16_t* row = reinterpret_cast<const bf16_t*>(p.w) + (long long)(cw * 32 + (lane & 31)) * p.ldw + (lane >> 5) * 8 + grp * (36 * 16);
#pragma unroll
    for (int c = 0; c < 36; ++c) wreg[c] = *reinterpret_cast<const u32x4*>(row + c * 16);
  }

  // ---- patch DMA (geometry as in conv_ws128.hip; 6 pieces per wave) -------------------------------------------------------
  const unsigned frame_bytes = (unsigned)H * (unsigned)W * 256u;
  auto tile_coords = [&](int tile, int& f, int& h0, int& w0) {
    f = tile / tiles_pf;
    const int r = tile - f * tiles_pf;
    const int th = r / tiles_w;
    h0 = th * W2_TH;
    w0 = (r - th * tiles_w) * W2_TW;
  };
  auto issue_patch = [&](int tile, int bufoff) {
    int f, h0, w0;
    tile_coords(tile, f, h0, w0);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(xg) + (long long)f * H * W * 128, 0, frame_bytes, 0x00020000);
    char* dst = smem + bufoff + wave * (W2_QPW * 1024);
    const unsigned tmask = (w0 == 0 ? 1u : 0u) | (w0 + W2_TW == W ? 2u : 0u);
    const int toff = (h0 * W + w0) * 256;
    w2_static_for<0, W2_QPW>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      const int b = (wave * W2_QPW + q) * 1024 + lane * 16;
      const int pp = b / W2_ROWP;
      const int unit = (b - pp * W2_ROWP) >> 4;
      const int pr = pp / W2_PW, pc = pp - pr * W2_PW;
      const bool ok = (pp < W2_NPIX) & (unit < 16) & !((pc == 0) & ((tmask & 1u) != 0)) & !((pc == W2_PW - 1) & ((tmask & 2u) != 0));
      const unsigned off = ok ? (unsigned)(((pr - 1) * W + (pc - 1)) * 256 + unit * 16 + toff) : kOob;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dst + q * 1024), 16, off, 0, 0, 0);
    });
  };

  // ---- per-lane constants (lane -> pixel map of a sub-tile: conv_ws128.hip, conflict-free ds_read_b128 groups) ----------
  auto subtile_pixel = [](int m, int& rsel, int& col) {
    const bool g0 = (m < 4) | ((m >= 12) & (m < 16)) | ((m >= 20) & (m < 28));
    rsel = g0 ? 0 : 1;
    col = g0 ? (m < 4 ? m : (m < 16 ? m - 8 : m - 12)) : (m < 12 ? m - 4 : (m < 20 ? m - 8 : m - 16));
  };
  int f_rsel, f_col;
  subtile_pixel(lane & 31, f_rsel, f_col);
  const int frag_off = (f_rsel * W2_PW + f_col) * W2_ROWP + (lane >> 5) * 16;
  // row phase: a group's 256 threads handle 32 T rows of a unit in two iterations; lane slot: row row_l + 16 it, channels [8 oct_j, +8)
  const int tg = tid & 255;
  const int oct_j = tg & 15, row_l = tg >> 4;
  int tp_r[2], tp_c[2];
  subtile_pixel(row_l, tp_r[0], tp_c[0]);
  subtile_pixel(row_l + 16, tp_r[1], tp_c[1]);

  const bool has_res = p.res_mode == VT_RES_ADD;   // uniform

  // unit v = (local tile v >> 1, half v & 1); T row r = 32 jj + m of the unit = pixel (2 (2 half + jj) + rsel(m), col(m)) of the tile
  auto unit_pix0 = [&](int v) -> long long {       // element-row index of the tile origin of unit v
    int f, h0, w0;
    tile_coords(t_begin + (v >> 1), f, h0, w0);
    return ((long long)f * H + h0) * W + w0;
  };
  auto row_pixel = [&](int v, int jj, int it) -> long long {
    return unit_pix0(v) + (long long)(2 * (2 * (v & 1) + jj) + tp_r[it]) * W + tp_c[it];
  };
  // residual rows of the row slot (unit v, sub-tile jj): requested at the start of the MFMA phase in front of it
  Oct<bf16_t> rq[2];
  rq[0].w[0] = rq[0].w[1] = rq[0].w[2] = rq[0].w[3] = 0u;
  rq[1].w = rq[0].w;
  auto prefetch_rows = [&](int v, int jj) {
    if (has_res) {
#pragma unroll
      for (int it = 0; it < 2; ++it) rq[it].load(rg + row_pixel(v, jj, it) * p.ldr + 8 * oct_j);
    }
  };
  // row slot: T rows [32 jj, 32 jj + 32) of unit v (bias already in): + residual -> y, LayerNorm(+SiLU) -> n.  Row arithmetic
  // in explicit-rounding intrinsics, the operation order of conv_ws128.hip
  auto row_slot = [&](int v, int jj) {
    const float* T = Tb + (v & 1) * (64 * 128);
    // LayerNorm affine of this lane's 8 channels: fetched per slot (L1-resident, 64 B per lane) rather than kept -- 16
    // registers less through the MFMA phases, which otherwise spill (144 stationary weight registers per wave)
    f32x4 g0, g1, b0, b1;
    if constexpr (LN != 0) {
      g0 = *reinterpret_cast<const f32x4*>(p.ln_gamma + 8 * oct_j);
      g1 = *reinterpret_cast<const f32x4*>(p.ln_gamma + 8 * oct_j + 4);
      b0 = *reinterpret_cast<const f32x4*>(p.ln_beta + 8 * oct_j);
      b1 = *reinterpret_cast<const f32x4*>(p.ln_beta + 8 * oct_j + 4);
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int row = 32 * jj + row_l + 16 * it;
      const int sw = row & 31;
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(T + row * 128 + (((2 * oct_j) ^ sw) << 2));
      const f32x4 t1 = *reinterpret_cast<const f32x4*>(T + row * 128 + (((2 * oct_j + 1) ^ sw) << 2));
      const long long pix = row_pixel(v, jj, it);
      float rv[8], s = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        rv[e] = __fadd_rn(rq[it].get(e), e < 4 ? t0[e] : t1[e - 4]);
        s = __fadd_rn(s, rv[e]);
      }
      if constexpr (KEEP) {
        u32x4 w4;
#pragma unroll
        for (int e = 0; e < 4; ++e) w4[e] = f32_to_bf16_bits(rv[2 * e]) | (f32_to_bf16_bits(rv[2 * e + 1]) << 16);
        *reinterpret_cast<u32x4*>(yg + pix * p.ldy + 8 * oct_j) = w4;
      }
      if constexpr (LN != 0) {
        const float mean = group_sum_dpp<16>(s) * (1.0f / 128.0f);
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          rv[e] = __fsub_rn(rv[e], mean);
          q = __fmaf_rn(rv[e], rv[e], q);
        }
        const float rstd = __builtin_amdgcn_rsqf(__fmaf_rn(group_sum_dpp<16>(q), 1.0f / 128.0f, p.ln_eps));
        u32x4 w4;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float a = __fmaf_rn(__fmul_rn(rv[e], rstd), e < 4 ? g0[e] : g1[e - 4], e < 4 ? b0[e] : b1[e - 4]);
          rv[e] = (LN == 2) ? silu_fast(a) : a;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) w4[e] = f32_to_bf16_bits(rv[2 * e]) | (f32_to_bf16_bits(rv[2 * e + 1]) << 16);
        *reinterpret_cast<u32x4*>(ng + pix * p.ldn + 8 * oct_j) = w4;
      }
    }
  };

  // ---- MFMA phase of unit u for K-half KH: 36 groups x 2 sub-tiles, fragments three MFMAs ahead ------------------------
  f32x16 acc[2];
  const int hq = lane >> 5;
  // this lane's quad of (sub-tile jj, channel quad g) in the P / T buffer.  Recomputed from the lane id at every use (made
  // opaque so the eight offsets are not hoisted out of the unit loop): kept resident they were the registers the allocator
  // spilled, and a scratch reload waits behind vmcnt(0) -- i.e. behind the residual rows just requested
  auto t_slot = [&](int u, int jj, int g) -> float* {
    int l = lane;
    asm volatile("" : "+v"(l));
    const int prow = 32 * jj + (l & 31);
    const int c4 = cw * 8 + 2 * g + (l >> 5);                  // 16-B chunk of channels [32 cw + 8 g + 4 (lane / 32), +4)
    return Tb + (u & 1) * (64 * 128) + prow * 128 + ((c4 ^ (l & 31)) << 2);
  };
  auto mfma_phase = [&](auto kh_c, int u, int bufoff) {
    constexpr int KH = decltype(kh_c)::value;
    f32x4 bq[4];                                               // K-half 1 adds the bias before it parks the finished sums
    if constexpr (KH == 1) {                                   // continue the sum of K-half 0
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(t_slot(u, jj, g));
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[jj][4 * g + e] = v[e];
        }
    }
    const char* pb = smem + bufoff + frag_off + (4 * (u & 1)) * W2_PW * W2_ROWP;   // sub-tile 2 half + jj starts at patch row 4 half + 2 jj
    auto frag_addr = [&](int m) -> const u32x4* {              // MFMA m = 2 g + jj of the phase
      const int gg = 36 * KH + (m >> 1), jj = m & 1;
      const int tap = gg >> 3, c = gg & 7;
      const int kh = tap / 3, kw = tap - 3 * kh;
      return reinterpret_cast<const u32x4*>(pb + ((2 * jj + kh) * W2_PW + kw) * W2_ROWP + c * 32);
    };
    u32x4 xf[4];
#pragma unroll
    for (int m = 0; m < 3; ++m) xf[m] = *frag_addr(m);
    __builtin_amdgcn_s_setprio(1);
    w2_static_for<0, 72>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      w2_mfma<(KH == 0 && m < 2), ((m >> 1) < W2_WA)>(wreg[m >> 1], xf[m % 4], acc[m & 1]);
      if constexpr (m + 3 < 72) xf[(m + 3) % 4] = *frag_addr(m + 3);
      if constexpr (KH == 1 && m == 58) {                      // requested late: 16 registers that must not live through the phase
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (p.bias) bq[g] = *reinterpret_cast<const f32x4*>(p.bias + cw * 32 + 8 * g + 4 * hq);
          else bq[g][0] = bq[g][1] = bq[g][2] = bq[g][3] = 0.0f;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");         // last MFMA -> first reader of its accumulator
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = KH == 1 ? acc[jj][4 * g + e] + bq[g][e] : acc[jj][4 * g + e];
        *reinterpret_cast<f32x4*>(t_slot(u, jj, g)) = v;
      }
  };
  auto publish = [&]() {                                       // my LDS writes are retired, then the barrier hands the buffers over
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  issue_patch(t_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  publish();
  for (int u = 0; u <= U; ++u) {
    const int tl = u >> 1;                                     // local tile of unit u
    const int bufoff = (tl & 1) * W2_PATCH;
    const bool next_patch = (u & 1) == 0 && t_begin + tl + 1 < t_end && u < U;   // uniform: this iteration requests the next tile's patch
    // ---- first half: group 0 M0(u) | group 1 rows [0,32) of unit u-1
    if (grp == 0) {
      if (u >= 1) prefetch_rows(u - 1, 1);                     // for my row slot in the second half
      if (u < U) mfma_phase(std::integral_constant<int, 0>{}, u, bufoff);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // residual rows here; stores / DMA pieces of my last row slot long retired
    } else {
      if (u >= 1) row_slot(u - 1, 0);
      if (next_patch) issue_patch(t_begin + tl + 1, ((tl + 1) & 1) * W2_PATCH);
    }
    publish();
    // ---- second half: group 1 M1(u) | group 0 rows [32,64) of unit u-1
    if (grp == 1) {
      if (u < U) {
        prefetch_rows(u, 0);                                   // for my row slot in the next iteration's first half
        mfma_phase(std::integral_constant<int, 1>{}, u, bufoff);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (u >= 1) row_slot(u - 1, 1);
      if (next_patch) issue_patch(t_begin + tl + 1, ((tl + 1) & 1) * W2_PATCH);
    }
    publish();
  }
#endif
}

}  // namespace

// conv_igemm.hip's dispatcher hands over launches that qualify (ws128_eligible there) when option conv_ws = 2
extern "C" __attribute__((visibility("hidden"))) int vt_ws2_launch(const void* args, void* stream_) {
  const ConvArgs& a = *reinterpret_cast<const ConvArgs*>(args);
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  VT_CHECK_ARG(a.prof == nullptr, "vt_conv_profile: the two-group weight-stationary kernel carries no stamps (use conv_ws = 1)");
  const bool keep = a.ln_mode == 0 || a.ln_keep_y != 0;
  const int vi = a.ln_mode == 0 ? 0 : (a.ln_mode == 1 ? (keep ? 1 : 2) : (keep ? 3 : 4));
  static const void* const kerns[5] = {
      reinterpret_cast<const void*>(&conv3x3_ws2_kernel<0, true>), reinterpret_cast<const void*>(&conv3x3_ws2_kernel<1, true>),
      reinterpret_cast<const void*>(&conv3x3_ws2_kernel<1, false>), reinterpret_cast<const void*>(&conv3x3_ws2_kernel<2, true>),
      reinterpret_cast<const void*>(&conv3x3_ws2_kernel<2, false>)};
  static std::atomic<int> cus[kMaxDevices];         // 0 = not set up on that device yet; else its CU count
  int dev = 0;
  VT_CHECK_HIP(hipGetDevice(&dev));
  const bool dev_ok = dev >= 0 && dev < kMaxDevices;
  int ncu = dev_ok ? cus[dev].load(std::memory_order_acquire) : 0;
  if (ncu == 0) {
    for (int k = 0; k < 5; ++k) VT_CHECK_HIP(hipFuncSetAttribute(kerns[k], hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS));
    VT_CHECK_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    if (ncu <= 0) ncu = 256;
    if (dev_ok) cus[dev].store(ncu, std::memory_order_release);
  }
  const int ntiles = (a.Wo / W2_TW) * (a.Ho / W2_TH) * a.B * a.To;
  const int grid = ntiles < ncu ? ntiles : ncu;     // one persistent workgroup per CU (all of its LDS)
  ConvArgs args_copy = a;
  void* kargs[] = {&args_copy};
  VT_CHECK_HIP(hipLaunchKernel(kerns[vi], dim3((unsigned)grid), dim3(512), kargs, W2_LDS, stream));
  return VT_OK;
}
