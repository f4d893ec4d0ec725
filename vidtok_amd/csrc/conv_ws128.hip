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
//   * epilogue = the 128 x 128 fp32 tile (+ bias) transposed through a third LDS buffer T; its row phase -- rows read
//     back as whole 256-B lines, + residual, y store, LayerNorm(+SiLU) store, the arithmetic of conv_epilogue_lds128 -- is
//     cut into 56 pieces that ride in the MFMA gaps of the NEXT tile's K loop (with one wave per SIMD nothing else would
//     hide its ~900 VALU instructions; run on their own they cost half a K loop: 1.64 -> measured below);
//   * the DMA of tile i+1 flies during the K loop of tile i; two barriers per tile.
// The pipeline relies on no ordering between loads and stores: every wave drains its own vmcnt after its K loop (the
// patch of the next tile has had the whole loop to land, the row phase's stores were issued >= 16 groups earlier) and
// the barriers publish it.
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
[[maybe_unused]] constexpr int WS_LDS = 2 * WS_PATCH + 128 * 128 * 4;   // 163 840 = all of the CU's LDS: [patch0][patch1][T]
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
#ifndef WS_DEPTH
#define WS_DEPTH 1   // 2 left the LayerNorm instantiations with scratch spills (3 fragment sets + the row phase in 256 VGPRs)
#endif
// ACC_A = true: accumulators (64) + the first 47 weight fragments in the accumulator half of the register file, 25
// fragments (100 registers) in the architectural half; false: accumulators in the architectural half (no copies in
// the epilogue), 63 fragments in the other one.  VT_WS_ACC selects at run time (A/B: an MFMA whose C / D operands sit
// in the architectural file shares its ports with the B operand).
template <bool ACC_A>
struct WsSplit {
  static constexpr int W_AGPR = ACC_A ? 47 : 64;
};
template <bool W_IN_AGPR, bool FIRST, bool ACC_A>
__device__ __forceinline__ void ws_mfma(const u32x4& w, const u32x4& x, f32x16& acc) {
  if constexpr (ACC_A) {
    if constexpr (FIRST) {
      if constexpr (W_IN_AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc) : "a"(w), "v"(x));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(w), "v"(x));
    } else {
      if constexpr (W_IN_AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(w), "v"(x));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(x));
    }
  } else {
    if constexpr (FIRST) {
      if constexpr (W_IN_AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc) : "a"(w), "v"(x));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc) : "v"(w), "v"(x));
    } else {
      if constexpr (W_IN_AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(x));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(x));
    }
  }
}

// "This value exists HERE": an empty volatile statement that reads and writes it.  Instruction selection places pure
// arithmetic next to its use, not where the source put it -- without pins the ~900 VALU instructions of a row phase,
// cut into 224 slices for the MFMA shadows, all sank to the stores they feed (sched_barrier only binds the later
// machine scheduler).  Volatile statements keep their order, so a pinned slice stays between "its" two MFMAs.
__device__ __forceinline__ void pin(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin_u(uint32_t& v) { asm volatile("" : "+v"(v)); }

template <int LN, bool KEEP, bool ACC_A, bool PROF = false>
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
  float* T = reinterpret_cast<float*>(smem + 2 * WS_PATCH);

  // ---- stationary weights: 72 A-fragments (rows n = 32 wave + lane%32, k = 16 c + 8 (lane/32) .. +8) -----------------
  u32x4 wreg[72];
  {
    const bf16_t* row = reinterpret_cast<const bf16_t*>(p.w) + (long long)(wave * 32 + (lane & 31)) * p.ldw + (lane >> 5) * 8;
#pragma unroll
    for (int c = 0; c < 72; ++c) wreg[c] = *reinterpret_cast<const u32x4*>(row + c * 16);
  }

  // ---- patch DMA: the descriptor is rebased to the tile's FRAME, so rows above / below the image are out of range by
  // themselves (negative or >= frame bytes: hardware zero fill); only the left / right halo columns of border tiles
  // need a test (frame bytes <= 2^30 by the launcher) ----
  const unsigned frame_bytes = (unsigned)H * (unsigned)W * 256u;
  auto tile_coords = [&](int tile, int& f, int& h0, int& w0) {
    f = tile / tiles_pf;
    const int r = tile - f * tiles_pf;
    const int th = r / tiles_w;
    h0 = th * WS_TH;
    w0 = (r - th * tiles_w) * WS_TW;
  };
  // per-tile scalars of the patch DMA (set by patch_begin), then one piece at a time (patch_piece): the K loop issues
  // the pieces of the NEXT tile's patch in its MFMA gaps, two per row iteration, instead of 12 back to back up front
  __amdgpu_buffer_rsrc_t pd_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(xg), 0, 0, 0x00020000);
  char* pd_dst = smem;
  unsigned pd_tmask = 0;
  int pd_toff = 0;
  bool pd_on = false;
  // PROF (vt_conv_profile): shader-clock stamps of workgroup 0's fourth tile, written straight to memory (the LDS is full)
  bool prof_tile = false;
  auto stamp = [&](int k) {
    if constexpr (PROF) {
      if (prof_tile) {
        const unsigned long long ts = __builtin_amdgcn_s_memtime();
        if (lane == 0) p.prof[wave * 16 + k] = ts;
      }
    }
  };
  auto patch_begin = [&](int tile, int bufoff) {
    int f, h0, w0;
    tile_coords(tile, f, h0, w0);
    pd_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(xg) + (long long)f * H * W * 128, 0, frame_bytes, 0x00020000);
    pd_dst = smem + bufoff + wave * (WS_QPW * 1024);
    pd_tmask = (w0 == 0 ? 1u : 0u) | (w0 + WS_TW == W ? 2u : 0u);
    pd_toff = (h0 * W + w0) * 256;
  };
  // Piece q of this wave writes LDS bytes [(wave*12+q)*1024 + 16 lane, +16) = 16-B unit `unit` of patch pixel (pr, pc).
  // The geometry is recomputed from the lane id at every use (~18 VALU in an otherwise idle MFMA shadow): twelve
  // resident offsets were the first thing the register allocator spilled, and a scratch reload waits behind vmcnt(0).
  auto patch_piece = [&](auto qc) {
    constexpr int q = decltype(qc)::value;
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));             // opaque: keeps the arithmetic below inside the tile loop
    const int b = (wave * WS_QPW + q) * 1024 + lane_o * 16;
    const int pp = b / WS_ROWP;
    const int unit = (b - pp * WS_ROWP) >> 4;
    const int pr = pp / WS_PW, pc = pp - pr * WS_PW;
    const bool ok = (pp < WS_NPIX) & (unit < 16) & !((pc == 0) & ((pd_tmask & 1u) != 0)) & !((pc == WS_PW - 1) & ((pd_tmask & 2u) != 0));
    const unsigned off = ok ? (unsigned)(((pr - 1) * W + (pc - 1)) * 256 + unit * 16 + pd_toff) : kOob;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(pd_rsrc, (lds_ptr_t)(pd_dst + q * 1024), 16, off, 0, 0, 0);
  };
  auto issue_patch = [&](int tile, int bufoff) {
    patch_begin(tile, bufoff);
    static_for<0, WS_QPW>([&](auto qc) { patch_piece(qc); });
  };

  // ---- per-lane constants of the fragment reads and of the row phase ----------------------------------------------------
  // MFMA column m = lane % 32 of a pixel sub-tile (2 rows x 16 columns of the tile) -> which pixel.  ds_read_b128
  // serves a wave in four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 -- one LDS
  // cycle each when the 16 addresses fall on 16 different 16-B bank quads.  With the 272-B pixel stride the quad of
  // patch pixel (r, c) is (2r + c + unit) mod 16, so a group must hold 16 pixels of ONE patch row: group 0 reads row 0
  // of the sub-tile, group 1 row 1 (the natural m -> (m / 16, m % 16) puts 8 + 8 pixels of both rows into each group:
  // two 2-way conflicts per group = half the LDS rate, and the K loop needs half of the full rate).
  auto subtile_pixel = [](int m, int& rsel, int& col) {
    const bool g0 = (m < 4) | ((m >= 12) & (m < 16)) | ((m >= 20) & (m < 28));
    rsel = g0 ? 0 : 1;
    col = g0 ? (m < 4 ? m : (m < 16 ? m - 8 : m - 12)) : (m < 12 ? m - 4 : (m < 20 ? m - 8 : m - 16));
  };
  int f_rsel, f_col;
  subtile_pixel(lane & 31, f_rsel, f_col);
  const int frag_off = (f_rsel * WS_PW + f_col) * WS_ROWP + (lane >> 5) * 16;
  // rows of T = MFMA order (32 j + m); the row phase handles T rows row0 + 16 it (it < 8), channels [8 oct_j, +8):
  // T row -> tile pixel (2 (it / 2) + rsel, col) with (rsel, col) of m = row0 + 16 (it % 2)
  const int oct_j = tid & 15, row0 = tid >> 4;
  int tp_r[2], tp_c[2];
  subtile_pixel(row0, tp_r[0], tp_c[0]);
  subtile_pixel(row0 + 16, tp_r[1], tp_c[1]);
  auto tile_pixel_off = [&](int it) -> long long {     // element-row offset of T row (row0 + 16 it) from the tile origin
    return (long long)(2 * (it >> 1) + tp_r[it & 1]) * W + tp_c[it & 1];
  };
  float lg[8], lb[8];        // LayerNorm affine of this lane's 8 channels
  if constexpr (LN != 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      lg[e] = p.ln_gamma[8 * oct_j + e];
      lb[e] = p.ln_beta[8 * oct_j + e];
    }
  }

  // ---- row phase of a FINISHED tile (its biased fp32 result sits in T, transposed): + residual, y store, LayerNorm
  // (+SiLU) store.  It is cut into 7 stages per row iteration x 8 iterations = 56 pieces, so that the K loop of the NEXT
  // tile can carry one piece per MFMA group: with one wave per SIMD nothing else would hide its ~900 VALU instructions
  // per tile (LayerNorm + SiLU of 64 elements per lane), which cost as much as half a K loop when run on their own.
  float rv[8], rmean = 0.f, rrstd = 0.f;
  f32x4 rt0, rt1;
  // residual rows: with no residual the registers stay zero and the adds are spent anyway -- the instantiations without
  // these 8 registers came out of the register allocator WITH scratch spills (the ones with them did not)
  Oct<bf16_t> rq[3];      // row iteration `it` uses rq[it % 3]; the other two are in flight for it + 1, it + 2
  rq[0].w[0] = rq[0].w[1] = rq[0].w[2] = rq[0].w[3] = 0u;
  rq[1].w = rq[0].w;
  rq[2].w = rq[0].w;
  float rsum = 0.f;
  const bool has_res = p.res_mode == VT_RES_ADD;   // uniform
  long long pix0_prev = 0;
  auto res_row_ptr = [&](int it) -> const bf16_t* {
    return rg + (pix0_prev + tile_pixel_off(it)) * p.ldr + 8 * oct_j;
  };
  // All row arithmetic is spelled with explicit-rounding intrinsics: the same source is instantiated inside the K loop and
  // once more for the last tile of a workgroup, and context-dependent FMA contraction would make a pixel's bits depend on
  // which of the two computed it (i.e. on how tiles were split over workgroups).
  // The row phase runs as TWO sweeps over the 8 row iterations, 28 groups each (piece = group index):
  //   sweep 1 (groups 0..27, 3.5 per iteration -> 14 shadows): T row + residual -> y, LayerNorm(+SiLU) -> both packed to
  //            bf16 and written back IN PLACE over the row's own two 16-B units of T.  Only loads are in flight.
  //   sweep 2 (groups 28..55): packed rows re-read from T and stored to HBM, plus the next patch's DMA pieces.
  // Loads and stores of a wave never overlap in time that way: with both pending hipcc does not trust a counted vmcnt
  // and drains everything before the first use of a loaded value, i.e. every residual / affine wait became a store
  // round trip (measured on the single-sweep version: SQ_WAIT_ANY = 41 % of the wave cycles).
  // slot s = 4 * (group % 7) + sub within a row iteration's 14 shadows (sweep 1) -- ~6 VALU instructions each
  auto row_piece = [&](auto piece_c, auto sub_c) {
    constexpr int piece = decltype(piece_c)::value, sub = decltype(sub_c)::value;
    if constexpr (piece < 28) {
      constexpr int lin = piece * 4 + sub;            // 0 .. 111
      constexpr int it = lin / 14, s = lin % 14;
      const int row = row0 + 16 * it;
      const int sw = row & 31;
      if constexpr (s == 0) {
        rt0 = *reinterpret_cast<const f32x4*>(T + row * 128 + (((2 * oct_j) ^ sw) << 2));
        rt1 = *reinterpret_cast<const f32x4*>(T + row * 128 + (((2 * oct_j + 1) ^ sw) << 2));
        if constexpr (it + 2 < 8) { if (has_res) rq[(it + 2) % 3].load(res_row_ptr(it + 2)); }   // two iterations (7 groups) ahead
      } else if constexpr (s >= 1 && s <= 4) {         // rv = T row + residual; running sum
        constexpr int u = s - 1;
        if constexpr (u == 0) rsum = 0.f;
#pragma unroll
        for (int e = 2 * u; e < 2 * u + 2; ++e) {
          rv[e] = __fadd_rn(rq[it % 3].get(e), e < 4 ? rt0[e] : rt1[e - 4]);
          rsum = __fadd_rn(rsum, rv[e]);
          pin(rv[e]);
        }
        pin(rsum);
      } else if constexpr (s == 5) {                   // y packed and parked in the row's first unit; mean
        if constexpr (KEEP) {
          u32x4 w4;
#pragma unroll
          for (int e = 0; e < 4; ++e) w4[e] = f32_to_bf16_bits(rv[2 * e]) | (f32_to_bf16_bits(rv[2 * e + 1]) << 16);
          *reinterpret_cast<u32x4*>(T + row * 128 + (((2 * oct_j) ^ sw) << 2)) = w4;
        }
        if constexpr (LN != 0) {
          rmean = group_sum_dpp<16>(rsum) * (1.0f / 128.0f);
          pin(rmean);
        }
      } else if constexpr (s >= 6 && s <= 7) {         // centred values, squares
        if constexpr (LN != 0) {
          constexpr int u = s - 6;
          if constexpr (u == 0) rsum = 0.f;
#pragma unroll
          for (int e = 4 * u; e < 4 * u + 4; ++e) {
            rv[e] = __fsub_rn(rv[e], rmean);
            rsum = __fmaf_rn(rv[e], rv[e], rsum);
            pin(rv[e]);
          }
          pin(rsum);
        }
      } else if constexpr (s == 8) {
        if constexpr (LN != 0) {
          rrstd = __builtin_amdgcn_rsqf(__fmaf_rn(group_sum_dpp<16>(rsum), 1.0f / 128.0f, p.ln_eps));
          pin(rrstd);
        }
      } else if constexpr (s >= 9 && s <= 12) {        // affine (+SiLU), two channels per shadow
        if constexpr (LN != 0) {
          constexpr int u = s - 9;
#pragma unroll
          for (int e = 2 * u; e < 2 * u + 2; ++e) {
            const float a = __fmaf_rn(__fmul_rn(rv[e], rrstd), lg[e], lb[e]);
            rv[e] = (LN == 2) ? silu_fast(a) : a;
            pin(rv[e]);
          }
        }
      } else {                                         // s == 13: the normalised row, packed, into the row's second unit
        if constexpr (LN != 0) {
          u32x4 w4;
#pragma unroll
          for (int e = 0; e < 4; ++e) w4[e] = f32_to_bf16_bits(rv[2 * e]) | (f32_to_bf16_bits(rv[2 * e + 1]) << 16);
          *reinterpret_cast<u32x4*>(T + row * 128 + (((2 * oct_j + 1) ^ sw) << 2)) = w4;
        }
      }
    } else {
      // sweep 2: groups 28..55; row iteration it = (piece - 28) * 4 + sub over 112 shadows -> one iteration per 14
      constexpr int lin = (piece - 28) * 4 + sub;
      constexpr int it = lin / 14, s = lin % 14;
      const int row = row0 + 16 * it;
      const int sw = row & 31;
      if constexpr (s == 0) {
        if constexpr (KEEP) rt0 = *reinterpret_cast<const f32x4*>(T + row * 128 + (((2 * oct_j) ^ sw) << 2));
        if constexpr (LN != 0) rt1 = *reinterpret_cast<const f32x4*>(T + row * 128 + (((2 * oct_j + 1) ^ sw) << 2));
      } else if constexpr (s == 4) {
        if constexpr (KEEP) *reinterpret_cast<f32x4*>(yg + (pix0_prev + tile_pixel_off(it)) * p.ldy + 8 * oct_j) = rt0;
      } else if constexpr (s == 8) {
        if constexpr (LN != 0) *reinterpret_cast<f32x4*>(ng + (pix0_prev + tile_pixel_off(it)) * p.ldn + 8 * oct_j) = rt1;
      } else if constexpr (s == 11) {
        if constexpr (it < 6) { if (pd_on) patch_piece(std::integral_constant<int, 2 * (it % 6)>{}); }
      } else if constexpr (s == 13) {
        if constexpr (it < 6) { if (pd_on) patch_piece(std::integral_constant<int, 2 * (it % 6) + 1>{}); }
      }
    }
  };
  auto row_piece_all = [&](auto piece_c) {
    static_for<0, 4>([&](auto sc) { row_piece(piece_c, sc); });
  };

  // ---- K loop of one tile: 72 groups (tap, 16-channel chunk) x 4 pixel sub-tiles, an explicit software pipeline: the four
  // B-fragments of group g+2 are requested before the MFMAs of group g (one group = 128 matrix-pipe cycles, LDS
  // latency under load ~ 1.5 groups); sched_barrier pins that order -- left to itself the scheduler hoists ~60 fragment
  // reads to the top of the tile and spills the stationary weights.  WITH_ROWS: groups 0..55 also carry the row phase
  // of the previous tile, one piece each; the last 16 groups stay bare so its stores have retired by the vmcnt drain.
  f32x16 acc[4];
  f32x4 bq[4];
  auto k_loop = [&](auto with_rows_c, int bufoff) {
    constexpr bool WITH_ROWS = decltype(with_rows_c)::value;
    const char* pb = smem + bufoff + frag_off;
    auto frag_addr = [&](int g, int j) -> const u32x4* {
      const int tap = g >> 3, c = g & 7;
      const int kh = tap / 3, kw = tap - 3 * kh;
      return reinterpret_cast<const u32x4*>(pb + ((2 * j + kh) * WS_PW + kw) * WS_ROWP + c * 32);
    };
    constexpr int D = WS_DEPTH;          // fragment prefetch distance in groups
    u32x4 xf[D + 1][4];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
      for (int j = 0; j < 4; ++j) xf[d][j] = *frag_addr(d, j);
    static_for<0, 72>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      static_for<0, 4>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        ws_mfma<(g < WsSplit<ACC_A>::W_AGPR), (g == 0), ACC_A>(wreg[g], xf[g % (D + 1)][j], acc[j]);
        // in this MFMA's 32-cycle shadow: the same sub-tile's fragment of group g + D, then ~5 VALU of the row phase
        if constexpr (g + D < 72) xf[(g + D) % (D + 1)][j] = *frag_addr(g + D, j);
        if constexpr (WITH_ROWS && g < 56) {
          row_piece(gc, jc);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (PROF && (g == 17 || g == 35 || g == 55)) stamp(2 + (g + 1) / 18);   // 3, 4, 5: after 72 / 144 / 224 MFMAs
      if constexpr (g == 60) {             // bias quads for the transposition below: requested in a bare stretch
        const int h = lane >> 5;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          if (p.bias) bq[q4] = *reinterpret_cast<const f32x4*>(p.bias + wave * 32 + 8 * q4 + 4 * h);
          else bq[q4][0] = bq[q4][1] = bq[q4][2] = bq[q4][3] = 0.0f;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");   // last MFMA -> first reader of its accumulator (hipcc pads nothing around asm)
  };
  // accumulators (+ bias) -> T, transposed: MFMA layout lane = pixel 32 j + lane%32, channels 32 wave + 8 g + 4 (lane/32) + e
  auto acc_to_T = [&]() {
    const int h = lane >> 5;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
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
  };

  issue_patch(t_begin, 0);
  wait_vmcnt<0>();
  int cur = 0;
  for (int tile = t_begin; tile < t_end; ++tile, cur ^= 1) {
    // (A) patch[cur] has landed for every wave (each drained its vmcnt before barrier B of the previous tile) and the
    //     previous tile's T is complete
    if constexpr (PROF) prof_tile = blockIdx.x == 0 && tile == t_begin + 3;
    stamp(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stamp(1);
    pd_on = tile + 1 < t_end;            // uniform
    if (tile == t_begin) {
      if (pd_on) issue_patch(tile + 1, cur ? 0 : WS_PATCH);
      k_loop(std::false_type{}, cur ? WS_PATCH : 0);
    } else {
      if (pd_on) patch_begin(tile + 1, cur ? 0 : WS_PATCH);
      if (has_res) {
        rq[0].load(res_row_ptr(0));
        rq[1].load(res_row_ptr(1));
      }
      stamp(2);
      k_loop(std::true_type{}, cur ? WS_PATCH : 0);
    }
    stamp(6);
    wait_vmcnt<0>();                 // own DMA pieces of the next patch, the row phase's loads and stores: long done
    stamp(7);
    __builtin_amdgcn_s_barrier();    // (B) every wave is done with the previous tile's T (and with patch[cur])
    asm volatile("" ::: "memory");
    stamp(8);
    acc_to_T();
    stamp(9);
    int f, h0, w0;
    tile_coords(tile, f, h0, w0);
    pix0_prev = ((long long)f * H + h0) * W + w0;
  }
  // row phase of the last tile, on its own
  __syncthreads();
  if (has_res) {
    rq[0].load(res_row_ptr(0));
    rq[1].load(res_row_ptr(1));
  }
  static_for<0, 56>([&](auto pc) { row_piece_all(pc); });
#endif
}

}  // namespace

// conv_igemm.hip's dispatcher hands over launches that qualify (ws128_eligible there); `args` is its ConvArgs
extern "C" __attribute__((visibility("hidden"))) int vt_ws128_launch(const void* args, void* stream_) {
  const ConvArgs& a = *reinterpret_cast<const ConvArgs*>(args);
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  // one instantiation per epilogue shape: LayerNorm none / plain / +SiLU, y kept or not
  const bool keep = a.ln_mode == 0 || a.ln_keep_y != 0;
  const int vi = a.ln_mode == 0 ? 0 : (a.ln_mode == 1 ? (keep ? 1 : 2) : (keep ? 3 : 4));
  const int acc_a = vt_opt(OPT_WS_ACC) != 0 ? 1 : 0;   // measured: no difference; the architectural placement compiles without spills
  static const void* const kerns[2][5] = {
      {reinterpret_cast<const void*>(&conv3x3_ws128_kernel<0, true, false>), reinterpret_cast<const void*>(&conv3x3_ws128_kernel<1, true, false>),
       reinterpret_cast<const void*>(&conv3x3_ws128_kernel<1, false, false>), reinterpret_cast<const void*>(&conv3x3_ws128_kernel<2, true, false>),
       reinterpret_cast<const void*>(&conv3x3_ws128_kernel<2, false, false>)},
      {reinterpret_cast<const void*>(&conv3x3_ws128_kernel<0, true, true>), reinterpret_cast<const void*>(&conv3x3_ws128_kernel<1, true, true>),
       reinterpret_cast<const void*>(&conv3x3_ws128_kernel<1, false, true>), reinterpret_cast<const void*>(&conv3x3_ws128_kernel<2, true, true>),
       reinterpret_cast<const void*>(&conv3x3_ws128_kernel<2, false, true>)}};
  const void* kern = kerns[acc_a][vi];
  if (a.prof != nullptr) {           // vt_conv_profile: the plain and the LayerNorm+SiLU (y kept) instantiations carry stamps
    VT_CHECK_ARG(vi == 0 || vi == 3, "vt_conv_profile (weight-stationary kernel): ln_mode 0, or 2 with ln_keep_y");
    kern = vi == 0 ? reinterpret_cast<const void*>(&conv3x3_ws128_kernel<0, true, false, true>)
                   : reinterpret_cast<const void*>(&conv3x3_ws128_kernel<2, true, false, true>);
    VT_CHECK_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS));
  }
  static std::atomic<int> cus[kMaxDevices];         // 0 = not queried yet on that device; else its CU count
  static std::atomic<bool> attr_done[2][5][kMaxDevices];
  int dev = 0;
  VT_CHECK_HIP(hipGetDevice(&dev));
  const bool dev_ok = dev >= 0 && dev < kMaxDevices;
  if (!dev_ok || !attr_done[acc_a][vi][dev].load(std::memory_order_acquire)) {
    VT_CHECK_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS));
    if (dev_ok) attr_done[acc_a][vi][dev].store(true, std::memory_order_release);
  }
  int ncu = dev_ok ? cus[dev].load(std::memory_order_acquire) : 0;
  if (ncu == 0) {
    VT_CHECK_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    if (ncu <= 0) ncu = 256;
    if (dev_ok) cus[dev].store(ncu, std::memory_order_release);
  }
  const int ntiles = (a.Wo / WS_TW) * (a.Ho / WS_TH) * a.B * a.To;
  const int grid = ntiles < ncu ? ntiles : ncu;     // one persistent workgroup per CU (114 KiB of LDS, 512 registers)
  ConvArgs args_copy = a;
  void* kargs[] = {&args_copy};
  VT_CHECK_HIP(hipLaunchKernel(kern, dim3((unsigned)grid), dim3(256), kargs, WS_LDS, stream));
  return VT_OK;
}
