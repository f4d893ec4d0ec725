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
// A-fragments = 144 registers, 128 of them in the accumulator half of the file), w = wave % 4, kh = wave / 4.  Tiles are
// 4 x 16 pixels (64); per tile u
//     M0(u): group 0 (kh = 0): 72 MFMAs over K-half 0 from zero                  -> partial sums P(u)      (LDS, fp32)
//     M1(u): group 1 (kh = 1): accumulators <- P(u), 72 MFMAs over K-half 1     -> T(u), in place over P(u)
//     R(u):  rows of T(u) + bias + residual -> y, LayerNorm(+SiLU) -> n; rows [0,32) by group 1, rows [32,64) by group 0
// and the two groups alternate, one barrier per half-step, so that a SIMD always has one wave in an MFMA phase and one in
// a row slot (plain-fp32 VALU work of a wave overlaps the other wave's MFMAs).  Since round 6 group 0's phase straddles the
// end-of-iteration barrier: its first w2_head MFMAs ("head", issue priority 0) run beside group 1's phase, in the window in which
// group 1 hands over its sums; the rest ("tail") and the hand-over of P beside group 1's rows:
//     iteration u, first half:    group 0: tail of M0(u), P(u) -> LDS           |  group 1: request patch(u+1), R(u-1) rows [0,32), await the patch
//     iteration u, second half:   group 1: M1(u) (priority 2)                   |  group 0: request residual(u), R(u-1) rows [32,64), head of M0(u+1)
// The fp32 sum of an output element is the same chain as in the first generation (K groups 0..35 from zero, then 36..71
// on top) and the row arithmetic is the same, so results are bit-identical to conv_ws128.hip (and, without LayerNorm,
// to the tile-per-workgroup kernel).
//
// Memory side.  Loads and stores of a wave retire through ONE in-order counter, and under this kernel's write traffic a
// store needs ~5 000 cycles to retire -- nearly two MFMA phases.  A wave that waits for a load it issued behind a store
// therefore stalls for thousands of cycles (first version of this file: rows requested at the head of an MFMA phase,
// vmcnt(0) at its end -- no faster than the first generation), and registers with a load in flight across a phase get
// copied by the allocator at block ends before the load has landed (second version).  So NOTHING is loaded into
// registers here: the halo patch of the next tile AND the residual rows of the current one come in by LDS-DMA
// (buffer_load ... lds), requested at the head of a row slot, in front of that slot's stores -- the patch by the waves
// of group 1, the residual rows by those of group 0 -- and awaited by a COUNTED wait that leaves exactly those stores
// outstanding (group 1: at the end of its following MFMA phase; group 0: at the end of the slot itself, ~3 000 cycles
// after the request); the barrier behind it publishes the buffers for the next iteration.
//
// LDS: two halo patches (6 x 18 pixel rows padded to 272 B: 2 x 29 696) + two P/T buffers (64 rows x 512 B) + two residual
// tiles (64 rows x 272 B) + the LayerNorm affine and the bias (1.5 KB) = 161 280 B.  Buffers are handed over by barriers
// only; every LDS write is retired (lgkmcnt(0)) before the barrier that publishes it.
#include <atomic>
#include <type_traits>

#include "conv_select.h"

#define VT_STORE_AUX 2      // cache policy bits of the y / n stores: nt (streaming; - 0.6 ... - 2.7 % per launch, profiles/r06_c128_kernel_variants.txt)

namespace {

[[maybe_unused]] constexpr int W2_TH = 4, W2_TW = 16;
[[maybe_unused]] constexpr int W2_PH = W2_TH + 2, W2_PW = W2_TW + 2;
[[maybe_unused]] constexpr int W2_NPIX = W2_PH * W2_PW;                // 108 pixel rows
[[maybe_unused]] constexpr int W2_ROWP = 272;                          // bytes per patch / residual pixel row: 256 + 16 pad
[[maybe_unused]] constexpr int W2_PPIECES = 29;                        // 1-KiB DMA pieces per patch (108 * 272 = 29 376 B)
[[maybe_unused]] constexpr int W2_PATCH = W2_PPIECES * 1024;           // 29 696
[[maybe_unused]] constexpr int W2_TBUF = 64 * 128 * 4;                 // one P / T buffer: 64 rows x 128 fp32
[[maybe_unused]] constexpr int W2_RPIECES = 17;                        // 1-KiB DMA pieces per residual tile (64 * 272 = 17 408 B)
[[maybe_unused]] constexpr int W2_RBUF = W2_RPIECES * 1024;            // 17 408
[[maybe_unused]] constexpr int W2_OFF_T = 2 * W2_PATCH;
[[maybe_unused]] constexpr int W2_OFF_R = W2_OFF_T + 2 * W2_TBUF;
[[maybe_unused]] constexpr int W2_OFF_PRM = W2_OFF_R + 2 * W2_RBUF;    // 159 744: LayerNorm gamma | beta | bias, 3 x 128 fp32
[[maybe_unused]] constexpr int W2_LDS = W2_OFF_PRM + 3 * 128 * 4;      // 161 280
// MFMAs of group 0's phase that run in front of the end-of-iteration barrier (even, 2 .. 70): what fits beside group 1's phase without
// lengthening it -- the window in which group 1 hands over its sums, less what group 0's own rows take (profiles/r06_ws2_split_phase.txt)
template <int LN, bool KEEP> [[maybe_unused]] constexpr int w2_head = LN == 0 ? 24 : (KEEP ? 8 : 16);
[[maybe_unused]] constexpr int W2_FD = 6;                               // fragment prefetch distance of an MFMA phase, in MFMAs
[[maybe_unused]] constexpr int W2_PSLOTS = 8, W2_RSLOTS = 5;           // DMA pieces per wave of group 1: patch pieces w + 4 q (< 29), residual pieces w + 4 q (< 17)

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
template <typename H, bool FIRST, bool W_IN_AGPR>
__device__ __forceinline__ void w2_mfma(const u32x4& w, const u32x4& x, f32x16& acc) {
  if constexpr (std::is_same<H, bf16_t>::value) {
    if constexpr (FIRST) {
      if constexpr (W_IN_AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc) : "a"(w), "v"(x));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc) : "v"(w), "v"(x));
    } else {
      if constexpr (W_IN_AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(x));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(x));
    }
  } else {
    if constexpr (FIRST) {
      if constexpr (W_IN_AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(acc) : "a"(w), "v"(x));
      else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(acc) : "v"(w), "v"(x));
    } else {
      if constexpr (W_IN_AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(x));
      else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(x));
    }
  }
}

// H = the 16-bit storage type (bf16_t / f16_t: common.h)
template <typename H, int LN, bool KEEP, bool PROF = false>
__global__ __launch_bounds__(512, 1) void conv3x3_ws2_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cw = wave & 3;                                             // channel group: output channels [32 cw, +32)
  const int grp = wave >> 2;                                           // K half and ping-pong group
  const int FH = p.Ho, W = p.Wo;
  const int tiles_w = W / W2_TW;
  const int tiles_pf = tiles_w * (FH / W2_TH);
  const int ntiles = tiles_pf * p.B * p.To;
  const int G = gridDim.x;
  const int slot = xcd_remap(blockIdx.x, G);
  const int tq = ntiles / G, tr = ntiles - tq * G;
  const int t_begin = slot * tq + min(slot, tr);
  const int t_end = t_begin + tq + (slot < tr ? 1 : 0);
  if (t_begin >= t_end) return;
  const int U = t_end - t_begin;                                       // tiles of this workgroup

  const H* __restrict__ xg = reinterpret_cast<const H*>(p.x);
  H* __restrict__ yg = reinterpret_cast<H*>(p.y);
  const H* __restrict__ rg = reinterpret_cast<const H*>(p.res);
  H* __restrict__ ng = reinterpret_cast<H*>(p.ln_out);
  constexpr unsigned kOob = 0xFFFF0000u;
  const bool has_res = p.res_mode == VT_RES_ADD;                        // uniform

  // ---- stationary weights: K groups [36 grp, 36 grp + 36) of the 72 (group g = tap * 8 + 16-channel chunk) -------------
  u32x4 wreg[36];
  {
    const H* row = reinterpret_cast<const H*>(p.w) + (long long)(cw * 32 + (lane & 31)) * p.ldw + (lane >> 5) * 8 + grp * (36 * 16);
#pragma unroll
    for (int c = 0; c < 36; ++c) wreg[c] = *reinterpret_cast<const u32x4*>(row + c * 16);
  }
  // ---- LayerNorm affine and bias of the 128 channels, parked in the LDS (a row slot reads its 24 values with six
  // ds_read_b128 instead of keeping 24 registers through the MFMA phases)
  float* prm = reinterpret_cast<float*>(smem + W2_OFF_PRM);
  if (tid < 384) {
    float v = 0.0f;
    if (tid < 128) v = LN != 0 ? ln_fold(p.ln_gamma[tid], LN == 2) : 1.0f;             // LayerNorm + SiLU: the affine carries -log2(e) (ln_row8, common.h)
    else if (tid < 256) v = LN != 0 ? ln_fold(p.ln_beta[tid - 128], LN == 2) : 0.0f;
    else v = p.bias ? p.bias[tid - 256] : 0.0f;
    // channel c = 8 oct + 4 half + e of an array at [half][oct][e]: the 16 lanes of a row read 256 contiguous bytes (in channel
    // order lanes oct and oct + 8 of a ds_read_b128 group share their banks: every parameter read 2-way conflicted)
    const int c = tid & 127;
    prm[(tid & ~127) + ((c >> 2) & 1) * 64 + (c >> 3) * 4 + (c & 3)] = v;
  }

  // Tile cursor: (frame, h0, w0) of a tile, stepped through the workgroup's run of tiles by additions -- a wave issues one
  // instruction every four cycles whatever the instruction is, and the two divisions of "tile index -> coordinates" were
  // ~60 scalar instructions a call, three calls an iteration (profiles/r03_ws2_iteration_cycles.txt: the row slots and the
  // phase prologues were issue-bound, 4 cycles x instruction count, not MFMA- or memory-bound).
  struct TileCur {
    int f, h0, w0;
  };
  auto cur_at = [&](int tile) {
    TileCur c;
    c.f = tile / tiles_pf;
    const int r = tile - c.f * tiles_pf;
    const int th = r / tiles_w;
    c.h0 = th * W2_TH;
    c.w0 = (r - th * tiles_w) * W2_TW;
    return c;
  };
  auto cur_step = [&](TileCur& c) {
    c.w0 += W2_TW;
    if (c.w0 == W) {
      c.w0 = 0;
      c.h0 += W2_TH;
      if (c.h0 == FH) {
        c.h0 = 0;
        c.f += 1;
      }
    }
  };
  // ---- LDS-DMA of a tile's operands by the waves of group 1: the 6 x 18 halo patch of x (29 pieces of 1 KiB of the
  // 272-B-row image; descriptor rebased to the tile's frame: rows above / below the image are out of range by themselves,
  // the left / right halo columns of border tiles are tested) and its 4 x 16 residual rows (17 pieces).  Wave w sends
  // patch pieces w, w + 4, ... and residual pieces w, w + 4, ...  What a lane fetches for a piece depends on the tile only
  // through the tile's byte offset in its frame, so the geometry is worked out ONCE per lane and piece slot (13 registers:
  // bits 4.. = byte offset relative to the tile origin, bit 0 / 1 = left / right halo column, bit 2 = never fetched: pad
  // bytes, image tail); a request then costs five instructions instead of the ~100 of the divide-by-272 arithmetic (first
  // measurement of this file: 5 000 cycles of request code per tile).
  const unsigned frame_bytes = (unsigned)FH * (unsigned)W * 256u;
  unsigned pgeo[W2_PSLOTS], rgeo[W2_RSLOTS];
#pragma unroll
  for (int q = 0; q < W2_PSLOTS; ++q) {
    const int b = (cw + 4 * q) * 1024 + lane * 16;
    const int pp = b / W2_ROWP;
    const int unit = (b - pp * W2_ROWP) >> 4;
    const int pr = pp / W2_PW, pc = pp - pr * W2_PW;
    const int rel = ((pr - 1) * W + (pc - 1)) * 256 + unit * 16;         // may be negative: wraps, and wraps back when the tile offset is added
    pgeo[q] = (unsigned)rel | (pc == 0 ? 1u : 0u) | (pc == W2_PW - 1 ? 2u : 0u) | ((pp >= W2_NPIX || unit >= 16) ? 4u : 0u);
  }
#pragma unroll
  for (int q = 0; q < W2_RSLOTS; ++q) {
    const int b = (cw + 4 * q) * 1024 + lane * 16;
    const int rr = b / W2_ROWP;                                          // residual row 16 r + c of the tile
    const int unit = (b - rr * W2_ROWP) >> 4;
    rgeo[q] = (unsigned)(((rr >> 4) * W + (rr & 15)) * 256 + unit * 16) | ((rr >= 64 || unit >= 16) ? 4u : 0u);
  }
  auto issue_dma = [&](const TileCur& pt, int pbuf, const TileCur& rt, int rbuf, bool want_patch, bool want_res) {
    if constexpr (PROF) {
      if (p.prof_mode & 2) return;
    }
    if (want_patch) {
      const int f = pt.f, h0 = pt.h0, w0 = pt.w0;
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<H*>(xg) + (long long)f * FH * W * 128, 0, frame_bytes, 0x00020000);
      const unsigned tmask = (w0 == 0 ? 1u : 0u) | (w0 + W2_TW == W ? 2u : 0u) | 4u;
      const unsigned toff = (unsigned)((h0 * W + w0) * 256);
      w2_static_for<0, W2_PSLOTS>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        if (cw + 4 * q < W2_PPIECES) {                                     // uniform
          const unsigned off = (pgeo[q] & tmask) ? kOob : (pgeo[q] & ~15u) + toff;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(smem + pbuf * W2_PATCH + (cw + 4 * q) * 1024), 16, off, 0, 0, 0);
        }
      });
    }
    if (want_res) {
      const int f = rt.f, h0 = rt.h0, w0 = rt.w0;
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<H*>(rg) + (long long)f * FH * W * 128, 0, frame_bytes, 0x00020000);
      const unsigned toff = (unsigned)((h0 * W + w0) * 256);
      w2_static_for<0, W2_RSLOTS>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        if (cw + 4 * q < W2_RPIECES) {                                     // uniform
          const unsigned off = (rgeo[q] & 4u) ? kOob : (rgeo[q] & ~15u) + toff;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(smem + W2_OFF_R + rbuf * W2_RBUF + (cw + 4 * q) * 1024), 16, off, 0, 0, 0);
        }
      });
    }
  };

  // ---- lane -> pixel map of a sub-tile (conv_ws128.hip: conflict-free ds_read_b128 groups) -------------------------------------
  // quads of m alternate between the two pixel rows in the order 0 1 1 0 1 0 0 1 (bit m/4 of 0x96), columns advance by 4
  // every second quad -- in bit operations, not compares: the compare form compiled into ~60 instructions of exec-mask
  // branches at the head of every MFMA phase and row slot
  auto subtile_pixel = [](int m, int& rsel, int& col) {
    rsel = (0x96 >> (m >> 2)) & 1;
    col = (m & 3) | ((m >> 1) & 12);
  };
  // PROF (vt_conv_profile): shader-clock stamps of workgroup 0's iterations 8 and 9, straight to memory
  int prof_u = -1;
  auto stamp = [&](int k) {
    if constexpr (PROF) {
      if (prof_u >= 0) {
        const unsigned long long ts = __builtin_amdgcn_s_memtime();
        if (lane == 0) p.prof[wave * 16 + 8 * prof_u + k] = ts;
      }
    }
  };

  // ---- row slot: T rows [32 jj, 32 jj + 32) of tile v: + bias + residual -> y, LayerNorm(+SiLU) -> n.  A group's 256 threads:
  // lane slot = row row_l + 16 it of sub-tile jj, channels [8 oct_j, +8).  Per-lane geometry is recomputed from an opaque
  // lane id in every slot instead of living in registers across the MFMA phases (144 weight registers per wave: anything
  // resident pushed the allocator into scratch, and a scratch reload waits -- vmcnt(0) -- behind the slot's own stores).
  // Row arithmetic in explicit-rounding intrinsics, the operation order of conv_ws128.hip (bias joined the tile before the
  // transposition there); every intermediate is pinned so the vectoriser cannot pair the elements -- packed fp32 does not
  // run beside the other wave's MFMAs (scripts/mfma_issue_bench.hip).
  // y / n go out through buffer descriptors rebased to the tile's frame (as the DMA requests come in): the per-lane byte
  // offset is tile-invariant and 32-bit, the tile's own offset is one more addition -- no 64-bit address arithmetic per
  // store (rows of y and n are 128 channels wide: ws128_eligible).  The tile offset does NOT ride in the instruction's
  // scalar-offset operand: a 16-byte store with an SGPR offset reads its data registers late, and the compiler's one
  // wait state did not cover it here -- the second y store of a slot went out with registers 1 of lanes 12-15 (mod 16)
  // already overwritten by the LayerNorm arithmetic behind it, in a few tiles per launch (tests/test_gpu_ops.py:
  // test_weight_stationary_kernels_are_split_independent caught it).
  auto row_slot = [&](const TileCur& tc, int par, int jj) {
    if constexpr (PROF) {
      if (p.prof_mode & 1) return;
    }
    int tt = tid;
    asm volatile("" : "+v"(tt));
    const int tg = tt & 255;
    const int oct_j = tg & 15, row_l = tg >> 4;
    const int tile_pix = tc.h0 * W + tc.w0;
    __amdgpu_buffer_rsrc_t yrs, nrs;
    if constexpr (KEEP) yrs = __builtin_amdgcn_make_buffer_rsrc(yg + (long long)tc.f * FH * W * 128, 0, frame_bytes, 0x00020000);
    if constexpr (LN != 0) nrs = __builtin_amdgcn_make_buffer_rsrc(ng + (long long)tc.f * FH * W * 128, 0, frame_bytes, 0x00020000);
    const float* pl = prm + 4 * oct_j;                         // gamma of channels [8 oct_j, +4) and, 64 floats on, [8 oct_j + 4, +4); beta + 128, bias + 256
    f32x4 g0, g1, b0, b1;
    if constexpr (LN != 0) {
      g0 = *reinterpret_cast<const f32x4*>(pl);
      g1 = *reinterpret_cast<const f32x4*>(pl + 64);
      b0 = *reinterpret_cast<const f32x4*>(pl + 128);
      b1 = *reinterpret_cast<const f32x4*>(pl + 192);
    }
    const f32x4 o0 = *reinterpret_cast<const f32x4*>(pl + 256);
    const f32x4 o1 = *reinterpret_cast<const f32x4*>(pl + 320);
    // lane slot `it`: row m = row_l + 16 it of the sub-tile; subtile_pixel(m + 16) = (1 - rsel(m), col(m) + 8)
    int rsel0, col0;
    subtile_pixel(row_l, rsel0, col0);
    const int tbase = W2_OFF_T + par * W2_TBUF + (32 * jj + row_l) * 512 + (((2 * oct_j) ^ row_l) << 4);   // T row, my first 16-B chunk
    const int rbase = W2_OFF_R + par * W2_RBUF + oct_j * 16;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int prow = 2 * jj + (it == 0 ? rsel0 : 1 - rsel0);   // pixel (prow, col) of the 4 x 16 tile
      const int col = col0 + 8 * it;
      // T row of slot 1 = row + 16: chunk index ^ 16 (the swizzle is row & 31), 16 rows further
      const int toff = it == 0 ? tbase : ((tbase ^ 256) + 16 * 512);
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(smem + toff);
      const f32x4 t1 = *reinterpret_cast<const f32x4*>(smem + (toff ^ 16));
      u32x4 rw;
      if (has_res) rw = *reinterpret_cast<const u32x4*>(smem + rbase + (16 * prow + col) * W2_ROWP);
      else rw[0] = rw[1] = rw[2] = rw[3] = 0u;
      const int pix = prow * W + col;                          // relative to the tile's first pixel
      float rv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint32_t w2 = rw[e >> 1];
        const float r = (e & 1) ? h16<H>::hi(w2) : h16<H>::lo(w2);
        rv[e] = __fadd_rn(r, __fadd_rn(e < 4 ? t0[e] : t1[e - 4], e < 4 ? o0[e] : o1[e - 4]));
      }
      if constexpr (KEEP) {
        u32x4 w4;
#pragma unroll
        for (int e = 0; e < 4; ++e) w4[e] = h16<H>::pack(rv[2 * e], rv[2 * e + 1]);
        __builtin_amdgcn_raw_buffer_store_b128(w4, yrs, (tile_pix + pix) * 256 + oct_j * 16, 0, VT_STORE_AUX);
      }
      if constexpr (LN != 0) {                                 // the row arithmetic every 16-bit LayerNorm site shares (round 6: one-pass moments)
        const float gg[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
        const float bb[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        float ov[8];
        ln_row8<16, LN == 2>(rv, gg, bb, p.ln_eps, ov);
        u32x4 w4;
#pragma unroll
        for (int e = 0; e < 4; ++e) w4[e] = h16<H>::pack(ov[2 * e], ov[2 * e + 1]);
        __builtin_amdgcn_raw_buffer_store_b128(w4, nrs, (tile_pix + pix) * 256 + oct_j * 16, 0, VT_STORE_AUX);
      }
    }
  };

  // ---- MFMA phase of tile u for K-half KH: 36 groups x 2 sub-tiles ------------------------
  f32x16 acc[2];
  // this lane's quad of (sub-tile jj, channel quad g) in the P / T buffer: row 32 jj + lane % 32, 16-B chunk
  // (8 cw + 2 g + lane / 32) ^ (lane % 32) -- 8 cw + lane / 32 and 2 g share no bits, so the chunk of quad g is the chunk of
  // quad 0 with g << 1 XORed in: ONE base per phase (recomputed from an opaque lane id, see row_slot), one XOR per quad,
  // the sub-tile in the instruction's offset field (the eight independent address computations of the first version
  // were 64 instructions between the last MFMA of a phase and its barrier)
  auto t_base = [&](int par) -> int {
    int l = lane;
    asm volatile("" : "+v"(l));
    const int l31 = l & 31;
    return W2_OFF_T + par * W2_TBUF + l31 * 512 + (((cw * 8 + (l >> 5)) ^ l31) << 4);
  };
  auto t_slot = [&](int tb, int jj, int g) -> float* { return reinterpret_cast<float*>(smem + ((tb ^ (g << 5)) + jj * (32 * 512))); };
  // Group 0's phase is split across the end-of-iteration barrier (round 6): MFMAs [0, w2_head) of M0(u+1) run at the tail of group 0's row
  // half of iteration u -- beside group 1's M1(u), the matrix pipe shared -- and MFMAs [w2_head, 72) + the hand-over of the sums in the first
  // half of iteration u+1, beside group 1's rows.  Before, the two groups' MFMA phases were strictly serial (each 2 440 cycles of MFMAs inside
  // ~3 600 of fragments-in / sums-out / barrier: the matrix pipe idled a third of every iteration while group 0 waited ~2 800 cycles at the
  // barrier, profiles/r04_ws2_iteration_cycles.txt).  The fragment ring and the accumulators stay in registers across the barrier; group 1
  // awaits its patch requests at the end of its row half, so that the mid barrier publishes patch(u+1) for the head.  An output's fp32 chain
  // is unchanged (K groups in order): the same bits.
  u32x4 xf[W2_FD + 1];
  auto frag_base = [&](int u) -> const char* {
    int f_rsel, f_col, lo = lane;
    asm volatile("" : "+v"(lo));
    subtile_pixel(lo & 31, f_rsel, f_col);
    const int frag_off = (f_rsel * W2_PW + f_col) * W2_ROWP + (lo >> 5) * 16;
    return smem + (u & 1) * W2_PATCH + frag_off;
  };
  auto frag_ptr = [&](auto kh_c, const char* pb, int m) -> const u32x4* {     // MFMA m = 2 g + jj of the phase
    constexpr int KH = decltype(kh_c)::value;
    const int gg = 36 * KH + (m >> 1), jj = m & 1;
    const int tap = gg >> 3, c = gg & 7;
    const int kh = tap / 3, kw = tap - 3 * kh;
    return reinterpret_cast<const u32x4*>(pb + ((2 * jj + kh) * W2_PW + kw) * W2_ROWP + c * 32);
  };
  // MFMAs [M0, M1) of the phase of K-half KH on tile u; the ring holds the fragments of [M0, M0 + W2_FD) on entry when M0 > 0
  auto mfma_run = [&](auto kh_c, auto m0_c, auto m1_c, auto prio_c, int u) {
    constexpr int KH = decltype(kh_c)::value, M0 = decltype(m0_c)::value, M1 = decltype(m1_c)::value, PRIO = decltype(prio_c)::value;
    const char* pb = frag_base(u);
    if constexpr (M0 == 0) {
      // fragment ring: a read is requested W2_FD MFMAs (~200 cycles) ahead of its use -- three ahead (~100 cycles) left every
      // MFMA waiting for the LDS: 51 cycles per MFMA instead of ~36 (profiles/r03_ws2_iteration_cycles.txt, first measurement)
#pragma unroll
      for (int m = 0; m < W2_FD; ++m) xf[m] = *frag_ptr(kh_c, pb, m);
      stamp(5);
    }
    __builtin_amdgcn_s_setprio(PRIO);
    w2_static_for<M0, M1>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      w2_mfma<H, (KH == 0 && m < 2), ((m >> 1) < W2_WA)>(wreg[m >> 1], xf[m % (W2_FD + 1)], acc[m & 1]);
      if constexpr (m + W2_FD < 72) xf[(m + W2_FD) % (W2_FD + 1)] = *frag_ptr(kh_c, pb, m + W2_FD);
      __builtin_amdgcn_sched_barrier(0);
    });
    __builtin_amdgcn_s_setprio(0);
  };
  auto sums_out = [&](int u) {
    const int tb = t_base(u & 1);
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");         // last MFMA -> first reader of its accumulator
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[jj][4 * g + e];
        *reinterpret_cast<f32x4*>(t_slot(tb, jj, g)) = v;
      }
  };
  using W2K0 = std::integral_constant<int, 0>;
  using W2K1 = std::integral_constant<int, 1>;
  using W2M0 = std::integral_constant<int, 0>;
  using W2MH = std::integral_constant<int, w2_head<LN, KEEP>>;
  using W2ME = std::integral_constant<int, 72>;
  // issue priorities: the head runs beside group 1's whole phase and must not delay it (that phase and group 1's rows are the iteration's
  // critical chain) -- it takes the matrix pipe while group 1 restores / hands over its sums and otherwise what is left
  auto m0_head = [&](int u) { mfma_run(W2K0{}, W2M0{}, W2MH{}, std::integral_constant<int, 0>{}, u); };
  auto m0_tail = [&](int u) {
    mfma_run(W2K0{}, W2MH{}, W2ME{}, std::integral_constant<int, 1>{}, u);
    stamp(6);
    sums_out(u);
  };
  auto m1_phase = [&](int u) {
    const int tb = t_base(u & 1);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)                               // continue the sum of K-half 0
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(t_slot(tb, jj, g));
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[jj][4 * g + e] = v[e];
      }
    mfma_run(W2K1{}, W2M0{}, W2ME{}, std::integral_constant<int, 2>{}, u);
    stamp(6);
    sums_out(u);
  };
  auto publish = [&]() {                                       // my LDS writes are retired, then the barrier hands the buffers over
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // prologue: patch (group 1) and residual rows (group 0) of the first tile, parameters, first barrier
  TileCur c_prev = cur_at(t_begin), c_cur = c_prev, c_next = c_prev;   // tiles u - 1 (from iteration 1 on), u, u + 1
  cur_step(c_next);
  if (grp == 1) issue_dma(c_cur, 0, c_cur, 0, true, false);
  else issue_dma(c_cur, 0, c_cur, 0, false, has_res);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  publish();
  constexpr int NS = 2 * ((KEEP ? 1 : 0) + (LN != 0 ? 1 : 0));  // stores of a row slot
  // iteration u: M phases of tile u (group 0's head ran in iteration u-1), rows of tile u-1; group 1 requests patch(u+1), group 0 the
  // residual rows of tile u.  One loop per group (the same two barriers an iteration in both): in a common loop the accumulators and the
  // fragment ring that group 0 carries across the back edge were live in group 1's row half as well, and 40 registers spilled.
  auto advance = [&]() {
    c_prev = c_cur;
    c_cur = c_next;
    cur_step(c_next);
  };
  if (grp == 0) {
    m0_head(0);
    for (int u = 0; u <= U; ++u) {
      if constexpr (PROF) prof_u = (blockIdx.x == 0 && (u == 8 || u == 9)) ? u - 8 : -1;
      stamp(0);
      // ---- first half: the rest of M0(u), sums -> LDS   (group 1: requests, rows [0,32) of tile u-1)
      if (u < U) m0_tail(u);
      stamp(1);
      publish();
      stamp(2);
      // ---- second half: rows [32,64) of tile u-1, head of M0(u+1)   (group 1: M1(u))
      // the residual rows of tile u go where those of tile u-2 were (read for the last time in iteration u-1); needed in
      // iteration u+1.  Requested ahead of my rows and stores, awaited behind them with the stores left outstanding.  (The
      // rows of the first tile came with the prologue.)
      if (u >= 1 && u < U) issue_dma(c_cur, 0, c_cur, u & 1, false, has_res);
      __builtin_amdgcn_sched_barrier(0);
      if (u >= 1) {
        row_slot(c_prev, (u - 1) & 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        wait_vmcnt<NS>();
      }
      if (u + 1 < U) m0_head(u + 1);
      stamp(3);
      publish();
      stamp(4);
      advance();
    }
  } else {
    for (int u = 0; u <= U; ++u) {
      if constexpr (PROF) prof_u = (blockIdx.x == 0 && (u == 8 || u == 9)) ? u - 8 : -1;
      stamp(0);
      // ---- first half: requests, rows [0,32) of tile u-1   (group 0: the rest of M0(u))
      // patch(u+1) goes where patch(u-1) was (read for the last time in iteration u-1); group 0 starts on it in the second half, so the
      // requests are awaited HERE, behind my rows: they have landed once only the stores I issued behind them are outstanding
      if (u < U) issue_dma(c_next, (u + 1) & 1, c_next, 0, u + 1 < U, false);
      __builtin_amdgcn_sched_barrier(0);                       // the counted waits below rely on "requests, then stores" in issue order
      if (u >= 1) row_slot(c_prev, (u - 1) & 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (u >= 1) wait_vmcnt<NS>();
      else wait_vmcnt<0>();
      stamp(1);
      publish();
      stamp(2);
      // ---- second half: M1(u)   (group 0: rows [32,64) of tile u-1, head of M0(u+1))
      if (u < U) m1_phase(u);
      stamp(7);
      stamp(3);
      publish();
      stamp(4);
      advance();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

}  // namespace

// conv_igemm.hip's dispatcher hands over launches that qualify (ws_eligible, conv_select.h); dtype = VT_BF16 / VT_F16
extern "C" __attribute__((visibility("hidden"))) int vt_ws2_launch(const void* args, int dtype, void* stream_) {
  const ConvArgs& a = *reinterpret_cast<const ConvArgs*>(args);
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const bool keep = a.ln_mode == 0 || a.ln_keep_y != 0;
  int vi = a.ln_mode == 0 ? 0 : (a.ln_mode == 1 ? (keep ? 1 : 2) : (keep ? 3 : 4));
  constexpr int NK = 12;
  static const void* const kerns[NK] = {
      reinterpret_cast<const void*>(&conv3x3_ws2_kernel<bf16_t, 0, true>), reinterpret_cast<const void*>(&conv3x3_ws2_kernel<bf16_t, 1, true>),
      reinterpret_cast<const void*>(&conv3x3_ws2_kernel<bf16_t, 1, false>), reinterpret_cast<const void*>(&conv3x3_ws2_kernel<bf16_t, 2, true>),
      reinterpret_cast<const void*>(&conv3x3_ws2_kernel<bf16_t, 2, false>), reinterpret_cast<const void*>(&conv3x3_ws2_kernel<bf16_t, 0, true, true>),
      reinterpret_cast<const void*>(&conv3x3_ws2_kernel<bf16_t, 2, true, true>),
      reinterpret_cast<const void*>(&conv3x3_ws2_kernel<f16_t, 0, true>), reinterpret_cast<const void*>(&conv3x3_ws2_kernel<f16_t, 1, true>),
      reinterpret_cast<const void*>(&conv3x3_ws2_kernel<f16_t, 1, false>), reinterpret_cast<const void*>(&conv3x3_ws2_kernel<f16_t, 2, true>),
      reinterpret_cast<const void*>(&conv3x3_ws2_kernel<f16_t, 2, false>)};
  if (a.prof != nullptr) {           // vt_conv_profile: the plain and the LayerNorm+SiLU (y kept) bf16 instantiations carry stamps [8 waves][16]
    VT_CHECK_ARG(dtype == VT_BF16 && (vi == 0 || vi == 3), "vt_conv_profile (weight-stationary kernel): bf16, ln_mode 0, or 2 with ln_keep_y");
    vi = vi == 0 ? 5 : 6;
  } else if (dtype == VT_F16) {
    vi += 7;
  }
  static std::atomic<int> cus[kMaxDevices];         // 0 = not set up on that device yet; else its CU count
  int dev = 0;
  VT_CHECK_HIP(hipGetDevice(&dev));
  const bool dev_ok = dev >= 0 && dev < kMaxDevices;
  int ncu = dev_ok ? cus[dev].load(std::memory_order_acquire) : 0;
  if (ncu == 0) {
    for (int k = 0; k < NK; ++k) VT_CHECK_HIP(hipFuncSetAttribute(kerns[k], hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS));
    VT_CHECK_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    if (ncu <= 0) ncu = 256;
    if (dev_ok) cus[dev].store(ncu, std::memory_order_release);
  }
  const int ntiles = (a.Wo / W2_TW) * (a.Ho / W2_TH) * a.B * a.To;   // 4 x 16-pixel tiles (ws_eligible guarantees Ho % 8 == 0, Wo % 16 == 0)
  const int grid = ntiles < ncu ? ntiles : ncu;     // one persistent workgroup per CU (nearly all of its LDS)
  ConvArgs args_copy = a;
  void* kargs[] = {&args_copy};
  VT_CHECK_HIP(hipLaunchKernel(kerns[vi], dim3((unsigned)grid), dim3(512), kargs, W2_LDS, stream));
  return VT_OK;
}
