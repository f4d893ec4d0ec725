// Implicit-GEMM convolution for NDHWC activations on gfx950 (MI355X) -- see include/vidtok_amd.h
// (vt_conv) for the operator contract and the list of reference call sites it replaces.
//
// GEMM view:  Y[m][n] = sum_k  X_gather[m][k] * W[n][k],   m = output pixel (b,to,ho,wo),
//             n = output channel, k = tap*Cin + c.
//
// Kernel structure (conv_igemm_glds_kernel):
//   * workgroup tile 128 pixels x 128 channels (4 wave64, two workgroups per CU), 256 x 256 (8 waves)
//     for Cout % 256 == 0 layers with many pixels, 256 x 32/64 for narrow outputs; K step = 128 BYTES per
//     tile row (64 bf16 / 32 fp32), so 8 consecutive lanes fetch one full 128-B line of a pixel's
//     channel vector (NDHWC keeps C innermost) or of a weight row;
//   * operands go global -> LDS by LDS-DMA (buffer_load ... lds / global_load_lds), no VGPR round trip;
//     bank conflicts are removed by an XOR swizzle applied on the SOURCE side; 2-stage ring, the DMA of
//     step s+1 flies during the MFMAs of step s (details at the kernel);
//   * MFMA 32x32 tiles with the operand roles SWAPPED: the weight fragment is the A operand (rows = n)
//     and the pixel fragment the B operand (cols = m).  Every lane then owns one pixel and 4
//     *consecutive* output channels per accumulator quad, so the epilogue issues 16-B (fp32) / 8-B
//     (bf16) vector loads/stores on the NDHWC rows;
//   * the K mapping inside a 16-B fragment is the same bijection for both operands, which is all an
//     inner product needs: bf16 uses v_mfma_f32_32x32x16_bf16 (one 16-B read = one MFMA), fp32 uses
//     v_mfma_f32_32x32x2_f32 (one 16-B read feeds 4 MFMAs); fp32 results are bit-wise an fmaf chain;
//   * padding, causal time padding (zero / replicate / cache), stride, and nearest-neighbour x2
//     up-sampling in space or time are folded into the gather addresses: nothing is materialised;
//   * epilogue: + bias, + residual or alpha-mix, dtype conversion, NDHWC vector store or NCTHW (fp32,
//     with front time trim) store; the 128 x 128 tile transposes through the LDS first so rows are
//     written with whole-line 16-B accesses, and can emit LayerNorm(+SiLU) of the result from there
//     (conv_epilogue_lds128);
//   * XCD-aware tile order: consecutive tiles of one XCD are neighbouring pixel tiles of the same
//     channel tile, so halo rows and the weight slab are shared in that XCD's L2; temporal convs walk
//     the frames innermost (launch_variant).
//
// Measured dead ends (kept out of the code): register-staged operand tiles
// (ds_write_b128 pass: 627 vs 440 TFLOP/s aggregate), a 4-stage ring of 64-B rows (no gain), 256x128 tiles
// with 8 or 4 waves (slower), 256x256 on 4 waves with a 128x128 wave tile and all 512 registers (7-9 % slower: round 2),
// 3-4 waves/SIMD with 64-B rows (slower), LayerNorm in the MFMA-layout epilogue
// (+1.6 ms per conv), a kw-innermost K walk (less fabric traffic, more time), a persistent variant with a deferred
// epilogue (conv_stream.hip in the git history: +3 % only, and its counted-vmcnt drain was not race-free in fp32).
#include <atomic>
#include <type_traits>

#pragma once
#include "conv_select.h"

namespace {

// Epilogue shared by both staging variants: + bias, + residual / alpha-mix, dtype conversion, NDHWC
// vector store (lane = one pixel, 4 consecutive channels per accumulator quad) or NCTHW fp32 store.
// Two straight-line paths chosen by ONE uniform branch: the fast path (full channel tile, NDHWC,
// 4-aligned strides: every layer of the resolution pyramid) issues ALL bias / residual vector loads
// back to back, then the arithmetic, then vector stores -- one memory latency per tile; the general
// path (ragged channel tails, NCTHW, narrow outputs) does the same per 32-pixel row group with
// clamped scalar loads.  (The first version branched and waited per element: ~26 us per tile.)
template <typename TOut, int TM, int TN, bool GENERAL = true>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, f32x16 (&acc)[TN][TM], int m_blk, int n_blk, int bn_tile,
                                              int wm, int wn, int lane, long long z) {
  TOut* __restrict__ yg = reinterpret_cast<TOut*>(p.y) + z * p.ys_z;
  const TOut* __restrict__ rg = reinterpret_cast<const TOut*>(p.res) + z * p.rs_z;
  const long long HWo = (long long)p.Ho * p.Wo;
  const bool has_res = p.res_mode != VT_RES_NONE;
  const int nlast = p.Cout - 1;
  const int nq = n_blk + wn * TN * 32 + 4 * (lane >> 5);   // + 32*a + 8*g : first channel of quad (a,g)

  float alpha = 0.0f;
  if (p.res_mode == VT_RES_MIX) alpha = 1.0f / (1.0f + __expf(-p.mix_factor[0]));

  // per-pixel addresses (rows beyond M are clamped for loads; their stores are masked)
  bool store_ok[TM];
  long long mr[TM], ybase[TM];
  long long mrow[TM];
#pragma unroll
  for (int b = 0; b < TM; ++b) {
    int m = m_blk + (wm * TM + b) * 32 + (lane & 31);
    store_ok[b] = m < p.M;
    if (m >= p.M) m = p.M - 1;
    mrow[b] = out_row(p, m);
    mr[b] = m;
    ybase[b] = 0;
    if (p.res_tshift != 0 || p.Tr != p.To || p.out_layout == VT_NCTHW) {
      const long long hw = m % HWo;
      const long long r = m / HWo;
      const int to = (int)(r % p.To);
      const int bb = (int)(r / p.To);
      mr[b] = ((long long)bb * p.Tr + (to >> p.res_tshift)) * HWo + hw;
      if (p.out_layout == VT_NCTHW) {
        const int Tout = p.To - p.t_trim;
        store_ok[b] = store_ok[b] && (to >= p.t_trim);
        ybase[b] = ((long long)bb * p.Cout * Tout + (to - p.t_trim)) * HWo + hw;
      }
    }
  }

  // GENERAL == false: the launcher guarantees the fast-path conditions (256x256 tile), so the scalar
  // path -- 128 address computations that would spill next to 128 accumulators -- is compiled out
  const bool fast = !GENERAL || ((p.out_layout == VT_NDHWC) && (n_blk + bn_tile <= p.Cout) && ((p.ldy & 3) == 0) &&
                                 (!has_res || (p.ldr & 3) == 0));
  if (fast) {
    // ---- fast path: straight-line vector code, GA channel sub-tiles (<= 64 channels) per batch ----
    constexpr int GA = (TN >= 2 && TM * TN <= 4) ? 2 : 1;   // big tiles: one sub-tile per batch (register budget)
#pragma unroll
    for (int a0 = 0; a0 < TN; a0 += GA) {
      Quad<TOut> rq[GA][TM][4];
      f32x4 bq[GA][4];
      if (has_res) {
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
          for (int a = 0; a < GA; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g)
              rq[a][b][g].v = *reinterpret_cast<const decltype(rq[a][b][g].v)*>(rg + mr[b] * p.ldr + nq + 32 * (a0 + a) + 8 * g);
      }
      if (p.bias) {
#pragma unroll
        for (int a = 0; a < GA; ++a)
#pragma unroll
          for (int g = 0; g < 4; ++g) bq[a][g] = *reinterpret_cast<const f32x4*>(p.bias + nq + 32 * (a0 + a) + 8 * g);
      } else {
#pragma unroll
        for (int a = 0; a < GA; ++a)
#pragma unroll
          for (int g = 0; g < 4; ++g) bq[a][g][0] = bq[a][g][1] = bq[a][g][2] = bq[a][g][3] = 0.0f;
      }
#pragma unroll
      for (int b = 0; b < TM; ++b)
#pragma unroll
        for (int a = 0; a < GA; ++a)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] = acc[a0 + a][b][4 * g + e] + bq[a][g][e];
              if (p.res_mode == VT_RES_ADD) v[e] = rq[a][b][g].get(e) + v[e];
              if (p.res_mode == VT_RES_MIX) v[e] = alpha * rq[a][b][g].get(e) + (1.0f - alpha) * v[e];
            }
            if (store_ok[b]) store_quad<TOut>(yg + mrow[b] * p.ldy + nq + 32 * (a0 + a) + 8 * g, v);
          }
    }
    return;
  }
  if (!GENERAL) return;

  // ---- general path: scalar, clamped loads first, masked stores after; one 32x32 sub-tile at a time ----
#pragma unroll
  for (int b = 0; b < TM; ++b) {
#pragma unroll
    for (int a = 0; a < TN; ++a) {
      float bv[4][4], rv[4][4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int n = min(nq + 32 * a + 8 * g + e, nlast);
          bv[g][e] = p.bias ? p.bias[n] : 0.0f;
          rv[g][e] = has_res ? to_f32<TOut>(rg[mr[b] * p.ldr + n]) : 0.0f;
        }
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int n = nq + 32 * a + 8 * g + e;
          float v = acc[a][b][4 * g + e] + bv[g][e];
          if (p.res_mode == VT_RES_ADD) v = rv[g][e] + v;
          if (p.res_mode == VT_RES_MIX) v = alpha * rv[g][e] + (1.0f - alpha) * v;
          if (store_ok[b] && n < p.Cout) {
            if (p.out_layout == VT_NCTHW)
              yg[ybase[b] + (long long)n * (p.To - p.t_trim) * HWo] = from_f32<TOut>(v);
            else
              yg[mrow[b] * p.ldy + n] = from_f32<TOut>(v);
          }
        }
    }
  }
}

// Coalesced epilogue of the 128 x 128 tile (4 waves).  In the MFMA layout a lane owns one pixel and 4 consecutive
// channels per quad, so a wave-level store touches 32 pixel rows with 8 B (bf16) each -- 64 separate 8-B L2
// transactions per instruction, and the residual read likewise.  Measured on the widest level (same-run, stores /
// residual switched off one by one): the 2.7 GB of epilogue traffic of a 128->128 conv cost 0.5-0.6 ms of a 2.5 ms
// launch and did not overlap with the co-resident workgroup's K loop: the L2 path is transaction-bound, not
// byte-bound.  Here the tile (+ bias) is transposed through the LDS the K loop no longer needs (128 x 128 fp32 =
// the 64 KiB of the two stages): every lane then handles 8 consecutive channels of a pixel row with 16 lanes per
// 128 channels, so a wave instruction covers whole 128-B lines (4 rows x 256 B bf16, 4 rows x 512 B fp32).
//   LDS layout: T[pixel][32 chunks of 4 floats], chunk index XOR (pixel & 31): the 8 lanes a ds_write_b128 services
//   together hold 8 different pixels of one chunk -> 8 different columns; a ds_read_b128 group reads 16 different
//   chunks of one row -> 16 different bank quads.
// (ALLOW_RES = false: an instantiation for launches that never carry a residual, without the eight residual rows' registers.
// NA x NB: the wave's accumulators cover channels [c_base, c_base + 32 NA) x pixel rows [prow_base, prow_base + 32 NB) of the tile:
// 2 x 2 at (64 wn, 64 wm) for the implicit-GEMM kernel's wave grid.  conv_in8_kernel's 1 x 4 grid has its own two-pass copy of this
// row phase, conv_epilogue_in8, on a 64-row buffer.)
template <typename TOut, bool ALLOW_RES = true, int NA = 2, int NB = 2>
__device__ __forceinline__ void conv_epilogue_lds128_at(const ConvArgs& p, f32x16 (&acc)[NA][NB], int m_blk, int n_blk, int c_base,
                                                        int prow_base, int lane, int tid, char* smem, long long z) {
  TOut* __restrict__ yg = reinterpret_cast<TOut*>(p.y) + z * p.ys_z;
  const TOut* __restrict__ rg = reinterpret_cast<const TOut*>(p.res) + z * p.rs_z;
  float* T = reinterpret_cast<float*>(smem);
  const bool has_res = ALLOW_RES && p.res_mode != VT_RES_NONE;
  float alpha = 0.0f;
  if (p.res_mode == VT_RES_MIX) alpha = 1.0f / (1.0f + __expf(-p.mix_factor[0]));
  // read-back mapping: 16 lanes per pixel row, 8 consecutive channels each (two adjacent 16-B chunks of the row:
  // the XOR swizzle keeps a pair adjacent), rows row0 + 16*it.  A wave instruction covers 4 rows x 256 B (bf16);
  // the LayerNorm statistics of a row reduce over 16 lanes with DPP only.
  constexpr int NT = 8;
  const int oct_j = tid & 15;
  const int row0 = tid >> 4;
  // residual first: its latency rides under the transposition
  Oct<TOut> rq[ALLOW_RES ? NT : 1];
  if (has_res) {
    const bool remap = p.res_tshift != 0 || p.Tr != p.To;   // uniform
    const long long HWo = (long long)p.Ho * p.Wo;
    long long mres[NT];
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      long long m = m_blk + row0 + 16 * it;
      if (remap) {
        const long long hw = m % HWo;
        const long long r = m / HWo;
        const int to = (int)(r % p.To);
        const int bb = (int)(r / p.To);
        m = ((long long)bb * p.Tr + (to >> p.res_tshift)) * HWo + hw;
      }
      mres[it] = m;
    }
#pragma unroll
    for (int it = 0; it < NT; ++it) rq[it].load(rg + mres[it] * p.ldr + n_blk + 8 * oct_j);
  }
  __syncthreads();                            // every wave has finished reading the last stage
  {
    const int h = lane >> 5;
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = c_base + 32 * a + 8 * g + 4 * h;         // first channel of the quad inside the tile
        f32x4 bq;
        if (p.bias) bq = *reinterpret_cast<const f32x4*>(p.bias + n_blk + c);
        else bq[0] = bq[1] = bq[2] = bq[3] = 0.0f;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int prow = prow_base + b * 32 + (lane & 31);
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[a][b][4 * g + e] + bq[e];
          *reinterpret_cast<f32x4*>(T + prow * 128 + (((c >> 2) ^ (prow & 31)) << 2)) = v;
        }
      }
  }
  __syncthreads();
  // fused LayerNorm (+SiLU) of the finished rows (launcher: Cout = 128, so the 16 lanes of a row hold all of it)
  float lg[8], lb[8];
  TOut* __restrict__ ng = reinterpret_cast<TOut*>(p.ln_out);
  if (p.ln_mode) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      lg[e] = ln_fold(p.ln_gamma[8 * oct_j + e], sizeof(TOut) == 2 && p.ln_mode == 2);     // 16-bit storage + SiLU: the affine carries -log2(e)
      lb[e] = ln_fold(p.ln_beta[8 * oct_j + e], sizeof(TOut) == 2 && p.ln_mode == 2);
    }
  }
#pragma unroll
  for (int it = 0; it < NT; ++it) {
    const int row = row0 + 16 * it;
    const int sw = row & 31;
    const f32x4 t0 = *reinterpret_cast<const f32x4*>(T + row * 128 + (((2 * oct_j) ^ sw) << 2));
    const f32x4 t1 = *reinterpret_cast<const f32x4*>(T + row * 128 + (((2 * oct_j + 1) ^ sw) << 2));
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] = e < 4 ? t0[e] : t1[e - 4];
      if (ALLOW_RES && p.res_mode == VT_RES_ADD) v[e] = rq[it].get(e) + v[e];
      if (ALLOW_RES && p.res_mode == VT_RES_MIX) v[e] = alpha * rq[it].get(e) + (1.0f - alpha) * v[e];
    }
    const long long orow = out_row(p, m_blk + row);
    if (!p.ln_mode || p.ln_keep_y) Oct<TOut>::store(yg + orow * p.ldy + n_blk + 8 * oct_j, v, p.nt_store != 0);
    if (p.ln_mode) {   // uniform; statistics of the fp32 row, taken before the rounding to TOut
      float o[8];
      if constexpr (sizeof(TOut) == 2) {   // 16-bit storage: the one-pass form every fused LayerNorm site of these modes shares (ln_row8, common.h; lg / lb folded above)
        if (p.ln_mode == 2) ln_row8<16, true>(v, lg, lb, p.ln_eps, o);
        else ln_row8<16, false>(v, lg, lb, p.ln_eps, o);
      } else {                              // fp32 storage (fp32 / split-bf16 modes): two-pass statistics like layernorm_act_kernel
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
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float u = v[e] * rstd * lg[e] + lb[e];
          o[e] = (p.ln_mode == 2) ? silu_fast(u) : u;
        }
      }
      Oct<TOut>::store(ng + orow * p.ldn + 8 * oct_j, o, p.nt_store != 0);
    }
  }
}

template <typename TOut>
__device__ __forceinline__ void conv_epilogue_lds128(const ConvArgs& p, f32x16 (&acc)[2][2], int m_blk, int n_blk, int wm,
                                                     int wn, int lane, int tid, char* smem, long long z) {
  conv_epilogue_lds128_at<TOut, true, 2, 2>(p, acc, m_blk, n_blk, wn * 64, wm * 64, lane, tid, smem, z);
}

// LayerNorm-fusing / row-coalescing epilogue of the 8-wave 256 x 256 tile (Cout % 256 == 0; with a LayerNorm: Cout = 256, so the two
// waves that share a pixel's channels meet in the LDS anyway).  The tile goes through the LDS (the ring the K loop no longer needs) as
// the waves' pixel sub-tiles b = 0 / 1 one after the other -- every wave parks 64 accumulators per half, nothing spills (the round-2
// form sent the upper / lower 128 pixel rows: half of the waves carried their 128 accumulators through the other half's row loop,
// 48 spilled registers) --, the residual rows a lane handles in a half are requested before the transposition and its two barriers,
// a lane owns 8 consecutive channels of a row (16-byte accesses in the 16-bit types, 2 x 16 in fp32: a row is one contiguous run of
// 32 lanes), and the row arithmetic runs on channel pairs (v_pk_add / mul / fma_f32: no MFMA runs on this CU during an epilogue, so
// packed fp32 costs what it says, 3 cycles per element instead of 5).
// Also the epilogue of launches WITHOUT a LayerNorm (ln_mode 0; any Cout % 256 == 0: n_blk = the tile's first channel): the
// transposition alone turns the 8-byte pieces of the MFMA layout into 16-byte accesses covering whole lines, for the stores and for
// the residual / alpha-mix operand (the vector epilogue's wave store is 64 separate 8-B transactions).
// LDS layout: T[128 rows][64 chunks of 4 floats], chunk index XOR (row & 63); row w*32 + l of half b = tile pixel
// w*64 + b*32 + l.  A lane's two chunks (2j, 2j+1) make its ds_read_b128 pair 2-way bank-conflicted (16 lanes of a service
// group hit 8 bank quads); the LDS is idle here, the global accesses are what counts.
template <typename TOut>
__device__ __forceinline__ void conv_epilogue_lds256(const ConvArgs& p, f32x16 (&acc)[4][2], int m_blk, int n_blk, int wm, int wn, int lane,
                                                     int tid, char* smem, long long z) {
#pragma clang fp contract(off)
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  TOut* __restrict__ yg = reinterpret_cast<TOut*>(p.y) + z * p.ys_z;
  const TOut* __restrict__ rg = reinterpret_cast<const TOut*>(p.res) + z * p.rs_z;
  TOut* __restrict__ ng = reinterpret_cast<TOut*>(p.ln_out);
  float* T = reinterpret_cast<float*>(smem);
  const bool has_res = p.res_mode != VT_RES_NONE;
  const bool has_ln = p.ln_mode != 0;                    // uniform
  float alpha = 0.0f;
  if (p.res_mode == VT_RES_MIX) alpha = 1.0f / (1.0f + __expf(-p.mix_factor[0]));
  constexpr int RS = 16;                                // T rows per sweep of the block (32 lanes per row)
  const int j = tid & 31, rsub = tid >> 5;              // lane j: channels [8j, 8j+8) of T rows rsub + RS it
  f32x2 lg[4], lb[4];
  if (has_ln) {
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.ln_gamma + 8 * j), g1 = *reinterpret_cast<const f32x4*>(p.ln_gamma + 8 * j + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.ln_beta + 8 * j), b1 = *reinterpret_cast<const f32x4*>(p.ln_beta + 8 * j + 4);
    lg[0] = f32x2{g0[0], g0[1]}; lg[1] = f32x2{g0[2], g0[3]}; lg[2] = f32x2{g1[0], g1[1]}; lg[3] = f32x2{g1[2], g1[3]};
    lb[0] = f32x2{b0[0], b0[1]}; lb[1] = f32x2{b0[2], b0[3]}; lb[2] = f32x2{b1[0], b1[1]}; lb[3] = f32x2{b1[2], b1[3]};
    if (sizeof(TOut) == 2 && p.ln_mode == 2) {            // 16-bit storage + SiLU: the affine carries -log2(e) (ln_row8, common.h)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        lg[q] = lg[q] * kNegLog2e;
        lb[q] = lb[q] * kNegLog2e;
      }
    }
  }
  const int h = lane >> 5;
  auto half = [&](auto pz_c) {
    constexpr int PZ = decltype(pz_c)::value;            // compile-time: acc[.][PZ] must not become a dynamic register index
    // tile pixel of T row r = rsub + RS it:  (r >> 5) * 64 + PZ * 32 + (r & 31)   (rsub < RS, RS divides 32: r >> 5 = it RS >> 5)
    Oct<TOut> rq[8];
    if (has_res) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const long long m = (long long)m_blk + ((RS * it) >> 5) * 64 + PZ * 32 + rsub + ((RS * it) & 31);
        rq[it].load(rg + m * p.ldr + n_blk + 8 * j);
      }
    }
    __syncthreads();                                    // K loop / previous half: everybody is done with this LDS
    {
      const int prl = wm * 32 + (lane & 31);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = wn * 128 + 32 * a + 8 * g + 4 * h;
          f32x4 bq;
          if (p.bias) bq = *reinterpret_cast<const f32x4*>(p.bias + n_blk + c);
          else bq[0] = bq[1] = bq[2] = bq[3] = 0.0f;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[a][PZ][4 * g + e] + bq[e];
          *reinterpret_cast<f32x4*>(T + prl * 256 + (((c >> 2) ^ (prl & 63)) << 2)) = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int r = rsub + RS * it;
      const int m = m_blk + ((RS * it) >> 5) * 64 + PZ * 32 + rsub + ((RS * it) & 31);
      const long long orow = out_row(p, m);
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(T + r * 256 + (((2 * j) ^ (r & 63)) << 2));
      const f32x4 t1 = *reinterpret_cast<const f32x4*>(T + r * 256 + (((2 * j + 1) ^ (r & 63)) << 2));
      f32x2 v[4] = {f32x2{t0[0], t0[1]}, f32x2{t0[2], t0[3]}, f32x2{t1[0], t1[1]}, f32x2{t1[2], t1[3]}};
      if (p.res_mode == VT_RES_ADD) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = f32x2{rq[it].get(2 * q), rq[it].get(2 * q + 1)} + v[q];
      } else if (p.res_mode == VT_RES_MIX) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = f32x2{rq[it].get(2 * q), rq[it].get(2 * q + 1)} * alpha + v[q] * (1.0f - alpha);
      }
      if (p.ln_keep_y || !has_ln) {
        const float yv[8] = {v[0][0], v[0][1], v[1][0], v[1][1], v[2][0], v[2][1], v[3][0], v[3][1]};
        Oct<TOut>::store(yg + orow * p.ldy + n_blk + 8 * j, yv, p.nt_store != 0);
      }
      if (!has_ln) continue;
      float o[8];
      if constexpr (sizeof(TOut) == 2) {
        // 16-bit storage: the one-pass form of ln_row8 (common.h) on channel pairs -- both moments together, t = x rstd - mean rstd, and
        // for SiLU a = t g' + b' with -log2(e) folded into lg / lb above: u sigmoid(u) = a * rcp(fma(2^a, -log2e, -log2e))
        const f32x2 s = (v[0] + v[1]) + (v[2] + v[3]);
        f32x2 qq = v[0] * v[0];
#pragma unroll
        for (int q = 1; q < 4; ++q) qq = __builtin_elementwise_fma(v[q], v[q], qq);
        float nm;
        const float rstd = ln_row_stats<32>(s[0] + s[1], qq[0] + qq[1], p.ln_eps, &nm);
        const f32x2 rstd2 = {rstd, rstd}, nm2 = {nm, nm}, c2 = {kNegLog2e, kNegLog2e};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x2 u = __builtin_elementwise_fma(__builtin_elementwise_fma(v[q], rstd2, nm2), lg[q], lb[q]);
          if (p.ln_mode == 2) {
            const f32x2 den = __builtin_elementwise_fma(f32x2{__builtin_amdgcn_exp2f(u[0]), __builtin_amdgcn_exp2f(u[1])}, c2, c2);
            u = u * f32x2{__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
          }
          o[2 * q] = u[0];
          o[2 * q + 1] = u[1];
        }
      } else {
        const f32x2 s = (v[0] + v[1]) + (v[2] + v[3]);
        const float mean = group_sum_dpp<32>(s[0] + s[1]) * (1.0f / 256.0f);
        f32x2 d[4], qq = {0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          d[q] = v[q] - mean;
          qq = __builtin_elementwise_fma(d[q], d[q], qq);
        }
        const float rstd = __builtin_amdgcn_rsqf(__builtin_fmaf(group_sum_dpp<32>(qq[0] + qq[1]), 1.0f / 256.0f, p.ln_eps));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x2 u = __builtin_elementwise_fma(d[q] * rstd, lg[q], lb[q]);
          if (p.ln_mode == 2) {                            // u * sigmoid(u), the arithmetic of silu_fast on a pair
            const f32x2 t = u * -1.4426950408889634f;
            const f32x2 e = f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + 1.0f;
            u = u * f32x2{__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
          }
          o[2 * q] = u[0];
          o[2 * q + 1] = u[1];
        }
      }
      Oct<TOut>::store(ng + orow * p.ldn + 8 * j, o, p.nt_store != 0);
    }
  };
  half(std::integral_constant<int, 0>{});
  half(std::integral_constant<int, 1>{});
}

// ------------------------------------------------------------------------------------------------
// The operand tiles go global -> LDS with LDS-DMA (no VGPR round trip, no ds_write pass -- a
// register-staged first version spent ~415 LDS cycles per K step on ds_write_b128 against 512 MFMA cycles).  The DMA writes each wave's 64
// lanes x 16 B to a contiguous 1 KiB, so a tile row is exactly 128 B (no pad) and instruction i of
// wave w fills rows 32*i + 8*w .. +8.  Bank conflicts are removed with an XOR swizzle applied on the
// SOURCE side (guide rule 21): the lane that writes 16-B slot `pos` of row r fetches logical K chunk
// pos ^ ((r>>1)&7); the fragment read of logical chunk c goes to slot c ^ ((r>>1)&7), which spreads
// every 16-lane ds_read_b128 service group over all 16 slots of the 256-B bank row.  Taps that fall
// in padding (and rows beyond M / Cout, K beyond taps*Cin) fetch from a zero page instead.
// Pipeline: 2 LDS stages; per K step  wait own DMA (vmcnt 0) -> barrier -> issue DMA of step s+1 ->
// MFMAs of step s, so a full step of MFMA work covers the DMA flight.  On the fast path (Cin a
// multiple of the K step) the gather addresses are recomputed only when the tap changes.
// ------------------------------------------------------------------------------------------------
[[maybe_unused]] __device__ u32x4 g_zero_page[8];   // 128 B of zeros (static device memory, zero-initialised)


// ROWB   bytes of K per tile row and pipeline step (128 or 64): BK = ROWB / sizeof(MT)
// STAGES LDS ring depth; STAGES-1 steps of DMA are kept in flight (counted vmcnt + raw s_barrier:
//        __syncthreads() would drain the DMA queue, guide section 5 "Pipelining across barriers")
// BUF    gather through buffer descriptors (buffer_load ... lds): the per-lane address is ONE 32-bit byte
//        offset against an SGPR descriptor and out-of-range offsets read zeros in hardware, so padding
//        taps / ragged rows need no zero-page select and no 64-bit pointer arithmetic (the K loop of the
//        short-K layers is instruction-issue bound: ~13 VALU per MFMA with pointers).  Needs the tensors
//        below 4 GiB and no cache-mode time padding; otherwise the pointer form is used.
// LN256  1: the 8-wave tile with the LDS-transposed epilogue (conv_epilogue_lds256: LayerNorm, or plain coalesced rows) -- its own
//        instantiations: the mere presence of a second epilogue path slowed every 256-tile convolution by 8 % through
//        register allocation (round 1)
// SCHED  K-step schedule of the 8-wave tile on descriptors: 0 plain loop (also every 4-wave tile and the pointer form), 1 software-
//        pipelined single body (fp32 operands), 2 two-group ping-pong (16-bit operands), 5 two groups for the split-bf16 arithmetic
template <typename MT, typename TOut, int WAVES_M, int WAVES_N, int TM, int TN, bool FAST, int ROWB, int STAGES, bool BUF, int LN256 = 0, bool PROF = false, int SCHED = 0>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, 2) void conv_igemm_glds_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)   // device pass only: the host pass needs just the launch stub (buffer-descriptor types are device-only)
  constexpr int THREADS = 64 * WAVES_M * WAVES_N;   // 4 waves (128x128, 256x32/64 tiles) or 8 waves (256x256)
  constexpr int NS = ROWB / 16;                     // 16-B slots per tile row
  constexpr int RSTEP = THREADS / NS;               // rows covered by one DMA instruction of the whole block
  constexpr int ROWS_PER_WAVE = 64 / NS;
  constexpr int SWZ_SHIFT = (ROWB == 128) ? 1 : 2;  // rows per 256-B bank window = 256 / ROWB
  constexpr int VEC = 16 / (int)sizeof(MT);
  constexpr int BK = ROWB / (int)sizeof(MT);
  constexpr int KS = ROWB / 32;                     // 32-B (one MFMA K group) sub-steps per stage
  constexpr int BM = WAVES_M * TM * 32;
  constexpr int BN = WAVES_N * TN * 32;
  constexpr int A_VECS = BM * NS / THREADS;
  constexpr int B_VECS = BN * NS / THREADS;
  constexpr int IPS = A_VECS + B_VECS;              // DMA instructions per thread and stage
  constexpr int A_BYTES = BM * ROWB;
  constexpr int STAGE_BYTES = (BM + BN) * ROWB;
  constexpr int D = STAGES - 1;                     // prefetch distance
  constexpr bool X3 = is_split3<MT>::value;         // split-bf16 arithmetic on fp32 storage (conv_common.h)
  static_assert(ROWB == 128 || ROWB == 64, "row bytes");
  static_assert(A_VECS >= 1 && B_VECS >= 1 && BM % RSTEP == 0 && BN % RSTEP == 0, "tile / workgroup mismatch");
  static_assert(D >= 1 && D <= 3 && (D - 1) * IPS <= 63, "pipeline depth");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* glb_ptr_t;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave % WAVES_M;
  const int wn = wave / WAVES_M;

  const int tile = xcd_remap(blockIdx.x, p.m_tiles * p.n_tiles);
  const int nt = tile / p.m_tiles;
  int mt = tile - nt * p.m_tiles;
  if (p.hw_tiles > 0) {   // frames innermost: the kt taps of a tile were fetched by its predecessors on this XCD
    const int per_b = p.hw_tiles * p.To;
    const int b = mt / per_b;
    const int r = mt - b * per_b;
    const int hwt = r / p.To;
    mt = (b * p.To + (r - hwt * p.To)) * p.hw_tiles + hwt;
  }
  const int m_blk = mt * BM;
  const int n_blk = nt * BN;

  const long long z = blockIdx.z;                  // problem of a batched launch, or (ksplit) the time-tap plane: y advances, x / w do not
  const MT* __restrict__ xg = reinterpret_cast<const MT*>(p.x) + z * p.xs_z;
  const MT* __restrict__ wg = reinterpret_cast<const MT*>(p.w) + z * p.ws_z;
  const MT* __restrict__ cg = reinterpret_cast<const MT*>(p.cache);
  const MT* zero = reinterpret_cast<const MT*>(g_zero_page);
  constexpr unsigned kOob = 0xFFFF0000u;   // BUF: offset beyond any descriptor's num_records -> hardware zero fill
  // BUF: the extents are loop state -- the steps after the last real prefetch still issue their DMA pieces,
  // against extent 0 (every lane out of range: zeros into a slot nobody reads again, no memory traffic), so
  // the K loop has no branch around any piece and ONE straight-line body
  unsigned ext_x = BUF ? p.x_bytes : 0u, ext_w = BUF ? p.w_bytes : 0u;
  // BUF in cache mode (v1.1 chunks after the first): a tile lies in ONE output frame (launch_variant checks), so a time tap
  // reads either the cache or x for all of its rows -- the descriptor of the activation pieces is switched per tap
  const MT* x_cur = xg;
  if constexpr (PROF) {            // measurement (option ws_prof_mode): bit 0 / bit 1 = activation / weight pieces become zero fills (no memory traffic)
    if (p.prof_mode & 1) ext_x = 0u;
    if (p.prof_mode & 2) ext_w = 0u;
  }

  const int pos = tid % NS;                                  // 16-B slot this lane writes in its rows
  const int srow = tid / NS;                                 // rows srow + RSTEP*i
  const int chunk = pos ^ ((srow >> SWZ_SHIFT) & (NS - 1));  // logical K chunk this lane fetches (same for all i)
  // wave-uniform part of the DMA destination, made provably uniform so the M0 set-up stays on the SALU
  const int lds_row_off = __builtin_amdgcn_readfirstlane(wave * ROWS_PER_WAVE * ROWB);

  const int Hv = p.Hi << p.ups_s, Wv = p.Wi << p.ups_s;
  const int Tv = p.Ti << p.ups_t;

  int a_b[A_VECS], a_t0[A_VECS], a_h0[A_VECS], a_w0[A_VECS];
#pragma unroll
  for (int i = 0; i < A_VECS; ++i) {
    const int m = m_blk + srow + RSTEP * i;
    if (m < p.M) {
      const unsigned r1 = fast_div((unsigned)m, p.fd_wo);
      const int wo = m - (int)r1 * p.Wo;
      const unsigned r2 = fast_div(r1, p.fd_ho);
      const int ho = (int)r1 - (int)r2 * p.Ho;
      const unsigned r3 = fast_div(r2, p.fd_to);
      const int to = (int)r2 - (int)r3 * p.To;
      a_b[i] = (int)r3;
      a_t0[i] = to * p.st - p.pt;
      a_h0[i] = ho * p.sh - p.ph;
      a_w0[i] = wo * p.sw - p.pw;
    } else {
      a_b[i] = -1;
      a_t0[i] = a_h0[i] = a_w0[i] = 0;
    }
  }
  const MT* b_row[B_VECS];
  unsigned b_off[B_VECS];
  // SCHED 2 stages the weight rows as two half-tiles (rows [0, 128) = the channel sub-tiles a = 0, 1 of both wave columns,
  // rows [128, 256) = a = 2, 3), each refilled as soon as its own last fragment read is over: LDS row
  // (a >> 1) * 128 + wn * 64 + (a & 1) * 32 + l holds channel wn * 128 + a * 32 + l of the tile
  constexpr bool S2 = SCHED == 2 && (WAVES_M * WAVES_N == 8) && FAST && BUF;
#pragma unroll
  for (int j = 0; j < B_VECS; ++j) {
    const int lr = srow + RSTEP * j;
    const int n = n_blk + (S2 ? ((lr >> 6) & 1) * 128 + ((((lr >> 7) & 1) << 1) | ((lr >> 5) & 1)) * 32 + (lr & 31) : lr);
    b_row[j] = (n < p.Cout) ? wg + (long long)n * p.ldw : nullptr;
    b_off[j] = (n < p.Cout) ? (unsigned)n * (unsigned)p.ldw * (unsigned)sizeof(MT) + (FAST ? (unsigned)chunk * 16u : 0u) : kOob;
    if (p.ksplit && n < p.Cout) b_off[j] += (unsigned)blockIdx.z * p.plane_bytes;       // my tap plane of the row (descriptor form only)
  }

  // Gather address of input row i for tap (kt,kh,kw).  Straight-line integer arithmetic (unsigned compares
  // fold the >= 0 tests, bitwise & instead of && so no exec-mask branches are generated); the only
  // branch left is the uniform cache-mode one of the pointer form (v1.1 later chunks).
  const bool replicate = p.tmode == VT_TPAD_REPLICATE;
  const unsigned pix_bytes = (unsigned)p.Cin * (unsigned)sizeof(MT);
  // descriptor form: byte offset into x, kOob for padding (tmode ZERO / REPLICATE only)
  auto row_off = [&](int i, int kt, int kh, int kw) -> unsigned {
    const int tv = a_t0[i] + kt;
    const int hv = a_h0[i] + kh;
    const int wv = a_w0[i] + kw;
    const bool ok = (a_b[i] >= 0) & ((unsigned)hv < (unsigned)Hv) & ((unsigned)wv < (unsigned)Wv) & (tv < Tv) &
                    ((tv >= 0) | replicate);
    const int ti = max(tv, 0) >> p.ups_t;
    const unsigned pix = (unsigned)(((a_b[i] * p.Ti + ti) * p.Hi + (hv >> p.ups_s)) * p.Wi + (wv >> p.ups_s));
    return ok ? pix * pix_bytes : kOob;
  };
  // pointer form: first element of the row, or nullptr when it reads padding
  auto row_ptr = [&](int i, int kt, int kh, int kw) -> const MT* {
    const int tv = a_t0[i] + kt;
    const int hv = a_h0[i] + kh;
    const int wv = a_w0[i] + kw;
    bool ok = (a_b[i] >= 0) & ((unsigned)hv < (unsigned)Hv) & ((unsigned)wv < (unsigned)Wv) & (tv < Tv);
    const MT* base = xg;
    int tstore = p.Ti;
    int ti = max(tv, 0) >> p.ups_t;
    if (p.tmode == VT_TPAD_CACHE) {            // uniform
      if (tv < 0) {
        base = cg;
        tstore = p.ncache;
        ti = p.ncache + tv;
        ok = ok & (ti >= 0);
      }
    } else {
      ok = ok & ((tv >= 0) | replicate);
    }
    const long long pix = (((long long)a_b[i] * tstore + ti) * p.Hi + (hv >> p.ups_s)) * p.Wi + (wv >> p.ups_s);
    return ok ? base + pix * p.Cin : nullptr;
  };

  const int khw = p.KH * p.KW;
  const int cpb = FAST ? (p.Cin / BK) : 1;
  const MT* a_ptr[A_VECS];       // FAST pointer form: cached per tap
  unsigned a_off[A_VECS];        // descriptor form: byte offset of this lane's 16 B in x, or kOob

  // FAST descriptor form: the gather offset of a tap is  time part (changes with kt only) + a wave-uniform
  // (kh,kw) displacement, and whether the tap reads padding is ONE bit test against a per-row mask built once
  // per tile -- 4 VALU per row and tap instead of the ~25 (three of them quarter-rate multiplies) of the
  // general formula: on the short-K layers (Cin = 128: a new tap every other K step) the address work was
  // more than half of the instructions of the K loop, which is instruction-issue bound.
  //   a_mask: bit kh = row (h0+kh) inside the image, bit 8+kw likewise for columns, bit 16 = time tap valid
  //           (set per kt), bits 17/18 = parity of h0 / w0 (x2 nearest up-sampling folded into the gather:
  //           (h0+kh)>>1 = (h0>>1) + ((h0&1)+kh)>>1, i.e. base + uniform part + parity * uniform part)
  unsigned a_mask[A_VECS], a_tb[A_VECS];
  int a_bt[A_VECS], a_hw[A_VECS];
  if constexpr (FAST && BUF) {
#pragma unroll
    for (int i = 0; i < A_VECS; ++i) {
      unsigned mk = 0;
      if (a_b[i] >= 0) {
        for (int kh = 0; kh < p.KH; ++kh) mk |= ((unsigned)(a_h0[i] + kh) < (unsigned)Hv) ? (1u << kh) : 0u;
        for (int kw = 0; kw < p.KW; ++kw) mk |= ((unsigned)(a_w0[i] + kw) < (unsigned)Wv) ? (1u << (8 + kw)) : 0u;
        mk |= (unsigned)(a_h0[i] & 1) << 17;
        mk |= (unsigned)(a_w0[i] & 1) << 18;
      }
      a_mask[i] = mk;
      a_bt[i] = a_b[i] * p.Ti;
      a_hw[i] = (a_h0[i] >> p.ups_s) * p.Wi + (a_w0[i] >> p.ups_s);   // arithmetic shifts: floor also for the -1 halo
      a_tb[i] = 0;
    }
  }

  int coff = 0, koff = 0;   // element offsets of this lane's chunk: in the pixel's channel vector / weight row
  bool kvalid = true;
  // FAST path: pipeline steps are prepared strictly in order, so the position in the K walk (tap-major, channel
  // chunks innermost) advances as scalar counters instead of being recovered from the step index with three
  // integer divisions per step.
  // (split-K: the walk starts at my tap plane -- kt = z, or kh = z for a convolution without time taps -- and nsteps ends it there)
  int q_step = 0, q_cc = 0, q_kt = p.ksplit == 1 ? (int)blockIdx.z : 0, q_kh = p.ksplit == 2 ? (int)blockIdx.z : 0, q_kw = 0;
  // Causal zero padding (tmode ZERO, the v1.0 models): the time taps in front of the clip multiply zeros.  Where a tile lies
  // inside ONE output frame (p.tskip: launch_variant checks) those taps are the same for all of its rows, so its K walk simply
  // starts at the first tap plane that reads a frame of the clip -- kt0 = -(to * st - pt) for the first frames, 0 elsewhere --
  // and is that many planes shorter: a third / two thirds of the K steps of frames 1 / 0 of a 3-tap convolution, 3 of the 3 T
  // tap planes of a clip (20 % at the T = 5 levels).  The fp32 sum of an output is unchanged (the skipped products are exact
  // zeros added to a partial sum that starts at +0).
  int nsteps = p.nsteps;
  if constexpr (FAST && BUF) {
    if (p.tskip) {
      const int kt0 = min(max(-__builtin_amdgcn_readfirstlane(a_t0[0]), 0), p.KT - 1);
      q_kt = kt0;
      q_step = kt0 * khw * cpb;                      // the weight rows are walked from that plane on (s_b = q_step * ROWB)
      nsteps -= q_step;
    }
  }
  unsigned s_a = 0, s_b = 0;   // BUF: wave-uniform byte offsets (soffset operand): chunk-in-tap for x, k offset for w
  const unsigned chunk_bytes = (unsigned)chunk * 16u;
  const unsigned HiWi = (unsigned)p.Hi * (unsigned)p.Wi;
  // addresses of the NEXT pipeline step (VALU/SALU only; the DMA pieces are fired separately so they can
  // be interleaved with the MFMAs of the stage being computed)
  auto prep_step = [&](int s) {
    if constexpr (PROF) {
      if (p.prof_mode & 8) return;             // measurement: no address arithmetic (the pieces keep their first addresses)
    }
    if (FAST) {
      if (q_cc == 0) {               // uniform branch: new tap -> new gather addresses
        if constexpr (BUF) {
          if ((q_kh | q_kw) == 0 || q_step == 0) {  // new kt (rare; or the first step of a walk that starts inside a plane): time part of the offsets, time-padding bit
            bool from_cache = false;
            if constexpr (!PROF) {
              if (p.tmode == VT_TPAD_CACHE) {            // uniform
                const int tv_u = __builtin_amdgcn_readfirstlane(a_t0[0]) + q_kt;     // the tile's frame: the same for every row
                from_cache = tv_u < 0;
                x_cur = from_cache ? cg : xg;
                ext_x = from_cache ? p.c_bytes : p.x_bytes;
              }
            }
            if (from_cache) {                            // frame ncache + tv of the cache [B][ncache][Hi][Wi][Cin] (ncache >= pt: vt_conv checks)
#pragma unroll
              for (int i = 0; i < A_VECS; ++i) {
                const unsigned ti = (unsigned)(p.ncache + a_t0[i] + q_kt);
                a_tb[i] = (((unsigned)(a_b[i] * p.ncache) + ti) * HiWi + (unsigned)a_hw[i]) * pix_bytes + chunk_bytes;
                a_mask[i] |= 1u << 16;
              }
            } else {
#pragma unroll
              for (int i = 0; i < A_VECS; ++i) {
                const int tv = a_t0[i] + q_kt;
                const bool ok = (tv < Tv) & ((tv >= 0) | replicate);
                const unsigned ti = (unsigned)(max(tv, 0) >> p.ups_t);
                a_tb[i] = (((unsigned)a_bt[i] + ti) * HiWi + (unsigned)a_hw[i]) * pix_bytes + chunk_bytes;
                a_mask[i] = (a_mask[i] & ~(1u << 16)) | (ok ? (1u << 16) : 0u);
              }
            }
          }
          const unsigned tm = (1u << q_kh) | (1u << (8 + q_kw)) | (1u << 16);
          if (p.ups_s == 0) {
            const unsigned delta = (unsigned)(q_kh * p.Wi + q_kw) * pix_bytes;
#pragma unroll
            for (int i = 0; i < A_VECS; ++i) a_off[i] = ((a_mask[i] & tm) == tm) ? a_tb[i] + delta : kOob;
            if constexpr (PROF) {
              // measurement (ws_prof_mode bit 7): every activation piece is gathered from the first 128 KiB of x -- live, non-zero data
              // (the matrix pipe's power draw stays what it is) that sits in every XCD's L2 after the first tiles: what the launch would
              // take if the activation gather cost no traffic beyond the L2 (the control the zero-fill probe lacks; wrong results)
              if (p.prof_mode & 128) {
#pragma unroll
                for (int i = 0; i < A_VECS; ++i) a_off[i] = a_off[i] == kOob ? kOob : (a_off[i] & 0x1FFF0u);
              }
            }
          } else {
            const unsigned delta = (unsigned)((q_kh >> 1) * p.Wi + (q_kw >> 1)) * pix_bytes;
            const unsigned dh = (q_kh & 1) ? (unsigned)p.Wi * pix_bytes : 0u;
            const unsigned dw = (q_kw & 1) ? pix_bytes : 0u;
#pragma unroll
            for (int i = 0; i < A_VECS; ++i) {
              const unsigned ph = (unsigned)(((int)(a_mask[i] << 14)) >> 31) & dh;   // parity bit 17 -> all-ones mask
              const unsigned pw = (unsigned)(((int)(a_mask[i] << 13)) >> 31) & dw;   // parity bit 18
              a_off[i] = ((a_mask[i] & tm) == tm) ? a_tb[i] + delta + ph + pw : kOob;
            }
          }
        } else {
#pragma unroll
          for (int i = 0; i < A_VECS; ++i) a_ptr[i] = row_ptr(i, q_kt, q_kh, q_kw);
        }
      }
      coff = q_cc * BK + chunk * VEC;
      koff = q_step * BK + chunk * VEC;
      s_a = (unsigned)q_cc * (unsigned)ROWB;
      s_b = (unsigned)q_step * (unsigned)ROWB;
      if constexpr (PROF) {
        // measurement (ws_prof_mode bit 6): every K step reads the FIRST 128 bytes of its weight rows -- live, non-zero data (the matrix
        // pipe's power draw stays what it is) that sits in the L2 after the first step: what the launch would take if the weight slab cost
        // no traffic beyond the L2 (an upper bound for any slab-sharing scheme; wrong results)
        if (p.prof_mode & 64) s_b = 0u;
      }
      ++q_step;
      if (++q_cc == cpb) {
        q_cc = 0;
        if (++q_kw == p.KW) {
          q_kw = 0;
          if (++q_kh == p.KH) {
            q_kh = 0;
            ++q_kt;
          }
        }
      }
    } else {
      const int k = s * BK + chunk * VEC;
      const int tap = k / p.Cin;
      coff = k - tap * p.Cin;
      koff = k;
      kvalid = tap < p.ntaps;
      const int kt = tap / khw;
      const int r2 = tap - kt * khw;
      const int kh = r2 / p.KW;
      const int kw = r2 - kh * p.KW;
#pragma unroll
      for (int i = 0; i < A_VECS; ++i) {
        if constexpr (BUF) a_off[i] = (kvalid ? row_off(i, kt, kh, kw) : kOob) + (unsigned)coff * (unsigned)sizeof(MT);
        else a_ptr[i] = kvalid ? row_ptr(i, kt, kh, kw) : nullptr;
      }
      s_a = 0;
      s_b = 0;
    }
  };
  // DMA piece q (0 .. IPS-1) of the prepared step into ring slot `stage`
  auto fire_piece = [&](int q, int stage) {
    if constexpr (PROF) {
      if (p.prof_mode & 4) return;             // measurement: no DMA requests at all (the K loop computes on stale tiles)
    }
    char* As = smem + stage * STAGE_BYTES + lds_row_off;
    if (q < A_VECS) {
      if constexpr (BUF) {
        const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<MT*>(x_cur), 0, ext_x, 0x00020000);
        // (a non-temporal hint on these gathers -- "let the L2 drop x first, keep the weight slab" -- was measured and lost:
        // 833 -> 778-793 frames/s, profiles/r03_traffic_by_layer_group.txt; and a run-time switch in front of every piece is a
        // branch in the K loop)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(As + (RSTEP * q) * ROWB), 16, a_off[q], s_a, 0, 0);
      } else {
        const MT* src = a_ptr[q] ? a_ptr[q] + coff : zero;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(As + (RSTEP * q) * ROWB), 16, 0, 0);
      }
    } else {
      const int j = q - A_VECS;
      if constexpr (BUF) {
        // FAST: lane offset is fixed for the tile (row + chunk), the step advances through soffset;
        // general: the lane's k offset is folded in here
        // (split-bf16 weight rows are zero-padded to whole K steps and their 16-byte chunks hold other k than the x chunk at
        // the same offset: no per-chunk validity on the weight side)
        const unsigned off = FAST ? b_off[j] : ((kvalid || X3) ? b_off[j] + (unsigned)koff * (unsigned)sizeof(MT) : kOob);
        const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<MT*>(wg), 0, ext_w, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(As + A_BYTES + (RSTEP * j) * ROWB), 16, off, s_b, 0, 0);
      } else {
        const MT* src = (b_row[j] && (kvalid || X3)) ? b_row[j] + koff : zero;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(As + A_BYTES + (RSTEP * j) * ROWB), 16, 0, 0);
      }
    }
  };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  const int frag_row = (lane & 31) * ROWB;
  const int swz = ((lane & 31) >> SWZ_SHIFT) & (NS - 1);
  const int khalf = lane >> 5;
  constexpr int NM = KS * TM * TN;                    // MFMA groups (one per 16-B fragment pair) per stage
  constexpr bool EIGHT_WAVES = (WAVES_M * WAVES_N == 8);
  // SCHED 1 (8-wave tile; measured with vt_conv_profile, profiles/r02_igemm_step_cycles.txt): a K step of the plain
  // schedule lasts ~3 590 cycles against 2 048 of MFMA work per SIMD -- the two waves of a SIMD sit in the same phase,
  // the older one wins every arbitration, finishes its 32 MFMAs in ~1 550 cycles and then waits > 1 000 at the barrier
  // while the younger one finishes alone (its DMA issue and ds_reads covered by nobody); and ~700 cycles per step pass
  // with no MFMA at all: the last DMA piece is issued after the last MFMA and awaited at once, then barrier, then the
  // address set-up of the next step.  So: all pieces go out in the FIRST half of the stage (they have the second half
  // to land), the address set-up of the step after next runs in the middle of the stage (VALU in the other wave's MFMA
  // shadow), and the two waves of a SIMD swap issue priority at half time.
  constexpr bool S1 = SCHED == 1 && EIGHT_WAVES;
  constexpr int FIRE_SPAN = S1 ? NM / 2 : NM;         // MFMA groups over which the DMA pieces of the next stage are spread
  constexpr int MPP = (FIRE_SPAN + IPS - 1) / IPS;    // ... per DMA piece
  const int wgrp = __builtin_amdgcn_readfirstlane((int)(tid >> 8));   // 8 waves: 0 = first wave of its SIMD, 1 = second (uniform: a scalar branch, not an exec mask)

  // stage `stage` -> MFMAs; if FIRE, the IPS DMA pieces of the prepared step go to ring slot `dst`,
  // one after every MPP MFMA groups, so their issue cost hides under the matrix pipe.  Fragments are
  // read KSB sub-steps at a time (all of the stage for the 4-wave tiles, half for the 8-wave tile whose
  // 128 accumulators leave no room for 24 live fragments).
  constexpr int KSB = (TM * TN >= 8 && KS > 2) ? 2 : KS;
  auto compute_stage = [&](int stage, auto fire_tag, bool fire_rt, int dst) {
    constexpr bool FIRE = decltype(fire_tag)::value;
    const char* As = smem + stage * STAGE_BYTES + (wm * TM * 32) * ROWB + frag_row;
    const char* Bs = smem + stage * STAGE_BYTES + A_BYTES + (wn * TN * 32) * ROWB + frag_row;
    if constexpr (X3) {
      // a row holds ROWB / 64 groups of 16 k-values: x as 16 fp32 (chunks 4 kg .. 4 kg + 3; a lane's 8 values = chunks
      // 4 kg + 2 khalf, + 1), w as [hi 16 x bf16 | lo 16 x bf16] (a lane's hi fragment = chunk 4 kg + khalf, lo = + 2)
      constexpr int KG = ROWB / 64;
      constexpr int NMX = KG * 3 * TM * TN;
      constexpr int MPPX = (NMX + IPS - 1) / IPS;
#pragma unroll
      for (int kg = 0; kg < KG; ++kg) {
        u32x4 whi[TN], wlo[TN], xhi[TM], xlo[TM];
        const int sh = ((4 * kg + khalf) ^ swz) * 16, sl = ((4 * kg + 2 + khalf) ^ swz) * 16;
        const int s0 = ((4 * kg + 2 * khalf) ^ swz) * 16, s1 = ((4 * kg + 2 * khalf + 1) ^ swz) * 16;
#pragma unroll
        for (int a = 0; a < TN; ++a) {
          whi[a] = *reinterpret_cast<const u32x4*>(Bs + a * 32 * ROWB + sh);
          wlo[a] = *reinterpret_cast<const u32x4*>(Bs + a * 32 * ROWB + sl);
        }
#pragma unroll
        for (int b = 0; b < TM; ++b) {
          const u32x4 r0 = *reinterpret_cast<const u32x4*>(As + b * 32 * ROWB + s0);
          const u32x4 r1 = *reinterpret_cast<const u32x4*>(As + b * 32 * ROWB + s1);
          split3_x(r0, r1, xhi[b], xlo[b]);
        }
#pragma unroll
        for (int qq = 0; qq < 3 * TM * TN; ++qq) {       // small terms first: x_lo w_hi, x_hi w_lo, x_hi w_hi
          const int pr = qq / (TM * TN), a = (qq / TM) % TN, b = qq % TM;
          mma_bf16(pr == 1 ? wlo[a] : whi[a], pr == 0 ? xlo[b] : xhi[b], acc[a][b]);
          const int q = kg * 3 * TM * TN + qq;
          if (FIRE && (q + 1) % MPPX == 0) {
            const int piece = q / MPPX;
            if (piece < IPS && fire_rt) fire_piece(piece, dst);
          }
        }
      }
      if (FIRE && fire_rt) {
#pragma unroll
        for (int piece = NMX / MPPX; piece < IPS; ++piece) fire_piece(piece, dst);
      }
      return;
    } else {
#pragma unroll
    for (int k0 = 0; k0 < KS; k0 += KSB) {
      u32x4 wf[KSB][TN], xf[KSB][TM];
#pragma unroll
      for (int kk = 0; kk < KSB; ++kk) {
        const int slot = (((k0 + kk) * 2 + khalf) ^ swz) * 16;
#pragma unroll
        for (int a = 0; a < TN; ++a) wf[kk][a] = *reinterpret_cast<const u32x4*>(Bs + a * 32 * ROWB + slot);
#pragma unroll
        for (int b = 0; b < TM; ++b) xf[kk][b] = *reinterpret_cast<const u32x4*>(As + b * 32 * ROWB + slot);
      }
#pragma unroll
      for (int qq = 0; qq < KSB * TM * TN; ++qq) {
        const int kk = qq / (TM * TN), a = (qq / TM) % TN, b = qq % TM;
        const int q = k0 * TM * TN + qq;
        mma_step<MT>(wf[kk][a], xf[kk][b], acc[a][b]);
        if (FIRE && (q + 1) % MPP == 0) {
          const int piece = q / MPP;
          if (piece < IPS && fire_rt) fire_piece(piece, dst);
        }
      }
    }
    if (FIRE && fire_rt) {   // pieces the MFMA groups did not cover (more DMA pieces than MFMA groups: narrow tiles)
#pragma unroll
      for (int piece = FIRE_SPAN / MPP; piece < IPS; ++piece) fire_piece(piece, dst);
    }
    }
  };

  // SCHED 5: the two-group schedule of the split-bf16 arithmetic on the 8-wave tile (below)
  constexpr bool S3 = SCHED == 5 && X3 && (WAVES_M * WAVES_N == 8) && FAST && BUF;
  // how many of the 4 DMA pieces of step s + 3 go out in LOAD(s); the others between the MFMAs of COMPUTE(s).  (All four in the LOAD
  // phase, all four between the MFMAs, and a one-barrier "stream" schedule with every wave in the same phase were measured and lost:
  // profiles/r04_bf16x3_step_cost_by_mode_sched*.txt)
  constexpr int S3_NL = 2;
  if constexpr (!S2 && !S3) {
#pragma unroll
    for (int d = 0; d < D; ++d)
      if (d < nsteps) {
        prep_step(d);
#pragma unroll
        for (int q = 0; q < IPS; ++q) fire_piece(q, d);
      }
  }
  if constexpr (S1) {                        // the addresses of a step are ready one stage before its pieces are fired
    if (D < nsteps) prep_step(D);
  }
  int stage = 0;
  const int n_fire = nsteps - D;          // steps that still have a successor to prefetch
  // PROF (vt_conv_profile): s_memtime at the phase boundaries of K steps [8, 12) of workgroup 0, parked in the LDS
  // behind the ring and copied out after the loop
  unsigned long long* stamps = reinterpret_cast<unsigned long long*>(smem + STAGES * STAGE_BYTES);
  int s_cur = 0;
  auto stamp = [&](int k) {
    if constexpr (PROF) {
      if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && s_cur >= 8 && s_cur < 12) {
        const unsigned long long ts = __builtin_amdgcn_s_memtime();
        if (lane == 0) stamps[((s_cur - 8) * (THREADS / 64) + (tid >> 6)) * 8 + k] = ts;
      }
    }
  };
  if constexpr (S3) {
    // Schedule 5: split-bf16 arithmetic (X3) on the 8-wave 256 x 256 tile.  A K step is ONE group of 16 k-values (rows of
    // 64 bytes: 16 fp32 of a pixel / [hi | lo] bf16 planes of a weight row), the ring has four slots, and a wave alternates
    //   LOAD(s)     12 fragment reads (x: 2 pixel sub-tiles x 2 chunks of fp32; w: 4 channel sub-tiles x (hi, lo)), the
    //               addresses and the 4 DMA pieces of step s + 3, the split of the 32 fp32 values into bf16 hi / lo
    //               fragments (~100 VALU) -- no MFMA
    //   COMPUTE(s)  24 MFMAs (x_lo w_hi, x_hi w_lo, x_hi w_hi for the 8 accumulator tiles) = 768 matrix-pipe cycles
    // with one barrier after each; waves 4-7 (the second wave of every SIMD) run one barrier behind waves 0-3, so a SIMD
    // always has one wave feeding the matrix pipe while the other owns the remaining issue slots (the structure of
    // schedule 2; with three MFMAs per fragment pair instead of one the LOAD phase fits under the partner's COMPUTE phase).
    // Slot s & 3 is read in LOAD(s) by both groups (the later one before barrier 2 s + 2, reads retired by lgkmcnt(0) in
    // front of it), so the pieces of step s + 3 = slot (s - 1) & 3 may go out in LOAD(s); a wave's pieces are issued
    // 4 per step in step order, so "all but the youngest 8" = vmcnt(8) at the end of LOAD(s) says its pieces of step
    // s + 1 have landed -- one barrier before anyone reads them, two steps after they were requested.
    static_assert(ROWB == 64 && STAGES == 4 && TM == 2 && TN == 4 && A_VECS == 2 && B_VECS == 2, "schedule 3: 8-wave tile, 64-byte rows, 4 slots");
    const int grp = __builtin_amdgcn_readfirstlane((int)(tid >> 8));     // 0: waves 0-3, 1: waves 4-7 (one barrier behind)
    const char* a_base = smem + (wm * TM * 32) * ROWB + frag_row;
    const char* b_base = smem + A_BYTES + (wn * TN * 32) * ROWB + frag_row;
    const int sh = (khalf ^ swz) * 16, sl = ((2 + khalf) ^ swz) * 16;
    const int s0 = ((2 * khalf) ^ swz) * 16, s1 = ((2 * khalf + 1) ^ swz) * 16;
    u32x4 whi[TN], wlo[TN], xhi[TM], xlo[TM];
    // prologue: steps 0, 1, 2 -- 12 pieces whatever nsteps is (pieces of steps that do not exist go out against extent 0)
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if (d < nsteps) prep_step(d);
      else ext_x = ext_w = 0u;
#pragma unroll
      for (int q = 0; q < IPS; ++q) fire_piece(q, d);
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 1) {
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    // PROF stamps per step: 0 LOAD start, 1 reads / pieces issued + x split, 2 COMPUTE start (waits + barrier over), 3 COMPUTE end,
    // 4 trailing barrier over
    for (int s = 0; s < nsteps; ++s) {
      const int stg = s & 3;
      s_cur = s;
      stamp(0);
      // ---- LOAD(s)
      {
        const char* As = a_base + stg * STAGE_BYTES;
        const char* Bs = b_base + stg * STAGE_BYTES;
        u32x4 r[TM][2];
#pragma unroll
        for (int b = 0; b < TM; ++b) {
          r[b][0] = *reinterpret_cast<const u32x4*>(As + b * 32 * ROWB + s0);
          r[b][1] = *reinterpret_cast<const u32x4*>(As + b * 32 * ROWB + s1);
        }
#pragma unroll
        for (int a = 0; a < TN; ++a) {
          whi[a] = *reinterpret_cast<const u32x4*>(Bs + a * 32 * ROWB + sh);
          wlo[a] = *reinterpret_cast<const u32x4*>(Bs + a * 32 * ROWB + sl);
        }
        if (s + 3 < nsteps) prep_step(s + 3);
        else ext_x = ext_w = 0u;                     // past the last step: the pieces below turn into zero fills
#pragma unroll
        for (int q = 0; q < S3_NL; ++q) fire_piece(q, (s + 3) & 3);
        bool do_split = true;
        if constexpr (PROF) do_split = !(p.prof_mode & 16);       // measurement: no split (the fragments keep stale values)
        if (do_split) {
#pragma unroll
          for (int b = 0; b < TM; ++b) split3_x(r[b][0], r[b][1], xhi[b], xlo[b]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);             // the split stays in the LOAD phase (left alone, half of it sinks behind the barrier)
      stamp(1);
      // my pieces of step s + 1 have landed: all but the youngest 8 (steps s + 2, s + 3), or 4 where step s + 3 is still to be requested
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 + S3_NL) : "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      stamp(2);
      // ---- COMPUTE(s): 24 MFMAs, nothing else
      bool do_mma = true;
      if constexpr (PROF) do_mma = !(p.prof_mode & 32);           // measurement: no MFMAs
      if (do_mma) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int qq = 0; qq < 3 * TM * TN; ++qq) {
          const int pr = qq / (TM * TN), a = (qq / TM) % TN, b = qq % TM;
          mma_bf16(pr == 1 ? wlo[a] : whi[a], pr == 0 ? xlo[b] : xhi[b], acc[a][b]);
          if constexpr (S3_NL < 4) {
            constexpr int GAP = 24 / (4 - S3_NL);    // one piece behind MFMAs 1, 1 + GAP, ...
            if (qq % GAP == 1) {
              __builtin_amdgcn_sched_barrier(0);
              fire_piece(S3_NL + qq / GAP, (s + 3) & 3);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
        __builtin_amdgcn_s_setprio(0);
      }
      __builtin_amdgcn_sched_barrier(0);
      stamp(3);
      if (!(grp == 1 && s + 1 == nsteps)) {      // waves 4-7 entered one barrier late: they leave without the last one
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
      stamp(4);
    }
  } else if constexpr (S2) {
    // Schedule 2, "ping-pong" (the structure of the guide's 256 x 256 template, adapted to the gather).
    // What the stamps of schedule 1 show (profiles/r02_igemm_step_cycles_sched1.txt): the two waves of a SIMD run the same
    // mixed stream of fragment reads, DMA pieces and MFMAs side by side, the older one wins every arbitration, finishes
    // its 32 MFMAs after ~2 200 cycles and parks 850-980 cycles at the stage barrier while the younger one completes alone
    // -- 3 180 cycles per K step against 2 048 of matrix work.  Here a wave alternates between a LOAD phase (fragment
    // reads + DMA pieces, no MFMA) and a COMPUTE phase (16 back-to-back MFMAs, nothing else), a barrier after each, and
    // waves 4-7 (the second wave of every SIMD) run one barrier behind waves 0-3: at any time one wave of a SIMD feeds
    // the matrix pipe and the other owns the rest of the issue slots.
    //   phase 2s:    LOAD  x fragments of both pixel sub-tiles + w fragments of channel sub-tiles 0, 1  (16 ds_read_b128),
    //                      DMA of W1(s+1);      COMPUTE acc[0..1][*] += ...   (16 MFMAs)
    //   phase 2s+1:  LOAD  w fragments of channel sub-tiles 2, 3 (8 reads; x stays in registers),
    //                      address set-up of step s+2, DMA of XX(s+2) and W0(s+2);   COMPUTE acc[2..3][*]
    // Half-tiles of a stage: XX = the 256 pixel rows (4 DMA pieces per lane), W0 / W1 = weight rows [0,128) / [128,256) (2
    // pieces each).  A half-tile's LDS region is refilled for step s+2 right behind its last read of step s (both
    // groups' reads are retired -- lgkmcnt(0) in front of the barrier that ends a LOAD phase -- one barrier before the
    // first piece goes out), so a piece has three phases (~3 000 cycles) to land instead of the half stage of schedule 1.
    // In issue order a wave's pieces are ... [XX(s) W0(s)] [W1(s)] [XX(s+1) W0(s+1)] [W1(s+1)] ...: with 6 + 2 pieces per
    // two phases, "everything but the youngest 8" = vmcnt(8) at the end of EVERY load phase is exactly what the next
    // phase reads, and a barrier lies between that wait and any other wave's read (guide: read a staged buffer one
    // phase after the wait that retires it).
    static_assert(STAGES == 2 && D == 1 && KS == 4 && TM == 2 && TN == 4 && A_VECS == 4 && B_VECS == 4, "schedule 2: the 8-wave 256 x 256 tile");
    // Two variations were measured and dropped (history: commits 1d8f768..c099519, profiles/r03_igemm_step_cycles_sched{3,4}.txt):
    // the DMA pieces moved from the load phases into the compute phases (847 against 863 frames/s), and that plus the
    // barrier in front of the fragment wait (818 against 841).  What the K loop costs when nothing but MFMAs, fragment reads
    // and one barrier per step is in it: scripts/mfma_tile_bench.hip (2 130-2 260 cycles per step against 2 048).
    const int grp = __builtin_amdgcn_readfirstlane((int)(tid >> 8));     // 0: waves 0-3, 1: waves 4-7 (one barrier behind)
    const char* a_base = smem + (wm * TM * 32) * ROWB + frag_row;
    const char* b_base = smem + A_BYTES + (wn * 64) * ROWB + frag_row;
    u32x4 xf[TM][KS], wf[2][KS];
    auto read_x = [&](int stg) {
      const char* As = a_base + stg * STAGE_BYTES;
#pragma unroll
      for (int b = 0; b < TM; ++b)
#pragma unroll
        for (int k = 0; k < KS; ++k) xf[b][k] = *reinterpret_cast<const u32x4*>(As + b * 32 * ROWB + (((k * 2 + khalf) ^ swz) * 16));
    };
    auto read_w = [&](int stg, int hf) {
      const char* Bs = b_base + stg * STAGE_BYTES + hf * 128 * ROWB;
#pragma unroll
      for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
        for (int k = 0; k < KS; ++k) wf[a2][k] = *reinterpret_cast<const u32x4*>(Bs + a2 * 32 * ROWB + (((k * 2 + khalf) ^ swz) * 16));
    };
    auto end_load = [&]() {   // my pieces for the next phase's reads have landed, my reads of this phase are retired
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    // COMPUTE phase of channel half HF: 16 MFMAs, nothing else
    auto compute = [&](auto hf_c) {
      constexpr int HF = decltype(hf_c)::value;
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int k = 0; k < KS; ++k)
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
          for (int b = 0; b < TM; ++b) mma_step<MT>(wf[a2][k], xf[b][k], acc[2 * HF + a2][b]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // prologue: [XX(0) W0(0)] [W1(0)] [XX(1) W0(1)] -- 14 pieces whatever nsteps is (pieces of steps that do not exist go out
    // against extent 0: zero fill, no traffic), then the common start barrier
    prep_step(0);
#pragma unroll
    for (int q = 0; q < IPS; ++q) fire_piece(q, 0);
    if (1 < nsteps) prep_step(1);
    else ext_x = ext_w = 0u;
#pragma unroll
    for (int q = 0; q < A_VECS + 2; ++q) fire_piece(q, 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 1) {
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    // PROF stamps per step: 0 L(2s) start, 1 reads / pieces issued, 2 C(2s) start (wait + barrier over), 3 C(2s) end,
    // 4 L(2s+1) start (barrier over), 5 issued, 6 C(2s+1) start, 7 C(2s+1) end
    for (int s = 0; s < nsteps; ++s) {
      const int stg = s & 1;
      s_cur = s;
      stamp(0);
      // ---- phase 2s
      read_x(stg);
      read_w(stg, 0);
      fire_piece(A_VECS + 2, stg ^ 1);             // W1(s+1): addresses of step s+1 are the last ones prepared
      fire_piece(A_VECS + 3, stg ^ 1);
      stamp(1);
      end_load();
      stamp(2);
      compute(I0{});
      stamp(3);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      stamp(4);
      // ---- phase 2s+1
      read_w(stg, 1);
      if (s + 2 < nsteps) prep_step(s + 2);
      else ext_x = ext_w = 0u;                     // past the last step: the pieces below turn into zero fills
      // XX(s+2), W0(s+2) into the regions read for the last time in phase 2s
#pragma unroll
      for (int q = 0; q < A_VECS + 2; ++q) fire_piece(q, stg);
      stamp(5);
      end_load();
      stamp(6);
      compute(I1{});
      stamp(7);
      if (!(grp == 1 && s + 1 == nsteps)) {      // waves 4-7 entered one barrier late: they leave without the last one
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else if constexpr (S1) {
    // Schedule 1: one K loop body, software-pipelined across stages.  Fragments are requested one 32-B sub-step ahead
    // of their MFMAs (two register sets).  The stage barrier sits in front of the LAST FOUR MFMA groups of a stage: by
    // then this wave has every fragment of the current slot in registers (so the slot may be refilled) and its own
    // pieces of the next stage have had most of a stage to land (measured: in front of the last EIGHT groups they had
    // not -- 400 cycles of vmcnt wait); after the barrier the first fragments of the next stage are requested and the
    // remaining MFMAs cover their latency: no MFMA waits for an LDS round trip.
    static_assert(STAGES == 2 && D == 1 && KS % 2 == 0, "schedule 1: one stage in flight on two slots");
    constexpr int BAR_AT = TM * TN - 4;      // MFMA groups of the last sub-step issued before the barrier
    constexpr int MPP1 = (NM / 2) / IPS >= 1 ? (NM / 2) / IPS : 1;       // MFMA groups per DMA piece: every piece is out before prep_step at half time
    static_assert(IPS * MPP1 <= NM / 2, "schedule 1: the pieces of a stage must fit its first half");
    const char* a_base = smem + (wm * TM * 32) * ROWB + frag_row;
    const char* b_base = smem + A_BYTES + (wn * TN * 32) * ROWB + frag_row;
    u32x4 wf[2][TN], xf[2][TM];
    auto read_frags = [&](int stg, int k, int set) {
      const char* As = a_base + stg * STAGE_BYTES;
      const char* Bs = b_base + stg * STAGE_BYTES;
      const int slot = ((k * 2 + khalf) ^ swz) * 16;
#pragma unroll
      for (int a = 0; a < TN; ++a) wf[set][a] = *reinterpret_cast<const u32x4*>(Bs + a * 32 * ROWB + slot);
#pragma unroll
      for (int b = 0; b < TM; ++b) xf[set][b] = *reinterpret_cast<const u32x4*>(As + b * 32 * ROWB + slot);
    };
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_frags(0, 0, 0);
    for (int s = 0; s < nsteps; ++s) {
      s_cur = s;
      stamp(0);
      const bool fire = s < n_fire;
      if (!fire && BUF) ext_x = ext_w = 0u;    // past the last prefetch: the pieces below turn into zero fills
      const bool fire_rt = BUF || fire;
      const int dst = (stage + D) % STAGES;
      const int nxt = (stage + 1 == STAGES) ? 0 : stage + 1;
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        // issue priority alternates between the two waves of a SIMD every sub-step (left alone the older wave wins
        // every arbitration, finishes early and idles at the barrier while its sibling runs uncovered)
        if (((k ^ wgrp) & 1) == 0) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(0);
        if (k + 1 < KS) read_frags(stage, k + 1, (k + 1) & 1);
#pragma unroll
        for (int qq = 0; qq < TM * TN; ++qq) {
          const int a = qq / TM, b = qq % TM;
          const int q = k * TM * TN + qq;
          if (k + 1 == KS && qq == BAR_AT) {
            // also in the last step (nothing in the loop body is conditional): the barrier is matched by every wave, the
            // fragments read behind it come from a slot nobody writes any more and are never used
            stamp(1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my last fragments of this slot are in registers
            wait_vmcnt<(D - 1) * IPS>();                          // my pieces of the next stage have landed (those of the one after may be in flight)
            stamp(2);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            stamp(3);
            read_frags(nxt, 0, 0);
          }
          mma_step<MT>(wf[k & 1][a], xf[k & 1][b], acc[a][b]);
          if ((q + 1) % MPP1 == 0) {
            const int piece = q / MPP1;
            if (piece < IPS && fire_rt) fire_piece(piece, dst);
          }
          if (q + 1 == NM / 2) {
            if (s + 1 < n_fire) prep_step(s + 1 + D);   // all pieces of step s + D are out: their registers are free
          }
        }
      }
      stage = nxt;
      stamp(4);
    }
    __builtin_amdgcn_s_setprio(0);
  } else {
  for (int s = 0; s < nsteps; ++s) {
    s_cur = s;
    stamp(0);
    // my DMA pieces of step s have landed once at most `newer` younger steps are still outstanding
    const int newer = min(D - 1, nsteps - 1 - s);
    if (D >= 3 && newer >= 2) wait_vmcnt<2 * IPS>();
    else if (D >= 2 && newer >= 1) wait_vmcnt<IPS>();
    else wait_vmcnt<0>();
    stamp(1);
    __builtin_amdgcn_s_barrier();   // everyone's have; and everyone finished reading the slot refilled next
    asm volatile("" ::: "memory");
    stamp(2);
    // exactly ONE instantiation of the MFMA body per kernel: with two (a firing and a non-firing copy)
    // the register allocator parked the accumulators in VGPRs across the loop edge and copied all of
    // them to AGPRs and back every step (128 v_accvgpr moves per 16 MFMAs).
    const bool fire = s < n_fire;
    if (fire) prep_step(s + D);
    else if (BUF) ext_x = ext_w = 0u;        // past the last prefetch: the pieces below turn into zero fills
    const bool fire_rt = BUF || fire;        // pointer form keeps the uniform branch around its pieces
    stamp(3);
    if (EIGHT_WAVES) {
      // 8-wave tile: the DMA pieces are issued between MFMA groups; measured 7 % faster than issuing them up front
      compute_stage(stage, TagTrue{}, fire_rt, (stage + D) % STAGES);
    } else {
      // 4-wave tiles: two workgroups share the CU and cover each other's DMA issue, and the prefetch is
      // only one step deep, so the whole next stage is requested first (interleaving measured 10-20 % slower)
      if (fire_rt) {
#pragma unroll
        for (int q = 0; q < IPS; ++q) fire_piece(q, (stage + D) % STAGES);
      }
      compute_stage(stage, TagFalse{}, false, 0);
    }
    stage = (stage + 1 == STAGES) ? 0 : stage + 1;
    stamp(4);
  }
  }
  if constexpr (BUF) wait_vmcnt<0>();   // the trailing zero-fill pieces must land before the LDS allocation is released
  if constexpr (PROF) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0 && p.prof != nullptr) {
      constexpr int NW = THREADS / 64;
      for (int i = 0; i < 4 * 8; ++i) p.prof[(tid >> 6) * 32 + i] = stamps[((i / 8) * NW + (tid >> 6)) * 8 + (i % 8)];
    }
  }
  if constexpr (LN256 != 0) {
    static_assert(WAVES_M == 4 && WAVES_N == 2 && TM == 2 && TN == 4 && std::is_same<typename storage_of<MT>::type, TOut>::value, "LN256: the 8-wave 256 x 256 tile");
    static_assert(STAGES * STAGE_BYTES >= 128 * 256 * 4, "LN256: the transposition tile must fit the ring");
    if constexpr (!BUF) wait_vmcnt<0>();
    conv_epilogue_lds256<TOut>(p, acc, m_blk, n_blk, wm, wn, lane, tid, smem, z);
    return;
  }
  if constexpr (WAVES_M == 2 && WAVES_N == 2 && TM == 2 && TN == 2 && STAGES * STAGE_BYTES >= 128 * 128 * 4) {
    // full tile, NDHWC, 4-aligned strides (uniform): the coalesced epilogue through the LDS
    if (p.lds_epi && m_blk + BM <= p.M && n_blk + BN <= p.Cout) {
      if constexpr (!BUF) wait_vmcnt<0>();
      conv_epilogue_lds128<TOut>(p, acc, m_blk, n_blk, wm, wn, lane, tid, smem, z);
      return;
    }
  }
  conv_epilogue<TOut, TM, TN, (TM * TN < 8)>(p, acc, m_blk, n_blk, BN, wm, wn, lane, z);
#endif
}

template <typename MT, typename TOut, int WAVES_M, int WAVES_N, int TM, int TN, bool FAST, int LN256 = 0, int STAGES = 2, int ROWB = kRowBytes>
int launch_variant(const ConvArgs& a_in, int nbatch, hipStream_t stream) {
  constexpr int BM = WAVES_M * TM * 32;
  constexpr int BN = WAVES_N * TN * 32;
  constexpr int BK = ROWB / (int)sizeof(MT);
  constexpr int THREADS = 64 * WAVES_M * WAVES_N;
  constexpr int LDS = STAGES * (BM + BN) * ROWB;
  ConvArgs a = a_in;
  a.m_tiles = (a.M + BM - 1) / BM;
  a.n_tiles = (a.Cout + BN - 1) / BN;
  a.nsteps = FAST ? a.ntaps * (a.Cin / BK) : (a.K + BK - 1) / BK;
  if (a.ksplit) {
    VT_CHECK_ARG(FAST, "vt_conv: split-K needs the tap-walk form");
    a.nsteps = (a.ksplit == 1 ? a.KH * a.KW : a.KW) * (a.Cin / BK);
  }
  // Temporal taps: in pixel order (b,t,h,w) the frames t-1, t-2 a tile reads were last touched one whole frame of
  // tiles earlier -- far beyond the 4 MiB L2 of its XCD -- so every kt tap came from the fabric again (the k3
  // temporal conv of the widest level moved 3x its input).  Walking the tiles as (b, hw tile, t) puts the
  // producers of those lines right before their consumer on the same XCD (xcd_remap keeps the sequence
  // contiguous): 298 -> 340 TFLOP/s on that layer, +3 % on the 27-tap up-sampler conv (same-run A/B).
  constexpr int kOctAlign = 16 / (int)sizeof(TOut) > 4 ? 8 : 4;   // elements per 16 bytes, at least a quad
  a.lds_epi = (vt_opt(OPT_CONV_LDSEPI) != 0 && a.out_layout == VT_NDHWC && a.ldy % kOctAlign == 0 &&
               (a.res_mode == VT_RES_NONE || a.ldr % kOctAlign == 0) && (a.ln_mode == 0 || a.ldn % kOctAlign == 0)) ? 1 : 0;
  a.hw_tiles = 0;
  if (conv_tinner() && a.KT > 1 && a.To > 1 && ((long long)a.Ho * a.Wo) % BM == 0) a.hw_tiles = (int)(((long long)a.Ho * a.Wo) / BM);
  // descriptor gather needs the tensors under 4 GiB (minus the out-of-range marker)
  const unsigned long long xb = (unsigned long long)a.B * a.Ti * a.Hi * a.Wi * a.Cin * sizeof(MT);
  const unsigned long long wb = (unsigned long long)a.Cout * a.ldw * sizeof(MT);
  // cache mode (v1.1 chunks after the first): descriptors only where a tile lies in one output frame, so that a time tap reads
  // the cache or x for the whole tile (the kernel switches the descriptor per tap); other shapes gather through pointers
  const unsigned long long cb = a.tmode == VT_TPAD_CACHE ? (unsigned long long)a.B * a.ncache * a.Hi * a.Wi * a.Cin * sizeof(MT) : 0ull;
  const bool cache_ok = a.tmode != VT_TPAD_CACHE ||
                        (FAST && (nbatch == 1 || a.ksplit) && a.prof == nullptr && cb < 0xFFFF0000ull && ((long long)a.Ho * a.Wo) % BM == 0 && a.ups_t == 0);
  const bool buf = conv_buf() && xb < 0xFFFF0000ull && wb < 0xFFFF0000ull && cache_ok &&
                   a.KH <= 8 && a.KW <= 8;   // the per-row padding mask of the FAST form holds 8 bits per axis
  VT_CHECK_ARG(!a.ksplit || buf, "vt_conv: split-K needs the descriptor gather");
  // zero-padded time taps skipped per tile: the tap-walk form on descriptors, a tile inside one output frame
  a.tskip = (FAST && buf && vt_opt(OPT_CONV_TSKIP) != 0 && a.tmode == VT_TPAD_ZERO && a.KT > 1 && a.pt > 0 && a.ups_t == 0 && !a.ksplit &&
             nbatch == 1 && a.prof == nullptr && ((long long)a.Ho * a.Wo) % BM == 0) ? 1 : 0;
  // K-step schedule of the 8-wave tile on descriptors (see the kernel): 16-bit operands 2 (two-group ping-pong), fp32 operands 1
  // (software-pipelined single body), split-bf16 on its 64-byte-row ring 5; everything else -- the 4-wave tiles, the pointer form,
  // the general (non tap-walk) form -- runs the plain loop.  (Earlier rounds kept every schedule selectable: their measurements are in
  // DESIGN section 6, the instantiations are gone.)
  constexpr bool EIGHT = WAVES_M * WAVES_N == 8;
  constexpr int SCHED_BUF = !(EIGHT && FAST) ? 0
                            : (is_split3<MT>::value ? ((ROWB == 64 && STAGES == 4) ? 5 : 0)
                               : ((ROWB == kRowBytes && STAGES == 2) ? (is_h16<MT>::value ? 2 : 1) : 0));
  const void* kern;
  if (buf) {
    a.x_bytes = (unsigned)xb;
    a.w_bytes = (unsigned)wb;
    a.c_bytes = (unsigned)cb;
    kern = reinterpret_cast<const void*>(&conv_igemm_glds_kernel<MT, TOut, WAVES_M, WAVES_N, TM, TN, FAST, ROWB, STAGES, true, LN256, false, SCHED_BUF>);
  } else {
    kern = reinterpret_cast<const void*>(&conv_igemm_glds_kernel<MT, TOut, WAVES_M, WAVES_N, TM, TN, FAST, ROWB, STAGES, false, LN256, false, 0>);
  }
  int lds_bytes = LDS;
  if (a.prof != nullptr) {   // vt_conv_profile: the scheduled 8-wave instantiations of bf16 and split-bf16 (no LayerNorm) carry the stamps
    if constexpr (EIGHT && FAST && LN256 == 0 && (SCHED_BUF == 2 || SCHED_BUF == 5) && !std::is_same<MT, f16_t>::value &&
                  std::is_same<typename storage_of<MT>::type, TOut>::value) {
      VT_CHECK_ARG(buf, "vt_conv_profile: descriptor gather only");
      kern = reinterpret_cast<const void*>(&conv_igemm_glds_kernel<MT, TOut, WAVES_M, WAVES_N, TM, TN, FAST, ROWB, STAGES, true, 0, true, SCHED_BUF>);
      lds_bytes = LDS + 4096;
      VT_CHECK_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    } else {
      VT_CHECK_ARG(false, "vt_conv_profile: only the bf16 / split-bf16 8-wave 256 x 256 tile without fused LayerNorm is instrumented");
    }
  }
  // the attribute is per device: one flag per (instantiation, gather form, device), set race-free
  static std::atomic<bool> attr_done[2][kMaxDevices];
  const int ki = buf ? 1 : 0;
  int dev = 0;
  VT_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices || !attr_done[ki][dev].load(std::memory_order_acquire)) {
    VT_CHECK_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    if (dev >= 0 && dev < kMaxDevices) attr_done[ki][dev].store(true, std::memory_order_release);
  }
  const long long nblk = (long long)a.m_tiles * a.n_tiles;
  VT_CHECK_ARG(nblk < (1ll << 31), "vt_conv: too many tiles (%lld)", nblk);
  void* kargs[] = {&a};
  VT_CHECK_HIP(hipLaunchKernel(kern, dim3((unsigned)nblk, 1, (unsigned)nbatch), dim3(THREADS), kargs, lds_bytes, stream));
  return VT_OK;
}

template <typename MT, typename TOut, int WAVES_M, int WAVES_N, int TM, int TN>
int launch_fast_or_general(const ConvArgs& a, int nbatch, hipStream_t stream) {
  constexpr int BK = kRowBytes / (int)sizeof(MT);
  return (a.Cin % BK) == 0 ? launch_variant<MT, TOut, WAVES_M, WAVES_N, TM, TN, true>(a, nbatch, stream)
                           : launch_variant<MT, TOut, WAVES_M, WAVES_N, TM, TN, false>(a, nbatch, stream);
}


template <typename MT, typename TOut>
int dispatch_tile(const ConvArgs& a, int nbatch, hipStream_t stream) {
  switch (select_tile(a, nbatch)) {
    case TILE_256x32: return launch_fast_or_general<MT, TOut, 4, 1, 2, 1>(a, nbatch, stream);
    case TILE_256x64: return launch_fast_or_general<MT, TOut, 4, 1, 2, 2>(a, nbatch, stream);
    case TILE_256x256:                                                                           // 8 waves
      if constexpr (is_split3<MT>::value) {
        // split-bf16: 64-byte rows (one group of 16 k-values per K step) on a 4-slot ring, schedule 5
        if (a.Cin % 16 == 0) {
          if (a.ln_mode != 0) return launch_variant<MT, TOut, 4, 2, 2, 4, true, 1, 4, 64>(a, nbatch, stream);
          return launch_variant<MT, TOut, 4, 2, 2, 4, true, 0, 4, 64>(a, nbatch, stream);
        }
        return launch_variant<MT, TOut, 4, 2, 2, 4, false>(a, nbatch, stream);
      } else if constexpr (std::is_same<MT, TOut>::value) {
        if (a.ln_mode != 0) return launch_variant<MT, TOut, 4, 2, 2, 4, true, 1>(a, nbatch, stream);   // conv_prepare checked Cin % BK
        // the same instantiation with ln_mode = 0: coalesced stores and residual reads (-8 % on the time up-sampler's
        // parity convolutions, -10 % on the K = 1 024 / 1 536 layers)
        if constexpr (is_h16<TOut>::value) {
          if (lds256_plain_eligible(a, nbatch, true)) return launch_variant<MT, TOut, 4, 2, 2, 4, true, 1>(a, nbatch, stream);
        }
      }
      return launch_fast_or_general<MT, TOut, 4, 2, 2, 4>(a, nbatch, stream);
    default:
      if (deep_ring_eligible(a, nbatch, (int)sizeof(MT))) return launch_variant<MT, TOut, 2, 2, 2, 2, true, 0, 4>(a, nbatch, stream);
      return launch_fast_or_general<MT, TOut, 2, 2, 2, 2>(a, nbatch, stream);
  }
}

}  // namespace
