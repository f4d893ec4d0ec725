// Loader-fed implicit-GEMM convolution for the Cout % 256 == 0 layers in a 16-bit type (round 6): a persistent workgroup of
// FOUR MATRIX waves (one per SIMD, a 128-pixel x 256-channel tile, nothing but fragment reads and MFMAs in their K loop) and
// FOUR LOADER waves (all LDS-DMA requests and all gather arithmetic).  Why: the 8-wave tile of conv_igemm_kernel.h runs its K
// step of 64 in ~3 480 cycles against 2 048 of matrix work (profiles/r04_igemm_step_cycles_sched2.txt) -- every wave alternates
// LOAD and COMPUTE phases, four barriers a step, and a DMA request issued from a wave that also feeds the matrix pipe costs that
// wave ~60 cycles of MFMA issue.  Here a matrix wave never issues a memory request and meets ONE barrier per K step; the loaders'
// request stream (48 pieces of 1 KiB per step) runs beside it on the other wave slot of each SIMD.
// Operator contract, gather, swizzle and fragment layout are conv_igemm_kernel.h's (tap-walk form on buffer descriptors): results
// are the 8-wave tile's bit for bit where the epilogue arithmetic is the same (it is: conv_epilogue_lds256's row phase).
//
// LDS: ring of two stages [X 128 rows x 128 B | W 256 rows x 128 B] = 96 KiB; the epilogue transposes the tile through the first
// 128 KiB (fp32, 128 rows x 256 channels) once the K loop is over.
#include <algorithm>
#include <atomic>
#include <type_traits>

#include "conv_select.h"

namespace {

[[maybe_unused]] constexpr int TR_BM = 128, TR_BN = 256, TR_ROWB = 128, TR_BK = 64;
[[maybe_unused]] constexpr int TR_XB = TR_BM * TR_ROWB;        // 16 384
[[maybe_unused]] constexpr int TR_WB = TR_BN * TR_ROWB;        // 32 768
[[maybe_unused]] constexpr int TR_STAGE = TR_XB + TR_WB;       // 49 152
[[maybe_unused]] constexpr int TR_LDS = 160 * 1024;

template <typename H>
__global__ __launch_bounds__(512, 1) void conv_tr256_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_loader = wave >= 4;                          // uniform per wave

  // ---- the tiles of this workgroup: a contiguous run of the launch's tile sequence (channel tile outermost, pixel tiles in
  // conv_igemm_kernel.h's order: frames innermost for temporal taps), XCD-contiguous like every persistent kernel here
  const int ntile = p.m_tiles * p.n_tiles;
  const int G = gridDim.x;
  const int slot = xcd_remap(blockIdx.x, G);
  const int tq = ntile / G, tr = ntile - tq * G;
  const int t_begin = slot * tq + min(slot, tr);
  const int t_end = t_begin + tq + (slot < tr ? 1 : 0);
  const int khw = p.KH * p.KW;
  const int cpb = p.Cin / TR_BK;
  auto tile_geom = [&](int tile, int& m_blk, int& n_blk, int& kt0) {
    const int nt = tile / p.m_tiles;
    int mt = tile - nt * p.m_tiles;
    if (p.hw_tiles > 0) {
      const int per_b = p.hw_tiles * p.To;
      const int b = mt / per_b;
      const int r = mt - b * per_b;
      const int hwt = r / p.To;
      mt = (b * p.To + (r - hwt * p.To)) * p.hw_tiles + hwt;
    }
    m_blk = mt * TR_BM;
    n_blk = nt * TR_BN;
    kt0 = 0;
    if (p.tskip) {                                           // the tile lies in one output frame: its leading zero-padded time taps are skipped
      const unsigned f = fast_div((unsigned)m_blk, p.fd_hw);
      const unsigned bb = fast_div(f, p.fd_to);
      const int to = (int)f - (int)bb * p.To;
      kt0 = min(max(-(to * p.st - p.pt), 0), p.KT - 1);
    }
  };

  // One tile, as seen by a wave of role LOADER (compile-time): the two roles run separate tile loops over this body, so that the
  // per-lane state of one role is not carried (or spilled) through the other's loops.
  auto run_tile = [&](auto loader_c, int tile) __attribute__((always_inline)) {
    constexpr bool LOADER = decltype(loader_c)::value;
    f32x16 acc[4][2];                                        // matrix waves: channels 32 a + ..., pixels 32 b + ... of the wave's 128 x 64 sub-tile
    int m_blk, n_blk, kt0;
    tile_geom(tile, m_blk, n_blk, kt0);
    const int step0 = kt0 * khw * cpb;
    const int S = p.nsteps - step0;                          // K steps of this tile

    if constexpr (LOADER) {
      // =============================================== loader waves ===============================================
      const int lt = tid - 256;
      const int pos = lt & 7;                                // 16-B slot this lane writes in its rows
      const int srow = lt >> 3;                              // rows srow + 32 i
      const int chunk = pos ^ ((srow >> 1) & 7);             // logical K chunk this lane fetches (the same for every i: 32 i keeps (row >> 1) & 7)
      const int lds_row_off = __builtin_amdgcn_readfirstlane((wave - 4) * 8 * TR_ROWB);
      const H* __restrict__ xg = reinterpret_cast<const H*>(p.x);
      const H* __restrict__ wg = reinterpret_cast<const H*>(p.w);
      const H* __restrict__ cg = reinterpret_cast<const H*>(p.cache);
      constexpr unsigned kOob = 0xFFFF0000u;
      const int Hv = p.Hi << p.ups_s, Wv = p.Wi << p.ups_s, Tv = p.Ti << p.ups_t;
      const bool replicate = p.tmode == VT_TPAD_REPLICATE;
      const unsigned pix_bytes = (unsigned)p.Cin * 2u;
      const unsigned chunk_bytes = (unsigned)chunk * 16u;
      const unsigned HiWi = (unsigned)p.Hi * (unsigned)p.Wi;
      unsigned ext_x = p.x_bytes;
      const unsigned ext_w = p.w_bytes;
      const H* x_cur = xg;
      int a_b[4], a_t0[4], a_hw[4], a_bt[4];
      unsigned a_mask[4], a_tb[4], a_off[4], b_off[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m_blk + srow + 32 * i;                 // < M: the launcher takes full tiles only
        const unsigned r1 = fast_div((unsigned)m, p.fd_wo);
        const int wo = m - (int)r1 * p.Wo;
        const unsigned r2 = fast_div(r1, p.fd_ho);
        const int ho = (int)r1 - (int)r2 * p.Ho;
        const unsigned r3 = fast_div(r2, p.fd_to);
        const int to = (int)r2 - (int)r3 * p.To;
        a_b[i] = (int)r3;
        a_t0[i] = to * p.st - p.pt;
        const int h0 = ho * p.sh - p.ph, w0 = wo * p.sw - p.pw;
        unsigned mk = 0;
        for (int kh = 0; kh < p.KH; ++kh) mk |= ((unsigned)(h0 + kh) < (unsigned)Hv) ? (1u << kh) : 0u;
        for (int kw = 0; kw < p.KW; ++kw) mk |= ((unsigned)(w0 + kw) < (unsigned)Wv) ? (1u << (8 + kw)) : 0u;
        mk |= (unsigned)(h0 & 1) << 17;
        mk |= (unsigned)(w0 & 1) << 18;
        a_mask[i] = mk;
        a_bt[i] = a_b[i] * p.Ti;
        a_hw[i] = (h0 >> p.ups_s) * p.Wi + (w0 >> p.ups_s);
        a_tb[i] = 0;
        a_off[i] = kOob;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) b_off[j] = (unsigned)(n_blk + srow + 32 * j) * (unsigned)p.ldw * 2u + chunk_bytes;
      int q_step = step0, q_cc = 0, q_kt = kt0, q_kh = 0, q_kw = 0;
      bool first = true;
      unsigned s_a = 0, s_b = 0;
      // gather addresses of the next K step in the walk (conv_igemm_kernel.h's tap-walk form: new addresses only when the tap changes)
      auto prep_step = [&]() {
        if (q_cc == 0) {
          if ((q_kh | q_kw) == 0 || first) {                 // new time tap (or the first step of the walk): time part of the offsets
            bool from_cache = false;
            if (p.tmode == VT_TPAD_CACHE) {                  // uniform: the tile lies in one output frame (launcher)
              const int tv_u = __builtin_amdgcn_readfirstlane(a_t0[0]) + q_kt;
              from_cache = tv_u < 0;
              x_cur = from_cache ? cg : xg;
              ext_x = from_cache ? p.c_bytes : p.x_bytes;
            }
            if (from_cache) {
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const unsigned ti = (unsigned)(p.ncache + a_t0[i] + q_kt);
                a_tb[i] = (((unsigned)(a_b[i] * p.ncache) + ti) * HiWi + (unsigned)a_hw[i]) * pix_bytes + chunk_bytes;
                a_mask[i] |= 1u << 16;
              }
            } else {
#pragma unroll
              for (int i = 0; i < 4; ++i) {
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
            for (int i = 0; i < 4; ++i) a_off[i] = ((a_mask[i] & tm) == tm) ? a_tb[i] + delta : kOob;
          } else {
            const unsigned delta = (unsigned)((q_kh >> 1) * p.Wi + (q_kw >> 1)) * pix_bytes;
            const unsigned dh = (q_kh & 1) ? (unsigned)p.Wi * pix_bytes : 0u;
            const unsigned dw = (q_kw & 1) ? pix_bytes : 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const unsigned ph = (unsigned)(((int)(a_mask[i] << 14)) >> 31) & dh;
              const unsigned pw = (unsigned)(((int)(a_mask[i] << 13)) >> 31) & dw;
              a_off[i] = ((a_mask[i] & tm) == tm) ? a_tb[i] + delta + ph + pw : kOob;
            }
          }
        }
        first = false;
        s_a = (unsigned)q_cc * (unsigned)TR_ROWB;
        s_b = (unsigned)q_step * (unsigned)TR_ROWB;
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
      };
      auto fire = [&](int stage) {
        char* Xs = smem + stage * TR_STAGE + lds_row_off;
        const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<H*>(x_cur), 0, ext_x, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<H*>(wg), 0, ext_w, 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(Xs + (32 * i) * TR_ROWB), 16, a_off[i], s_a, 0, 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(Xs + TR_XB + (32 * j) * TR_ROWB), 16, b_off[j], s_b, 0, 0);
      };
      prep_step();
      fire(0);
      if (S > 1) prep_step();                                // the addresses of step 1 are ready before its slot is
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();                          // B(0): step 0 has landed
      for (int s = 0; s < S; ++s) {
        // after B(s): every matrix wave holds the fragments it still needs of slot (s + 1) & 1 = (s - 1) & 1 in registers
        if (s + 1 < S) {
          fire((s + 1) & 1);
          if (s + 2 < S) prep_step();                        // step s + 2's addresses while step s + 1's pieces fly
        }
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                        // B(s + 1)
      }
    } else {
      // =============================================== matrix waves ===============================================
      const int wm = wave & 1, wn = wave >> 1;               // pixels [64 wm, +64), channels [128 wn, +128) of the tile
      const int frag_row = (lane & 31) * TR_ROWB;
      const int swz = ((lane & 31) >> 1) & 7;
      const int khalf = lane >> 5;
      const char* a_base = smem + (wm * 64) * TR_ROWB + frag_row;
      const char* b_base = smem + TR_XB + (wn * 128) * TR_ROWB + frag_row;
      u32x4 wf[2][4], xf[2][2];
      auto read_frags = [&](int stg, int k, int set) __attribute__((always_inline)) {
        const int sl = ((k * 2 + khalf) ^ swz) * 16;
#pragma unroll
        for (int a = 0; a < 4; ++a) wf[set][a] = *reinterpret_cast<const u32x4*>(b_base + stg * TR_STAGE + a * 32 * TR_ROWB + sl);
#pragma unroll
        for (int b = 0; b < 2; ++b) xf[set][b] = *reinterpret_cast<const u32x4*>(a_base + stg * TR_STAGE + b * 32 * TR_ROWB + sl);
      };
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
      __builtin_amdgcn_s_barrier();                          // B(0)
      asm volatile("" ::: "memory");
      read_frags(0, 0, 0);
      __builtin_amdgcn_s_setprio(2);
      for (int s = 0; s < S; ++s) {
        const int stage = s & 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (k + 1 < 4) read_frags(stage, k + 1, (k + 1) & 1);
#pragma unroll
          for (int qq = 0; qq < 8; ++qq) {
            const int a = qq >> 1, b = qq & 1;
            if (k == 3 && qq == 4) {
              // every fragment of this slot is in registers (the slot may be refilled); the next slot has landed once everybody is here
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
              __builtin_amdgcn_s_barrier();                  // B(s + 1)
              asm volatile("" ::: "memory");
              read_frags(stage ^ 1, 0, 0);                   // (after the last step: stale bytes, never used)
            }
            acc[a][b] = h16<H>::mfma32(wf[k & 1][a], xf[k & 1][b], acc[a][b]);
          }
        }
      }
      __builtin_amdgcn_s_setprio(0);
    }

    // =================================== epilogue: the tile through the LDS, rows by all eight waves ===================================
    __syncthreads();                                         // the K loop is over for everybody, no DMA request is outstanding
    float* T = reinterpret_cast<float*>(smem);               // [128 rows][64 chunks of 4 floats], chunk index XOR (row & 63)
    if constexpr (!LOADER) {
      const int wm = wave & 1, wn = wave >> 1, h = lane >> 5;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int prl = wm * 64 + b * 32 + (lane & 31);
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
            for (int e = 0; e < 4; ++e) v[e] = acc[a][b][4 * g + e] + bq[e];
            *reinterpret_cast<f32x4*>(T + prl * 256 + (((c >> 2) ^ (prl & 63)) << 2)) = v;
          }
      }
    }
    __syncthreads();
    {
#pragma clang fp contract(off)
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      H* __restrict__ yg = reinterpret_cast<H*>(p.y);
      const H* __restrict__ rg = reinterpret_cast<const H*>(p.res);
      H* __restrict__ ng = reinterpret_cast<H*>(p.ln_out);
      const bool has_ln = p.ln_mode != 0;
      float alpha = 0.0f;
      if (p.res_mode == VT_RES_MIX) alpha = 1.0f / (1.0f + __expf(-p.mix_factor[0]));
      int tt = tid;
      asm volatile("" : "+v"(tt));                           // opaque: the row geometry is recomputed per tile, not carried through the K loop
      const int j = tt & 31, rsub = tt >> 5;                 // lane j: channels [8 j, +8) of rows rsub + 16 it
      f32x2 lg[4], lb[4];
      if (has_ln) {
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.ln_gamma + 8 * j), g1 = *reinterpret_cast<const f32x4*>(p.ln_gamma + 8 * j + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.ln_beta + 8 * j), b1 = *reinterpret_cast<const f32x4*>(p.ln_beta + 8 * j + 4);
        lg[0] = f32x2{g0[0], g0[1]}; lg[1] = f32x2{g0[2], g0[3]}; lg[2] = f32x2{g1[0], g1[1]}; lg[3] = f32x2{g1[2], g1[3]};
        lb[0] = f32x2{b0[0], b0[1]}; lb[1] = f32x2{b0[2], b0[3]}; lb[2] = f32x2{b1[0], b1[1]}; lb[3] = f32x2{b1[2], b1[3]};
      }
      Oct<H> rq[8];
      if (p.res_mode != VT_RES_NONE) {
#pragma unroll
        for (int it = 0; it < 8; ++it) rq[it].load(rg + (long long)(m_blk + rsub + 16 * it) * p.ldr + n_blk + 8 * j);
      }
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int r = rsub + 16 * it;
        const long long orow = out_row(p, m_blk + r);
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
          Oct<H>::store(yg + orow * p.ldy + n_blk + 8 * j, yv);
        }
        if (!has_ln) continue;
        const f32x2 s = (v[0] + v[1]) + (v[2] + v[3]);
        const float mean = group_sum_dpp<32>(s[0] + s[1]) * (1.0f / 256.0f);
        f32x2 d[4], qq = {0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          d[q] = v[q] - mean;
          qq = __builtin_elementwise_fma(d[q], d[q], qq);
        }
        const float rstd = __builtin_amdgcn_rsqf(__builtin_fmaf(group_sum_dpp<32>(qq[0] + qq[1]), 1.0f / 256.0f, p.ln_eps));
        float o[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x2 u = __builtin_elementwise_fma(d[q] * rstd, lg[q], lb[q]);
          if (p.ln_mode == 2) {
            const f32x2 t = u * -1.4426950408889634f;
            const f32x2 e = f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + 1.0f;
            u = u * f32x2{__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
          }
          o[2 * q] = u[0];
          o[2 * q + 1] = u[1];
        }
        Oct<H>::store(ng + orow * p.ldn + 8 * j, o);
      }
    }
    __syncthreads();                                         // T is read: the next tile's DMA may overwrite it
  };
  if (is_loader) {
#pragma unroll 1
    for (int tile = t_begin; tile < t_end; ++tile) run_tile(std::true_type{}, tile);
  } else {
#pragma unroll 1
    for (int tile = t_begin; tile < t_end; ++tile) run_tile(std::false_type{}, tile);
  }
#endif
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// The same tile with the row phase OVERLAPPED (option conv_tr256 = 2).  In the form above -- as in the 8-wave tile -- a workgroup's K loop
// and its epilogue follow one another: on the short-K layers of the 256-channel level (K = 768: 12 K steps against an epilogue that reads a
// residual, normalises and writes two tensors) the matrix pipe and the VALU idle half of the time each.  Here the rows of tile t are worked
// on while the K loop of tile t + 1 runs:
//   * the matrix waves end a tile by parking it in the LDS as 16-bit rows (R: 128 rows x 512 B, behind the ring) and go straight on;
//   * the four HYBRID waves (the loaders above) keep requesting the operand pieces and, between the requests, walk R in blocks of 8 rows:
//     + bias, + residual / alpha-mix (its rows fetched a step ahead), LayerNorm (+ SiLU) -- and write the finished 16-bit rows back IN
//     PLACE.  They never store to memory: a wave's loads and stores retire through one in-order counter and a store takes thousands of
//     cycles under this write traffic (conv_ws2.hip), so a wave that stores cannot also be the one whose requests a K step waits for;
//   * the matrix waves copy finished blocks from R to memory, one 1-KiB store per wave and block, between their MFMAs (a wave that never
//     waits for a load can leave any number of stores in flight).
// A launch that emits y AND LayerNorm(y) walks R twice: y (stored a step later), then LayerNorm of the stored 16-bit y -- the arithmetic of
// the two-launch form (convolution, then vt_layernorm_act on its rounded result).  Pipeline of block q with P blocks per step: y rows in
// step q / P, their store in the next, LayerNorm in the one after, its store in the fourth; P = the smallest power of two that fits the
// tile's own step count (tiles shortened by skipped time taps walk more blocks per step).  After the last tile a few steps without matrix
// work drain the pipeline.
// ---------------------------------------------------------------------------------------------------------------------------------------
[[maybe_unused]] constexpr int TR_OFF_R = 2 * TR_STAGE;        // 98 304: the row buffer, 128 x 512 B (exactly the rest of the 160 KiB)

__device__ __forceinline__ int tr_blocks_per_step(int S) {     // smallest P in {1, 2, 4, 8, 16} with 16 / P + 3 <= S  (S >= 4)
  int P = 1;
  while (16 / P + 3 > S) P <<= 1;
  return P;
}

template <typename H>
__global__ __launch_bounds__(512, 1) void conv_tr256o_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_hybrid = wave >= 4;
  const int ntile = p.m_tiles * p.n_tiles;
  const int G = gridDim.x;
  const int slot = xcd_remap(blockIdx.x, G);
  const int tq = ntile / G, tr = ntile - tq * G;
  const int t_begin = slot * tq + min(slot, tr);
  const int U = tq + (slot < tr ? 1 : 0);                    // tiles of this workgroup (>= 1: the grid never exceeds the tile count)
  const int khw = p.KH * p.KW;
  const int cpb = p.Cin / TR_BK;
  auto tile_geom = [&](int tile, int& m_blk, int& n_blk, int& kt0) {
    const int nt = tile / p.m_tiles;
    int mt = tile - nt * p.m_tiles;
    if (p.hw_tiles > 0) {
      const int per_b = p.hw_tiles * p.To;
      const int b = mt / per_b;
      const int r = mt - b * per_b;
      const int hwt = r / p.To;
      mt = (b * p.To + (r - hwt * p.To)) * p.hw_tiles + hwt;
    }
    m_blk = mt * TR_BM;
    n_blk = nt * TR_BN;
    kt0 = 0;
    if (p.tskip) {
      const unsigned f = fast_div((unsigned)m_blk, p.fd_hw);
      const unsigned bb = fast_div(f, p.fd_to);
      const int to = (int)f - (int)bb * p.To;
      kt0 = min(max(-(to * p.st - p.pt), 0), p.KT - 1);
    }
  };
  const bool has_ln = p.ln_mode != 0;                        // uniform
  const bool two_phase = has_ln && p.ln_keep_y != 0;         // y and LayerNorm(y): R is walked twice
  constexpr int S_DRAIN = 7;                                 // steps behind the last tile (P = 4)
  H* __restrict__ yg = reinterpret_cast<H*>(p.y);
  H* __restrict__ ng = reinterpret_cast<H*>(p.ln_out);

  if (!is_hybrid) {
    // =================================================== matrix waves ===================================================
    const int wm = wave & 1, wn = wave >> 1;
    const int frag_row = (lane & 31) * TR_ROWB;
    const int swz = ((lane & 31) >> 1) & 7;
    const int khalf = lane >> 5;
    const char* a_base = smem + (wm * 64) * TR_ROWB + frag_row;
    const char* b_base = smem + TR_XB + (wn * 128) * TR_ROWB + frag_row;
    f32x16 acc[4][2];
    u32x4 wf[2][4], xf[2][2];
    auto read_frags = [&](int stg, int k, int set) __attribute__((always_inline)) {
      const int sl = ((k * 2 + khalf) ^ swz) * 16;
#pragma unroll
      for (int a = 0; a < 4; ++a) wf[set][a] = *reinterpret_cast<const u32x4*>(b_base + stg * TR_STAGE + a * 32 * TR_ROWB + sl);
#pragma unroll
      for (int b = 0; b < 2; ++b) xf[set][b] = *reinterpret_cast<const u32x4*>(a_base + stg * TR_STAGE + b * 32 * TR_ROWB + sl);
    };
    // finished block `blk` of the tile at (pm_blk, pn_blk): my two rows of it, R -> memory (kind 0: the first walk's tensor, 1: the LayerNorm's)
    int pm_blk = 0, pn_blk = 0;
    u32x4 pend;                                              // a block's row data between its LDS read and its store (one slot apart)
    auto row_of = [&](int blk) __attribute__((always_inline)) { return 8 * blk + 2 * wave + (lane >> 5); };
    auto op_read = [&](int blk) __attribute__((always_inline)) {
      const int r = row_of(blk);
      pend = *reinterpret_cast<const u32x4*>(smem + TR_OFF_R + r * 512 + (((lane & 31) ^ (r & 31)) << 4));
    };
    auto op_store = [&](int blk, int kind) __attribute__((always_inline)) {
      const int r = row_of(blk);
      const long long orow = out_row(p, pm_blk + r);
      const bool to_n = kind == 1 || (has_ln && !two_phase);
      H* dst = to_n ? ng + orow * p.ldn + 8 * (lane & 31) : yg + orow * p.ldy + pn_blk + 8 * (lane & 31);
      *reinterpret_cast<u32x4*>(dst) = pend;
    };
    // the store operations of step s of an iteration whose rows belong to the previous tile: blocks [P (s - 1), +P) of the first walk,
    // [P (s - 3), +P) of the second
    int n_ops = 0, op_b0 = 0, op_b1 = 0, op_n0 = 0;          // ops [0, op_n0): first-walk blocks op_b0 + o; ops [op_n0, n_ops): second-walk blocks op_b1 + (o - op_n0)
    auto plan_ops = [&](int s, int P, bool rows) __attribute__((always_inline)) {
      const int nst = 16 / P;                                // steps a walk takes
      const bool w0 = rows && s >= 1 && s - 1 < nst;
      const bool w1 = rows && two_phase && s >= 3 && s - 3 < nst;
      op_n0 = w0 ? P : 0;
      n_ops = op_n0 + (w1 ? P : 0);
      op_b0 = P * (s - 1);
      op_b1 = P * (s - 3);
    };
    auto op_blk = [&](int o) __attribute__((always_inline)) { return o < op_n0 ? op_b0 + o : op_b1 + (o - op_n0); };
    auto op_kind = [&](int o) __attribute__((always_inline)) { return o < op_n0 ? 0 : 1; };

    __builtin_amdgcn_s_barrier();                            // B: step 0 of the first tile has landed
    asm volatile("" ::: "memory");
    read_frags(0, 0, 0);
    int v = 0;                                               // global K step: its slot is v & 1
#pragma unroll 1
    for (int i = 0; i <= U; ++i) {
      int m_blk = 0, n_blk = 0, kt0 = 0;
      if (i < U) tile_geom(t_begin + i, m_blk, n_blk, kt0);
      const int S = i < U ? p.nsteps - kt0 * khw * cpb : S_DRAIN;
      const int P = tr_blocks_per_step(S);
      if (i >= 1) {
        // ---- park the finished tile in R as 16-bit rows (bias and everything behind it belong to the row walk): row = pixel, 16-byte chunk
        // c / 8 at slot chunk ^ (row & 31); a lane's quad = 8 bytes at half (lane / 32) of chunk 16 wn + 4 a + g
        const int h = lane >> 5;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int prow = wm * 64 + b * 32 + (lane & 31);
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              u32x2 w2;
              w2[0] = h16<H>::pack(acc[a][b][4 * g], acc[a][b][4 * g + 1]);
              w2[1] = h16<H>::pack(acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]);
              *reinterpret_cast<u32x2*>(smem + TR_OFF_R + prow * 512 + (((16 * wn + 4 * a + g) ^ (prow & 31)) << 4) + 8 * h) = w2;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // D: R holds tile i - 1
        asm volatile("" ::: "memory");
      }
      if (i < U) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
        __builtin_amdgcn_s_setprio(2);
#pragma unroll 1
        for (int s = 0; s < S; ++s, ++v) {
          const int stage = v & 1;
          plan_ops(s, P, i >= 1);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (k + 1 < 4) read_frags(stage, k + 1, (k + 1) & 1);
#pragma unroll
            for (int qq = 0; qq < 8; ++qq) {
              const int a = qq >> 1, b = qq & 1;
              const int mi = k * 8 + qq;                     // MFMA index in the step
              if (k == 3 && qq == 4) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                // B: the next slot has landed, this one may be refilled; the hybrids' rows of this step are in R
                asm volatile("" ::: "memory");
                read_frags(stage ^ 1, 0, 0);
              }
              acc[a][b] = h16<H>::mfma32(wf[k & 1][a], xf[k & 1][b], acc[a][b]);
              // store operations between the MFMAs: op o is read from R behind MFMA 1 + 4 o and stored behind MFMA 5 + 4 o (o < 6);
              // whatever is left (tiles with very few steps) goes out in a loop before the barrier
              if ((mi & 3) == 1 && mi <= 25) {
                const int o = mi >> 2;
                if (o >= 1 && o - 1 < n_ops) op_store(op_blk(o - 1), op_kind(o - 1));
                if (o < 6 && o < n_ops) op_read(op_blk(o));
              }
              if (mi == 26) {
                for (int o = 6; o < n_ops; ++o) {
                  op_read(op_blk(o));
                  op_store(op_blk(o), op_kind(o));
                }
              }
            }
          }
        }
        __builtin_amdgcn_s_setprio(0);
      } else {
        // ---- behind the last tile: the remaining blocks, no matrix work
#pragma unroll 1
        for (int s = 0; s < S; ++s) {
          plan_ops(s, P, true);
          for (int o = 0; o < n_ops; ++o) {
            op_read(op_blk(o));
            op_store(op_blk(o), op_kind(o));
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
        }
      }
      pm_blk = m_blk;
      pn_blk = n_blk;
    }
  } else {
    // =================================================== hybrid waves ===================================================
    const int lt = tid - 256;
    const int hw_ = wave - 4;
    const int pos = lt & 7, srow = lt >> 3;
    const int chunk = pos ^ ((srow >> 1) & 7);
    const int lds_row_off = __builtin_amdgcn_readfirstlane(hw_ * 8 * TR_ROWB);
    const H* __restrict__ xg = reinterpret_cast<const H*>(p.x);
    const H* __restrict__ wg = reinterpret_cast<const H*>(p.w);
    const H* __restrict__ cg = reinterpret_cast<const H*>(p.cache);
    const H* __restrict__ rg = reinterpret_cast<const H*>(p.res);
    constexpr unsigned kOob = 0xFFFF0000u;
    const int Hv = p.Hi << p.ups_s, Wv = p.Wi << p.ups_s, Tv = p.Ti << p.ups_t;
    const bool replicate = p.tmode == VT_TPAD_REPLICATE;
    const unsigned pix_bytes = (unsigned)p.Cin * 2u;
    const unsigned chunk_bytes = (unsigned)chunk * 16u;
    const unsigned HiWi = (unsigned)p.Hi * (unsigned)p.Wi;
    unsigned ext_x = p.x_bytes;
    const unsigned ext_w = p.w_bytes;
    const H* x_cur = xg;
    // ---- the request stream: a walk over (tile, K step) of the whole run, prepared one step ahead of its requests
    int a_b[4], a_t0[4], a_hw[4], a_bt[4];
    unsigned a_mask[4], a_tb[4], a_off[4], b_off[8];
    int w_tile = -1, w_left = 0;                             // tile under the walk (index in my run), its steps still to prepare
    int q_step = 0, q_cc = 0, q_kt = 0, q_kh = 0, q_kw = 0;
    bool first = true, prepared = false;
    unsigned s_a = 0, s_b = 0;
    auto setup_tile = [&](int i) {
      int m_blk, n_blk, kt0;
      tile_geom(t_begin + i, m_blk, n_blk, kt0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = m_blk + srow + 32 * q;
        const unsigned r1 = fast_div((unsigned)m, p.fd_wo);
        const int wo = m - (int)r1 * p.Wo;
        const unsigned r2 = fast_div(r1, p.fd_ho);
        const int ho = (int)r1 - (int)r2 * p.Ho;
        const unsigned r3 = fast_div(r2, p.fd_to);
        const int to = (int)r2 - (int)r3 * p.To;
        a_b[q] = (int)r3;
        a_t0[q] = to * p.st - p.pt;
        const int h0 = ho * p.sh - p.ph, w0 = wo * p.sw - p.pw;
        unsigned mk = 0;
        for (int kh = 0; kh < p.KH; ++kh) mk |= ((unsigned)(h0 + kh) < (unsigned)Hv) ? (1u << kh) : 0u;
        for (int kw = 0; kw < p.KW; ++kw) mk |= ((unsigned)(w0 + kw) < (unsigned)Wv) ? (1u << (8 + kw)) : 0u;
        mk |= (unsigned)(h0 & 1) << 17;
        mk |= (unsigned)(w0 & 1) << 18;
        a_mask[q] = mk;
        a_bt[q] = a_b[q] * p.Ti;
        a_hw[q] = (h0 >> p.ups_s) * p.Wi + (w0 >> p.ups_s);
        a_tb[q] = 0;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) b_off[j] = (unsigned)(n_blk + srow + 32 * j) * (unsigned)p.ldw * 2u + chunk_bytes;
      q_step = kt0 * khw * cpb;
      q_cc = 0; q_kt = kt0; q_kh = 0; q_kw = 0;
      first = true;
      w_left = p.nsteps - q_step;
    };
    auto advance_prep = [&]() {
      if (w_left == 0) {
        if (w_tile + 1 >= U) {
          prepared = false;
          return;
        }
        ++w_tile;
        setup_tile(w_tile);
      }
      --w_left;
      prepared = true;
      if (q_cc == 0) {
        if ((q_kh | q_kw) == 0 || first) {
          bool from_cache = false;
          if (p.tmode == VT_TPAD_CACHE) {
            const int tv_u = __builtin_amdgcn_readfirstlane(a_t0[0]) + q_kt;
            from_cache = tv_u < 0;
            x_cur = from_cache ? cg : xg;
            ext_x = from_cache ? p.c_bytes : p.x_bytes;
          }
          if (from_cache) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const unsigned ti = (unsigned)(p.ncache + a_t0[q] + q_kt);
              a_tb[q] = (((unsigned)(a_b[q] * p.ncache) + ti) * HiWi + (unsigned)a_hw[q]) * pix_bytes + chunk_bytes;
              a_mask[q] |= 1u << 16;
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int tv = a_t0[q] + q_kt;
              const bool ok = (tv < Tv) & ((tv >= 0) | replicate);
              const unsigned ti = (unsigned)(max(tv, 0) >> p.ups_t);
              a_tb[q] = (((unsigned)a_bt[q] + ti) * HiWi + (unsigned)a_hw[q]) * pix_bytes + chunk_bytes;
              a_mask[q] = (a_mask[q] & ~(1u << 16)) | (ok ? (1u << 16) : 0u);
            }
          }
        }
        const unsigned tm = (1u << q_kh) | (1u << (8 + q_kw)) | (1u << 16);
        if (p.ups_s == 0) {
          const unsigned delta = (unsigned)(q_kh * p.Wi + q_kw) * pix_bytes;
#pragma unroll
          for (int q = 0; q < 4; ++q) a_off[q] = ((a_mask[q] & tm) == tm) ? a_tb[q] + delta : kOob;
        } else {
          const unsigned delta = (unsigned)((q_kh >> 1) * p.Wi + (q_kw >> 1)) * pix_bytes;
          const unsigned dh = (q_kh & 1) ? (unsigned)p.Wi * pix_bytes : 0u;
          const unsigned dw = (q_kw & 1) ? pix_bytes : 0u;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const unsigned ph = (unsigned)(((int)(a_mask[q] << 14)) >> 31) & dh;
            const unsigned pw = (unsigned)(((int)(a_mask[q] << 13)) >> 31) & dw;
            a_off[q] = ((a_mask[q] & tm) == tm) ? a_tb[q] + delta + ph + pw : kOob;
          }
        }
      }
      first = false;
      s_a = (unsigned)q_cc * (unsigned)TR_ROWB;
      s_b = (unsigned)q_step * (unsigned)TR_ROWB;
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
    };
    auto fire = [&](int stage) {
      char* Xs = smem + stage * TR_STAGE + lds_row_off;
      const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<H*>(x_cur), 0, ext_x, 0x00020000);
      const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<H*>(wg), 0, ext_w, 0x00020000);
#pragma unroll
      for (int q = 0; q < 4; ++q) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(Xs + (32 * q) * TR_ROWB), 16, a_off[q], s_a, 0, 0);
#pragma unroll
      for (int j = 0; j < 8; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(Xs + TR_XB + (32 * j) * TR_ROWB), 16, b_off[j], s_b, 0, 0);
    };

    // ---- the row walk: this wave's two rows of a block (8 rows: 2 per hybrid wave), lane j = channels [8 j, +8) of the row
    const int j = lane & 31;
    float alpha = 0.0f;
    if (p.res_mode == VT_RES_MIX) alpha = 1.0f / (1.0f + __expf(-p.mix_factor[0]));
    float lgm[8], lbt[8], bia[8];
    if (has_ln) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        lgm[e] = p.ln_gamma[8 * j + e];
        lbt[e] = p.ln_beta[8 * j + e];
      }
    }
    u32x4 rq[16];                                            // residual rows of the NEXT step's first-walk blocks
    int rm_blk = 0, rn_blk = 0;                              // the tile R holds
    auto row_addr = [&](int blk) __attribute__((always_inline)) {
      const int r = 8 * blk + 2 * hw_ + (lane >> 5);
      return smem + TR_OFF_R + r * 512 + ((j ^ (r & 31)) << 4);
    };
    auto load_res = [&](int u, int blk, int tm_blk, int tn_blk) __attribute__((always_inline)) {
      const int r = 8 * blk + 2 * hw_ + (lane >> 5);
      rq[u] = *reinterpret_cast<const u32x4*>(rg + (long long)(tm_blk + r) * p.ldr + tn_blk + 8 * j);
    };
    auto layer_norm = [&](float (&vv)[8]) __attribute__((always_inline)) {      // in place: LayerNorm over the row's 256 channels (32 lanes x 8) (+ SiLU)
      float sm = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) sm += vv[e];
      const float mean = group_sum_dpp<32>(sm) * (1.0f / 256.0f);
      float qs = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        vv[e] -= mean;
        qs = __builtin_fmaf(vv[e], vv[e], qs);
      }
      const float rstd = __builtin_amdgcn_rsqf(__builtin_fmaf(group_sum_dpp<32>(qs), 1.0f / 256.0f, p.ln_eps));
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float u_ = __builtin_fmaf(vv[e] * rstd, lgm[e], lbt[e]);
        if (p.ln_mode == 2) {
          const float ex = __builtin_amdgcn_exp2f(u_ * -1.4426950408889634f);
          u_ = u_ * __builtin_amdgcn_rcpf(ex + 1.0f);
        }
        vv[e] = u_;
      }
    };
    // (nxt >= 0: the block whose residual row takes this one's place in rq[u] -- requested as soon as the old row is consumed, a step ahead of its use)
    auto walk0 = [&](int u, int blk, int nxt) __attribute__((always_inline)) {   // first walk: + bias, + residual; y rows (or, single walk with LayerNorm, its rows)
      char* ad = row_addr(blk);
      const u32x4 w4 = *reinterpret_cast<const u32x4*>(ad);
      float vv[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        vv[2 * e] = h16<H>::lo(w4[e]) + bia[2 * e];
        vv[2 * e + 1] = h16<H>::hi(w4[e]) + bia[2 * e + 1];
      }
      if (p.res_mode == VT_RES_ADD) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          vv[2 * e] = h16<H>::lo(rq[u][e]) + vv[2 * e];
          vv[2 * e + 1] = h16<H>::hi(rq[u][e]) + vv[2 * e + 1];
        }
      } else if (p.res_mode == VT_RES_MIX) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          vv[2 * e] = h16<H>::lo(rq[u][e]) * alpha + vv[2 * e] * (1.0f - alpha);
          vv[2 * e + 1] = h16<H>::hi(rq[u][e]) * alpha + vv[2 * e + 1] * (1.0f - alpha);
        }
      }
      if (p.res_mode != VT_RES_NONE && nxt >= 0) load_res(u, nxt, rm_blk, rn_blk);
      if (has_ln && !two_phase) layer_norm(vv);
      u32x4 o4;
#pragma unroll
      for (int e = 0; e < 4; ++e) o4[e] = h16<H>::pack(vv[2 * e], vv[2 * e + 1]);
      *reinterpret_cast<u32x4*>(ad) = o4;
    };
    auto walk1 = [&](int blk) __attribute__((always_inline)) {                   // second walk: LayerNorm (+ SiLU) of the stored y rows
      char* ad = row_addr(blk);
      const u32x4 w4 = *reinterpret_cast<const u32x4*>(ad);
      float vv[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        vv[2 * e] = h16<H>::lo(w4[e]);
        vv[2 * e + 1] = h16<H>::hi(w4[e]);
      }
      layer_norm(vv);
      u32x4 o4;
#pragma unroll
      for (int e = 0; e < 4; ++e) o4[e] = h16<H>::pack(vv[2 * e], vv[2 * e + 1]);
      *reinterpret_cast<u32x4*>(ad) = o4;
    };

    // ---- prologue: step 0 of the first tile, the addresses of step 1
    advance_prep();
    fire(0);
    advance_prep();
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                            // B
    int v = 0;
#pragma unroll 1
    for (int i = 0; i <= U; ++i) {
      int m_blk = 0, n_blk = 0, kt0 = 0;
      if (i < U) tile_geom(t_begin + i, m_blk, n_blk, kt0);
      const int S = i < U ? p.nsteps - kt0 * khw * cpb : S_DRAIN;
      const int P = tr_blocks_per_step(S);
      const int nst = 16 / P;
      if (i >= 1) {                                          // (bias of the tile R holds: its channel block may differ from the last one's)
#pragma unroll
        for (int e = 0; e < 8; ++e) bia[e] = p.bias ? p.bias[rn_blk + 8 * j + e] : 0.0f;
      }
#pragma unroll 1
      for (int s = 0; s < S; ++s) {
        // ---- requests of the next K step (its slot was read for the last time before the barrier just passed)
        if (i < U) {
          if (prepared) fire((v + 1) & 1);
          ++v;
        }
        // ---- residual rows of the first-walk blocks of the NEXT step: blocks [P (s + 1), +P) of the tile R holds -- or, at a tile's last
        // step, blocks [0, P') of the tile being computed now (its walk starts with the next iteration's step 0, P' = that iteration's P)
        // (within a tile's walk the next step's rows are requested by walk0 itself, each into the register its own row just left)
        if (p.res_mode != VT_RES_NONE) {
          if (s + 1 == S && i < U) {
            int nm, nn, nk;
            int Sn = S_DRAIN;
            if (i + 1 < U) {
              tile_geom(t_begin + i + 1, nm, nn, nk);
              Sn = p.nsteps - nk * khw * cpb;
            }
            const int Pn = tr_blocks_per_step(Sn);
#pragma unroll
            for (int u = 0; u < 16; ++u)
              if (u < Pn) load_res(u, u, m_blk, n_blk);
          }
        }
        if (i < U) advance_prep();                           // the addresses of the step after next, while the requests fly
        if (s == 0 && i >= 1) {
          __builtin_amdgcn_s_barrier();                      // D: R holds tile i - 1
          asm volatile("" ::: "memory");
        }
        // ---- this step's share of the row walks
        if (i >= 1) {
          if (s < nst) {                                     // (the residual rows were requested a step ago and have landed: that step's closing wait covered them)
            const bool more = s + 1 < nst;
#pragma unroll
            for (int u = 0; u < 16; ++u)
              if (u < P) walk0(u, P * s + u, more ? P * (s + 1) + u : -1);
          }
          if (two_phase && s >= 2 && s - 2 < nst) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
              if (u < P) walk1(P * (s - 2) + u);
          }
        }
        // my rows are in R; my pieces of the next step and the residual rows have landed (the builtin, not asm: the compiler's own wait
        // bookkeeping then knows, and does not wait again -- for the NEXT step's pieces -- in front of the first use of a residual row)
        __builtin_amdgcn_s_waitcnt(0x0070);                  // vmcnt(0) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();                        // B
        asm volatile("" ::: "memory");
      }
      rm_blk = m_blk;
      rn_blk = n_blk;
    }
  }
#endif
}

}  // namespace

// vt_conv hands over launches that qualify (tr256_eligible, conv_select.h); `args` = its ConvArgs (validated), dtype = VT_BF16 / VT_F16
extern "C" __attribute__((visibility("hidden"))) int vt_conv_tr256_launch(const void* args, int dtype, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ConvArgs a = *reinterpret_cast<const ConvArgs*>(args);
  a.m_tiles = a.M / TR_BM;
  a.n_tiles = a.Cout / TR_BN;
  a.nsteps = a.ntaps * (a.Cin / TR_BK);
  a.hw_tiles = 0;
  if (conv_tinner() && a.KT > 1 && a.To > 1 && ((long long)a.Ho * a.Wo) % TR_BM == 0) a.hw_tiles = (int)(((long long)a.Ho * a.Wo) / TR_BM);
  a.x_bytes = (unsigned)((unsigned long long)a.B * a.Ti * a.Hi * a.Wi * a.Cin * 2);
  a.w_bytes = (unsigned)((unsigned long long)a.Cout * a.ldw * 2);
  a.c_bytes = a.tmode == VT_TPAD_CACHE ? (unsigned)((unsigned long long)a.B * a.ncache * a.Hi * a.Wi * a.Cin * 2) : 0u;
  a.tskip = (vt_opt(OPT_CONV_TSKIP) != 0 && a.tmode == VT_TPAD_ZERO && a.KT > 1 && a.pt > 0 && a.ups_t == 0 && ((long long)a.Ho * a.Wo) % TR_BM == 0) ? 1 : 0;
  // the overlapped form needs at least 4 K steps in every tile (its shortest row pipeline); a tile's steps: all, less the skipped time-tap planes
  const int s_min = a.nsteps - (a.tskip ? std::min(a.pt, a.KT - 1) * a.KH * a.KW * (a.Cin / TR_BK) : 0);
  const bool overlapped = vt_opt(OPT_CONV_TR256) >= 2 && s_min >= 4;
  const void* kern = overlapped ? (dtype == VT_F16 ? reinterpret_cast<const void*>(&conv_tr256o_kernel<f16_t>) : reinterpret_cast<const void*>(&conv_tr256o_kernel<bf16_t>))
                                : (dtype == VT_F16 ? reinterpret_cast<const void*>(&conv_tr256_kernel<f16_t>) : reinterpret_cast<const void*>(&conv_tr256_kernel<bf16_t>));
  const int ki = (dtype == VT_F16 ? 1 : 0) + (overlapped ? 2 : 0);
  static std::atomic<int> cus[4][kMaxDevices];
  int dev = 0;
  VT_CHECK_HIP(hipGetDevice(&dev));
  const bool dev_ok = dev >= 0 && dev < kMaxDevices;
  int ncu = dev_ok ? cus[ki][dev].load(std::memory_order_acquire) : 0;
  if (ncu == 0) {
    VT_CHECK_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, TR_LDS));
    VT_CHECK_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    if (ncu <= 0) ncu = 256;
    if (dev_ok) cus[ki][dev].store(ncu, std::memory_order_release);
  }
  const long long ntile = (long long)a.m_tiles * a.n_tiles;
  const int grid = ntile < ncu ? (int)ntile : ncu;
  void* kargs[] = {&a};
  VT_CHECK_HIP(hipLaunchKernel(kern, dim3((unsigned)grid), dim3(512), kargs, TR_LDS, stream));
  return VT_OK;
}
