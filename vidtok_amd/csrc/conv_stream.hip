// Streaming variant of the implicit-GEMM convolution (see conv_igemm.hip for the operator and the tile
// mechanics) for the layers whose tiles are SHORT: the widest pyramid level (Cout = 128, K = 384 .. 2304).
//
// Why: with one workgroup per 128 x 128 tile the per-tile fixed cost -- workgroup launch, coordinate set-up,
// the first DMA's memory latency, the residual read and the output write of the epilogue -- was measured at
// ~13 us against ~1 us per K step (same-run experiment with stores / residual / K loop switched off one by
// one: 3x3 128->128 conv 2.52 ms = 1.42 ms K loop + 1.10 ms fixed; k3 temporal conv 1.53 ms, 70 % of it
// fixed), and none of it overlaps with anything: the co-resident workgroup runs at its own pace.
//
// Here a workgroup is persistent (2 per CU) and walks a sequence of tiles:
//   * the DMA pipeline runs across tile boundaries: during the LAST K step of tile i the coordinates of
//     tile i+1 are set up and its first stage is requested, so the memory latency hides under MFMAs;
//   * the epilogue is DEFERRED: the accumulators of tile i (started at the bias) are parked in a second
//     register set and drained in slices of SPS accumulator quads (a quad = 4 channels of one pixel per lane)
//     during the first K steps of tile i+1: add the residual, convert, store;
//   * vmcnt bookkeeping: within a step the program order is  DMA pieces -> (MFMAs) -> stores of this step's
//     slice -> residual loads of the NEXT step's slice, and VM operations retire in order, so "my DMA pieces
//     have landed" is vmcnt(#younger operations) -- stores and residual loads stay in flight across the barrier.
// STATUS: experimental, off by default (VT_CONV_STREAM=1).  Results are identical to conv_igemm's (same op tests),
// the bare persistent K loop is 1.3-1.7x faster on the widest level, but the drain's stores delay the next DMA
// wait (in-order retirement, one step of slack with a 2-stage ring) and the set-up is serial: net +3 %
// (DESIGN.md section 6).
// Restrictions (the launcher falls back to conv_igemm otherwise): descriptor gather, Cin a multiple of the
// K step, Cout a multiple of 128, NDHWC vector epilogue, plain residual (no time shift), one problem per launch.
#include <type_traits>

#include "conv_common.h"

namespace {

template <typename MT, typename TOut, int SPS>   // SPS = accumulator quads drained per K step (16 / SPS steps drain a tile)
__global__ __launch_bounds__(256, 2) void conv_stream_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int ROWB = 128, NS = 8, RSTEP = 32, ROWS_PER_WAVE = 8;
  constexpr int VEC = 16 / (int)sizeof(MT);
  constexpr int BK = ROWB / (int)sizeof(MT);
  constexpr int KS = ROWB / 32;
  constexpr int BM = 128, BN = 128, TM = 2, TN = 2;
  constexpr int A_VECS = 4, B_VECS = 4;
  constexpr int A_BYTES = BM * ROWB;
  constexpr int STAGE_BYTES = (BM + BN) * ROWB;
  constexpr int NQ = 16;                      // accumulator quads per lane: (a, b, g) = 2 x 2 x 4
  constexpr int NGRP = NQ / SPS;              // K steps that carry a drain slice
  static_assert(NQ % SPS == 0, "SPS divides 16");
  (void)VEC;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ __attribute__((aligned(16))) float s_bias[1024];   // bias vector (launcher: Cout <= 1024)
  typedef __attribute__((address_space(3))) void* lds_ptr_t;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave & 1;
  const int wn = wave >> 1;

  // tile sequence of this workgroup: XCD x owns a contiguous chunk of the tile order (as xcd_remap), its
  // workgroups sweep the chunk side by side (stride = workgroups per XCD), so neighbours in the order -- which
  // share halo rows, temporal taps and the weight slab -- are in flight together in one L2
  const int nblk = p.m_tiles * p.n_tiles;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int cq = nblk >> 3, cr = nblk & 7;
  const int c_start = (xcd < cr) ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
  const int c_len = cq + (xcd < cr ? 1 : 0);
  if (loc >= c_len) return;

  const MT* __restrict__ xg = reinterpret_cast<const MT*>(p.x);
  const MT* __restrict__ wg = reinterpret_cast<const MT*>(p.w);
  constexpr unsigned kOob = 0xFFFF0000u;
  unsigned ext_x = p.x_bytes, ext_w = p.w_bytes;

  const int pos = tid % NS;
  const int srow = tid / NS;
  const int chunk = pos ^ ((srow >> 1) & (NS - 1));
  const unsigned chunk_bytes = (unsigned)chunk * 16u;
  const int lds_row_off = __builtin_amdgcn_readfirstlane(wave * ROWS_PER_WAVE * ROWB);

  const int Hv = p.Hi << p.ups_s, Wv = p.Wi << p.ups_s, Tv = p.Ti << p.ups_t;
  const bool replicate = p.tmode == VT_TPAD_REPLICATE;
  const unsigned pix_bytes = (unsigned)p.Cin * (unsigned)sizeof(MT);
  const unsigned HiWi = (unsigned)p.Hi * (unsigned)p.Wi;
  const int cpb = p.Cin / BK;

  // ---- gather state of the tile whose K walk is being prepared (conv_igemm.hip: FAST descriptor form) ----
  unsigned a_mask[A_VECS], a_tb[A_VECS], a_off[A_VECS], b_off[B_VECS];
  int a_bt[A_VECS], a_hw[A_VECS], a_t0[A_VECS];
  int q_step = 0, q_cc = 0, q_kt = 0, q_kh = 0, q_kw = 0;
  unsigned s_a = 0, s_b = 0;
  int m_blk = 0, n_blk = 0;

  auto tile_origin = [&](int seq, int& mb, int& nb) {
    const int tile = c_start + seq;
    const int nt = tile / p.m_tiles;
    int mt = tile - nt * p.m_tiles;
    if (p.hw_tiles > 0) {
      const int per_b = p.hw_tiles * p.To;
      const int b = mt / per_b;
      const int r = mt - b * per_b;
      const int hwt = r / p.To;
      mt = (b * p.To + (r - hwt * p.To)) * p.hw_tiles + hwt;
    }
    mb = mt * BM;
    nb = nt * BN;
  };
  auto setup_tile = [&](int mb, int nb) {
#pragma unroll
    for (int i = 0; i < A_VECS; ++i) {
      const int m = mb + srow + RSTEP * i;
      unsigned mk = 0;
      int t0 = 0, bt = 0, hw = 0;
      if (m < p.M) {
        const int wo = m % p.Wo;
        int r = m / p.Wo;
        const int ho = r % p.Ho;
        r /= p.Ho;
        const int to = r % p.To;
        const int h0 = ho * p.sh - p.ph, w0 = wo * p.sw - p.pw;
        t0 = to * p.st - p.pt;
        bt = (r / p.To) * p.Ti;
        for (int kh = 0; kh < p.KH; ++kh) mk |= ((unsigned)(h0 + kh) < (unsigned)Hv) ? (1u << kh) : 0u;
        for (int kw = 0; kw < p.KW; ++kw) mk |= ((unsigned)(w0 + kw) < (unsigned)Wv) ? (1u << (8 + kw)) : 0u;
        mk |= (unsigned)(h0 & 1) << 17;
        mk |= (unsigned)(w0 & 1) << 18;
        hw = (h0 >> p.ups_s) * p.Wi + (w0 >> p.ups_s);
      }
      a_mask[i] = mk; a_t0[i] = t0; a_bt[i] = bt; a_hw[i] = hw; a_tb[i] = 0;
    }
#pragma unroll
    for (int j = 0; j < B_VECS; ++j)
      b_off[j] = (unsigned)(nb + srow + RSTEP * j) * (unsigned)p.ldw * (unsigned)sizeof(MT) + chunk_bytes;   // Cout % 128 == 0: always a row
    q_step = q_cc = q_kt = q_kh = q_kw = 0;
  };
  auto prep_step = [&]() {
    if (q_cc == 0) {
      if ((q_kh | q_kw) == 0) {
#pragma unroll
        for (int i = 0; i < A_VECS; ++i) {
          const int tv = a_t0[i] + q_kt;
          const bool ok = (tv < Tv) & ((tv >= 0) | replicate);
          const unsigned ti = (unsigned)(max(tv, 0) >> p.ups_t);
          a_tb[i] = (((unsigned)a_bt[i] + ti) * HiWi + (unsigned)a_hw[i]) * pix_bytes + chunk_bytes;
          a_mask[i] = (a_mask[i] & ~(1u << 16)) | (ok ? (1u << 16) : 0u);
        }
      }
      const unsigned tm = (1u << q_kh) | (1u << (8 + q_kw)) | (1u << 16);
      if (p.ups_s == 0) {
        const unsigned delta = (unsigned)(q_kh * p.Wi + q_kw) * pix_bytes;
#pragma unroll
        for (int i = 0; i < A_VECS; ++i) a_off[i] = ((a_mask[i] & tm) == tm) ? a_tb[i] + delta : kOob;
      } else {
        const unsigned delta = (unsigned)((q_kh >> 1) * p.Wi + (q_kw >> 1)) * pix_bytes;
        const unsigned dh = (q_kh & 1) ? (unsigned)p.Wi * pix_bytes : 0u;
        const unsigned dw = (q_kw & 1) ? pix_bytes : 0u;
#pragma unroll
        for (int i = 0; i < A_VECS; ++i) {
          const unsigned ph = (unsigned)(((int)(a_mask[i] << 14)) >> 31) & dh;
          const unsigned pw = (unsigned)(((int)(a_mask[i] << 13)) >> 31) & dw;
          a_off[i] = ((a_mask[i] & tm) == tm) ? a_tb[i] + delta + ph + pw : kOob;
        }
      }
    }
    s_a = (unsigned)q_cc * (unsigned)ROWB;
    s_b = (unsigned)q_step * (unsigned)ROWB;
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
  auto fire_stage = [&](int stage) {
    char* As = smem + stage * STAGE_BYTES + lds_row_off;
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<MT*>(xg), 0, ext_x, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<MT*>(wg), 0, ext_w, 0x00020000);
#pragma unroll
    for (int q = 0; q < A_VECS; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(As + (RSTEP * q) * ROWB), 16, a_off[q], s_a, 0, 0);
#pragma unroll
    for (int j = 0; j < B_VECS; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(As + A_BYTES + (RSTEP * j) * ROWB), 16, b_off[j], s_b, 0, 0);
  };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  const int frag_row = (lane & 31) * ROWB;
  const int swz = ((lane & 31) >> 1) & (NS - 1);
  const int khalf = lane >> 5;
  // Fragments are double-buffered one k16 group (4 reads, 4 MFMAs) deep: the reads of group k+1 are issued, then
  // the MFMAs of group k run and cover their LDS latency.  The order is pinned with sched_barrier: left alone, the
  // scheduler (short of registers next to the parked tile) read each fragment right before its use and exposed
  // the LDS latency on every group (the bare K loop ran 15-20 % below the tile-per-workgroup kernel's); reading
  // the whole stage up front, as that kernel does, needs 64 fragment registers and costs the second wave per SIMD.
  auto compute_stage = [&](int stage) {
    const char* As = smem + stage * STAGE_BYTES + (wm * TM * 32) * ROWB + frag_row;
    const char* Bs = smem + stage * STAGE_BYTES + A_BYTES + (wn * TN * 32) * ROWB + frag_row;
    u32x4 wf[2][TN], xf[2][TM];
    auto read_group = [&](int k, int buf) {
      const int slot = ((k * 2 + khalf) ^ swz) * 16;
#pragma unroll
      for (int a = 0; a < TN; ++a) wf[buf][a] = *reinterpret_cast<const u32x4*>(Bs + a * 32 * ROWB + slot);
#pragma unroll
      for (int b = 0; b < TM; ++b) xf[buf][b] = *reinterpret_cast<const u32x4*>(As + b * 32 * ROWB + slot);
    };
    read_group(0, 0);
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      if (k + 1 < KS) read_group(k + 1, (k + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) mma_step<MT>(wf[k & 1][a], xf[k & 1][b], acc[a][b]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- deferred epilogue of the parked tile ----
  // The bias is folded into the accumulator start values (read from an LDS copy of the bias vector when a tile
  // begins), so a quad's epilogue is  residual load -> add -> convert -> store.  Placement of the residual loads
  // matters: VM operations retire in order, so a load that misses to HBM and sits OLDER than a step's DMA pieces
  // hides their completion behind ~2 us of memory latency -- issued at the top of the step, the drain cost 10 us
  // per tile that way.  They are issued at the END of the step before the one that consumes them: younger than
  // that step's DMA pieces (the counted wait at the next barrier skips them), a full K step of latency cover.
  // The parked accumulators are read with a uniform dynamic register index (s_set_gpr_idx), 4 moves per quad.
  typedef __attribute__((ext_vector_type(32))) float f32x32;
  f32x32 pacc_lo, pacc_hi;              // parked tile: [a=0 | a=1], element b*16 + 4g + e
#pragma unroll
  for (int r = 0; r < 32; ++r) pacc_lo[r] = pacc_hi[r] = 0.0f;
  constexpr unsigned ESZ = (unsigned)sizeof(TOut);
  const bool has_res = p.res_mode != VT_RES_NONE;
  const unsigned ystep = 32u * (unsigned)p.ldy * ESZ, rstep = 32u * (unsigned)p.ldr * ESZ;
  // byte offsets of this lane's first pixel row (+ first channel of its quads); the second row (b = 1) is 32
  // pixels further, a uniform displacement.  n*: the tile being computed, plain: the parked tile
  unsigned yrow = 0, rrow = 0, nyrow = 0, nrrow = 0;
  auto tile_rows = [&](int mb, int nb) {
    const unsigned pn = (unsigned)(nb + wn * TN * 32 + 4 * (lane >> 5));
    const unsigned m = (unsigned)(mb + wm * TM * 32 + (lane & 31));   // M % 128 == 0 (launcher): every row exists
    nyrow = (m * (unsigned)p.ldy + pn) * ESZ;
    nrrow = (m * (unsigned)p.ldr + pn) * ESZ;
  };
  auto init_acc = [&](int nb) {         // accumulators start at the bias of their channel
    const float* bl = s_bias + nb + wn * TN * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(bl + 32 * a + 8 * g);
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[a][b][4 * g + e] = t[e];
      }
  };
  auto park_tile = [&]() {
    yrow = nyrow;
    rrow = nrrow;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      pacc_lo[r] = acc[0][0][r]; pacc_lo[16 + r] = acc[0][1][r];
      pacc_hi[r] = acc[1][0][r]; pacc_hi[16 + r] = acc[1][1][r];
    }
  };
  // Quad order of the drain: a step's SPS quads must complete whole memory segments -- the 4 quads g = 0..3 of one
  // (a, b) are 64 contiguous bytes (bf16) of a pixel row, with a = 0, 1 a full 128-B line.  Drained one quad per
  // step, every line was written in 8 partial pieces microseconds apart.
  //   SPS <= 4: q = a*8 + b*4 + g      SPS = 8: q = b*8 + a*4 + g (one step = full lines of 32 pixel rows)
  auto QA = [](int q) { return SPS == 8 ? (q >> 2) & 1 : q >> 3; };
  auto QB = [](int q) { return SPS == 8 ? q >> 3 : (q >> 2) & 1; };
  Quad<TOut> rq[SPS];
#pragma unroll
  for (int u = 0; u < SPS; ++u) rq[u].v = decltype(rq[u].v){};
  auto res_load = [&](int grp, unsigned row) {      // grp uniform, 0 .. NGRP-1
    const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<TOut*>(reinterpret_cast<const TOut*>(p.res)), 0, p.r_bytes, 0x00020000);
#pragma unroll
    for (int u = 0; u < SPS; ++u) {
      const int q = grp * SPS + u;
      const int qa = QA(q), qb = QB(q), qg = q & 3;
      const unsigned rs = (unsigned)(32 * qa + 8 * qg) * ESZ + (unsigned)qb * rstep;
      if constexpr (sizeof(TOut) == 4) rq[u].v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, row, rs, 0));
      else rq[u].v = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc_r, row, rs, 0));
    }
  };
  auto drain_store = [&](int grp) {
    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<TOut*>(p.y), 0, p.y_bytes, 0x00020000);
#pragma unroll
    for (int u = 0; u < SPS; ++u) {
      const int q = grp * SPS + u;
      const int qa = QA(q), qb = QB(q), qg = q & 3;
      const unsigned ys = (unsigned)(32 * qa + 8 * qg) * ESZ + (unsigned)qb * ystep;
      const int idx = __builtin_amdgcn_readfirstlane(qb * 16 + qg * 4);
      const bool hi = qa != 0;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float l = pacc_lo[idx + e], h = pacc_hi[idx + e];
        v[e] = hi ? h : l;
        if (has_res) v[e] = rq[u].get(e) + v[e];
      }
      if constexpr (sizeof(TOut) == 4) {
        f32x4 t; t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, t), rsrc_y, yrow, ys, 0);
      } else {
        u32x2 t;
        t[0] = f32_to_bf16_bits(v[0]) | (f32_to_bf16_bits(v[1]) << 16);
        t[1] = f32_to_bf16_bits(v[2]) | (f32_to_bf16_bits(v[3]) << 16);
        __builtin_amdgcn_raw_buffer_store_b64(t, rsrc_y, yrow, ys, 0);
      }
    }
  };

  // ---- the tile stream ----
  for (int i = tid; i < p.Cout; i += 256) s_bias[i] = p.bias ? p.bias[i] : 0.0f;
  __syncthreads();
  int seq = loc;
  tile_origin(seq, m_blk, n_blk);
  setup_tile(m_blk, n_blk);
  init_acc(n_blk);
  prep_step();
  fire_stage(0);
  int stage = 0;
  bool have_prev = false;
  int allowed = 0;                       // VM operations younger than the DMA pieces of the step just issued
  const int nsteps = p.nsteps;
  // one K step; LAST selects what is prepared for the stage after it: the next step of this tile, or (peeled
  // last step) the first step of the next tile -- its set-up writes two dozen loop-carried registers, and as a
  // branch inside the step loop it cost ~40 register copies on every step
  auto k_step = [&](int s, auto last_tag, bool more, int nxt) {
    constexpr bool LAST = decltype(last_tag)::value;
    if (allowed == 2 * SPS) wait_vmcnt<2 * SPS>();   // uniform: everything but those younger operations has retired,
    else if (allowed == SPS) wait_vmcnt<SPS>();      // i.e. my DMA pieces have landed
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr (!LAST) {
      prep_step();
    } else {
      if (more) {                            // the next tile's first stage rides under these MFMAs
        tile_origin(nxt, m_blk, n_blk);
        setup_tile(m_blk, n_blk);
        prep_step();
      } else {
        ext_x = ext_w = 0u;                   // nothing follows: zero fills (no memory traffic)
      }
    }
    asm volatile("" ::: "memory");
    fire_stage(stage ^ 1);
    asm volatile("" ::: "memory");          // everything below stays younger than the DMA pieces (counted wait above)
    compute_stage(stage);
    asm volatile("" ::: "memory");
    const bool drain = have_prev && s < NGRP;   // uniform
    if (drain) drain_store(s);
    allowed = drain ? SPS : 0;
    if (has_res) {
      if constexpr (LAST) {                  // next step = step 0 of the next tile, which drains THIS tile
        res_load(0, nrrow);
        allowed += SPS;
      } else if (have_prev && s + 1 < NGRP) {
        res_load(s + 1, rrow);
        allowed += SPS;
      }
    }
    stage ^= 1;
  };
  for (;;) {
    tile_rows(m_blk, n_blk);
    const int nxt = seq + per;
    const bool more = nxt < c_len;
    for (int s = 0; s < nsteps - 1; ++s) k_step(s, TagFalse{}, false, 0);
    k_step(nsteps - 1, TagTrue{}, more, nxt);
    // nsteps >= NGRP is guaranteed by the launcher, so the previously parked tile is fully drained here
    park_tile();
    init_acc(n_blk);
    have_prev = true;
    if (!more) break;
    seq = nxt;
  }
  wait_vmcnt<0>();   // trailing zero fills must land before the LDS allocation is released
  // last tile: drain everything (group 0's residual was requested in the last step)
  for (int g0 = 0; g0 < NGRP; ++g0) {
    if (has_res && g0 > 0) res_load(g0, rrow);
    drain_store(g0);
  }
#endif
}

template <typename MT, typename TOut, int SPS>
int launch_stream(const ConvArgs& a, hipStream_t stream) {
  constexpr int LDS = 2 * (128 + 128) * 128;
  const void* kern = reinterpret_cast<const void*>(&conv_stream_kernel<MT, TOut, SPS>);
  static bool attr_done = false;
  if (!attr_done) {
    VT_CHECK_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_done = true;
  }
  ConvArgs args = a;
  void* kargs[] = {&args};
  const int grid = env_int("VT_CONV_STREAM_GRID", 512);   // 2 workgroups per CU (LDS- and register-limited), 64 per XCD
  VT_CHECK_HIP(hipLaunchKernel(kern, dim3((unsigned)grid), dim3(256), kargs, LDS, stream));
  return VT_OK;
}

template <typename MT, typename TOut>
int dispatch_stream(const ConvArgs& a, hipStream_t stream) {
  const int sps = env_int("VT_STREAM_SPS", 0);
  if (sps == 8) return launch_stream<MT, TOut, 8>(a, stream);
  if (sps == 4) return launch_stream<MT, TOut, 4>(a, stream);
  if (sps == 1 && a.nsteps >= 16) return launch_stream<MT, TOut, 1>(a, stream);
  if (a.nsteps >= 16) return launch_stream<MT, TOut, 1>(a, stream);
  return launch_stream<MT, TOut, 4>(a, stream);   // launcher guarantees nsteps >= 4
}

}  // namespace

// a: fully populated ConvArgs for the 128 x 128 tile (m_tiles, n_tiles, nsteps, hw_tiles, x_bytes, w_bytes);
// dtype_code 0 = fp32 -> fp32, 1 = bf16 -> fp32, 2 = bf16 -> bf16
int vt_conv_stream_launch(const void* args, int dtype_code, hipStream_t stream) {
  const ConvArgs& a = *reinterpret_cast<const ConvArgs*>(args);
  if (dtype_code == 0) return dispatch_stream<float, float>(a, stream);
  if (dtype_code == 1) return dispatch_stream<bf16_t, float>(a, stream);
  return dispatch_stream<bf16_t, bf16_t>(a, stream);
}
