// Fused temporal residual block of the widest level (C = 128, bf16):
//     y = x + conv2( SiLU(LN2( conv1( SiLU(LN1(x)) ) )) ),   conv = causal Conv1d over T, k = 3
// = ResnetCausalBlock1D._forward of the reference (model_3dcausal.py:473-499 with CausalConv1d :144-159), optionally
// followed by the LayerNorm(+SiLU) the NEXT block starts with (emitted like vt_conv's `ln_mode`).
//
// Unfused, the two K = 384 convolutions of such a block are HBM-shaped (95 FLOP/B): per pixel they move the normalised
// input, the intermediate twice, the residual, the result and the next norm -- ~2 KiB -- in two launches of 1.45 ms.
// Here a block reads x once and writes y (+ the next norm) once: 512-768 B per pixel, one launch.
//
// Structure (weight-stationary like conv_ws128.hip): one workgroup per CU; the weights of BOTH convolutions stay in
// registers for the lifetime of the kernel.  A workgroup owns pixel columns -- 64 consecutive pixels of one clip -- and
// walks each column through time:
//   step t:  x[t] rows (prefetched into registers) -> LN1+SiLU -> ring1[t % 3]                         (LDS, bf16)
//            GEMM1: taps = ring1 slots of frames t-2, t-1, t                 -> T1 (bf16, transposed)  (LDS)
//            rows of T1 + b1 -> LN2+SiLU -> ring2[t % 3]
//            GEMM2: taps = ring2 slots                                        -> T2 (fp32)
//            rows of T2 + b2 + x[t] -> y[t], LayerNorm_next -> n[t]                                     (HBM)
// Causal padding: frames before the clip are zeros (v1.0: their taps are skipped) or the first frame repeated (v1.1
// first chunk / un-tiled).  Chunk-to-chunk caches (v1.1 tiling) are not handled here: the host keeps those blocks on
// the unfused path.  (The first generation of this kernel -- four waves, one per SIMD, every phase on the same wave:
// 14 960 cycles per step -- is in the history at commit da80542.)
#include <atomic>
#include <type_traits>

#include "conv_common.h"

namespace {

[[maybe_unused]] constexpr int TB_PIX = 64;                      // pixels per column step
[[maybe_unused]] constexpr int TB_ROWP = 272;                    // ring row: 256 B of channels + 16 B pad (bank spread)
[[maybe_unused]] constexpr int TB_SLOT = TB_PIX * TB_ROWP;       // 17 408
[[maybe_unused]] constexpr int TB_RING = 3 * TB_SLOT;            // 52 224
[[maybe_unused]] constexpr int TB_T = TB_PIX * 128 * 4;          // 32 768

struct TBlockArgs {
  const bf16_t* x;
  bf16_t* y;
  bf16_t* n_out;
  const bf16_t* w1;
  const bf16_t* w2;
  const float* b1;
  const float* b2;
  const float* g1; const float* be1;
  const float* g2; const float* be2;
  const float* gn; const float* ben;
  int B, T, HW;
  int replicate;      // 0: zero frames before the clip, 1: first frame repeated
  int keep_y;         // write y (0 only with ln_next != 0: the consumer needs just the normalised tensor)
  int ln_next;        // 0 none, 1 LayerNorm, 2 LayerNorm + SiLU
  float eps;
  unsigned long long* prof;   // PROF instantiation only (vt_temporal_block_profile): cycle stamps of workgroup 0
  int prof_mode;              // PROF only: 1 = row jobs skipped (wrong results; times the bare GEMMs), VT_TBLOCK_PROF_MODE
};

template <int I, int N, typename F>
__device__ __forceinline__ void tb_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    tb_static_for<I + 1, N>(f);
  }
}

// LayerNorm (+SiLU) of one pixel row held by 16 lanes x 8 channels; two-pass statistics like layernorm_act_kernel.
// Every row phase exists twice -- sliced into MFMA shadows, and plain (first / last step of a workgroup, steps with
// skipped taps) -- so nothing here may depend on FMA contraction (off for this file; fused multiply-adds are spelled
// out): a pixel's bits must not depend on which version computed it, i.e. on how columns were split over workgroups.
//
// The block is bound by its VALU work, not by the MFMAs (three LayerNorm+SiLU per element against 2 x 24 MFMAs per 64
// pixels: per step and wave ~1200 VALU + 200 transcendental instructions = ~8 000 issue cycles against 3 072 MFMA
// cycles, counted on the ISA), so the row arithmetic runs on channel PAIRS: v_pk_add / v_pk_mul / v_pk_fma_f32 carry two
// elements per instruction, one v_cvt_pk_bf16_f32 packs a pair; only exp and rcp stay per element.
#pragma clang fp contract(off)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 tb_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 tb_unpack2(uint32_t w) {            // bf16 pair (low half = even channel) -> fp32 pair
  f32x2 r;
  r[0] = __uint_as_float(w << 16);
  r[1] = __uint_as_float(w & 0xffff0000u);
  return r;
}
__device__ __forceinline__ uint32_t tb_pack2(f32x2 v) {              // round-to-nearest-even, one v_cvt_pk_bf16_f32
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, tb_bf16x2));
}
__device__ __forceinline__ f32x2 tb_silu2(f32x2 u) {                 // u * sigmoid(u), the arithmetic of silu_fast
  const f32x2 t = u * -1.4426950408889634f;
  f32x2 e;
  e[0] = __builtin_amdgcn_exp2f(t[0]);
  e[1] = __builtin_amdgcn_exp2f(t[1]);
  const f32x2 d = e + 1.0f;
  f32x2 r;
  r[0] = __builtin_amdgcn_rcpf(d[0]);
  r[1] = __builtin_amdgcn_rcpf(d[1]);
  return u * r;
}
__device__ __forceinline__ f32x2 tb_affine_act2(f32x2 d, float rstd, f32x2 g, f32x2 b, bool silu) {
  const f32x2 u = __builtin_elementwise_fma(d * rstd, g, b);
  return silu ? tb_silu2(u) : u;
}
template <bool SILU>
__device__ __forceinline__ void tb_row_norm2(f32x2 (&v)[4], const f32x2 (&g)[4], const f32x2 (&b)[4], float eps, f32x2 (&o)[4]) {
  f32x2 s = v[0];
#pragma unroll
  for (int q = 1; q < 4; ++q) s = s + v[q];
  const float mean = group_sum_dpp<16>(s[0] + s[1]) * (1.0f / 128.0f);
  f32x2 d[4], qq = {0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    d[q] = v[q] - mean;
    qq = __builtin_elementwise_fma(d[q], d[q], qq);
  }
  const float rstd = __builtin_amdgcn_rsqf(__builtin_fmaf(group_sum_dpp<16>(qq[0] + qq[1]), 1.0f / 128.0f, eps));
#pragma unroll
  for (int q = 0; q < 4; ++q) o[q] = tb_affine_act2(d[q], rstd, g[q], b[q], SILU);
}

// "This value exists HERE" (see conv_ws128.hip): keeps a slice of row arithmetic in the MFMA shadow the source put it in
__device__ __forceinline__ void tb_pin(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void tb_pin2(f32x2& v) { asm volatile("" : "+v"(v)); }

// ---------------------------------------------------------------------------------------------------------------------
// The block on EIGHT waves with two roles.  What the cycle stamps of the one-role kernel showed
// (profiles/r02_tblock_phase_cycles.txt): a 64-pixel step costs 14 960 cycles of which 3 072 are
// MFMA; the three LayerNorm+SiLU per element are ~6 300 issue cycles of VALU work that a single wave per SIMD cannot put
// behind its own MFMAs (an MFMA hides at most ~6 plain VALU instructions, and no packed-fp32 one).  So the SIMD gets a
// second wave: waves 0-3 ("matrix waves", one per SIMD) keep the stationary weights and do nothing but the two GEMMs and
// the accumulator transposes; waves 4-7 ("row waves") do every row phase.  The dependency chain of a step,
//     L1(t) -> G1(t) -> L2(t) -> G2(t) -> O(t)        (L = LayerNorm+SiLU rows, G = GEMM, O = + x, y / next-norm stores)
// is software-pipelined over the virtual steps k of a workgroup, two phases per step, one barrier each:
//     phase A_k:  matrix: T2 <- acc2 (conv2 of step k-2), G1(k) -> acc1        rows: L2(k-1): T1 -> ring2
//     phase B_k:  matrix: T1 <- acc1 (conv1 of step k), G2(k-1) -> acc2        rows: O(k-2): T2 (+ x) -> y, n;  L1(k+1) -> ring1
// Every buffer is written in one phase and read in the next one (or later): ring slots by virtual step mod 3 (slot k+1 of
// ring1 is rewritten in B_k, last read by G1(k) in A_k; slot k-1 of ring2 in A_k, last read by G2(k-2) in B_{k-1}); T1
// holds conv1 as bf16 (272-B rows; b1 is added by the row waves), T2 conv2 in fp32.
// LDS: 2 x 52 224 + 17 408 + 32 768 = 154 624 B.  Registers: 256 per wave (two waves per SIMD): 192 weights + 2 x 32
// accumulators would not fit, so G1 and G2 share ONE accumulator set (each is parked in its T buffer right after the
// barrier that ends its phase).
// ---------------------------------------------------------------------------------------------------------------------
[[maybe_unused]] constexpr int T3_T1P = 272;                              // bytes per T1 row: 128 bf16 + 16 pad (16-B aligned rows)
[[maybe_unused]] constexpr int T3_OFF_T1 = 2 * TB_RING;
[[maybe_unused]] constexpr int T3_OFF_T2 = 2 * TB_RING + TB_PIX * T3_T1P;
[[maybe_unused]] constexpr int T3_LDS = T3_OFF_T2 + TB_T;                 // 154 624
#ifndef T3_WA
#define T3_WA 32     // weight fragments (of 48) kept in the accumulator half; the rest + the accumulators in the architectural half
#endif
// Row arithmetic of the row waves: plain (unpacked) fp32.  A packed-fp32 instruction does not execute next to an MFMA of
// the OTHER wave of the SIMD either: with `v_pk_*` rows the row waves' LayerNorm passes stretched by the length of the
// matrix waves' GEMMs (L2 4 430 cycles next to G1, 2 020 with the matrix pipe idle; profiles/r02_tblock_v3_phase_cycles.txt).
// The pins keep the SLP vectoriser from re-pairing the elements.
template <bool SILU>
__device__ __forceinline__ void t3_row_norm(float (&v)[8], const float (&g)[8], const float (&b)[8], float eps, float (&o)[8]) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) s = s + v[e];
  const float mean = group_sum_dpp<16>(s) * (1.0f / 128.0f);
  float q = 0.f, d[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    d[e] = v[e] - mean;
    tb_pin(d[e]);
    q = __builtin_fmaf(d[e], d[e], q);
  }
  const float rstd = __builtin_amdgcn_rsqf(__builtin_fmaf(group_sum_dpp<16>(q), 1.0f / 128.0f, eps));
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float u = __builtin_fmaf(d[e] * rstd, g[e], b[e]);
    tb_pin(u);
    if constexpr (SILU) {
      const float ex = __builtin_amdgcn_exp2f(u * -1.4426950408889634f);
      u = u * __builtin_amdgcn_rcpf(ex + 1.0f);
      tb_pin(u);
    }
    o[e] = u;
  }
}
__device__ __forceinline__ void t3_unpack8(const u32x4& w, float (&v)[8]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    v[2 * q] = __uint_as_float(w[q] << 16);
    v[2 * q + 1] = __uint_as_float(w[q] & 0xffff0000u);
  }
}
__device__ __forceinline__ u32x4 t3_pack8(const float (&o)[8]) {
  u32x4 w;
#pragma unroll
  for (int q = 0; q < 4; ++q) w[q] = tb_pack2(f32x2{o[2 * q], o[2 * q + 1]});
  return w;
}

template <bool W_IN_AGPR>
__device__ __forceinline__ void t3_mfma(const u32x4& w, const u32x4& x, f32x16& acc) {
  if constexpr (W_IN_AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(x));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(x));
}

template <int LNN, bool KEEP, bool PROF = false>
__global__ __launch_bounds__(512, 1) void tblock_split_kernel(const TBlockArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ptiles = p.HW / TB_PIX;
  const int ncols = p.B * ptiles;
  const int G = gridDim.x;
  const int slot_id = xcd_remap(blockIdx.x, G);
  const int cq = ncols / G, cr = ncols - cq * G;
  const int c_begin = slot_id * cq + min(slot_id, cr);
  const int c_end = c_begin + cq + (slot_id < cr ? 1 : 0);
  if (c_begin >= c_end) return;
  const int n = (c_end - c_begin) * p.T;                             // virtual steps of this workgroup
  char* ring1 = smem;
  char* ring2 = smem + TB_RING;
  char* T1 = smem + T3_OFF_T1;
  float* T2 = reinterpret_cast<float*>(smem + T3_OFF_T2);
  auto coords = [&](int k, int& col, int& t) {
    const int q = k / p.T;
    col = c_begin + q;
    t = k - q * p.T;
  };
  auto col_base = [&](int col) -> long long {                        // element offset of (b, frame 0, first pixel of the tile)
    const int b = col / ptiles;
    const int pt = col - b * ptiles;
    return ((long long)b * p.T * p.HW + (long long)pt * TB_PIX) * 128;
  };
  const long long frame_stride = (long long)p.HW * 128;
  // PROF: stamps of workgroup 0, steps [8, 12): [wave][step][8] in the LDS behind T2, copied out at the end
  unsigned long long* stamps = reinterpret_cast<unsigned long long*>(smem + T3_LDS);
  int kcur = 0;
  auto stamp = [&](int i) {
    if constexpr (PROF) {
      if (blockIdx.x == 0 && kcur >= 8 && kcur < 12) {
        const unsigned long long ts = __builtin_amdgcn_s_memtime();
        if (lane == 0) stamps[((kcur - 8) * 8 + wave) * 8 + i] = ts;
      }
    }
  };
  auto dump = [&]() {
    if constexpr (PROF) {
      if (blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < 32; ++i) p.prof[wave * 32 + i] = stamps[((i / 8) * 8 + wave) * 8 + (i % 8)];
    }
  };

  if (wave < 4) {
    // =================================================== matrix waves ===================================================
    u32x4 wreg[48];
    {
      const long long roff = (long long)(wave * 32 + (lane & 31)) * 384 + (lane >> 5) * 8;
#pragma unroll
      for (int g = 0; g < 24; ++g) wreg[g] = *reinterpret_cast<const u32x4*>(p.w1 + roff + g * 16);
#pragma unroll
      for (int g = 0; g < 24; ++g) wreg[24 + g] = *reinterpret_cast<const u32x4*>(p.w2 + roff + g * 16);
    }
    const int h = lane >> 5;
    const int frag_off = (lane & 31) * TB_ROWP + h * 16;             // B-fragment of pixel lane%32, k half lane/32
    f32x16 acc[2];
    // GEMM of virtual step k (frame t of its column) over the live taps [KT0, 3) of `ring`
    auto gemm = [&](auto kt0_c, auto wbase_c, const char* ring, int k, int t) {
      constexpr int KT0 = decltype(kt0_c)::value, WB = decltype(wbase_c)::value;
      constexpr int G0 = KT0 * 8;
      const char* sp[3];
#pragma unroll
      for (int kt = 0; kt < 3; ++kt) {
        const int vs = (t - 2 + kt >= 0) ? k - 2 + kt : k - t;         // replicate: frames before the clip = its frame 0
        sp[kt] = ring + (vs % 3) * TB_SLOT + frag_off;
      }
      auto faddr = [&](int g, int j) -> const u32x4* {
        return reinterpret_cast<const u32x4*>(sp[g >> 3] + j * (32 * TB_ROWP) + (g & 7) * 32);
      };
      // fragments in MFMA order m = 2 g + j, a ring of four: the one used three MFMAs from now is requested behind each MFMA
      // (two whole groups ahead, as in the 4-wave kernel, would take 24 registers; this path has 16 to spare)
      constexpr int M0 = 2 * G0;
      u32x4 xf[4];
#pragma unroll
      for (int m = M0; m < M0 + 3; ++m) xf[m % 4] = *faddr(m >> 1, m & 1);
      tb_static_for<M0, 48>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        t3_mfma<(WB + (m >> 1) < T3_WA)>(wreg[WB + (m >> 1)], xf[m % 4], acc[m & 1]);
        if constexpr (m + 3 < 48) xf[(m + 3) % 4] = *faddr((m + 3) >> 1, (m + 3) & 1);
        __builtin_amdgcn_sched_barrier(0);
      });
      asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");             // last MFMA -> first VALU reader of its accumulator
    };
    auto run_gemm = [&](auto wbase_c, const char* ring, int k) {
      int col, t;
      coords(k, col, t);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
      const int first = p.replicate ? 0 : max(0, 2 - t);             // uniform
      if (first == 0) gemm(std::integral_constant<int, 0>{}, wbase_c, ring, k, t);
      else if (first == 1) gemm(std::integral_constant<int, 1>{}, wbase_c, ring, k, t);
      else gemm(std::integral_constant<int, 2>{}, wbase_c, ring, k, t);
    };
    // accumulators: lane = pixel 32 j + lane % 32, channels 32 wave + 8 g + 4 h + e
    auto acc_to_T1 = [&]() {                                         // rounded to bf16 (8 B per quad); b1 is added by the row waves
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int prow = 32 * j + (lane & 31);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = wave * 32 + 8 * g + 4 * h;
          u32x2 w;
          w[0] = tb_pack2(f32x2{acc[j][4 * g], acc[j][4 * g + 1]});
          w[1] = tb_pack2(f32x2{acc[j][4 * g + 2], acc[j][4 * g + 3]});
          *reinterpret_cast<u32x2*>(T1 + prow * T3_T1P + c * 2) = w;
        }
      }
    };
    auto acc_to_T2 = [&]() {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int prow = 32 * j + (lane & 31);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = wave * 32 + 8 * g + 4 * h;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[j][4 * g + e];
          *reinterpret_cast<f32x4*>(T2 + prow * 128 + (((c >> 2) ^ (prow & 31)) << 2)) = v;
        }
      }
    };
    __syncthreads();                                                 // L1(0) is in ring1
    for (int k = 0; k < n + 2; ++k) {
      kcur = k;
      stamp(0);
      // ---- phase A_k: G1(k); its result stays in the accumulators across the barrier
      if (k < n && !(PROF && p.prof_mode == 1)) run_gemm(std::integral_constant<int, 0>{}, ring1, k);
      stamp(1);
      __syncthreads();
      stamp(2);
      // ---- phase B_k: park conv1(k) in T1, G2(k-1), park conv2(k-1) in T2 (read by the row waves in B_{k+1})
      if (k < n) acc_to_T1();
      stamp(3);
      if (k >= 1 && k - 1 < n && !(PROF && p.prof_mode == 1)) {
        run_gemm(std::integral_constant<int, 24>{}, ring2, k - 1);
      }
      stamp(4);
      __syncthreads();
      stamp(5);
      // (start of A_{k+1}) conv2(k-1) -> T2: nobody reads T2 in an A phase
      if (k >= 1 && k - 1 < n) acc_to_T2();
      stamp(6);
    }
    dump();
  } else {
    // ===================================================== row waves =====================================================
    // The row waves issue first: left at equal priority the older matrix wave of a SIMD wins every arbitration -- also
    // while its next MFMA cannot issue yet -- and the row wave next to it stands still for the length of a GEMM (measured:
    // L2 next to G1 = L2 alone + G1).  The matrix wave needs one issue slot per 32 cycles; it gets them in the row wave's
    // dependency stalls.
    __builtin_amdgcn_s_setprio(3);
    const int vt = tid - 256;
    const int oct_j = vt & 15, row0 = vt >> 4;                        // rows row0 + 16 it (it < 4), channels [8 oct_j, +8)
    float lg1[8], lb1[8], lg2[8], lb2[8], lgn[8], lbn[8], bo1[8], bo2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = 8 * oct_j + e;
      bo1[e] = p.b1 ? p.b1[c] : 0.0f;
      lg1[e] = p.g1[c]; lb1[e] = p.be1[c];
      lg2[e] = p.g2[c]; lb2[e] = p.be2[c];
      lgn[e] = LNN ? p.gn[c] : 1.0f;
      lbn[e] = LNN ? p.ben[c] : 0.0f;
      bo2[e] = p.b2 ? p.b2[c] : 0.0f;
    }
    const int row_lds = oct_j * 16;
    auto ring_store = [&](char* ring, int vs, int row, const float (&o)[8]) {
      *reinterpret_cast<u32x4*>(ring + (vs % 3) * TB_SLOT + row * TB_ROWP + row_lds) = t3_pack8(o);
    };
    auto load_rows = [&](Oct<bf16_t> (&dst)[4], int k) {            // x rows of virtual step k
      int col, t;
      coords(k, col, t);
      const bf16_t* src = p.x + col_base(col) + (long long)t * frame_stride + 8 * oct_j;
#pragma unroll
      for (int it = 0; it < 4; ++it) dst[it].load(src + (long long)(row0 + 16 * it) * 128);
    };
    auto ln1_rows = [&](Oct<bf16_t> (&xr)[4], int k) {              // L1(k): LayerNorm1 + SiLU of x rows -> ring1
      // L1 runs behind O, when the matrix waves have finished G2 and wait at the barrier: nothing for packed fp32 to
      // collide with, and a packed row costs 2 080 cycles against 2 700
      f32x2 g2v[4], b2v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        g2v[q] = f32x2{lg1[2 * q], lg1[2 * q + 1]};
        b2v[q] = f32x2{lb1[2 * q], lb1[2 * q + 1]};
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        f32x2 v[4], o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = tb_unpack2(xr[it].w[q]);
        tb_row_norm2<true>(v, g2v, b2v, p.eps, o);
        u32x4 w;
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = tb_pack2(o[q]);
        *reinterpret_cast<u32x4*>(ring1 + (k % 3) * TB_SLOT + (row0 + 16 * it) * TB_ROWP + row_lds) = w;
      }
    };
    // x rows: of step k+1 for L1 and -- read a second time, from the L2 -- of step k-2 for the residual of O.  They are
    // requested one whole step ahead, at the top of phase B and BEFORE that phase's y / n stores, into the register set
    // the phase does not use (two sets, the loop body exists twice): a load issued behind stores would make its first use
    // (and any reuse of the stores' operand registers) wait for those stores, and under this kernel's write traffic a
    // store takes ~5 000 cycles to retire -- with the requests at the top of phase A, right behind the stores of
    // phase B, L2 took 5 000 cycles instead of 2 100 (profiles/r02_tblock_v3_phase_cycles.txt).
    struct XSet { Oct<bf16_t> xn[4], xr[4]; };
    XSet sa, sb;
    load_rows(sa.xn, 0);
    ln1_rows(sa.xn, 0);
    if (1 < n) load_rows(sa.xn, 1);                                  // x(1) for L1(1) in B_0
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      sa.xr[it].w = sa.xn[it].w;
      sb.xn[it].w = sa.xn[it].w;
      sb.xr[it].w = sa.xn[it].w;
    }
    __syncthreads();                                                 // L1(0) is in ring1
    // GUARD = false: a step in the middle of the walk -- every load, store and phase is live, the body is straight-line
    // code (with conditional loads / stores in it hipcc cannot count what is in flight and waits for everything)
    auto body = [&](auto guard_c, int k, XSet& cur, XSet& nxt) {
      constexpr bool GUARD = decltype(guard_c)::value;
      kcur = k;
      stamp(0);
      // ---- phase A_k: L2(k-1): rows of T1 (conv1, bf16) + b1 -> LayerNorm2 + SiLU -> ring2
      if (!GUARD || (k >= 1 && k - 1 < n)) {
        u32x4 tw[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) tw[it] = *reinterpret_cast<const u32x4*>(T1 + (row0 + 16 * it) * T3_T1P + row_lds);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          float v[8], o[8];
          t3_unpack8(tw[it], v);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = v[e] + bo1[e];
          t3_row_norm<true>(v, lg2, lb2, p.eps, o);
          ring_store(ring2, k - 1, row0 + 16 * it, o);
        }
      }
      stamp(1);
      __syncthreads();
      stamp(2);
      // ---- phase B_k: requests for B_{k+1}; O(k-2): rows of T2 + b2 + x(k-2) -> y, LayerNorm_next -> n;  L1(k+1) -> ring1
      if (!GUARD || k + 2 < n) load_rows(nxt.xn, k + 2);
      if (!GUARD || (k >= 1 && k - 1 < n)) load_rows(nxt.xr, k - 1);
      if (!GUARD || (k >= 2 && k - 2 < n)) {
        int col, t;
        coords(k - 2, col, t);
        const long long ob = col_base(col) + (long long)t * frame_stride + 8 * oct_j;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          f32x4 tq[2][2];
#pragma unroll
          for (int i2 = 0; i2 < 2; ++i2) {
            const int row = row0 + 16 * (2 * half + i2);
            const int sw = row & 31;
            tq[i2][0] = *reinterpret_cast<const f32x4*>(T2 + row * 128 + (((2 * oct_j) ^ sw) << 2));
            tq[i2][1] = *reinterpret_cast<const f32x4*>(T2 + row * 128 + (((2 * oct_j + 1) ^ sw) << 2));
          }
#pragma unroll
          for (int i2 = 0; i2 < 2; ++i2) {
            const int it = 2 * half + i2;
            const int row = row0 + 16 * it;
            const f32x4 t0 = tq[i2][0], t1 = tq[i2][1];
            float v[8], xv[8];
            t3_unpack8(cur.xr[it].w, xv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              v[e] = xv[e] + ((e < 4 ? t0[e] : t1[e - 4]) + bo2[e]);
              tb_pin(v[e]);
            }
            if constexpr (KEEP) *reinterpret_cast<u32x4*>(p.y + ob + (long long)row * 128) = t3_pack8(v);
            if constexpr (LNN != 0) {
              float o[8];
              t3_row_norm<(LNN == 2)>(v, lgn, lbn, p.eps, o);
              *reinterpret_cast<u32x4*>(p.n_out + ob + (long long)row * 128) = t3_pack8(o);
            }
          }
        }
      }
      stamp(3);
      if (!GUARD || k + 1 < n) ln1_rows(cur.xn, k + 1);
      stamp(4);
      __syncthreads();
      stamp(5);
      stamp(6);
    };
    // steps k in [2, n - 2) need no guards: k-2 >= 0 and k+2 < n
    for (int k = 0; k < n + 2; k += 2) {
      if (k >= 2 && k + 3 < n) {
        body(std::false_type{}, k, sa, sb);
        body(std::false_type{}, k + 1, sb, sa);
      } else {
        body(std::true_type{}, k, sa, sb);
        if (k + 1 < n + 2) body(std::true_type{}, k + 1, sb, sa);
      }
    }
    dump();
  }
#endif
}

}  // namespace

// Fused temporal residual block, see include/vidtok_amd.h (vt_temporal_block).  Returns VT_ERR_ARG with a message when
// the shape is not one this kernel covers; vt_temporal_block_supported lets the host ask first.
extern "C" int vt_tblock_desc_size(void) { return (int)sizeof(vt_tblock_desc); }

extern "C" int vt_temporal_block_supported(const vt_tblock_desc* d) {
  if (!d) return 0;
  if (d->dtype != VT_BF16 || d->C != 128 || d->ld != 128) return 0;
  if (d->B <= 0 || d->T <= 0 || d->HW <= 0 || d->HW % TB_PIX != 0) return 0;
  if (d->tmode != VT_TPAD_ZERO && d->tmode != VT_TPAD_REPLICATE) return 0;
  if (d->ln_next_mode < 0 || d->ln_next_mode > 2) return 0;
  if (vt_opt(OPT_TBLOCK_FUSED) == 0) return 0;
  return 1;
}

namespace {
int tblock_launch(const vt_tblock_desc* d, vt_stream stream_, unsigned long long* prof) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  VT_CHECK_ARG(d != nullptr, "vt_temporal_block: null descriptor");
  VT_CHECK_ARG(vt_temporal_block_supported(d),
               "vt_temporal_block: only bf16, C = ld = 128, HW %% 64 == 0, zero / replicate time padding (got dtype %d C %d ld %d "
               "HW %lld tmode %d)", d->dtype, d->C, d->ld, (long long)d->HW, d->tmode);
  VT_CHECK_ARG(d->x && d->w1 && d->w2 && d->norm1_gamma && d->norm1_beta && d->norm2_gamma && d->norm2_beta,
               "vt_temporal_block: null tensor pointer");
  VT_CHECK_ARG(d->keep_y || d->ln_next_mode != 0, "vt_temporal_block: nothing to write (keep_y = 0 and no next norm)");
  VT_CHECK_ARG(!d->keep_y || d->y, "vt_temporal_block: y is null");
  VT_CHECK_ARG(d->ln_next_mode == 0 || (d->next_gamma && d->next_beta && d->n_out), "vt_temporal_block: next norm needs gamma, beta, n_out");
  const uintptr_t al = reinterpret_cast<uintptr_t>(d->x) | reinterpret_cast<uintptr_t>(d->y) | reinterpret_cast<uintptr_t>(d->n_out) |
                       reinterpret_cast<uintptr_t>(d->w1) | reinterpret_cast<uintptr_t>(d->w2);
  VT_CHECK_ARG((al & 15) == 0, "vt_temporal_block: tensors must be 16-byte aligned");
  TBlockArgs a;
  a.x = (const bf16_t*)d->x; a.y = (bf16_t*)d->y; a.n_out = (bf16_t*)d->n_out;
  a.w1 = (const bf16_t*)d->w1; a.w2 = (const bf16_t*)d->w2; a.b1 = d->b1; a.b2 = d->b2;
  a.g1 = d->norm1_gamma; a.be1 = d->norm1_beta; a.g2 = d->norm2_gamma; a.be2 = d->norm2_beta;
  a.gn = d->next_gamma; a.ben = d->next_beta;
  a.B = d->B; a.T = d->T; a.HW = (int)d->HW;
  a.replicate = d->tmode == VT_TPAD_REPLICATE ? 1 : 0;
  a.keep_y = d->keep_y ? 1 : 0;
  a.ln_next = d->ln_next_mode;
  a.eps = d->eps;
  a.prof = prof;
  a.prof_mode = prof ? vt_opt(OPT_TBLOCK_PROF_MODE) : 0;
  // one instantiation per output shape: next norm none / LayerNorm / LayerNorm+SiLU, y kept or not
  static const void* const kerns[6] = {
      reinterpret_cast<const void*>(&tblock_split_kernel<0, true>), reinterpret_cast<const void*>(&tblock_split_kernel<1, true>),
      reinterpret_cast<const void*>(&tblock_split_kernel<1, false>), reinterpret_cast<const void*>(&tblock_split_kernel<2, true>),
      reinterpret_cast<const void*>(&tblock_split_kernel<2, false>), reinterpret_cast<const void*>(&tblock_split_kernel<2, true, true>)};
  int ki = a.ln_next == 0 ? 0 : (a.ln_next == 1 ? (a.keep_y ? 1 : 2) : (a.keep_y ? 3 : 4));
  int lds = T3_LDS;
  const unsigned threads = 512;
  if (prof != nullptr) {                      // measurement aid: stamps [wave 0..7][step][8] of the LayerNorm+SiLU, y kept instantiation
    VT_CHECK_ARG(a.ln_next == 2 && a.keep_y, "vt_temporal_block_profile: ln_next_mode 2 and keep_y only");
    ki = 5;
    lds = T3_LDS + 2048;
  }
  const void* kern = kerns[ki];
  // per device, once: the dynamic-LDS attribute of every instantiation and the CU count (not a per-launch runtime call)
  static std::atomic<int> cus[kMaxDevices];
  int dev = 0;
  VT_CHECK_HIP(hipGetDevice(&dev));
  int ncu = (dev >= 0 && dev < kMaxDevices) ? cus[dev].load(std::memory_order_acquire) : 0;
  if (ncu == 0) {
    for (int k = 0; k < 6; ++k)
      VT_CHECK_HIP(hipFuncSetAttribute(kerns[k], hipFuncAttributeMaxDynamicSharedMemorySize, k == 5 ? T3_LDS + 2048 : T3_LDS));
    VT_CHECK_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    if (ncu <= 0) ncu = 256;
    if (dev >= 0 && dev < kMaxDevices) cus[dev].store(ncu, std::memory_order_release);
  }
  const long long ncols = (long long)d->B * (d->HW / TB_PIX);
  const int grid = ncols < ncu ? (int)ncols : ncu;
  void* kargs[] = {&a};
  VT_CHECK_HIP(hipLaunchKernel(kern, dim3((unsigned)grid), dim3(threads), kargs, lds, stream));
  return VT_OK;
}
}  // namespace

extern "C" int vt_temporal_block(const vt_tblock_desc* d, vt_stream stream) { return tblock_launch(d, stream, nullptr); }

// Measurement aid (scripts/tblock_profile.py): the same launch with s_memtime stamps at the phase boundaries of four
// steps of workgroup 0; stamps [wave][step][16] (uint64 shader-clock ticks) land in `stamps_out` (4*4*16 values).
extern "C" int vt_temporal_block_profile(const vt_tblock_desc* d, uint64_t* stamps_out, vt_stream stream) {
  VT_CHECK_ARG(stamps_out != nullptr, "vt_temporal_block_profile: null output");
  return tblock_launch(d, stream, reinterpret_cast<unsigned long long*>(stamps_out));
}
