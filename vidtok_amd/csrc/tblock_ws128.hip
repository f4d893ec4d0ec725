// Fused temporal residual block of the widest level (C = 128, bf16):
//     y = x + conv2( SiLU(LN2( conv1( SiLU(LN1(x)) ) )) ),   conv = causal Conv1d over T, k = 3
// = ResnetCausalBlock1D._forward of the reference (model_3dcausal.py:473-499 with CausalConv1d :144-159), optionally
// followed by the LayerNorm(+SiLU) the NEXT block starts with (emitted like vt_conv's `ln_mode`).
//
// Unfused, the two K = 384 convolutions of such a block are HBM-shaped (95 FLOP/B): per pixel they move the normalised
// input, the intermediate twice, the residual, the result and the next norm -- ~2 KiB -- in two launches of 1.45 ms.
// Here a block reads x once and writes y (+ the next norm) once: 512-768 B per pixel, one launch.
//
// Structure (weight-stationary): one workgroup per CU; the weights of BOTH convolutions stay in registers for the
// lifetime of the kernel (conv1 in waves 0-3, conv2 in waves 4-7).  A workgroup owns pixel columns -- 64 consecutive
// pixels of one clip -- and walks each column through time:
//   step t:  x[t] rows (prefetched into registers) -> LN1+SiLU -> ring1[t % 3]                         (LDS, bf16)
//            GEMM1: taps = ring1 slots of frames t-2, t-1, t                 -> T1 (bf16, transposed)  (LDS)
//            rows of T1 + b1 -> LN2+SiLU -> ring2[t % 3]
//            GEMM2: taps = ring2 slots                                        -> T2 (fp32)
//            rows of T2 + b2 + x[t] -> y[t], LayerNorm_next -> n[t]                                     (HBM)
// Causal padding: frames before the clip are zeros (v1.0: their taps are skipped) or the first frame repeated (v1.1
// first chunk / un-tiled) or the two frames a previous chunk left behind (v1.1 tiling: CausalConv1d.causal_cache of both
// convolutions, reference model_3dcausal_v1_1.py:159-178 -- here the INPUTS of the two convolutions are intermediates
// that never reach memory, so the kernel keeps the caches itself: at the start of a column it fills the ring slots of
// frames -2, -1 from them, and it writes the rows of frames T - cache_offset - 2, T - cache_offset - 1 back).  Earlier arrangements are in the history: four waves, every phase on the same wave (14 960 cycles
// per step; commit da80542), and eight waves in two roles -- matrix waves / row waves (11 000 cycles; commit ee72952,
// profiles/r02_tblock_v3_phase_cycles.txt).
#include <atomic>
#include <type_traits>

#include "conv_common.h"

// cache policy bits of the y / n stores: nt (streaming).  The rows are read next by another launch, long after they have left every cache;
// stored plainly they displace x rows the O units read again two steps later (1.62 -> 1.58 ms per launch, profiles/r06_c128_kernel_variants.txt)
#define VT_STORE_AUX 2

namespace {

[[maybe_unused]] constexpr int TB_PIX = 64;                      // pixels per column step
[[maybe_unused]] constexpr int TB_ROWP = 272;                    // ring row: 256 B of channels + 16 B pad (bank spread)
[[maybe_unused]] constexpr int TB_SLOT = TB_PIX * TB_ROWP;       // 17 408
[[maybe_unused]] constexpr int TB_RING = 3 * TB_SLOT;            // 52 224
[[maybe_unused]] constexpr int TB_T = TB_PIX * 128 * 4;          // 32 768

// (tensor pointers as uint16_t*: elements of the launch's 16-bit storage type, bf16 or fp16 -- the kernel's template parameter)
typedef uint16_t tb_h;
struct TBlockArgs {
  const tb_h* x;
  tb_h* y;
  tb_h* n_out;
  const tb_h* w1;
  const tb_h* w2;
  const float* b1;
  const float* b2;
  const float* g1; const float* be1;
  const float* g2; const float* be2;
  const float* gn; const float* ben;
  int B, T, HW;
  int replicate;      // frames before the clip: 0 zeros, 1 the first frame repeated, 2 the two cached frames (cache1 / cache2)
  tb_h* cache1;       // [B][2][HW][128]: conv1's input (SiLU(LN1(x))) at frames -2, -1; rewritten in place with frames
  tb_h* cache2;       //   T - cache_off - 2, T - cache_off - 1 of this clip (NULL: no chunk state kept); conv2's input likewise
  int cache_off;
  int keep_y;         // write y (0 only with ln_next != 0: the consumer needs just the normalised tensor)
  int ln_next;        // 0 none, 1 LayerNorm, 2 LayerNorm + SiLU
  float eps;
  unsigned long long* prof;   // PROF instantiation only (vt_temporal_block_profile): cycle stamps of workgroup 0
  int prof_mode;              // PROF only (option tblock_prof_mode): bit 0 GEMMs skipped, bit 1 row units skipped, bit 4 no stores (wrong results)
};

template <int I, int N, typename F>
__device__ __forceinline__ void tb_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    tb_static_for<I + 1, N>(f);
  }
}

// Row arithmetic: LayerNorm (+SiLU) of a pixel row held by 16 lanes x 8 channels, two-pass statistics like
// layernorm_act_kernel, plain (unpacked) fp32 -- a packed-fp32 instruction does not execute while an MFMA of the OTHER
// wave of its SIMD is in flight (profiles/r02_tblock_v3_phase_cycles.txt: an L2 pass in v_pk_* form took 4 430 cycles next
// to a GEMM, 2 020 with the matrix pipe idle).  Nothing may depend on FMA contraction (off for this file; fused
// multiply-adds are spelled out): a pixel's bits must not depend on the code path that computed it (guarded first /
// last steps and straight-line middle steps are separate instantiations of the same source).
#pragma clang fp contract(off)
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <typename H>
__device__ __forceinline__ uint32_t tb_pack2(f32x2 v) {              // round-to-nearest-even, one v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32
  return h16<H>::pack(v[0], v[1]);
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS: two rings of three 64-row slots (bf16 rows of 272 B: 256 + 16 pad), T1 (conv1 rounded to bf16, 272-B rows; b1 is
// added by the rows that read it), T2 (conv2 in fp32, 16-B chunks XOR-swizzled by row), the LayerNorm affines and biases.
// ---------------------------------------------------------------------------------------------------------------------
// T1 rows: 128 bf16 = 256 B, the sixteen 16-B chunks of a row XOR-swizzled by row % 16 (a row read = 16 lanes x 16 B is then a
// permutation of one 256-B bank line, and the two rows a ds_read_b128 lane group spans -- lanes {0-3, 12-15} of row 4 w, {20-27} of
// row 4 w + 1 -- land in complementary bank quads; with 272-B padded rows one quad of every group was hit twice)
[[maybe_unused]] constexpr int T3_T1P = 256;
[[maybe_unused]] constexpr int T3_OFF_T1 = 2 * TB_RING;
[[maybe_unused]] constexpr int T3_OFF_T2 = 2 * TB_RING + TB_PIX * T3_T1P;
[[maybe_unused]] constexpr int T3_LDS = T3_OFF_T2 + TB_T;                 // 153 600
template <typename H>
__device__ __forceinline__ void t3_unpack8(const u32x4& w, float (&v)[8]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    v[2 * q] = h16<H>::lo(w[q]);
    v[2 * q + 1] = h16<H>::hi(w[q]);
  }
}
template <typename H>
__device__ __forceinline__ u32x4 t3_pack8(const float (&o)[8]) {
  u32x4 w;
#pragma unroll
  for (int q = 0; q < 4; ++q) w[q] = tb_pack2<H>(f32x2{o[2 * q], o[2 * q + 1]});
  return w;
}

// ---------------------------------------------------------------------------------------------------------------------
// Two EQUAL groups.  What the stamps of the two-role arrangement showed
// (profiles/r02_tblock_v3_phase_cycles.txt): a step takes ~11 000 cycles because the row waves carry 10 400 cycles of row
// work while the matrix waves are busy for 5 800 and idle for the rest -- and a wave issues one instruction every four
// cycles at best, whatever the instruction (profiles/r03_ws2_iteration_cycles.txt), so the row waves' ~2 000 instructions
// a step cannot go faster than that.  Here the two convolutions go to different waves of a SIMD and BOTH do rows:
//     group 0 (waves 0-3): the weights of conv1 (24 fragments, all in the accumulator half), G1, T1 <- acc, rows
//     group 1 (waves 4-7): the weights of conv2,                                             G2, T2 <- acc, rows
//     phase A_k:   group 0: G1(k)                                  | group 1: T2 <- acc2(k-2), L2(k-1) (4 row units)
//     phase B_k:   group 0: T1 <- acc1(k), O(k-2) (4 units), L1(k+1) unit 0 | group 1: G2(k-1), L1(k+1) units 1-3
// (row unit = 16 pixel rows x 128 channels on a group's 256 threads).  Buffers and their hand-over are those of the
// two-role kernel: written in one phase, read in a later one, one barrier per phase.  The x rows a unit needs (an L1 unit: the rows of
// step k + 1; an O unit: those of step k - 2, its residual) are requested a whole step ahead into ONE register set, each request right behind
// the unit that consumed the register's previous contents (round 6; rounds 3-5 kept two sets and two loop bodies and requested at the top of
// phase B: 40 registers of x rows in group 0, the allocator at 254 / 256).  The middle steps run in a loop of their own that is entered behind
// a real vmcnt(0), so the compiler's counted waits in front of the units leave the step's stores in flight (see the loops below).
// The LayerNorm affines and the biases live in the LDS (4 KiB; a unit reads what it needs: two ds_read_b128 an array),
// all addressing is 32-bit through buffer descriptors rebased to the frame, the (column, frame) of a virtual step is
// carried by additions.  Row arithmetic: plain fp32, element order of the two-role kernel (L1 sums the even and the odd
// channels separately, as its packed form did): the two arrangements agreed to the bit when both existed.
// LDS: 153 600 + 4 096 = 157 696 B.
// ---------------------------------------------------------------------------------------------------------------------
[[maybe_unused]] constexpr int T4_OFF_PRM = T3_LDS;                       // g1 | be1 | g2 | be2 | gn | ben | b1 | b2, 128 fp32 each
[[maybe_unused]] constexpr int T4_LDS = T4_OFF_PRM + 8 * 128 * 4;         // 157 696
[[maybe_unused]] constexpr int T4_FD = 6;                                 // fragment prefetch distance of a GEMM, in MFMAs

template <typename H, bool FIRST>
__device__ __forceinline__ void t4_mfma(const u32x4& w, const u32x4& x, f32x16& acc) {
  if constexpr (std::is_same<H, bf16_t>::value) {
    if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc) : "a"(w), "v"(x));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(x));
  } else {
    if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(acc) : "a"(w), "v"(x));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(x));
  }
}
// LayerNorm(+SiLU) of a row slice: ln_row8 of common.h (round 6: one-pass moments, centring + scaling as one fma, -log2(e) folded into the
// affines in the LDS; 7.5 plain + 2 transcendental instructions an element against 9.5 + 2 of the two-pass form this kernel had until round 5)
template <bool SILU>
__device__ __forceinline__ void t4_row_norm(float (&v)[8], const float (&g)[8], const float (&b)[8], float eps, float (&o)[8]) {
  ln_row8<16, SILU>(v, g, b, eps, o);
}

// CACHE: the instantiation that keeps v1.1 chunk state (cache1 / cache2).  Separate because its loads and stores are
// conditional (a column starts, a kept frame goes by), and with conditional memory operations in the step body the
// compiler no longer counts what is in flight -- it waits for everything.
// H = the 16-bit storage type (bf16_t / f16_t)
template <typename H, int LNN, bool KEEP, bool CACHE, bool PROF = false>
__global__ __launch_bounds__(512, 1) void tblock_pair_kernel(const TBlockArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cw = wave & 3, grp = wave >> 2;
  const int ptiles = p.HW / TB_PIX;
  const int ncols = p.B * ptiles;
  const int G = gridDim.x;
  const int slot_id = xcd_remap(blockIdx.x, G);
  const int cq = ncols / G, cr = ncols - cq * G;
  const int c_begin = slot_id * cq + min(slot_id, cr);
  const int c_end = c_begin + cq + (slot_id < cr ? 1 : 0);
  if (c_begin >= c_end) return;
  // Issue priority: group 1 is the step's critical path in both phases (T2 + three L2 units beside G1; G2 + two O units beside group 0's
  // rows -- group 0 waits ~600 + ~1 700 cycles a step at the two barriers, profiles/r06_tblock_pair_phase_cycles.txt), so its instructions
  // win the arbitration throughout (2, its GEMM 3); group 0 stays at 0, its GEMM included.  1.64 -> 1.61 ms per launch.
  if (grp == 1) __builtin_amdgcn_s_setprio(2);
  const int n = (c_end - c_begin) * p.T;                             // virtual steps of this workgroup
  constexpr int R1 = 0, R2 = TB_RING;                                // ring offsets in the LDS
  const unsigned frame_bytes = (unsigned)p.HW * 256u;

  // ---- parameters -> LDS
  float* prm = reinterpret_cast<float*>(smem + T4_OFF_PRM);
  {
    const int c = tid & 127, a = tid >> 7;                           // arrays a and a + 4
    // channel c = 8 oct + 4 half + e of an array sits at [half][oct][e]: the 16 lanes of a row read their first (second) four
    // values as 256 contiguous bytes.  In channel order (32 B per lane) lanes oct and oct + 8 of a ds_read_b128 group met in
    // the same banks -- every parameter read 2-way conflicted, 23 % of the kernel's LDS cycles (profiles/r03_bench_bf16_sq_pmc.txt)
    const int pc = ((c >> 2) & 1) * 64 + (c >> 3) * 4 + (c & 3);
    const float* src0 = a == 0 ? p.g1 : (a == 1 ? p.be1 : (a == 2 ? p.g2 : p.be2));
    prm[a * 128 + pc] = ln_fold(src0[c], true);                      // norm1 / norm2 are followed by SiLU: the affines carry -log2(e) (ln_row8)
    float v;
    if (a == 0) v = LNN ? ln_fold(p.gn[c], LNN == 2) : 1.0f;
    else if (a == 1) v = LNN ? ln_fold(p.ben[c], LNN == 2) : 0.0f;
    else if (a == 2) v = p.b1 ? p.b1[c] : 0.0f;
    else v = p.b2 ? p.b2[c] : 0.0f;
    prm[(a + 4) * 128 + pc] = v;
  }

  // ---- virtual step -> (batch, pixel tile, frame), carried by additions
  struct Cur {
    int b, pt, t;
  };
  auto cur_at = [&](int k) __attribute__((always_inline)) {
    Cur c;
    const int q = k / p.T;
    c.t = k - q * p.T;
    const int col = c_begin + q;
    c.b = col / ptiles;
    c.pt = col - c.b * ptiles;
    return c;
  };
  auto cur_step = [&](Cur& c) __attribute__((always_inline)) {
    c.t += 1;
    if (c.t == p.T) {
      c.t = 0;
      c.pt += 1;
      if (c.pt == ptiles) {
        c.pt = 0;
        c.b += 1;
      }
    }
  };
  auto frame_rsrc = [&](const tb_h* base, const Cur& c) __attribute__((always_inline)) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<tb_h*>(base) + ((long long)c.b * p.T + c.t) * p.HW * 128, 0, frame_bytes, 0x00020000);
  };

  // PROF: stamps of workgroup 0, steps [8, 12): [step][wave][8] in the LDS behind the parameters, copied out at the end
  unsigned long long* stamps = reinterpret_cast<unsigned long long*>(smem + T4_LDS);
  int kcur = 0;
  auto stamp = [&](int i) __attribute__((always_inline)) {
    if constexpr (PROF) {
      if (blockIdx.x == 0 && kcur >= 8 && kcur < 12) {
        const unsigned long long ts = __builtin_amdgcn_s_memtime();
        if (lane == 0) stamps[((kcur - 8) * 8 + wave) * 8 + i] = ts;
      }
    }
  };
  auto dump = [&]() __attribute__((always_inline)) {
    if constexpr (PROF) {
      if (blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < 32; ++i) p.prof[wave * 32 + i] = stamps[((i / 8) * 8 + wave) * 8 + (i % 8)];
    }
  };

  // ---- stationary weights of my convolution
  u32x4 wreg[24];
  {
    const tb_h* wsrc = grp == 0 ? p.w1 : p.w2;
    const long long roff = (long long)(cw * 32 + (lane & 31)) * 384 + (lane >> 5) * 8;
#pragma unroll
    for (int g = 0; g < 24; ++g) wreg[g] = *reinterpret_cast<const u32x4*>(wsrc + roff + g * 16);
    // a REAL s_waitcnt (the builtin, not asm): the compiler's wait-count bookkeeping then knows the weights have landed.
    // Left pending at the loop entry, it re-waited for them in front of every MFMA of every step -- vmcnt(7) ... vmcnt(0) --
    // i.e. for whatever the wave had in flight, including the stores this kernel wants to keep in flight across a GEMM.
    __builtin_amdgcn_s_waitcnt(0x0F70);                              // vmcnt(0)
  }
  f32x16 acc[2];
  // GEMM of the step at frame t whose ring slot is s3 (= virtual step % 3) over the live taps [KT0, 3) of the ring at `roff`
  auto gemm = [&](auto kt0_c, int roff, int s3, int t) __attribute__((always_inline)) {
    constexpr int KT0 = decltype(kt0_c)::value;
    constexpr int M0 = 16 * KT0;
    int lo = lane;
    asm volatile("" : "+v"(lo));
    const int frag_off = roff + (lo & 31) * TB_ROWP + (lo >> 5) * 16;   // B-fragment of pixel lane%32, k half lane/32
    int sp[3];
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      // tap kt = frame t - 2 + kt; replicate: frames before the clip = its frame 0 (virtual step k - t); cache: their own
      // slots (filled from the cache at the start of the column)
      const int back = (t - 2 + kt >= 0 || (CACHE && p.replicate == 2)) ? 2 - kt : t;
      int sl = s3 - back;
      sl = sl < 0 ? sl + 3 : sl;
      sp[kt] = frag_off + sl * TB_SLOT;
    }
    auto faddr = [&](int m) __attribute__((always_inline)) -> const u32x4* {
      const int g = m >> 1, j = m & 1;
      return reinterpret_cast<const u32x4*>(smem + sp[g >> 3] + j * (32 * TB_ROWP) + (g & 7) * 32);
    };
    u32x4 xf[T4_FD + 1];
#pragma unroll
    for (int m = M0; m < M0 + T4_FD; ++m) xf[m % (T4_FD + 1)] = *faddr(m);
    if (grp == 1) __builtin_amdgcn_s_setprio(3);
    tb_static_for<M0, 48>([&](auto mc) __attribute__((always_inline)) {
      constexpr int m = decltype(mc)::value;
      t4_mfma<H, (m < M0 + 2)>(wreg[m >> 1], xf[m % (T4_FD + 1)], acc[m & 1]);
      if constexpr (m + T4_FD < 48) xf[(m + T4_FD) % (T4_FD + 1)] = *faddr(m + T4_FD);
      __builtin_amdgcn_sched_barrier(0);
    });
    if (grp == 1) __builtin_amdgcn_s_setprio(2);
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");             // last MFMA -> first VALU reader of its accumulator
  };
  auto run_gemm = [&](int roff, int s3, int t) __attribute__((always_inline)) {
    const int first = p.replicate != 0 ? 0 : max(0, 2 - t);          // uniform
    if (first == 0) gemm(std::integral_constant<int, 0>{}, roff, s3, t);
    else if (first == 1) gemm(std::integral_constant<int, 1>{}, roff, s3, t);
    else gemm(std::integral_constant<int, 2>{}, roff, s3, t);
  };
  // accumulators: lane = pixel 32 j + lane % 32, channels 32 cw + 8 g + 4 h + e
  auto acc_to_T1 = [&]() __attribute__((always_inline)) {                                           // rounded to bf16 (8 B per quad); b1 is added by the rows
    int lo = lane;
    asm volatile("" : "+v"(lo));
    const int l31 = lo & 31;
    const int base = T3_OFF_T1 + l31 * T3_T1P + 8 * (lo >> 5);       // chunk 4 cw + g of the row, half lane / 32, at slot chunk ^ (row % 16)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 w;
        w[0] = tb_pack2<H>(f32x2{acc[j][4 * g], acc[j][4 * g + 1]});
        w[1] = tb_pack2<H>(f32x2{acc[j][4 * g + 2], acc[j][4 * g + 3]});
        *reinterpret_cast<u32x2*>(smem + base + j * (32 * T3_T1P) + (((4 * cw + g) ^ (l31 & 15)) << 4)) = w;
      }
  };
  auto acc_to_T2 = [&]() __attribute__((always_inline)) {                                           // fp32, 16-B chunk (8 cw + 2 g + h) ^ (row % 32) of row 32 j + lane % 32
    int lo = lane;
    asm volatile("" : "+v"(lo));
    const int l31 = lo & 31;
    const int base = T3_OFF_T2 + l31 * 512 + (((cw * 8 + (lo >> 5)) ^ l31) << 4);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[j][4 * g + e];
        *reinterpret_cast<f32x4*>(smem + ((base ^ (g << 5)) + j * (32 * 512))) = v;
      }
  };

  // ---- row units: thread (oct_j, row0) of a group handles channels [8 oct_j, +8) of row row0 + 16 it of unit it
  auto unit_geom = [&](int& oct_j, int& row0) __attribute__((always_inline)) {                      // recomputed per phase from an opaque id: nothing resident
    int tt = tid;
    asm volatile("" : "+v"(tt));
    const int vt = tt & 255;
    oct_j = vt & 15;
    row0 = vt >> 4;
  };
  auto ld_prm = [&](int arr, int oct_j, float (&o)[8]) __attribute__((always_inline)) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(prm + arr * 128 + 4 * oct_j);
    const f32x4 b = *reinterpret_cast<const f32x4*>(prm + arr * 128 + 64 + 4 * oct_j);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = a[e];
      o[4 + e] = b[e];
    }
  };
  // x rows of the step at cursor c: units [U0, U1) of my group
  auto load_rows = [&](auto u0_c, auto u1_c, u32x4 (&dst)[4], const Cur& c) __attribute__((always_inline)) {
    constexpr int U0 = decltype(u0_c)::value, U1 = decltype(u1_c)::value;
    int oct_j, row0;
    unit_geom(oct_j, row0);
    const __amdgpu_buffer_rsrc_t rs = frame_rsrc(p.x, c);
    const int vo = (c.pt * TB_PIX + row0) * 256 + oct_j * 16;
#pragma unroll
    for (int it = U0; it < U1; ++it) dst[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo + it * (16 * 256), 0, 0);
  };
  // L1 of units [U0, U1): LayerNorm1 + SiLU of x rows -> ring1 slot s3
  // chunk state: the row of frame c.t that a convolution's cache keeps (slot j = t - (T - cache_off - 2) in {0, 1})
  auto cache_put = [&](tb_h* cache, const Cur& c, int oct_j, int row, const u32x4& w) __attribute__((always_inline)) {
    if constexpr (!CACHE) return;
    const int j = c.t - (p.T - p.cache_off - 2);
    if (cache != nullptr && (j == 0 || j == 1)) {                    // uniform
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(cache + ((long long)c.b * 2 + j) * p.HW * 128, 0, frame_bytes, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b128(w, rs, (c.pt * TB_PIX + row) * 256 + oct_j * 16, 0, 0);
    }
  };
  // ... and the other direction: the cached frames -2, -1 of the column at cursor c into ring slots sl2, sl1 (a group's 256
  // threads move 2 x 64 rows of 256 B; once per column)
  auto cache_get = [&](const tb_h* cache, int roff, const Cur& c, int sl2, int sl1) __attribute__((always_inline)) {
    if constexpr (!CACHE) return;
    int oct_j, row0;
    unit_geom(oct_j, row0);
    u32x4 w[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<tb_h*>(cache) + ((long long)c.b * 2 + j) * p.HW * 128, 0, frame_bytes, 0x00020000);
#pragma unroll
      for (int it = 0; it < 4; ++it) w[j][it] = __builtin_amdgcn_raw_buffer_load_b128(rs, (c.pt * TB_PIX + row0 + 16 * it) * 256 + oct_j * 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int it = 0; it < 4; ++it)
        *reinterpret_cast<u32x4*>(smem + roff + (j == 0 ? sl2 : sl1) * TB_SLOT + (row0 + 16 * it) * TB_ROWP + oct_j * 16) = w[j][it];
  };
  auto ln1_units = [&](auto u0_c, auto u1_c, const u32x4 (&xr)[4], int s3, const Cur& c) __attribute__((always_inline)) {
    constexpr int U0 = decltype(u0_c)::value, U1 = decltype(u1_c)::value;
    int oct_j, row0;
    unit_geom(oct_j, row0);
    float g[8], b[8];
    ld_prm(0, oct_j, g);
    ld_prm(1, oct_j, b);
    const int wo = R1 + s3 * TB_SLOT + row0 * TB_ROWP + oct_j * 16;
#pragma unroll
    for (int it = U0; it < U1; ++it) {
      float v[8], o[8];
      t3_unpack8<H>(xr[it], v);
      t4_row_norm<true>(v, g, b, p.eps, o);
      const u32x4 w = t3_pack8<H>(o);
      *reinterpret_cast<u32x4*>(smem + wo + it * (16 * TB_ROWP)) = w;
      cache_put(p.cache1, c, oct_j, row0 + 16 * it, w);
      __builtin_amdgcn_sched_barrier(0);                                // one row at a time: interleaved rows double the live registers (fp16 spilled)
    }
  };

  auto out_store = [&](const u32x4& w, __amdgpu_buffer_rsrc_t rs, int vo) __attribute__((always_inline)) {
    if constexpr (PROF) {
      if (p.prof_mode & 16) return;                                  // measurement: no stores at all
    }
    __builtin_amdgcn_raw_buffer_store_b128(w, rs, vo, 0, VT_STORE_AUX);
  };
  // ---- prologue: L1(0) -> ring1 slot 0; x(1) rows for the L1 units of phase B_0
  // x rows in registers: ONE set (round 6).  A row unit's input -- the x rows of step k + 1 for an L1 unit, those of step k - 2 as the
  // residual of an O unit -- is requested a full step ahead, and the request goes out right BEHIND the unit that consumed the register's
  // previous contents, into the same registers.  Until round 5 the requests of a phase went out together at its top, in front of the phase's
  // stores, into a second register set (two sets, two bodies): 40 of the 128 architectural registers in group 0 held x rows, the allocator
  // sat at 254 / 256 with values parked in the accumulator half, and a fragment ring deeper than three MFMAs did not fit (the fp16
  // instantiations spilled outright).  A request behind stores returns only when they have retired (~5 000 cycles) -- its consumer is
  // a whole step (> 8 000 cycles) away.
  struct XSet {
    u32x4 xn[4], xr[4];
  };
  XSet xs;
  Cur c_p2 = cur_at(0);          // cursors of virtual steps k + 2, k + 1, k, k - 1, k - 2 (valid where those steps exist)
  Cur c_p1 = c_p2, c_0 = c_p2, c_m1 = c_p2, c_m2 = c_p2;
  __syncthreads();               // parameters are in the LDS
  constexpr int L1G0 = 3;                                            // L1 units [0, L1G0) belong to group 0, the rest to group 1 (4 : 0 measured the same)
  using IL = std::integral_constant<int, L1G0>;
  if (grp == 0) {
    load_rows(std::integral_constant<int, 0>{}, IL{}, xs.xn, c_0);
    ln1_units(std::integral_constant<int, 0>{}, IL{}, xs.xn, 0, c_0);
  } else {
    load_rows(IL{}, std::integral_constant<int, 4>{}, xs.xn, c_0);
    ln1_units(IL{}, std::integral_constant<int, 4>{}, xs.xn, 0, c_0);
    if (CACHE && p.replicate == 2) cache_get(p.cache1, R1, c_0, 1, 2);         // frames -2, -1 of the first column: slots (-2) % 3, (-1) % 3
  }
  cur_step(c_p1);
  c_p2 = c_p1;
  cur_step(c_p2);
  if (1 < n) {
    if (grp == 0) load_rows(std::integral_constant<int, 0>{}, IL{}, xs.xn, c_p1);
    else load_rows(IL{}, std::integral_constant<int, 4>{}, xs.xn, c_p1);
  }
  // every register of the set is defined before the loop (guarded steps skip loads, not uses); a group touches only
  // its own units: group 0 xn[0..2] and xr[0..1], group 1 xn[3] and xr[2..3]
  if (grp == 0) {
#pragma unroll
    for (int it = 0; it < 2; ++it) xs.xr[it] = xs.xn[0];
  } else {
#pragma unroll
    for (int it = 2; it < 4; ++it) xs.xr[it] = xs.xn[3];
  }
  int s3 = 0;                    // k % 3
  auto advance = [&]() __attribute__((always_inline)) {
    c_m2 = c_m1;
    c_m1 = c_0;
    c_0 = c_p1;
    c_p1 = c_p2;
    cur_step(c_p2);
    s3 = s3 == 2 ? 0 : s3 + 1;
  };
  auto slot_back = [&](int back) __attribute__((always_inline)) {                                   // (k - back) % 3
    const int s = s3 - back;
    return s < 0 ? s + 3 : s;
  };
  __syncthreads();               // L1(0) is in ring1

  // L2 of units [U0, U1) of the step whose ring slot is sl: rows of T1 (conv1, bf16) + b1 -> LayerNorm2 + SiLU -> ring2
  auto l2_units = [&](auto u0_c, auto u1_c, int sl, const Cur& c) __attribute__((always_inline)) {
    constexpr int U0 = decltype(u0_c)::value, U1 = decltype(u1_c)::value;
    int oct_j, row0;
    unit_geom(oct_j, row0);
    float bo1[8], g[8], b[8];
    ld_prm(6, oct_j, bo1);
    ld_prm(2, oct_j, g);
    ld_prm(3, oct_j, b);
    const int ro = T3_OFF_T1 + row0 * T3_T1P + ((oct_j ^ row0) << 4);    // rows row0 + 16 it: swizzle key row % 16 = row0
    const int wo = R2 + sl * TB_SLOT + row0 * TB_ROWP + oct_j * 16;
    u32x4 tw[4];                                                       // the row after the one being worked on is in flight, not all of them
    tw[U0] = *reinterpret_cast<const u32x4*>(smem + ro + U0 * (16 * T3_T1P));
#pragma unroll
    for (int it = U0; it < U1; ++it) {
      if (it + 1 < U1) tw[it + 1] = *reinterpret_cast<const u32x4*>(smem + ro + (it + 1) * (16 * T3_T1P));
      float v[8], o[8];
      t3_unpack8<H>(tw[it], v);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] + bo1[e];
      t4_row_norm<true>(v, g, b, p.eps, o);
      const u32x4 w = t3_pack8<H>(o);
      *reinterpret_cast<u32x4*>(smem + wo + it * (16 * TB_ROWP)) = w;
      cache_put(p.cache2, c, oct_j, row0 + 16 * it, w);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // O of unit IT of the step at cursor c: rows of T2 + b2 + x -> y, LayerNorm_next -> n
  auto o_unit = [&](auto it_c, const u32x4& xrow, const Cur& c) __attribute__((always_inline)) {
    constexpr int it = decltype(it_c)::value;
    int oct_j, row0;
    unit_geom(oct_j, row0);
    float bo2[8], gn[8], bn[8];
    ld_prm(7, oct_j, bo2);
    if constexpr (LNN != 0) {
      ld_prm(4, oct_j, gn);
      ld_prm(5, oct_j, bn);
    }
    const int vo = (c.pt * TB_PIX + row0) * 256 + oct_j * 16 + it * (16 * 256);
    // row row0 + 16 it: the swizzle key (row % 32) flips bit 4 for odd it
    const int tb = T3_OFF_T2 + row0 * 512 + (((2 * oct_j) ^ row0) << 4);
    const int to = ((it & 1) ? (tb ^ 256) : tb) + it * (16 * 512);
    const f32x4 t0 = *reinterpret_cast<const f32x4*>(smem + to);
    const f32x4 t1 = *reinterpret_cast<const f32x4*>(smem + (to ^ 16));
    float v[8], xv[8];
    t3_unpack8<H>(xrow, xv);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = xv[e] + ((e < 4 ? t0[e] : t1[e - 4]) + bo2[e]);
    if constexpr (KEEP) out_store(t3_pack8<H>(v), frame_rsrc(p.y, c), vo);
    if constexpr (LNN != 0) {
      float o[8];
      t4_row_norm<(LNN == 2)>(v, gn, bn, p.eps, o);
      out_store(t3_pack8<H>(o), frame_rsrc(p.n_out, c), vo);
    }
    __builtin_amdgcn_sched_barrier(0);                                  // a unit at a time (see ln1_units)
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  using I4 = std::integral_constant<int, 4>;

  // Who does which row unit (unit = 16 pixel rows on a group's 256 threads; ~630-730 cycles each, a GEMM ~2 100-2 400):
  //   phase A_k:  group 0: G1(k), L2(k-1) unit 3                         | group 1: T2 <- conv2(k-2), L2(k-1) units 0-2
  //   phase B_k:  group 0: T1 <- conv1(k), O0, L1_0, O1, L1_1, L1_2      | group 1: G2(k-1), O2, L1_3, O3
  // The y / n stores (two per O unit and thread) are kept apart on purpose: a CU retires ~13 B of stores a cycle
  // (scripts/hbm_bw_bench.hip: 6.7 TB/s over 256 CUs), so a step's 32 KiB take 2 500 cycles of the store path, and a wave
  // issuing into a full store queue stands still -- with all of O in one group back to back its eight stores cost it
  // 2 500 cycles on top of 2 900 of arithmetic (profiles/r03_tblock_pair_phase_cycles.txt).
  if (grp == 0) {
    // ======================================================= group 0 =======================================================
    auto body = [&](auto guard_c, int k) __attribute__((always_inline)) {
      constexpr bool GUARD = decltype(guard_c)::value;
      kcur = k;
      stamp(0);
      // ---- phase A_k: G1(k) (the result stays in the accumulators across the barrier), one unit of L2(k-1)
      if ((!GUARD || k < n) && !(PROF && (p.prof_mode & 1))) run_gemm(R1, s3, c_0.t);
      stamp(1);
      if ((!GUARD || (k >= 1 && k - 1 < n)) && !(PROF && (p.prof_mode & 2))) l2_units(I3{}, I4{}, slot_back(1), c_m1);
      // step k-1 opened a column: conv2's cached frames -2, -1 go into ring2 slots (k-3) % 3, (k-2) % 3 -- read for the last
      // time by G2(k-2) in phase B_{k-1}, needed by G2(k-1) in phase B_k
      if (CACHE && p.replicate == 2 && k >= 1 && k - 1 < n && c_m1.t == 0) cache_get(p.cache2, R2, c_m1, s3, slot_back(2));
      stamp(2);
      __syncthreads();
      // ---- phase B_k: T1 <- conv1(k), my units of O(k-2) and L1(k+1); behind each unit the request for its successor of step k+1
      if (!GUARD || k < n) acc_to_T1();
      stamp(3);
      const bool do_o = (!GUARD || (k >= 2 && k - 2 < n)) && !(PROF && (p.prof_mode & 2));
      const bool do_l1 = (!GUARD || k + 1 < n) && !(PROF && (p.prof_mode & 2));
      const bool ld_r = !GUARD || (k >= 1 && k - 1 < n);                 // residual rows of step k-1, for O(k-1) in phase B_{k+1}
      const bool ld_n = !GUARD || k + 2 < n;                             // x rows of step k+2, for L1(k+2) in phase B_{k+1}
      if (do_o) o_unit(I0{}, xs.xr[0], c_m2);
      if (ld_r) load_rows(I0{}, I1{}, xs.xr, c_m1);
      if (do_l1) ln1_units(I0{}, I1{}, xs.xn, slot_back(2), c_p1);       // (k + 1) % 3 = (k - 2) % 3
      if (ld_n) load_rows(I0{}, I1{}, xs.xn, c_p2);
      if (do_o) o_unit(I1{}, xs.xr[1], c_m2);
      if (ld_r) load_rows(I1{}, I2{}, xs.xr, c_m1);
      stamp(4);
      if (do_l1) ln1_units(I1{}, IL{}, xs.xn, slot_back(2), c_p1);
      if (ld_n) load_rows(I1{}, IL{}, xs.xn, c_p2);
      stamp(5);
      __syncthreads();
      stamp(6);
      advance();
    };
    // three loops, not one with a branch: the registers with requests in flight cross only the back edge of the middle loop, whose body has
    // no guarded twin to merge with (with both bodies in one loop the allocator moved the x rows at the loop header -- behind a vmcnt(0))
    int k = 0;
    for (; k < 2 && k < n + 2; ++k) body(std::true_type{}, k);
    // a REAL vmcnt(0) (the builtin): the compiler's wait-count bookkeeping enters the middle loop with nothing pending, so the waits it
    // places in front of the units count only what the loop itself issued behind the request (vmcnt(4..7): stores stay in flight).  With
    // the guarded steps' unknown state merged in at the loop header, every unit waited for vmcnt(0) -- i.e. for the previous step's STORES
    // to be acknowledged by the memory, ~1 500 cycles a unit (profiles/r06_tblock_pair_phase_cycles.txt, "no stores")
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (; k + 2 < n; ++k) body(std::false_type{}, k);              // straight-line middle steps
    for (; k < n + 2; ++k) body(std::true_type{}, k);
  } else {
    // ======================================================= group 1 =======================================================
    auto body = [&](auto guard_c, int k) __attribute__((always_inline)) {
      constexpr bool GUARD = decltype(guard_c)::value;
      kcur = k;
      stamp(0);
      // ---- phase A_k: T2 <- conv2(k-2) (nobody reads T2 in an A phase), three units of L2(k-1)
      if (!GUARD || (k >= 2 && k - 2 < n)) acc_to_T2();
      stamp(1);
      if ((!GUARD || (k >= 1 && k - 1 < n)) && !(PROF && (p.prof_mode & 2))) l2_units(I0{}, I3{}, slot_back(1), c_m1);
      stamp(2);
      __syncthreads();
      // ---- phase B_k: G2(k-1); my units of O(k-2) and L1(k+1), each followed by the request for its successor of step k+1
      if ((!GUARD || (k >= 1 && k - 1 < n)) && !(PROF && (p.prof_mode & 1))) run_gemm(R2, slot_back(1), c_m1.t);
      stamp(3);
      const bool do_o = (!GUARD || (k >= 2 && k - 2 < n)) && !(PROF && (p.prof_mode & 2));
      const bool do_l1 = (!GUARD || k + 1 < n) && !(PROF && (p.prof_mode & 2));
      const bool ld_r = !GUARD || (k >= 1 && k - 1 < n);
      const bool ld_n = !GUARD || k + 2 < n;
      if (do_o) o_unit(I2{}, xs.xr[2], c_m2);
      if (ld_r) load_rows(I2{}, I3{}, xs.xr, c_m1);
      stamp(4);
      if (do_l1) ln1_units(IL{}, I4{}, xs.xn, slot_back(2), c_p1);
      if (ld_n) load_rows(IL{}, I4{}, xs.xn, c_p2);
      if (do_o) o_unit(I3{}, xs.xr[3], c_m2);
      if (ld_r) load_rows(I3{}, I4{}, xs.xr, c_m1);
      // step k+1 opens a column: conv1's cached frames -2, -1 go into ring1 slots (k-1) % 3, k % 3 -- read for the last time
      // by G1(k) in phase A_k, needed by G1(k+1) in phase A_{k+1}
      if (CACHE && p.replicate == 2 && k + 1 < n && c_p1.t == 0) cache_get(p.cache1, R1, c_p1, slot_back(1), s3);
      stamp(5);
      __syncthreads();
      stamp(6);
      advance();
    };
    // three loops, not one with a branch: the registers with requests in flight cross only the back edge of the middle loop, whose body has
    // no guarded twin to merge with (with both bodies in one loop the allocator moved the x rows at the loop header -- behind a vmcnt(0))
    int k = 0;
    for (; k < 2 && k < n + 2; ++k) body(std::true_type{}, k);
    // a REAL vmcnt(0) (the builtin): the compiler's wait-count bookkeeping enters the middle loop with nothing pending, so the waits it
    // places in front of the units count only what the loop itself issued behind the request (vmcnt(4..7): stores stay in flight).  With
    // the guarded steps' unknown state merged in at the loop header, every unit waited for vmcnt(0) -- i.e. for the previous step's STORES
    // to be acknowledged by the memory, ~1 500 cycles a unit (profiles/r06_tblock_pair_phase_cycles.txt, "no stores")
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (; k + 2 < n; ++k) body(std::false_type{}, k);              // straight-line middle steps
    for (; k < n + 2; ++k) body(std::true_type{}, k);
  }
  dump();
#endif
}

}  // namespace

// Fused temporal residual block, see include/vidtok_amd.h (vt_temporal_block).  Returns VT_ERR_ARG with a message when
// the shape is not one this kernel covers; vt_temporal_block_supported lets the host ask first.
extern "C" int vt_tblock_desc_size(void) { return (int)sizeof(vt_tblock_desc); }

extern "C" int vt_temporal_block_supported(const vt_tblock_desc* d) {
  if (!d) return 0;
  if (!vt_is_h16(d->dtype) || d->C != 128 || d->ld != 128) return 0;
  if (d->B <= 0 || d->T <= 0 || d->HW <= 0 || d->HW % TB_PIX != 0) return 0;
  if (d->tmode != VT_TPAD_ZERO && d->tmode != VT_TPAD_REPLICATE && d->tmode != VT_TPAD_CACHE) return 0;
  if (d->tmode == VT_TPAD_CACHE && (d->cache1 == nullptr || d->cache2 == nullptr)) return 0;
  // chunk state is rewritten in place: the rows it keeps must come from frames >= 1 of this clip (frame 0's rows are written
  // in the phase that still reads the old ones)
  if ((d->cache1 != nullptr || d->cache2 != nullptr) && (d->cache_offset < 0 || d->T - d->cache_offset < 3)) return 0;
  if (d->ln_next_mode < 0 || d->ln_next_mode > 2) return 0;
  if (vt_opt(OPT_TBLOCK_FUSED) == 0) return 0;
  return 1;
}

namespace {
int tblock_launch(const vt_tblock_desc* d, vt_stream stream_, unsigned long long* prof) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  VT_CHECK_ARG(d != nullptr, "vt_temporal_block: null descriptor");
  VT_CHECK_ARG(vt_temporal_block_supported(d),
               "vt_temporal_block: only bf16 / fp16, C = ld = 128, HW %% 64 == 0; cache mode needs both caches, kept chunk state "
               "T - cache_offset >= 3 (got dtype %d C %d ld %d HW %lld T %d tmode %d cache_offset %d)", d->dtype, d->C, d->ld,
               (long long)d->HW, d->T, d->tmode, d->cache_offset);
  VT_CHECK_ARG(d->x && d->w1 && d->w2 && d->norm1_gamma && d->norm1_beta && d->norm2_gamma && d->norm2_beta,
               "vt_temporal_block: null tensor pointer");
  VT_CHECK_ARG(d->keep_y || d->ln_next_mode != 0, "vt_temporal_block: nothing to write (keep_y = 0 and no next norm)");
  VT_CHECK_ARG(!d->keep_y || d->y, "vt_temporal_block: y is null");
  VT_CHECK_ARG(d->ln_next_mode == 0 || (d->next_gamma && d->next_beta && d->n_out), "vt_temporal_block: next norm needs gamma, beta, n_out");
  const uintptr_t al = reinterpret_cast<uintptr_t>(d->x) | reinterpret_cast<uintptr_t>(d->y) | reinterpret_cast<uintptr_t>(d->n_out) |
                       reinterpret_cast<uintptr_t>(d->w1) | reinterpret_cast<uintptr_t>(d->w2);
  VT_CHECK_ARG((al & 15) == 0, "vt_temporal_block: tensors must be 16-byte aligned");
  TBlockArgs a;
  a.x = (const tb_h*)d->x; a.y = (tb_h*)d->y; a.n_out = (tb_h*)d->n_out;
  a.w1 = (const tb_h*)d->w1; a.w2 = (const tb_h*)d->w2; a.b1 = d->b1; a.b2 = d->b2;
  a.g1 = d->norm1_gamma; a.be1 = d->norm1_beta; a.g2 = d->norm2_gamma; a.be2 = d->norm2_beta;
  a.gn = d->next_gamma; a.ben = d->next_beta;
  a.B = d->B; a.T = d->T; a.HW = (int)d->HW;
  a.replicate = d->tmode == VT_TPAD_REPLICATE ? 1 : (d->tmode == VT_TPAD_CACHE ? 2 : 0);
  a.cache1 = (tb_h*)d->cache1; a.cache2 = (tb_h*)d->cache2; a.cache_off = d->cache_offset;
  VT_CHECK_ARG(((reinterpret_cast<uintptr_t>(d->cache1) | reinterpret_cast<uintptr_t>(d->cache2)) & 15) == 0, "vt_temporal_block: caches must be 16-byte aligned");
  a.keep_y = d->keep_y ? 1 : 0;
  a.ln_next = d->ln_next_mode;
  a.eps = d->eps;
  a.prof = prof;
  a.prof_mode = prof ? vt_opt(OPT_TBLOCK_PROF_MODE) : 0;
  VT_CHECK_ARG((long long)d->HW * 256 < (1ll << 31), "vt_temporal_block: frames of at most 2^23 pixels");
  // one instantiation per output shape: next norm none / LayerNorm / LayerNorm+SiLU, y kept or not
#define VT_TB_KERNS(H)                                                                                                              \
  reinterpret_cast<const void*>(&tblock_pair_kernel<H, 0, true, false>), reinterpret_cast<const void*>(&tblock_pair_kernel<H, 1, true, false>),      \
      reinterpret_cast<const void*>(&tblock_pair_kernel<H, 1, false, false>), reinterpret_cast<const void*>(&tblock_pair_kernel<H, 2, true, false>), \
      reinterpret_cast<const void*>(&tblock_pair_kernel<H, 2, false, false>), reinterpret_cast<const void*>(&tblock_pair_kernel<H, 0, true, true>),  \
      reinterpret_cast<const void*>(&tblock_pair_kernel<H, 1, true, true>), reinterpret_cast<const void*>(&tblock_pair_kernel<H, 1, false, true>),   \
      reinterpret_cast<const void*>(&tblock_pair_kernel<H, 2, true, true>), reinterpret_cast<const void*>(&tblock_pair_kernel<H, 2, false, true>)
  constexpr int NK = 21;             // [bf16: 5 output shapes x {plain, chunk state}] [fp16: the same] [bf16 with stamps]
  static const void* const kerns2[NK] = {VT_TB_KERNS(bf16_t), VT_TB_KERNS(f16_t), reinterpret_cast<const void*>(&tblock_pair_kernel<bf16_t, 2, true, false, true>)};
#undef VT_TB_KERNS
  const bool cached = a.replicate == 2 || a.cache1 != nullptr || a.cache2 != nullptr;
  int ki = a.ln_next == 0 ? 0 : (a.ln_next == 1 ? (a.keep_y ? 1 : 2) : (a.keep_y ? 3 : 4));
  int lds = T4_LDS;
  const unsigned threads = 512;
  if (cached) ki += 5;
  if (d->dtype == VT_F16) ki += 10;
  if (prof != nullptr) {                      // measurement aid: stamps [wave 0..7][step][8] of the bf16 LayerNorm+SiLU, y kept instantiation
    VT_CHECK_ARG(d->dtype == VT_BF16 && a.ln_next == 2 && a.keep_y && !cached, "vt_temporal_block_profile: bf16, ln_next_mode 2 and keep_y only, no chunk state");
    ki = 20;
    lds += 2048;
  }
  const void* kern = kerns2[ki];
  // per device, once: the dynamic-LDS attribute of every instantiation and the CU count (not a per-launch runtime call)
  static std::atomic<int> cus[kMaxDevices];
  int dev = 0;
  VT_CHECK_HIP(hipGetDevice(&dev));
  int ncu = (dev >= 0 && dev < kMaxDevices) ? cus[dev].load(std::memory_order_acquire) : 0;
  if (ncu == 0) {
    for (int k = 0; k < NK; ++k)
      VT_CHECK_HIP(hipFuncSetAttribute(kerns2[k], hipFuncAttributeMaxDynamicSharedMemorySize, k == 20 ? T4_LDS + 2048 : T4_LDS));
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
