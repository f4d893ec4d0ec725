// conv_in8_kernel: the encoder's conv_in (3 -> 128 channels, 3 x 3 x 3) in a 16-bit storage type.  Part of vt_conv (conv_igemm.hip
// dispatches; in8_eligible in conv_select.h says when).
#include <atomic>
#include <type_traits>

#include "conv_select.h"

namespace {

// ---- conv_in of the encoder: CausalConv3d 3 -> 128, 3 x 3 x 3 (model_3dcausal.py:568 with :162-197), in a 16-bit type ----------------------------------
// K = 27 taps x 8 stored channels = 216: on the general path of the kernel above (Cin = 8 is below a K-step row) a 128 x 128 tile takes
// four K steps, each a round trip of gathered 16-byte pieces, next to an epilogue that writes y and the consumer's LayerNorm -- 1.2 ms for
// 2.8 GB of output, 2.2 TB/s.  Here the same tile, the same MFMA sequence per accumulator and the same row arithmetic (the results are the
// general path's bit for bit: test_conv_in8_kernel_equals_general_path), with an operand path made for this shape:
//   * x: the tile's HALO PATCH -- 3 frames x (rows + 2) x (columns + 2) pixels of 16 bytes, <= 27 KiB -- arrives by LDS-DMA in ~27 wave
//     instructions (a patch row segment of 64 pixels each; padding = out-of-range offsets = hardware zero fill), once; a B fragment is then
//     ONE ds_read_b128 per lane (pixel lane % 32, tap 2 j + lane / 32) at base + a per-half constant -- no validity, no gather arithmetic;
//   * w: a wave keeps the 14 A fragments of its 32 channels (56 registers) in the accumulator half of the register file, loaded from the
//     L2-resident rows (55 KB) while the patch is in flight; 1 x 4 wave grid: 14 k-groups x 4 pixel fragments = 56 MFMAs per wave;
//   * the epilogue transposes through a 64-row buffer (32 KiB) in two passes, so patch + buffer leave room for two workgroups per CU.
// One tile per workgroup (the dispatcher overlaps one workgroup's loads with another's epilogue).  Zero / replicate causal padding (v1.0,
// v1.1 un-tiled and first chunks); cache mode (later chunks of a tiled pass) stays on the general path.  Option conv_in8.
// (Round-5 history: a first form loaded the fragments straight from memory, twelve 1-KB loads in flight per wave -- bit-equal and 14 % SLOWER
// than the general path, latency-bound; profiles/r05_conv_in8_ab.txt.)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

constexpr int kIn8TBytes = 64 * 128 * 4;      // the epilogue's 64-row transposition buffer

// conv_epilogue_lds128's row phase for a 1 x 4 wave grid (a wave: channels [32 wave, +32) of all 128 pixels), through a 64-row buffer in
// two passes (pixel fragments 0, 1 then 2, 3).  The transposition layout and the row arithmetic are conv_epilogue_lds128's, statement
// for statement: the bits must not depend on which kernel produced a pixel.
template <typename TOut>
__device__ __forceinline__ void conv_epilogue_in8(const ConvArgs& p, f32x16 (&acc)[4], int m_blk, int wave, int lane, int tid, char* smem) {
  TOut* __restrict__ yg = reinterpret_cast<TOut*>(p.y);
  TOut* __restrict__ ng = reinterpret_cast<TOut*>(p.ln_out);
  float* T = reinterpret_cast<float*>(smem);
  const int oct_j = tid & 15;
  const int row0 = tid >> 4;
  float lg[8], lb[8];
  if (p.ln_mode) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      lg[e] = ln_fold(p.ln_gamma[8 * oct_j + e], sizeof(TOut) == 2 && p.ln_mode == 2);     // 16-bit storage + SiLU: the affine carries -log2(e)
      lb[e] = ln_fold(p.ln_beta[8 * oct_j + e], sizeof(TOut) == 2 && p.ln_mode == 2);
    }
  }
  static_for<0, 2>([&](auto pc) __attribute__((always_inline)) {
    constexpr int pass = decltype(pc)::value;
    __syncthreads();                            // the patch reads / the previous pass's rows are done
    {
      const int h = lane >> 5;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = 32 * wave + 8 * g + 4 * h;               // first channel of the quad
        f32x4 bq;
        if (p.bias) bq = *reinterpret_cast<const f32x4*>(p.bias + c);
        else bq[0] = bq[1] = bq[2] = bq[3] = 0.0f;
#pragma unroll
        for (int bl = 0; bl < 2; ++bl) {
          const int prow = bl * 32 + (lane & 31);
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[2 * pass + bl][4 * g + e] + bq[e];
          *reinterpret_cast<f32x4*>(T + prow * 128 + (((c >> 2) ^ (prow & 31)) << 2)) = v;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = row0 + 16 * it;
      const int sw = row & 31;
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(T + row * 128 + (((2 * oct_j) ^ sw) << 2));
      const f32x4 t1 = *reinterpret_cast<const f32x4*>(T + row * 128 + (((2 * oct_j + 1) ^ sw) << 2));
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = e < 4 ? t0[e] : t1[e - 4];
      const long long orow = out_row(p, m_blk + 64 * pass + row);
      if (!p.ln_mode || p.ln_keep_y) Oct<TOut>::store(yg + orow * p.ldy + 8 * oct_j, v, p.nt_store != 0);
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
  });
}

template <typename H, bool REPL>
__global__ __launch_bounds__(256, 2) void conv_in8_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l32 = lane & 31;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m_blk = tile * 128;
  // ---- the tile's origin (uniform): frame (bb, t), first row h0, first column w0.  A tile is a 128-pixel segment of one row (Wo % 128
  // == 0) or 128 / Wo whole rows (launcher: Ho * Wo % 128 == 0)
  const unsigned fi = fast_div((unsigned)m_blk, p.fd_hw);
  const int rem = m_blk - (int)fi * (p.Ho * p.Wo);
  const int h0 = (int)fast_div((unsigned)rem, p.fd_wo);
  const int w0 = rem - h0 * p.Wo;
  const int bb = (int)fast_div(fi, p.fd_to);
  const int t = (int)fi - bb * p.To;
  const int CT = p.in8_ct, SEGS = p.in8_segs;
  const int PR = p.in8_rt + 2, RS = SEGS * 64;                 // patch rows per frame; LDS row stride in pixels
  char* patch = smem + kIn8TBytes;
  const int zslot = 3 * PR * RS * 16;                           // a 16-byte zero slot behind the patch: "tap 27"

  // ---- the halo patch by LDS-DMA: wave instruction q covers pixels [64 seg, 64 seg + 64) of patch row q / SEGS = (frame f, row r)
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
  const int nq = 3 * PR * SEGS;
  for (int q = wave; q < nq; q += 4) {
    const int prow = q / SEGS, seg = q - prow * SEGS;           // uniform
    const int f = prow / PR, r = prow - f * PR;
    const int tt = t + f - 2, hh = h0 + r - 1;
    const int c = seg * 64 + lane;
    const int ww = w0 + c - 1;
    const bool ok = (c < CT + 2) & ((unsigned)ww < (unsigned)p.Wi) & ((unsigned)hh < (unsigned)p.Hi) & (REPL | (tt >= 0));
    const unsigned off = ok ? (unsigned)(((bb * p.Ti + max(tt, 0)) * p.Hi + hh) * p.Wi + ww) * 16u : 0xFFFF0000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(patch + (prow * RS + seg * 64) * 16), 16, off, 0, 0, 0);
  }
  if (tid == 0) *reinterpret_cast<u32x4*>(patch + zslot) = u32x4{0u, 0u, 0u, 0u};

  // ---- stationary weights: A fragment j = channel 32 wave + l32, k = 16 j + 8 half .. + 8 of the packed row [Cout][216]
  u32x4 wf[14];
  {
    const H* row = reinterpret_cast<const H*>(p.w) + (long long)(32 * wave + l32) * p.ldw + 8 * half;
#pragma unroll
    for (int j = 0; j < 14; ++j) {
      if (j < 13 || half == 0) wf[j] = *reinterpret_cast<const u32x4*>(row + 16 * j);
      else wf[j] = u32x4{0u, 0u, 0u, 0u};                        // k 216 .. 223: beyond the 27 taps
    }
  }
  // ---- fragment addresses: pixel b * 32 + l32 of the tile = (row pr, column pc) of the tile's rectangle -> patch (pr + kh, pc + kw)
  int fb[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int pix = b * 32 + l32;
    const int pr = pix >> p.in8_ctl2, pc = pix & (CT - 1);
    fb[b] = (pr * RS + pc) * 16;
  }
  wait_vmcnt<0>();                                               // my patch pieces and weights have landed
  __syncthreads();                                               // everybody's have

  f32x16 acc[4];
  constexpr int PF = 2;                                          // k-groups of fragment reads in flight
  u32x4 xf[PF + 1][4];
  auto issue = [&](auto jc) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;
    constexpr int tpa = 2 * j, tpb = 2 * j + 1;
    const int da = (((tpa / 9) * PR + (tpa % 9) / 3) * RS + tpa % 3) * 16;
    const int db = tpb < 27 ? (((tpb / 9) * PR + (tpb % 9) / 3) * RS + tpb % 3) * 16 : 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      int a = fb[b] + (half ? db : da);
      if (tpb >= 27) a = half ? zslot : a;
      xf[j % (PF + 1)][b] = *reinterpret_cast<const u32x4*>(patch + a);
    }
  };
  static_for<0, PF>([&](auto jc) __attribute__((always_inline)) { issue(jc); });
  static_for<0, 14>([&](auto jc) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;
    if constexpr (j + PF < 14) issue(std::integral_constant<int, j + PF>{});
    // every accumulator takes its 16-k groups in K order, as in the general path (the sum of an output is the same chain).  Inline asm
    // with weights AND accumulators pinned to the accumulator half of the register file ("a"): left to the allocator they land next
    // to fragments and the epilogue's rows and spill
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      // ("=&a": the result may not share registers with the fragments -- the compiler's own MFMA definitions carry the same early-clobber)
      if constexpr (std::is_same<H, bf16_t>::value) {
        if constexpr (j == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&a"(acc[b]) : "a"(wf[j]), "v"(xf[j % (PF + 1)][b]));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[b]) : "a"(wf[j]), "v"(xf[j % (PF + 1)][b]));
      } else {
        if constexpr (j == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&a"(acc[b]) : "a"(wf[j]), "v"(xf[j % (PF + 1)][b]));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[b]) : "a"(wf[j]), "v"(xf[j % (PF + 1)][b]));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  });
  asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");              // last MFMA -> first reader of its accumulator
  conv_epilogue_in8<H>(p, acc, m_blk, wave, lane, tid, smem);
#endif
}


}  // namespace

// conv_igemm.hip's dispatcher hands over launches that qualify (in8_eligible, conv_select.h); `args` is its ConvArgs
extern "C" __attribute__((visibility("hidden"))) int vt_conv_in8_launch(const void* args, int dtype, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ConvArgs a = *reinterpret_cast<const ConvArgs*>(args);
  a.x_bytes = (unsigned)((unsigned long long)a.B * a.Ti * a.Hi * a.Wi * 8 * 2);
  a.in8_ct = a.Wo >= 128 ? 128 : a.Wo;
  a.in8_rt = 128 / a.in8_ct;
  a.in8_segs = (a.in8_ct + 2 + 63) / 64;
  a.in8_ctl2 = 0;
  while ((1 << a.in8_ctl2) < a.in8_ct) ++a.in8_ctl2;
  const int tiles = a.M / 128;
  const int lds = kIn8TBytes + 3 * (a.in8_rt + 2) * a.in8_segs * 64 * 16 + 16;
  static const void* const kerns[4] = {reinterpret_cast<const void*>(&conv_in8_kernel<bf16_t, false>), reinterpret_cast<const void*>(&conv_in8_kernel<bf16_t, true>),
                                       reinterpret_cast<const void*>(&conv_in8_kernel<f16_t, false>), reinterpret_cast<const void*>(&conv_in8_kernel<f16_t, true>)};
  const int ki = (dtype == VT_F16 ? 2 : 0) + (a.tmode == VT_TPAD_REPLICATE ? 1 : 0);
  static std::atomic<bool> attr_done[4][kMaxDevices];
  int dev = 0;
  VT_CHECK_HIP(hipGetDevice(&dev));
  const bool dev_ok = dev >= 0 && dev < kMaxDevices;
  if (!dev_ok || !attr_done[ki][dev].load(std::memory_order_acquire)) {
    VT_CHECK_HIP(hipFuncSetAttribute(kerns[ki], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (dev_ok) attr_done[ki][dev].store(true, std::memory_order_release);
  }
  void* kargs[] = {&a};
  VT_CHECK_HIP(hipLaunchKernel(kerns[ki], dim3((unsigned)tiles), dim3(256), kargs, lds, stream));
  return VT_OK;
}
