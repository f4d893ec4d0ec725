// Single-head self-attention over the pixels of a frame without the S x S score matrix in memory:
//     O[q][:] = sum_k softmax_k(scale * Q[q] . K[k]) V[k][:] + bias_v,        q, k < S = H * W,  head dimension C = 512
// = F.scaled_dot_product_attention of AttnBlock / AttnBlockWrapper (reference model_3dcausal.py:129-141: one "head" per frame,
// q / k / v = 1x1x1 projections of the normalised frame).  The operator path (vt_conv as batched GEMM -> vt_softmax_rows -> vt_conv)
// writes the fp32 scores [Z][S][S] and the probabilities to HBM: 84 MB at 256 x 256 input, 1 GiB per latent frame at 1024 x 1024
// (VERDICT r3 #9).  Here a workgroup owns 64 query rows of one frame and walks the keys in tiles of 32 with the online softmax:
//   * 4 waves, wave w = query rows 16 w .. 16 w + 15, ALL 512 output dimensions: its O^T accumulators are 32 MFMA tiles 16 x 16
//     (128 registers), its Q rows 16 B-operand fragments that never leave the registers (64);
//   * swapped products so that nothing is transposed between the two GEMMs (v_mfma_f32_16x16x32_bf16, D[m][n] = sum_k A[m][k] B[k][n]):
//         S^T[key][q]  = sum_d K[key][d]   Q[q][d]        A = K rows from the LDS,   B = Q (registers)
//         O^T[dim][q] += sum_key V^T[dim][key] P^T[key][q] A = V^T rows from the LDS, B = P^T = exp(S^T - max) -- in the C layout of
//     the first product a lane holds column q = lane % 16 and the keys 4 (lane / 16) + i of each 16-key tile: the four values of two
//     tiles ARE a B fragment of the second product if contraction slot (g, e) is taken to mean key 16 (e / 4) + 4 g + e % 4 -- and the
//     A fragment of V^T then is two 8-byte reads (keys 4 g .. 4 g + 3 and 16 + 4 g .. + 3 of the row of a dimension);
//   * V^T [Z][C][S] (keys contiguous per dimension) is what the host already produces by swapping the operands of the v projection
//     (W_v as the row operand); its bias is added at the end (the probabilities of a row sum to 1);
//   * the statistics of a query row (running max, running sum) live in the lanes that hold the row: a reduction over a lane's 8
//     values and across the four lane groups (two wave shuffles), fp32;
//   * K / V^T tiles (32 KiB each) arrive by LDS-DMA into a 2-slot ring, swizzled on the source side so that the A-fragment reads of 16
//     rows hit 16 different bank groups; one barrier per key tile.
// FLOPs are negligible (0.1 % of the path); the point is the traffic: Q, K, V^T once per 64 rows, O once.
#include <atomic>

#include "conv_common.h"

namespace {

constexpr int FA_D = 512;            // head dimension = channels of the attention level (ch * ch_mult[-1] of every shipped config)
constexpr int FA_BQ = 64;            // query rows per workgroup
constexpr int FA_BK = 32;            // keys per tile
constexpr int FA_KTILE = FA_BK * FA_D * 2;        // 32 768 B: [32 keys][1 024 B]
constexpr int FA_VTILE = FA_D * FA_BK * 2;        // 32 768 B: [512 dims][64 B]
constexpr int FA_STAGE = FA_KTILE + FA_VTILE;
constexpr int FA_LDS = 2 * FA_STAGE;              // 131 072 B

struct FlashArgs {     // (uint16_t: elements of the launch's 16-bit storage type, bf16 or fp16)
  const uint16_t* q;    // [Z][S][512]
  const uint16_t* k;    // [Z][S][512]
  const uint16_t* vt;   // [Z][512][ldv]: V^T, keys contiguous
  const float* bias_v;  // [512] or null
  uint16_t* o;          // [Z][S][512]
  int S, ldv;
  float scale_log2e;    // scale * log2(e): exp(scale * s - m) = exp2(scale_log2e * s - m')
};

template <typename H>
__global__ __launch_bounds__(256, 1) void flash_attn_kernel(const FlashArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int z = blockIdx.y, q0 = blockIdx.x * FA_BQ + wave * 16;
  const int S = p.S;
  const int n = lane & 15, g = lane >> 4;            // my query row (column of the products), my contraction group
  const uint16_t* qz = p.q + (long long)z * S * FA_D;
  const uint16_t* kz = p.k + (long long)z * S * FA_D;
  const uint16_t* vz = p.vt + (long long)z * FA_D * p.ldv;

  // ---- Q: B fragments of the first product, k-step ks = dims 32 ks + 8 g .. + 8 of row q0 + n
  u32x4 qf[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) qf[ks] = *reinterpret_cast<const u32x4*>(qz + (long long)(q0 + n) * FA_D + ks * 32 + g * 8);

  // ---- tile DMA.  K tile: row r (key) = 1 024 B = 64 slots of 16 B, logical chunk c at slot c ^ (r % 16); instruction i of wave w
  // fills row 8 w + i (lane = slot).  V^T tile: row d (dimension) = 64 B = 4 slots, logical chunk c at slot c ^ ((d / 4) % 4);
  // instruction j of wave w fills rows 16 (8 w + j) .. + 15 (lane / 4 = row, lane % 4 = slot).
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(kz), 0, (unsigned)S * FA_D * 2u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(vz), 0, (unsigned)FA_D * (unsigned)p.ldv * 2u, 0x00020000);
  auto issue_tile = [&](int t, int stg) {
    char* kd = smem + stg * FA_STAGE;
    char* vd = kd + FA_KTILE;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = wave * 8 + i;
      const unsigned off = (unsigned)(t * FA_BK + r) * (FA_D * 2u) + (unsigned)((lane ^ (r & 15)) * 16);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_ptr_t)(kd + r * 1024), 16, off, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int d = 16 * (wave * 8 + j) + (lane >> 2);
      const int c = (lane & 3) ^ ((d >> 2) & 3);
      const unsigned off = (unsigned)d * (unsigned)p.ldv * 2u + (unsigned)(t * FA_BK * 2 + c * 16);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_ptr_t)(vd + 16 * (wave * 8 + j) * 64), 16, off, 0, 0, 0);
    }
  };

  f32x4 acc[32];                                      // O^T: tile mt = dims 16 mt + 4 g + i, column q
#pragma unroll
  for (int mt = 0; mt < 32; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -3.0e38f, l_run = 0.f;                // running max (in the exp2 domain) and sum of my query row

  const int ntile = S / FA_BK;
  issue_tile(0, 0);
  for (int t = 0; t < ntile; ++t) {
    const int stg = t & 1;
    wait_vmcnt<0>();
    __syncthreads();                                  // tile t has landed for everyone; everyone is done with tile t - 1
    if (t + 1 < ntile) issue_tile(t + 1, stg ^ 1);
    const char* kt = smem + stg * FA_STAGE;
    const char* vt = kt + FA_KTILE;
    // ---- S^T = K Q^T: two 16-key tiles, 16 k-steps of 32 dims; A fragment: key row 16 mt + n, logical chunk 4 ks + g
    f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int slot = ((4 * ks + g) ^ n) * 16;       // (row % 16 = n for both tiles)
      const u32x4 a0 = *reinterpret_cast<const u32x4*>(kt + n * 1024 + slot);
      const u32x4 a1 = *reinterpret_cast<const u32x4*>(kt + (16 + n) * 1024 + slot);
      s0 = h16<H>::mfma16(a0, qf[ks], s0);
      s1 = h16<H>::mfma16(a1, qf[ks], s1);
    }
    // ---- online softmax of my query row over the 32 keys of the tile (my 8 values + the other three lane groups')
    float v[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[i] = s0[i] * p.scale_log2e;
      v[4 + i] = s1[i] * p.scale_log2e;
    }
    float mx = v[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) mx = fmaxf(mx, v[i]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float ps = 0.f, pv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      pv[i] = __builtin_amdgcn_exp2f(v[i] - m_new);
      ps += pv[i];
    }
    ps += __shfl_xor(ps, 16, 64);
    ps += __shfl_xor(ps, 32, 64);
    l_run = l_run * alpha + ps;
    m_run = m_new;
    // P^T as the B fragment of the second product: contraction slot (g, e) = key 16 (e / 4) + 4 g + e % 4
    u32x4 pf;
    pf[0] = h16<H>::pack(pv[0], pv[1]);
    pf[1] = h16<H>::pack(pv[2], pv[3]);
    pf[2] = h16<H>::pack(pv[4], pv[5]);
    pf[3] = h16<H>::pack(pv[6], pv[7]);
    // ---- O^T = alpha O^T + V^T P^T: 32 tiles of 16 dims; A fragment of dims row d = 16 mt + n: keys 4 g .. + 3 (logical chunk g / 2,
    // half g % 2) and 16 + 4 g .. + 3 (logical chunk 2 + g / 2)
    // (once the running maxima have settled alpha is exactly 1 for every row of the wave: the 128 multiplications are skipped --
    // by 1.0f they would change nothing)
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0ull) {
#pragma unroll
      for (int mt = 0; mt < 32; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[mt][i] *= alpha;
    }
#pragma unroll
    for (int mt = 0; mt < 32; ++mt) {
      const int d = 16 * mt + n;
      const int sw = (d >> 2) & 3;
      const char* row = vt + d * 64 + 8 * (g & 1);
      const u32x2 lo = *reinterpret_cast<const u32x2*>(row + (((g >> 1) ^ sw) * 16));
      const u32x2 hi = *reinterpret_cast<const u32x2*>(row + (((2 + (g >> 1)) ^ sw) * 16));
      const u32x4 a = u32x4{lo[0], lo[1], hi[0], hi[1]};
      acc[mt] = h16<H>::mfma16(a, pf, acc[mt]);
    }
  }
  // ---- O[q][dims] = O^T / l + bias_v: a lane owns 4 consecutive dims of its row per tile (8-byte stores)
  const float inv = 1.0f / l_run;
  H* orow = reinterpret_cast<H*>(p.o) + ((long long)z * S + q0 + n) * FA_D;
#pragma unroll
  for (int mt = 0; mt < 32; ++mt) {
    const int d0 = 16 * mt + 4 * g;
    f32x4 b = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias_v) b = *reinterpret_cast<const f32x4*>(p.bias_v + d0);
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = acc[mt][i] * inv + b[i];
    store_quad<H>(orow + d0, o);
  }
#endif
}

}  // namespace

extern "C" int vt_flash_attention_supported(int32_t dtype, int32_t S, int32_t C, int32_t ldv) {
  return vt_opt(OPT_ATTN_FLASH) != 0 && vt_is_h16(dtype) && C == FA_D && S > 0 && S % FA_BQ == 0 && ldv >= S && ldv % 8 == 0 &&
         (long long)S * FA_D * 2 < 0xFFFF0000ll && (long long)FA_D * ldv * 2 < 0xFFFF0000ll;
}

extern "C" int vt_flash_attention(const void* q, const void* k, const void* vt, const float* bias_v, void* o, int32_t dtype, int32_t Z, int32_t S,
                                  int32_t C, int32_t ldv, float scale, vt_stream stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  VT_CHECK_ARG(q && k && vt && o && Z > 0, "vt_flash_attention: null tensor or empty batch");
  VT_CHECK_ARG(vt_opt(OPT_ATTN_FLASH) != 0, "vt_flash_attention: switched off (option attn_flash = 0)");
  VT_CHECK_ARG(vt_flash_attention_supported(dtype, S, C, ldv), "vt_flash_attention: bf16 / fp16, C = 512, S %% 64 == 0, ldv >= S and %% 8 == 0 only (got dtype %d S %d C %d ldv %d)",
               dtype, S, C, ldv);
  const uintptr_t al = reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(vt) | reinterpret_cast<uintptr_t>(o) |
                       reinterpret_cast<uintptr_t>(bias_v);
  VT_CHECK_ARG((al & 15) == 0, "vt_flash_attention: tensors must be 16-byte aligned");
  FlashArgs a;
  a.q = (const uint16_t*)q; a.k = (const uint16_t*)k; a.vt = (const uint16_t*)vt; a.bias_v = bias_v; a.o = (uint16_t*)o;
  a.S = S; a.ldv = ldv;
  a.scale_log2e = scale * 1.4426950408889634f;
  const void* kern = dtype == VT_F16 ? reinterpret_cast<const void*>(&flash_attn_kernel<f16_t>) : reinterpret_cast<const void*>(&flash_attn_kernel<bf16_t>);
  const int ki = dtype == VT_F16 ? 1 : 0;
  static std::atomic<bool> attr_done[2][kMaxDevices];
  int dev = 0;
  VT_CHECK_HIP(hipGetDevice(&dev));
  const bool dev_ok = dev >= 0 && dev < kMaxDevices;
  if (!dev_ok || !attr_done[ki][dev].load(std::memory_order_acquire)) {
    VT_CHECK_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, FA_LDS));
    if (dev_ok) attr_done[ki][dev].store(true, std::memory_order_release);
  }
  // the frame index rides in grid.y (65 535 at most): more frames than that go out as slices of the batch
  for (int z0 = 0; z0 < Z; z0 += 65535) {
    const int zn = Z - z0 < 65535 ? Z - z0 : 65535;
    FlashArgs s = a;
    s.q = a.q + (long long)z0 * S * FA_D; s.k = a.k + (long long)z0 * S * FA_D; s.o = a.o + (long long)z0 * S * FA_D;
    s.vt = a.vt + (long long)z0 * FA_D * ldv;
    void* kargs[] = {&s};
    VT_CHECK_HIP(hipLaunchKernel(kern, dim3((unsigned)(S / FA_BQ), (unsigned)zn), dim3(256), kargs, FA_LDS, stream));
  }
  return VT_OK;
}
