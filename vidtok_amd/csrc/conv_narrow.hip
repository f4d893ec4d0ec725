// 3x3x3 convolution with a handful of output channels: the decoder's conv_out (128 -> 3 RGB planes, CausalConv3d,
// reference model_3dcausal.py:862-870 with :162-197) at full resolution -- M = 5.2 M pixels, K = 27 x 128, N = 3.  On
// the implicit-GEMM kernel its narrowest tile is 32 channels wide: 10.7x the MACs of the problem, 1.9 ms per call.
//
// Here N carries (output channel, kw) instead: one 16x16x32 MFMA multiplies the 16 rows m = 4 co + kw (co < 4, kw < 3)
// of the weights of ONE (kt, kh) with 16 neighbouring input pixels of one image row,
//     Q[kt,kh][(co,kw)][column] += W[co][kt][kh][kw][:] . x[frame][row][column][:]
// and the three kw rows of a channel are combined at the very end, shifted by one column each (out[col] = Q[kw=0][col-1]
// + Q[kw=1][col] + Q[kw=2][col+1]).  A wave owns a window of 16 input columns (14 output columns) x 8 output rows and
// walks the frames of its time segment:
//   * the 36 weight fragments of all (kt, kh, k-step) sit in registers for the whole kernel (144 VGPRs);
//   * an input row of the window (16 px x 256 B) is fetched ONCE per frame by four LDS-DMA instructions into a
//     per-wave ring and read back as four B fragments; each fragment feeds the 9 MFMAs of all (kt, kh): frame tau
//     contributes to the output frames tau + pt - kt, image row r to the output rows r + 1 - kh -- 3 x 8 accumulators of
//     4 registers hold the three output frames in flight;
//   * when a frame has received its third time tap it is finished: shift-sum over kw (two 16-lane shuffles), + bias,
//     fp32 store straight into the NCTHW result (t_trim leading frames never computed); the slot is zeroed for the frame
//     three steps on;
//   * no barriers, no cross-wave traffic: waves are independent streams (DMA depth NW_DEPTH rows), one per SIMD.
// MACs executed: 16/9 (unused rows of the MFMA) x 16/14 (window halo) x 10/8 (row halo) of the problem's -- 2.5x, against
// 10.7x; the kernel is bound by the one read of x.  Zero padding in space, and every time mode of vt_conv (zeros /
// first frame repeated / cache frames before the clip, zeros after it), come from out-of-range DMA offsets (hardware
// zero fill) and a per-frame source pointer -- the arithmetic is the same for every pixel, so results do not depend
// on the tiling.
#include <atomic>
#include <type_traits>

#include "conv_select.h"

namespace {

[[maybe_unused]] constexpr int NW_TH = 8;                 // output rows of a wave tile
[[maybe_unused]] constexpr int NW_PH = NW_TH + 2;         // input rows
[[maybe_unused]] constexpr int NW_OW = 14;                // output columns of a wave tile (16 input columns)
// per arithmetic: bytes of one input row of the window in LDS (bf16: 4 fragments of 1 KiB; fp32 for the split-bf16 form: 8),
// ring of rows per wave, rows in flight ahead of the one being multiplied -- 128 KiB per workgroup (4 waves) either way
template <int MODE> struct NwCfg {
  [[maybe_unused]] static constexpr int NDMA = MODE == 0 ? 4 : 8;
  [[maybe_unused]] static constexpr int ROWB = NDMA * 1024;
  [[maybe_unused]] static constexpr int RING = MODE == 0 ? 8 : 4;
  [[maybe_unused]] static constexpr int DEPTH = MODE == 0 ? 5 : 2;
};
[[maybe_unused]] constexpr int NW_LDS = 4 * 8 * 4096;     // 131 072 B

struct NarrowArgs {
  const char* x;        // [B][Ti][H][W][128] bf16
  const char* cache;    // [B][ncache][H][W][128] bf16 or null
  const char* w;        // [Cout][ldw] bf16, k = ((kt*3+kh)*3+kw)*128 + c
  const float* bias;
  float* y;             // [B][Cout][To - t_trim][H][W] fp32
  int B, Ti, To, H, W, Cout, ldw;
  int pt, tmode, ncache, t_trim;
  int nseg, seg_len;    // output frames [t_trim, To) cut into nseg segments of seg_len (the last may be shorter)
  int hblks, wgrps;     // 8-row blocks, groups of 4 windows
};

template <int I, int N, typename F>
__device__ __forceinline__ void nw_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    nw_static_for<I + 1, N>(f);
  }
}

// MODE 0: bf16 tensors.  MODE 1 / 2: the split-bf16 arithmetic (VT_BF16X3: x fp32, w = [hi 16 | lo 16] bf16 planes per 16 k) as
// TWO passes over x, each with 36 stationary fragments like the bf16 kernel (both planes at once would be 288 registers of
// weights): pass 1 keeps the hi plane and takes W_hi x_lo + W_hi x_hi (+ bias) -> y, pass 2 keeps the lo plane and adds
// W_lo x_hi to y.  The activations are split in registers behind their fragment read (split3_x, conv_common.h).
// HT: the 16-bit type of the tensors (MODE 0: bf16_t / f16_t; the split planes of MODE 1 / 2 are bf16)
template <int MODE, typename HT = bf16_t>
__global__ __launch_bounds__(256, 1) void conv3d_narrow_kernel(const NarrowArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NW_ROWB = NwCfg<MODE>::ROWB, NW_RING = NwCfg<MODE>::RING, NW_DEPTH = NwCfg<MODE>::DEPTH, NDMA = NwCfg<MODE>::NDMA;
  constexpr int PIXB = MODE == 0 ? 256 : 512;          // bytes of a pixel's 128 channels
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int H = p.H, W = p.W;

  // workgroup -> (clip, time segment, row block, window group); neighbours in the sequence share halo rows / columns
  int slot = xcd_remap(blockIdx.x, gridDim.x);
  const int wg = slot % p.wgrps; slot /= p.wgrps;
  const int hb = slot % p.hblks; slot /= p.hblks;
  const int seg = slot % p.nseg;
  const int b = slot / p.nseg;
  const int c0 = (wg * 4 + wave) * NW_OW;            // first output column of this wave
  if (c0 >= W) return;                               // (no barriers in this kernel)
  const int h0 = hb * NW_TH;
  const int ob = p.t_trim + seg * p.seg_len;         // first output frame of the segment
  const int nq = min(p.seg_len, p.To - ob);          // output frames of the segment
  const int nj = nq + 2;                             // input frames (steps): tau = ob - pt + j

  // ---- stationary weights: A fragment (kt, kh, ks): row m = lane & 15 = 4 co + kw, k = 32 ks + 8 (lane >> 4) .. +8 --------
  u32x4 wf[36];
  {
    const int m = lane & 15, co = m >> 2, kw = m & 3;
    const bool ok = co < p.Cout && kw < 3;
    if constexpr (MODE == 0) {
      const HT* row = reinterpret_cast<const HT*>(p.w) + (long long)(ok ? co : 0) * p.ldw + (ok ? kw : 0) * 128 + (lane >> 4) * 8;
#pragma unroll
      for (int f = 0; f < 36; ++f) {                 // f = (kt*3 + kh)*4 + ks
        u32x4 v = *reinterpret_cast<const u32x4*>(row + (f >> 2) * 3 * 128 + (f & 3) * 32);
        if (!ok) v = u32x4{0u, 0u, 0u, 0u};
        wf[f] = v;
      }
    } else {
      // k = tap * 128 + 32 ks + 8 (lane >> 4): group of 16 = k / 16 at byte (k / 16) * 64 of the row, plane 32 B apart, 8 values = 16 B
      const char* row = p.w + (long long)(ok ? co : 0) * p.ldw * 4 + (MODE == 2 ? 32 : 0);
#pragma unroll
      for (int f = 0; f < 36; ++f) {
        const int k0 = ((f >> 2) * 3 + (ok ? kw : 0)) * 128 + (f & 3) * 32 + (lane >> 4) * 8;
        u32x4 v = *reinterpret_cast<const u32x4*>(row + (k0 >> 4) * 64 + (k0 & 15) * 2);
        if (!ok) v = u32x4{0u, 0u, 0u, 0u};
        wf[f] = v;
      }
    }
  }

  // ---- DMA lane constants: lane l fetches 16 B: pixel l >> 2 of the window, unit 4 ks + (l & 3) of its 16 ----------------
  constexpr unsigned kOob = 0xFFFF0000u;
  const unsigned frame_bytes = (unsigned)H * (unsigned)W * (unsigned)PIXB;   // <= 2^31 (launcher)
  const int dcol = c0 - 1 + (lane >> 2);
  const bool dcol_ok = dcol >= 0 && dcol < W;
  // bf16: unit 4 ks + (l & 3) = 16 B; fp32: the 8 values of (ks, l & 3) are 32 B, fetched as two 16-B halves
  const unsigned dlane = (unsigned)(dcol * PIXB + (lane & 3) * (MODE == 0 ? 16 : 32));
  char* ring = smem + wave * (NW_RING * NW_ROWB);
  // B fragment lane (g = lane >> 4, n = lane & 15): pixel n, unit 4 ks + g -> byte (4 n + g) * 16 of fragment ks
  const int rd = ((lane & 15) * 4 + (lane >> 4)) * 16;

  // source frame of step j: pointer + validity (wave-uniform); `live` = false: a row past the end of the segment
  // (issued only to keep the in-flight count constant) -- zero extent, every lane out of range, no memory access
  auto frame_rsrc = [&](int j, bool live) {
    const int tau = ob - p.pt + j;
    const bool pre = tau < 0;
    const bool from_cache = pre && p.tmode == VT_TPAD_CACHE;
    const bool ok = live && tau < p.Ti && (!pre || p.tmode != VT_TPAD_ZERO);
    const char* base = from_cache ? p.cache : p.x;
    const long long fidx = from_cache ? (long long)b * p.ncache + (p.ncache + tau) : (long long)b * p.Ti + (pre ? 0 : tau);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base) + (ok ? fidx : 0) * (long long)frame_bytes, 0, ok ? frame_bytes : 0u, 0x00020000);
  };
  // issue the four DMA instructions of row r of step j into ring row `rs`
  auto issue_row = [&](int j, int r, int rs, bool live) {
    __amdgpu_buffer_rsrc_t rsrc = frame_rsrc(j, live);
    const int hi = h0 - 1 + r;
    const bool ok = hi >= 0 && hi < H;
    const unsigned off = (ok && dcol_ok) ? (unsigned)(hi * W) * (unsigned)PIXB + dlane : kOob;
    char* dst = ring + rs * NW_ROWB;
    // k-step ks = +64 B (bf16) / +128 B (fp32; second half of a lane's 8 values + 16 B) in memory through the scalar offset (an
    // instruction offset would move the LDS side as well)
#pragma unroll
    for (int i = 0; i < NDMA; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dst + i * 1024), 16, off, MODE == 0 ? i * 64 : (i >> 1) * 128 + (i & 1) * 16, 0, 0);
  };

  // ---- accumulators: [slot = output frame mod 3][output row] -----------------------------------------------------------------
  f32x4 acc[3][NW_TH];
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int r = 0; r < NW_TH; ++r) acc[s][r] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int g = lane >> 4, n = lane & 15;
  const float bias = (p.bias != nullptr && g < p.Cout) ? p.bias[g] : 0.f;
  const int Tout = p.To - p.t_trim;
  const bool st_lane = g < p.Cout && n >= 1 && n <= NW_OW && (c0 - 1 + n) < W;
  float* ybase = p.y + (((long long)b * p.Cout + (g < p.Cout ? g : 0)) * Tout) * (long long)H * W + (c0 - 1 + n);

  // ---- prologue: NW_DEPTH rows in flight ------------------------------------------------------------------------------------
  int pj = 0, pr = 0, prs = 0;                       // next row to issue: step, row, ring row
  const int total_rows = nj * NW_PH;
  int issued = 0;
  auto issue_next = [&]() {
    issue_row(pj, pr, prs, issued < total_rows);
    ++issued;
    const bool wrap = pr + 1 == NW_PH;
    pr = wrap ? 0 : pr + 1;
    pj += wrap ? 1 : 0;
    prs = prs + 1 == NW_RING ? 0 : prs + 1;
  };
#pragma unroll
  for (int i = 0; i < NW_DEPTH; ++i) issue_next();

  // B fragments of two rows: the row being multiplied and the next one, read from the ring one row ahead of its use
  // (with one wave per SIMD nobody else covers an LDS round trip in front of the MFMAs)
  u32x4 bf[2][NDMA];                                  // bf16: fragment ks; fp32: the two halves of fragment ks at 2 ks, 2 ks + 1
  int crs = 0;                                        // ring row to read next
  auto read_row = [&](auto parc) {
    constexpr int par = decltype(parc)::value;
    const char* src = ring + crs * NW_ROWB + rd;
#pragma unroll
    for (int i = 0; i < NDMA; ++i) bf[par][i] = *reinterpret_cast<const u32x4*>(src + i * 1024);
    crs = crs + 1 == NW_RING ? 0 : crs + 1;
  };
  wait_vmcnt<NDMA * (NW_DEPTH - 1)>();
  read_row(std::integral_constant<int, 0>{});

  // one step = one input frame: 10 rows x 4 fragments x 9 MFMAs, then the frame that got its last tap is stored
  auto step = [&](auto jpc, int j) {
    constexpr int jp = decltype(jpc)::value;          // j mod 3
    nw_static_for<0, NW_PH>([&](auto rc) {
      constexpr int r = decltype(rc)::value;          // (NW_PH is even: the parity of a row is the parity of r)
      issue_next();
      wait_vmcnt<NDMA * (NW_DEPTH - 1)>();            // the next row has landed; later ones may still fly (stores only add)
      read_row(std::integral_constant<int, (r + 1) & 1>{});
      nw_static_for<0, 4>([&](auto ksc) {
        constexpr int ks = decltype(ksc)::value;
        u32x4 xhi, xlo;
        if constexpr (MODE == 0) xhi = bf[r & 1][ks];
        else split3_x(bf[r & 1][2 * ks], bf[r & 1][2 * ks + 1], xhi, xlo);
        nw_static_for<0, 3>([&](auto ktc) {
          constexpr int kt = decltype(ktc)::value;
          constexpr int s = (jp - kt + 3) % 3;        // output frame j - kt
          nw_static_for<0, 3>([&](auto khc) {
            constexpr int kh = decltype(khc)::value;
            constexpr int orow = r - kh;              // input row h0 - 1 + r is tap kh of output row h0 + r - kh
            if constexpr (orow >= 0 && orow < NW_TH) {
              if constexpr (MODE == 1)                // small term first: W_hi x_lo
                acc[s][orow] = h16<HT>::mfma16(wf[(kt * 3 + kh) * 4 + ks], xlo, acc[s][orow]);
              acc[s][orow] = h16<HT>::mfma16(wf[(kt * 3 + kh) * 4 + ks], xhi, acc[s][orow]);
            }
          });
        });
      });
    });
    // output frame q = j - 2 is complete (slot (jp + 1) % 3); q < 0: only the reset
    constexpr int fs = (jp + 1) % 3;
    const int q = j - 2;
    const bool st_frame = q >= 0;
    float* yf = ybase + (long long)(ob - p.t_trim + q) * H * W;
#pragma unroll
    for (int r = 0; r < NW_TH; ++r) {
      const f32x4 v = acc[fs][r];
      const float left = __shfl_up(v[0], 1, 16);      // Q[kw=0] of column - 1
      const float right = __shfl_down(v[2], 1, 16);   // Q[kw=2] of column + 1
      float o = __fadd_rn(__fadd_rn(left, v[1]), right);
      if constexpr (MODE != 2) o = __fadd_rn(o, bias);
      if (st_frame && st_lane && (h0 + r) < H) {
        if constexpr (MODE == 2) o = __fadd_rn(yf[(long long)(h0 + r) * W], o);     // second pass: onto the first one's result
        yf[(long long)(h0 + r) * W] = o;
      }
      acc[fs][r] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };

  for (int j0 = 0; j0 < nj; j0 += 3) {
    step(std::integral_constant<int, 0>{}, j0);
    if (j0 + 1 < nj) step(std::integral_constant<int, 1>{}, j0 + 1);
    if (j0 + 2 < nj) step(std::integral_constant<int, 2>{}, j0 + 2);
  }
  wait_vmcnt<0>();                                    // the dummy rows of the tail target this wave's LDS
#endif
}

}  // namespace

namespace {
// launch geometry of a qualifying vt_conv call (narrow_eligible in conv_igemm.hip)
long long narrow_geometry(const ConvArgs& a, NarrowArgs& n) {
  n.x = a.x; n.cache = a.cache; n.w = a.w; n.bias = a.bias; n.y = reinterpret_cast<float*>(a.y);
  n.B = a.B; n.Ti = a.Ti; n.To = a.To; n.H = a.Ho; n.W = a.Wo; n.Cout = a.Cout; n.ldw = a.ldw;
  n.pt = a.pt; n.tmode = a.tmode; n.ncache = a.ncache; n.t_trim = a.t_trim;
  n.hblks = (a.Ho + NW_TH - 1) / NW_TH;
  const int nwin = (a.Wo + NW_OW - 1) / NW_OW;
  n.wgrps = (nwin + 3) / 4;
  // time segments: enough independent waves to fill the SIMDs a few times over, segments of >= 6 output frames (each
  // segment reads two input frames of its predecessor again)
  const int nout = a.To - n.t_trim;
  const long long waves1 = (long long)a.B * n.hblks * nwin;
  int nseg = 1;
  while (waves1 * nseg < 4096 && nout / (nseg + 1) >= 6) ++nseg;
  n.seg_len = (nout + nseg - 1) / nseg;
  n.nseg = (nout + n.seg_len - 1) / n.seg_len;
  return (long long)a.B * n.nseg * n.hblks * n.wgrps;
}
}  // namespace

// {pixels per wave tile, MFMA rows, waves per workgroup, workgroups} for vt_conv_plan
extern "C" __attribute__((visibility("hidden"))) void vt_conv_narrow_plan(const void* args, int32_t* plan4) {
  NarrowArgs n;
  const long long grid = narrow_geometry(*reinterpret_cast<const ConvArgs*>(args), n);
  plan4[0] = NW_TH * NW_OW; plan4[1] = 16; plan4[2] = 4; plan4[3] = (int32_t)grid;
}

// conv_igemm.hip's dispatcher hands over launches that qualify; `args` is its ConvArgs
// mode: 0 bf16, 1 fp16, 2 split-bf16 (fp32 x, two passes)
extern "C" __attribute__((visibility("hidden"))) int vt_conv_narrow_launch(const void* args, void* stream_, int mode) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  NarrowArgs n;
  const long long grid = narrow_geometry(*reinterpret_cast<const ConvArgs*>(args), n);
  VT_CHECK_ARG(grid > 0 && grid < (1ll << 31), "vt_conv (narrow): grid");
  const void* kerns[4] = {reinterpret_cast<const void*>(&conv3d_narrow_kernel<0>), reinterpret_cast<const void*>(&conv3d_narrow_kernel<1>),
                          reinterpret_cast<const void*>(&conv3d_narrow_kernel<2>), reinterpret_cast<const void*>(&conv3d_narrow_kernel<0, f16_t>)};
  static std::atomic<bool> attr_done[kMaxDevices];
  int dev = 0;
  VT_CHECK_HIP(hipGetDevice(&dev));
  const bool dev_ok = dev >= 0 && dev < kMaxDevices;
  if (!dev_ok || !attr_done[dev].load(std::memory_order_acquire)) {
    for (const void* k : kerns) VT_CHECK_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, NW_LDS));
    if (dev_ok) attr_done[dev].store(true, std::memory_order_release);
  }
  void* kargs[] = {&n};
  if (mode != 2) {
    VT_CHECK_HIP(hipLaunchKernel(kerns[mode == 1 ? 3 : 0], dim3((unsigned)grid), dim3(256), kargs, NW_LDS, stream));
    return VT_OK;
  }
  // split-bf16: two passes over x (hi weight plane -> y, lo weight plane onto y), stream-ordered
  VT_CHECK_HIP(hipLaunchKernel(kerns[1], dim3((unsigned)grid), dim3(256), kargs, NW_LDS, stream));
  VT_CHECK_HIP(hipLaunchKernel(kerns[2], dim3((unsigned)grid), dim3(256), kargs, NW_LDS, stream));
  return VT_OK;
}
