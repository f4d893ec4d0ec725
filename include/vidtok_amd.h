/*
 * vidtok_amd.h -- C-ABI of libvidtok_amd.so: the MI355X (gfx950) kernels behind the VidTok
 * causal tokenizer's encode / decode path.
 *
 * The reference (microsoft/VidTok) has no FFI of its own: every operator on the hot path is a
 * torch.nn call.  Each entry point below replaces the ATen operator(s)
 * the reference reaches from the cited file:line, so a maintainer binds them with ctypes
 * (INTEGRATION.md shows the stub) from the Python modules that mirror the reference classes.
 *
 * Conventions
 *   - plain C: pointers are DEVICE pointers unless said otherwise, sizes are element counts;
 *   - every function returns 0 on success, a negative vt_status otherwise; vt_last_error()
 *     returns a thread-local message; nothing throws across the boundary;
 *   - the library never allocates or frees activation memory, never synchronises the stream
 *     and is hipGraph-capture safe (all launches go to the stream argument);
 *   - activation layout is NDHWC ([B][T][H][W][C], C innermost, C stored padded to `ld*`
 *     elements); the public tensors of the reference API are NCTHW fp32 and are converted at
 *     the two ends by vt_ncthw_to_ndhwc / the NCTHW epilogue of vt_conv;
 *   - `dtype` selects the arithmetic: VT_F32 = fp32 storage + fp32-input MFMA
 *     (v_mfma_f32_32x32x2_f32, bit-wise an fmaf chain), VT_BF16 = bf16 storage + bf16 MFMA
 *     (v_mfma_f32_32x32x16_bf16) with fp32 accumulation, VT_F16 = fp16 storage + fp16 MFMA
 *     (v_mfma_f32_32x32x16_f16, the same rate) with fp32 accumulation: what the reference's
 *     README computes in under torch.autocast(device_type="cuda", dtype=torch.float16)
 *     (README.md:336-340,375-385; scripts/inference_*.py --precision autocast).  Results beyond
 *     fp16's range (|v| > 65 504) become +-inf when a tensor is stored, as they do in the
 *     reference's fp16 convolutions.  Every operator that takes VT_BF16 takes VT_F16.
 *     Statistics, softmax, the regularizers and all epilogue arithmetic are fp32 in every mode.
 *     vt_conv additionally takes
 *     VT_BF16X3 ("split-bf16"): fp32 storage (x, cache, y, res, ln_out exactly as for VT_F32,
 *     out_dtype = VT_F32) with the products taken on the bf16 matrix cores from bf16 hi / lo
 *     planes of both operands (x_hi w_hi + x_hi w_lo + x_lo w_hi, fp32 accumulation: ~2^-17
 *     relative per product against 2^-9 for VT_BF16, at 3/16 of the matrix-pipe time of
 *     VT_F32).  Activations are split inside the kernel; the WEIGHTS are handed over already
 *     split: row n of `w` holds, per group of 16 consecutive k, 16 bf16 hi values followed by
 *     16 bf16 lo values (hi = bf16_rne(w), lo = bf16_rne(w - hi)) -- 64 bytes, the size of the
 *     16 fp32 values they replace; rows are zero-padded to whole 128-byte K steps, so ldw (in
 *     4-byte units) = K rounded up to 32
 *     (vidtok_amd/packing.py::pack_split3).  Every other operator of a split-bf16 pass is the
 *     VT_F32 one.
 */
#ifndef VIDTOK_AMD_H
#define VIDTOK_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vt_stream;          /* hipStream_t */

typedef enum vt_status {
  VT_OK = 0,
  VT_ERR_ARG = -1,                /* bad / unsupported argument (message in vt_last_error) */
  VT_ERR_HIP = -2,                /* a HIP runtime call failed */
  VT_ERR_UNSUPPORTED = -3
} vt_status;

typedef enum vt_dtype { VT_F32 = 0, VT_BF16 = 1, VT_I32 = 2, VT_BF16X3 = 3 /* vt_conv arithmetic only, see above */, VT_F16 = 4 } vt_dtype;

/* time-axis treatment of taps that fall before the first frame of the input */
typedef enum vt_tmode {
  VT_TPAD_ZERO = 0,       /* v1.0 CausalConv{1,3}d, pad_mode "constant"  (model_3dcausal.py:156-159,193-197) */
  VT_TPAD_REPLICATE = 1,  /* v1.1 first chunk: first frame repeated        (model_3dcausal_v1_1.py:160-163,217-220) */
  VT_TPAD_CACHE = 2,      /* v1.1 later chunks: last frames of the cache   (model_3dcausal_v1_1.py:164-171,221-228) */
  VT_TPAD_ZERO_BACK = 3   /* vt_time_avgpool3s2 only: one zero frame AFTER the sequence, non-causal TimeDownsampleRes2x (model_3dnoncausal.py:86-89) */
} vt_tmode;

typedef enum vt_resmode {
  VT_RES_NONE = 0,
  VT_RES_ADD = 1,         /* y = conv + bias + res          (ResnetBlock x + h, model_3dcausal.py:337,424,499,118) */
  VT_RES_MIX = 2          /* y = a*res + (1-a)*(conv+bias), a = sigmoid(*mix_factor)
                             (TimeDown/UpsampleResCausal2x, model_3dcausal.py:248-252,268-273) */
} vt_resmode;

typedef enum vt_layout { VT_NDHWC = 0, VT_NCTHW = 1 } vt_layout;

const char* vt_last_error(void);
/* returns a version integer; also a cheap "is the library loadable" probe */
int vt_version(void);
/* hipGraphLaunch of an instantiated graph (hipGraphExec_t) on `stream`, nothing else: how the Python host replays the encoder /
 * decoder launch sequences it captured (vidtok_amd/graphs.py) -- a framework's own replay also launches bookkeeping kernels */
int vt_graph_launch(void* graph_exec, vt_stream stream);
/* number of bytes of dynamic LDS the biggest conv variant asks for (diagnostic) */
int vt_conv_max_lds_bytes(void);

/* ------------------------------------------------------------------------------------------
 * Process-wide tuning / test switches.  Which kernel a call runs is a function of its descriptor and of this
 * table only -- no launch path reads the environment.  The table is filled once, on first use, from the
 * environment (variable VT_<NAME>, e.g. VT_CONV_WS=0, so shell A/B runs work) and changes afterwards only
 * through vt_set_option; vt_reset_options() re-reads the environment defaults.  Every option selects between
 * implementations of the SAME operator contract (the parity tests run both sides of each switch), so none of them
 * is part of the reference's interface.  Names (default):
 *   conv_buf (1)        gather through buffer descriptors; 0 = 64-bit pointers (always used for > 4 GiB tensors, and in cache mode when a tile spans frames)
 *   conv_tinner (1)     temporal convolutions walk their tiles frames-innermost (L2 reuse of the kt taps)
 *   conv_ldsepi (1)     128 x 128 tile: epilogue transposed through the LDS (whole-line stores, carries the fused LayerNorm)
 *   conv_ws (2)         weight-stationary persistent kernel (conv_ws2.hip: two waves per SIMD splitting K) for the 3x3 128 -> 128
 *                       convolutions in a 16-bit type; 0 = off (the tile-per-workgroup kernel)
 *   conv_narrow (1)     conv3d_narrow_kernel for Cout <= 4 (the decoder's conv_out)
 *   conv_tile (0)       128 / 256: force that tile wherever it is legal; 0 = choose by size
 *   conv_tile_min (128) fewest 256 x 256 tiles for which the 8-wave tile is chosen
 *   conv_fuse_ln (1), conv_fuse_ln256 (1)   LayerNorm of the result inside the epilogue for Cout = 128 / 256
 *   conv_deep (1)       128 x 128 tile on a 4-slot ring (three K steps of DMA in flight) for launches with no more tiles than CUs
 *   conv_splitk (1)     split-K over the tap planes for launches with few pixels PER CLIP when the caller provides scratch
 *                       (vt_conv_work_bytes).  A split launch sums in another order than a whole one, so the decision is a function
 *                       of one clip's geometry (To, Ho, Wo, Cout, K) and never of B: a clip's bits do not depend on its batch.
 *                       + 2-3 % on a v1.1 tiled pass, neutral on the B = 4 benchmark step (profiles/r05_tskip_splitk_ab.txt)
 *   conv_tskip (1)      causal zero padding in time (tmode VT_TPAD_ZERO): a tile that lies inside one output frame starts its K walk
 *                       behind the tap planes that read only the zero frames in front of the clip (frames 0 / 1 of a 3-tap convolution
 *                       run a third / two thirds of the K steps); the skipped products are exact zeros: same bits as the full walk
 *   conv_in8 (1)        conv_in8_kernel for the encoder's conv_in (bf16, 3 x 3 x 3, 8 stored input channels -> 128, zero / replicate time
 *                       padding, frames whose width divides or is a multiple of 128): the tile's halo patch arrives by LDS-DMA once, a
 *                       fragment is one ds_read, the weights are register-stationary, the epilogue transposes through a 64-row buffer.
 *                       The same bits as the general path of the implicit-GEMM kernel (0), 16 % faster on the launch
 *   conv_tup_ln (1)     the LayerNorm the consumer of a v1.0 time up-sampler starts with is emitted by the up-sampler's two parity launches
 *                       (alpha-mix + interleaved output frames + LayerNorm together in the bf16 LDS epilogue of the 8-wave tile) instead of
 *                       running as its own pass (- 0.2 ... 0.3 ms of the benchmark step); hosts ask vt_conv_plan whether a launch fuses it; 0 = the separate pass.
 *                       (Gates the Cout = 256 epilogue only: the Cout = 128 LayerNorm epilogue takes an alpha-mix / interleaved output under conv_fuse_ln alone.)
 *   conv_nt_mb (64)     outputs (y and the fused LayerNorm) of at least this many MiB are written by the LDS epilogues with streaming (nt)
 *                       stores: rows nobody reads before they have left every cache do not displace the weight slabs and halo rows the next tiles
 *                       read again (whole benchmark step - 0.7 %; the fused temporal block and conv3x3_ws2 always store this way); 0 = plain stores
 *   attn_flash (1)      the attention block as one vt_flash_attention launch where it applies; 0: GEMM -> softmax -> GEMM operators
 *   tblock_prof_mode (0), ws_prof_mode (0)   measurement aids
 * (Rounds 1-5 also kept the superseded forms selectable -- K-step schedules 0 / 1 / 3 / 4, the first LayerNorm epilogue of the 8-wave
 * tile, the first-generation weight-stationary kernel, 128 x 256 half tiles: their measurements are in DESIGN.md section 6, the code is
 * in the history.)
 *   tblock_fused (1)   0: vt_temporal_block_supported answers no (blocks stay on the unfused operators)
 * Returns VT_ERR_ARG for an unknown name.
 * ---------------------------------------------------------------------------------------- */
int vt_set_option(const char* name, int32_t value);
int vt_get_option(const char* name, int32_t* value);
int vt_reset_options(void);
int vt_option_count(void);
const char* vt_option_name(int32_t i);    /* i < vt_option_count(), else NULL */

/* ------------------------------------------------------------------------------------------
 * vt_conv -- implicit-GEMM convolution on NDHWC, M = B*To*Ho*Wo pixels x N = Cout x K = taps*Cin.
 * One entry point covers every convolution of the path:
 *   nn.Conv2d 3x3 s1 p1      ResnetBlock.conv1/conv2            model_3dcausal.py:296,301
 *   nn.Conv2d 3x3 s1 p1 after nearest x2   Upsample               model_3dcausal.py:208-212  (ups_s=1)
 *   nn.Conv2d 3x3 s2 p0 after F.pad(0,1,0,1)  Downsample          model_3dcausal.py:223-227
 *   nn.Conv2d 1x1            nin_shortcut                       model_3dcausal.py:306
 *   nn.Conv1d k3 causal      CausalConv1d                       model_3dcausal.py:144-159
 *   nn.Conv3d 3x3x3 causal   CausalConv3d (stride 1 or (2,1,1)) model_3dcausal.py:162-197
 *   nn.Conv3d 1x1x1          AttnBlockWrapper q/k/v/proj_out    model_3dcausal.py:124-127
 *   nearest x2 in T folded   TimeUpsampleResCausal2x            model_3dcausal.py:267-273  (ups_t=1)
 * and, with KT=KH=KW=1 and nbatch>1, the batched GEMMs of the attention block
 * (F.scaled_dot_product_attention, model_3dcausal.py:140).
 *
 * Tap (kt,kh,kw) of output (b,to,ho,wo) reads the *virtual* input position
 *   tv = to*st + kt - pt,  hv = ho*sh + kh - ph,  wv = wo*sw + kw - pw
 * in a virtual input of size (Ti<<ups_t, Hi<<ups_s, Wi<<ups_s); the stored element is
 * (tv>>ups_t, hv>>ups_s, wv>>ups_s).  Spatial positions outside the virtual extent read 0;
 * tv<0 follows `tmode` (cache: frame ncache+tv of `cache`, shape [B][ncache][Hi][Wi][Cin]);
 * tv beyond the end reads 0.
 * Weights are pre-packed row-major [Cout][ldw], k = ((kt*KH+kh)*KW+kw)*Cin + c (see
 * vidtok_amd/packing.py), in the arithmetic dtype.
 * yt_mul = 2 serves the convolutions over a nearest-x2 frame-repeated input (TimeUpsampleRes*2x): with u[t] = x[t>>1]
 * the three temporal taps of an output frame hit only two input frames, so even and odd output frames are two 2-tap
 * convolutions over x with pre-summed weights -- 2/3 of the MACs of the folded ups_t form, no up-sampled tensor
 * either way (vidtok_amd/packing.py::time_upsample_parity_weights).
 * ys_mul = 2 is the same identity in space (Upsample: nearest x2 then 3x3): each of the four output parities (py, px)
 * is a 2x2 convolution over the stored frame with pre-summed taps, 4/9 of the MACs
 * (vidtok_amd/packing.py::space_upsample_parity_weights).
 * ln_mode != 0 additionally emits ln_out = [SiLU](LayerNorm_Cout(result) * gamma + beta), the norm that follows the
 * convolution inside the residual blocks (norm2 + nonlinearity after conv1, model_3dcausal.py:321-323,405-407,
 * 482-484).  When the tile spans the channel row (Cout = 128) the statistics are taken from the fp32 result inside
 * the epilogue and, with ln_keep_y = 0, y is never written: one activation write instead of write + read + write.
 * Otherwise the library runs the convolution into y and vt_layernorm_act on it -- same contract, y must always be
 * a full-size buffer.  NDHWC output, nbatch = 1 only.
 * ---------------------------------------------------------------------------------------- */
typedef struct vt_conv_desc {
  const void* x;            /* input  [B][Ti][Hi][Wi][Cin]     (dtype)                        */
  const void* w;            /* packed weights [Cout][ldw]       (dtype)                        */
  const float* bias;        /* [Cout] fp32 or NULL                                             */
  void* y;                  /* output, out_dtype; NDHWC [M][ldy] or NCTHW (see out_layout)     */
  const void* res;          /* residual / mix operand, out_dtype, [B][Tr][Ho][Wo][ldr] or NULL */
  const void* cache;        /* time cache [B][ncache][Hi][Wi][Cin] (dtype) or NULL             */
  const float* mix_factor;  /* device scalar; alpha = sigmoid(*mix_factor) for VT_RES_MIX      */
  const float* ln_gamma;    /* fused LayerNorm of the result (ln_mode != 0): affine, fp32 [Cout]  */
  const float* ln_beta;
  void* ln_out;             /* normalised (+SiLU) result, out_dtype, NDHWC [M][ldn]            */
  int32_t B, Ti, Hi, Wi, Cin;
  int32_t To, Ho, Wo, Cout;
  int32_t ldw;              /* row stride of w in elements (>= KT*KH*KW*Cin, multiple of 16 B) */
  int32_t ldy;              /* channel stride of y for NDHWC                                   */
  int32_t KT, KH, KW;
  int32_t st, sh, sw;
  int32_t pt, ph, pw;
  int32_t tmode, ncache;
  int32_t ups_t, ups_s;
  int32_t res_mode;         /* vt_resmode                                                      */
  int32_t res_tshift;       /* res time index = to >> res_tshift                               */
  int32_t Tr, ldr;
  int32_t out_layout;       /* vt_layout; VT_NCTHW needs out_dtype == VT_F32                   */
  int32_t t_trim;           /* NCTHW only: drop the first t_trim output frames                 */
  int32_t dtype;            /* arithmetic / input / weight dtype                               */
  int32_t out_dtype;        /* dtype or VT_F32                                                 */
  int32_t nbatch;           /* >=1: independent problems along grid.z                          */
  int32_t ln_mode;          /* 0 none, 1 LayerNorm over Cout, 2 LayerNorm + SiLU -> ln_out     */
  int32_t ln_keep_y;        /* 0: only ln_out is needed; y is then scratch (may stay unwritten) */
  int32_t ldn;              /* channel stride of ln_out                                        */
  int32_t yt_mul, yt_off;   /* output frame interleave: computed frame f (b*To + to) is stored as frame
                               f*yt_mul + yt_off of a y (and ln_out) holding To*yt_mul frames; 0/1 = off  */
  int32_t ys_mul, ys_oh, ys_ow; /* ys_mul = 2: computed pixel (ho, wo) is stored as pixel (2ho+ys_oh, 2wo+ys_ow) of
                               a y whose frames are 2Ho x 2Wo (spatial parity classes, see below); 0/1 = off     */
  float ln_eps;
  int64_t xs_z, ws_z, ys_z, rs_z;   /* element strides between problems                       */
  void* work;                  /* optional scratch for split-K launches (see vt_conv_work_bytes); NULL = never split */
  int64_t work_bytes;
} vt_conv_desc;

int vt_conv(const vt_conv_desc* d, vt_stream stream);
/* Split-K over tap planes (the three time taps; the three rows of a 3 x 3 without time taps).  A long-K convolution on few pixels (the 512-channel 3x3x3 layers on a chunk of a v1.1 tiled
 * pass: M = 4 096 pixels x N = 512 x K = 13 824 = 128 tiles of 128 x 128 with 216 K steps each on 256 CUs) is latency-bound: one tile per
 * CU walking a long K.  Given scratch, vt_conv runs it as three launches-in-one (grid.z = tap plane: each workgroup walks the KH x KW x Cin
 * -- or KW x Cin -- of ONE plane into an fp32 partial) and a reduction that owns bias / residual / rounding: 3 x the workgroups, a third of the steps.
 * vt_conv_work_bytes(d) = the scratch vt_conv(d) would use (0: this call does not split; decided from ONE clip's geometry in the
 * descriptor -- never from B -- and option conv_splitk, ignoring d->work); with d->work = NULL or d->work_bytes too small the call runs unsplit.  The fp32 sum of an output
 * is then taken in another order (per tap, then over the taps): results differ from the unsplit launch by fp32 rounding. */
int64_t vt_conv_work_bytes(const vt_conv_desc* d);
/* What vt_conv(d) would do, without launching (no GPU needed): out8 = {pixel tile, channel tile, waves per
 * workgroup, workgroups (tiles for the persistent kernel), 1 if LayerNorm comes from the conv epilogue, kernel
 * launches of the call, kernel (0 = tile-per-workgroup implicit GEMM; weight-stationary persistent 3x3 for
 * Cin = Cout = 128 bf16: 1 = conv_ws128.hip, 8 x 16-pixel tiles on 4 waves, 3 = conv_ws2.hip, 4 x 16-pixel tiles on 8 waves; 2 = the
 * narrow-output kernel; 4 = conv_in8_kernel, the encoder's conv_in from an LDS halo patch), 1 if the 8-wave tile's epilogue goes through the LDS (coalesced rows; always with a fused LayerNorm,
 * without one for bf16 full tiles), 2 if the 128 x 128 tile runs on its 4-slot ring (option conv_deep: launches with no
 * more tiles than the device has CUs), 3 if such an LDS-epilogue launch runs as 128 x 256 half tiles on 4 waves, two workgroups
 * per CU (option conv_half256)}.  Validates `d` exactly like vt_conv.  Test / measurement aid: which kernel and
 * instantiation does a parity case exercise. */
int vt_conv_plan(const vt_conv_desc* d, int32_t* out8);
/* Measurement aid (scripts/conv_profile.py): vt_conv(d) on the bf16 8-wave 256 x 256 tile (no LayerNorm) with shader-clock
 * stamps at the phase boundaries of K steps 8..11 of workgroup 0: stamps_out (device, 8*4*8 uint64) = [wave][step][stamp] */
int vt_conv_profile(const vt_conv_desc* d, uint64_t* stamps_out, vt_stream stream);
/* sizeof(vt_conv_desc) as compiled: lets a binding verify its struct mirror */
int vt_conv_desc_size(void);

/* ------------------------------------------------------------------------------------------
 * vt_temporal_block -- one fused launch for the temporal residual block of the reference,
 * ResnetCausalBlock1D._forward (vidtok/modules/model_3dcausal.py:473-499 with CausalConv1d :144-159 and, for the
 * first-frame-replicate padding of v1.1, model_3dcausal_v1_1.py:159-178):
 *     y = x + conv2( SiLU(LN2( conv1( SiLU(LN1(x)) ) )) )       conv = causal Conv1d over T, k = 3, C -> C
 * on an NDHWC activation x [B][T][HW][ld]; optionally also n_out = [SiLU](LayerNorm_next(y)), the norm the consumer
 * of y starts with (same contract as vt_conv's ln_mode / ln_keep_y).  LayerNorms are per position over C, eps as
 * given, statistics in fp32.  w1 / w2 are packed like vt_conv weights: [C][3*C], k = kt*C + c.
 * tmode: VT_TPAD_ZERO (frames before the clip are zeros), VT_TPAD_REPLICATE (first frame repeated) or VT_TPAD_CACHE
 * (the two frames a previous chunk left: v1.1 tiling, CausalConv1d.causal_cache of model_3dcausal_v1_1.py:159-178).
 * The cached tensors are the INPUTS of the two convolutions -- SiLU(LN1(x)) and SiLU(LN2(conv1)), which a fused launch
 * never writes out -- so the launch keeps them itself: cache1 / cache2 [B][2][HW][ld] are read in cache mode and, when
 * given (any tmode), rewritten in place with frames T - cache_offset - 2, T - cache_offset - 1 of this clip (the
 * reference's padded[:len - cache_offset][-2:]); T - cache_offset >= 3 is required.
 * Covered shapes: bf16, C = ld = 128, HW % 64 == 0 (the widest level of every 488 / 41616 / 288 config);
 * vt_temporal_block_supported(d) says so without launching (1 / 0); anything else returns VT_ERR_ARG.
 * ------------------------------------------------------------------------------------------ */
typedef struct vt_tblock_desc {
  const void* x;                 /* [B][T][HW][ld]                                            */
  void* y;                       /* same shape; may be NULL when keep_y == 0                   */
  void* n_out;                   /* same shape; LayerNorm_next(y) (+SiLU), or NULL             */
  const void* w1; const float* b1;   /* conv1: [C][3C] packed, bias [C] fp32 or NULL           */
  const void* w2; const float* b2;   /* conv2                                                   */
  const float* norm1_gamma; const float* norm1_beta;   /* [C] fp32                              */
  const float* norm2_gamma; const float* norm2_beta;
  const float* next_gamma; const float* next_beta;     /* used when ln_next_mode != 0           */
  int32_t dtype;                 /* VT_BF16 or VT_F16                                          */
  int32_t C, ld;
  int32_t B, T;
  int64_t HW;
  int32_t tmode;                 /* VT_TPAD_ZERO | VT_TPAD_REPLICATE | VT_TPAD_CACHE           */
  int32_t keep_y;                /* 0: only n_out is needed (y is never written)               */
  int32_t ln_next_mode;          /* 0 none, 1 LayerNorm, 2 LayerNorm + SiLU                    */
  float eps;
  int32_t cache_offset;          /* frames at the end of the clip that the kept chunk state skips */
  void* cache1;                  /* [B][2][HW][ld] chunk state of conv1 (see above), or NULL    */
  void* cache2;                  /* ... of conv2                                               */
} vt_tblock_desc;

/* sizeof(vt_tblock_desc) as compiled: lets a binding verify its struct mirror */
int vt_tblock_desc_size(void);
int vt_temporal_block_supported(const vt_tblock_desc* d);
int vt_temporal_block(const vt_tblock_desc* d, vt_stream stream);
/* Measurement aid (scripts/tblock_profile.py): the same launch (ln_next_mode 2, keep_y 1 only) with shader-clock
 * stamps at the phase boundaries of four steps of workgroup 0: stamps_out (device, 8*4*8 uint64) = [wave][step][stamp] */
int vt_temporal_block_profile(const vt_tblock_desc* d, uint64_t* stamps_out, vt_stream stream);

/* ------------------------------------------------------------------------------------------
 * vt_layernorm_act -- per-position LayerNorm over C (eps inside the sqrt, biased variance,
 * affine) optionally followed by SiLU x*sigmoid(x).
 * Replaces LayerNorm wrapper + nonlinearity: model_3dcausal.py:62-80, 26-27 (every norm site of
 * ResnetBlock / ResnetCausalBlock / ResnetCausalBlock1D / AttnBlock / norm_out).
 * x: [M][ldx] in `in_dtype`, y: [M][ldy] in `out_dtype`; gamma, beta fp32 [C].
 * ---------------------------------------------------------------------------------------- */
int vt_layernorm_act(const void* x, int in_dtype, int64_t ldx, void* y, int out_dtype, int64_t ldy,
                     const float* gamma, const float* beta, int64_t M, int32_t C, float eps,
                     int32_t silu, vt_stream stream);

/* ------------------------------------------------------------------------------------------
 * vt_groupnorm_act -- torch.nn.GroupNorm(32, C, eps, affine) optionally followed by SiLU: the `norm_type: groupnorm`
 * branch of Normalize() (model_3dcausal.py:30-34; no shipped config selects it).  GroupNorm normalises over
 * (C/groups, *spatial) of the view its call site passes, so the reduction domain is a parameter:
 *   VT_GN_FRAME  spatial ResnetBlock on "(b t) c h w"  (model_3dcausal.py:14-19, 317-337)  per (b,t,g) over (C/G,H,W)
 *   VT_GN_PIXEL  temporal blocks on "(b h w) c t"       (model_3dcausal.py:20-23, 473-499)  per (b,h,w,g) over (C/G,T)
 *   VT_GN_CLIP   3-D blocks, attention norm, norm_out on "b c t h w"                          per (b,g) over (C/G,T,H,W)
 * x [B][T][HW][ldx], y [B][T][HW][ldy]; gamma, beta fp32 [C]; work: vt_groupnorm_work_bytes() bytes (fp64 sums).
 * ---------------------------------------------------------------------------------------- */
typedef enum { VT_GN_FRAME = 0, VT_GN_PIXEL = 1, VT_GN_CLIP = 2 } vt_gn_scope;
int64_t vt_groupnorm_work_bytes(int32_t B, int32_t T, int32_t groups, int32_t scope);
int vt_groupnorm_act(const void* x, int in_dtype, int64_t ldx, void* y, int out_dtype, int64_t ldy,
                     const float* gamma, const float* beta, int32_t B, int32_t T, int64_t HW, int32_t C,
                     int32_t groups, int32_t scope, float eps, int32_t silu, void* work, vt_stream stream);

/* Single-head self-attention of the pixels of a frame in ONE launch, nothing S x S in memory (F.scaled_dot_product_attention
 * of AttnBlock, model_3dcausal.py:129-141): o[z][q][:] = sum_k softmax_k(scale * q[z][q] . k[z][k]) v[z][k][:] + bias_v.
 * q, k, o: [Z][S][C]; vt: V TRANSPOSED, [Z][C][ldv] with the keys contiguous (the v projection with its operands swapped, as the
 * hosts already compute it; its bias goes in as bias_v, fp32 [C] or NULL).  bf16, C = 512, S a multiple of 64
 * (vt_flash_attention_supported says whether a shape is covered and option attn_flash is on; otherwise the same contract is
 * vt_conv as batched GEMM -> vt_softmax_rows -> vt_conv).  Online softmax in fp32; P is rounded to bf16 before the second
 * product, un-normalised, and the division by the row sum comes last. */
int vt_flash_attention_supported(int32_t dtype, int32_t S, int32_t C, int32_t ldv);
int vt_flash_attention(const void* q, const void* k, const void* vt, const float* bias_v, void* o, int32_t dtype, int32_t Z, int32_t S,
                       int32_t C, int32_t ldv, float scale, vt_stream stream);

/* row softmax(scale * s) over the last dim; s fp32 [rows][cols] -> p (out_dtype) [rows][ldp].
 * The softmax inside F.scaled_dot_product_attention (model_3dcausal.py:140), scale = C^-0.5. */
/* x = tanh(x) in place on n fp32 elements: the decoders' `tanh_out` option (model_3dcausal.py:866-869) */
int vt_tanh_inplace(float* x, int64_t n, vt_stream stream);
int vt_softmax_rows(const float* s, void* p, int out_dtype, int64_t rows, int32_t cols,
                    int64_t ldp, float scale, vt_stream stream);

/* ------------------------------------------------------------------------------------------
 * layout conversion at the two ends of the path
 * vt_ncthw_to_ndhwc: x fp32 NCTHW [B][C][T][H][W] -> y (out_dtype) [B][tpad+T][H][W][ldy],
 *   channels C..ldy-1 zero-filled, the first `tpad` frames replicate frame 0
 *   (EncoderCausal3DPadding.forward pad_at_dim(..., "replicate"), model_3dcausal.py:685-689,
 *   v1.1 model_3dcausal_v1_1.py:755-760).
 * vt_ndhwc_to_ncthw: x (in_dtype) [B][T][H][W][ldx] -> y fp32 [B][C][T-ttrim][H][W]
 *   (DecoderCausal3DPadding.forward trim, model_3dcausal.py:883-885).
 * ---------------------------------------------------------------------------------------- */
int vt_ncthw_to_ndhwc(const float* x, void* y, int out_dtype, int32_t B, int32_t C, int32_t T,
                      int32_t H, int32_t W, int32_t ldy, int32_t tpad, vt_stream stream);
int vt_ndhwc_to_ncthw(const void* x, int in_dtype, float* y, int32_t B, int32_t C, int32_t T,
                      int32_t H, int32_t W, int32_t ldx, int32_t ttrim, vt_stream stream);

/* ------------------------------------------------------------------------------------------
 * time resamplers (the non-conv branches)
 * vt_time_avgpool3s2: y[to] = (xp[2to]+xp[2to+1]+xp[2to+2])/3 over the front-padded sequence
 *   xp = [pad, x]; pad frame = 0 (tmode ZERO, v1.0 model_3dcausal.py:249-250), x[0]
 *   (REPLICATE, v1.1 first chunk) or `cache` (one frame [B][1][H][W][C], v1.1 later chunks,
 *   model_3dcausal_v1_1.py:293-300); VT_TPAD_ZERO_BACK: xp = [x, 0] instead (non-causal family,
 *   model_3dnoncausal.py:86-89).  x [B][Ti][HW][C] -> y [B][Ti/2][HW][C] (floor: an odd Ti loses its
 *   last frame, as avg_pool3d does), same dtype.
 * vt_time_lerp2x: F.interpolate(scale (2,1,1), "trilinear", align_corners=False) along T of a
 *   sequence of Ti frames -> 2*Ti frames (model_3dcausal_v1_1.py:327-341); fp32 arithmetic.
 * vt_time_lerp2x_cat: the same of the sequence [head (nh frames, [B][nh][HWC]) | x ([B][Tx][HWC])] per clip without
 *   assembling it, leaving out the first `skip` output frames: y [B][2 (nh + Tx) - skip][HWC].  The chunks after the
 *   first of a v1.1 tiled pass: head = the frames cached from the previous chunk, skip = 2 * num_temp_upsample
 *   (torch.cat + interpolate + slice, model_3dcausal_v1_1.py:331-341).  nh = 0, skip = 0 is vt_time_lerp2x.
 * ---------------------------------------------------------------------------------------- */
int vt_time_avgpool3s2(const void* x, const void* cache, void* y, int dtype, int32_t B, int32_t Ti,
                       int64_t HW, int32_t C, int32_t tmode, vt_stream stream);
int vt_time_lerp2x(const void* x, void* y, int dtype, int32_t B, int32_t Ti, int64_t HWC,
                   vt_stream stream);
int vt_time_lerp2x_cat(const void* head, int32_t nh, const void* x, void* y, int dtype, int32_t B, int32_t Tx, int32_t skip,
                       int64_t HWC, vt_stream stream);

/* ------------------------------------------------------------------------------------------
 * vt_pack_conv_weight -- a convolution parameter in the reference's layout, w fp32 [Cout][Cin][taps_in] (nn.Conv3d / Conv2d /
 * Conv1d weight, taps = kT*kH*kW row-major), to the rows vt_conv reads: out [Cout][ldw], k = tap * cin_p + c,
 * channels zero-padded to cin_p (the activation's stored count), the row tail zero.  out_dtype VT_F32 / VT_BF16 / VT_F16 = plain rows
 * (round to nearest even); VT_BF16X3 = the split-bf16 container: per 16 k [hi 16 x bf16 | lo 16 x bf16], hi = bf16(w),
 * lo = bf16(w - hi), 4 bytes per k, ldw a multiple of 32.  mix_host (host, [taps_out][4], -1 = absent; NULL = identity): output
 * tap j = (w[m0] + w[m1]) + (w[m2] + w[m3]) in fp32 -- the pre-summed taps of the up-samplers' parity classes (a 3-tap window
 * over a nearest-x2 repeated input touches two inputs: [W0 + W1, W2] / [W0, W1 + W2] per axis).  One-time work per weight. */
int vt_pack_conv_weight(const float* w, void* out, int32_t out_dtype, int32_t Cout, int32_t Cin, int32_t cin_p, int32_t taps_in,
                        int32_t taps_out, const int32_t* mix_host, int64_t ldw, vt_stream stream);

/* ------------------------------------------------------------------------------------------
 * regularizers; all tensors NCTHW fp32 as in the reference API
 * vt_kl_sample: h [B][2*zc][S] (S = T*H*W); mean,logvar = chunk(h,2,dim=1); logvar clamped to
 *   [-30,20]; z = mean + exp(0.5*logvar)*noise (noise [B][zc][S] or NULL -> mode());
 *   *kl_out = 0.5*sum(mean^2 + var - 1 - logvar) / B     (distributions.py:8-28,
 *   regularizers.py:82-92; the reference draws `noise` on the host with torch.randn).
 * vt_fsq_quantize: h [B][D][S], levels[D]; bound -> round-half-even -> codes z [B][D][S]
 *   (values q/(L//2)) and int32 indices [B][S] (regularizers.py:153-178, 225-230).
 * vt_fsq_indices_to_codes: indices [B][S] -> z [B][D][S] (regularizers.py:180-198).
 * vt_fsq_aux_stats: the two entropies and the commitment term of FSQRegularizer.forward
 *   (regularizers.py:231-246) for h [B][D][S]: out[0] = per-sample entropy (mean), out[1] =
 *   codebook entropy of the batch-mean distribution, out[2] = commit loss.  `work` is a device
 *   scratch of vt_fsq_aux_work_floats(...) floats.
 * vt_fsq_aux_stats_avg: the same, and (avg_out != NULL) the batch-mean code distribution avg_prob[prod(levels)] the
 *   codebook entropy is taken of -- the tensor the reference averages across ranks when torch.distributed runs
 *   with world > 1 (maybe_distributed_mean, regularizers.py:49-59,240); vt_entropy(avg, J, out) then recomputes
 *   out[0] = sum_j -avg_j * log(max(avg_j, 1e-5)) (regularizers.py:40-45) of the all-reduced distribution.
 * ---------------------------------------------------------------------------------------- */
/* host-only helper: writes half_l[D], offset[D], shift[D], basis[D] (as floats) exactly as the
 * FSQ kernels use them (FSQRegularizer.bound constants, regularizers.py:153-158) -- no GPU needed. */
int vt_fsq_consts(const int32_t* levels_host, int32_t D, float* out_host);
int vt_kl_sample(const float* h, const float* noise, float* z, float* kl_out, int32_t B,
                 int32_t zc, int64_t S, vt_stream stream);
int vt_fsq_quantize(const float* h, float* z, int32_t* indices, const int32_t* levels_host,
                    int32_t D, int32_t B, int64_t S, vt_stream stream);
int vt_fsq_indices_to_codes(const int32_t* indices, float* z, const int32_t* levels_host,
                            int32_t D, int32_t B, int64_t S, vt_stream stream);
/* The same with `ncb` codebooks (FSQRegularizer num_codebooks, "b n (c d) -> b n c d", regularizers.py:131-133,227): h, z
 * [B][ncb * D][S] -- the D channels of codebook c of a clip are contiguous --, each codebook quantised on its own, and the
 * codebook axis LAST on the indices as the reference keeps it: indices [B][S][ncb].  ncb = 1 is the plain form. */
int vt_fsq_quantize_cb(const float* h, float* z, int32_t* indices, const int32_t* levels_host,
                       int32_t D, int32_t B, int32_t ncb, int64_t S, vt_stream stream);
int vt_fsq_indices_to_codes_cb(const int32_t* indices, float* z, const int32_t* levels_host,
                               int32_t D, int32_t B, int32_t ncb, int64_t S, vt_stream stream);
int64_t vt_fsq_aux_work_floats(const int32_t* levels_host, int32_t D, int32_t B, int64_t S);
int vt_fsq_aux_stats(const float* h, const int32_t* levels_host, int32_t D, int32_t B, int64_t S,
                     float inv_temperature, float* work, float* out3, vt_stream stream);
int vt_fsq_aux_stats_avg(const float* h, const int32_t* levels_host, int32_t D, int32_t B, int64_t S,
                         float inv_temperature, float* work, float* out3, float* avg_out, vt_stream stream);
/* out[0] = (stats3[0] - diversity_gamma * ce) * entropy_weight + stats3[2] * commitment_weight, ce = codebook_entropy[0] or
 * (NULL) stats3[1]: the auxiliary loss FSQRegularizer.forward returns (regularizers.py:241,264-266), every product / sum
 * rounded on its own like the reference's separate tensor operations */
int vt_fsq_aux_loss(const float* stats3, const float* codebook_entropy, float diversity_gamma, float entropy_weight,
                    float commitment_weight, float* out, vt_stream stream);
int vt_entropy(const float* avg, int64_t J, float* out, vt_stream stream);

/* ------------------------------------------------------------------------------------------
 * Model handle: AutoencodingEngine.encode / decode of the causal (v1.0, v1.1) and non-causal tokenizers driven from C++ over the operators above
 * (reference vidtok/models/autoencoder.py:197-229: encode = encoder -> regularization, decode = decoder, forward = both;
 * the module tree of vidtok/modules/model_3dcausal.py:502-885 with `norm_type: layernorm`, `resamp_with_conv: true`).
 * Same stage graph, descriptors and fusion decisions as the Python host (vidtok_amd/modules.py), hence the same bits.
 *   vt_create(cfg, VT_BF16 | VT_F16 | VT_F32 | VT_BF16X3, &h)   the fields of cfg are the constructor arguments of the reference's YAML;
 *                                            VT_BF16X3 = fp32 storage with every convolution in split-bf16 arithmetic
 *   vt_load_weight(h, key, data, shape, n)   key = the reference state_dict key ("encoder.down.0.block.0.conv1.weight",
 *                                            ...), data = fp32 on the HOST in the reference's parameter layout; weights
 *                                            are re-packed when first used.  vt_weight_count / vt_weight_name /
 *                                            vt_weight_shape list the parameters; unknown keys and wrong shapes are refused.
 *   vt_workspace_bytes(h, B, T, H, W)        device bytes vt_encode of a [B][in_channels][T][H][W] clip and vt_decode of
 *                                            its latent need (activations; the handle owns its packed weights)
 *   vt_latent_dims(h, T, H, W, out4)         {channels of the encoder output, T', H', W'}
 *   vt_encode(h, x, B, T, H, W, h_out, ws, ws_bytes, stream)     x fp32 NCTHW -> h_out fp32 [B][C'][T'][H'][W'] (the
 *                                            Gaussian moments for KL, the pre-quantisation latent for FSQ)
 *   vt_regularize_kl(h, moments, noise | NULL, z, kl_out, B, T', H', W', stream)     = vt_kl_sample
 *   vt_regularize_fsq(h, pre, z, indices, B, T', H', W', stream)                     = [project_in ->] vt_fsq_quantize_cb [-> project_out]
 *   vt_indices_to_latent(h, indices, z, B, T', H', W', stream)                       = vt_fsq_indices_to_codes_cb [-> project_out]
 *                                            (indices int32 [B][T'][H'][W'], with fsq_num_codebooks = c > 1: [B][T'][H'][W'][c]; with
 *                                            projections the handle keeps a scratch of the projected latent, grown on first use)
 *   vt_decode(h, z, B, T', H', W', x_out, ws, ws_bytes, stream)  z fp32 NCTHW -> x_out fp32 [B][out_ch][T][H][W]
 *   vt_regularize_fsq_aux(h, pre, B, T', H', W', inv_temperature, work, out3, stream)   the three statistics of FSQ's auxiliary
 *                                            loss = vt_fsq_aux_stats with the handle's levels
 *   v1.1 (version 1): the encoder pads T up to a multiple of time_downsample_factor in front and vt_decode returns all
 *   T' * factor frames -- the caller keeps the last T, as AutoencodingEngine.forward does (autoencoder_v1_1.py:339-341).
 * Temporal tiling of the v1.1 models (AutoencodingEngine.tile_encode / tile_decode, autoencoder_v1_1.py:218-331): the clip
 * runs as chunks [0,1), [1,1+c), [1+c,1+2c) ... through the same graph, every causal convolution and time resampler
 * keeping its last frames in device buffers owned by the handle (the reference's `causal_cache` attributes):
 *   vt_tile_latent_frames(h, T, t_chunk_enc)      T' of a tiled encode = sum over the chunks of ceil(frames / factor)
 *   vt_tile_workspace_bytes(h, B, T, H, W, t_chunk_enc, use_overlap)   workspace of vt_tile_encode + vt_tile_decode
 *   vt_tile_encode(h, x, B, T, H, W, t_chunk_enc, h_out, ws, ws_bytes, stream)     h_out fp32 [B][C'][T'][H'][W']; the
 *                                            regularizer then runs on the whole latent (it is per position)
 *   vt_tile_decode(h, z, B, T', H', W', t_chunk_dec, use_overlap, x_out, ws, ws_bytes, stream)   t_chunk_dec = t_chunk_enc /
 *                                            factor; use_overlap = one look-ahead latent frame per chunk whose output frames
 *                                            are dropped, with the doubling cache offsets of the reference (:307-320);
 *                                            x_out fp32 [B][out_ch][T' * factor][H][W]
 *   vt_reset_cache(h)                        drops the chunk state and frees its buffers (synchronises the device); the
 *                                            tiled calls reset the state themselves at the start of a clip
 *   vt_prepare(h)                            packs and uploads every weight now instead of at its first use
 * All device pointers; the calls are asynchronous on `stream` except that (a) the FIRST use of a weight packs it on the host
 * and uploads it with a blocking copy (vt_prepare does all of them up front) and (b) a chunk cache is hipMalloc'ed the first time a chunk kind needs it -- run one
 * warm-up call before capturing a stream.  A handle keeps per-call state (arenas, caches): one call at a time per handle.
 * The workspace must outlive the work queued on it.
 * ---------------------------------------------------------------------------------------- */
typedef struct vt_model_config {
  int32_t version;               /* 0 = v1.0 causal, 1 = v1.1 causal (replicate padding; one pass or temporal tiling), 2 = non-causal */
  int32_t ch, num_res_blocks, in_channels, out_ch, z_channels, double_z;
  int32_t num_resolutions;       /* len(ch_mult)                                                                */
  int32_t ch_mult[8];
  int32_t n_spatial_ds, spatial_ds[8], n_tempo_ds, tempo_ds[8];     /* encoder levels that end in a down-sampler */
  int32_t n_spatial_us, spatial_us[8], n_tempo_us, tempo_us[8];     /* decoder levels that end in an up-sampler  */
  int32_t time_downsample_factor;
  int32_t regularizer;           /* 0 DiagonalGaussianRegularizer, 1 FSQRegularizer                              */
  int32_t n_levels, levels[8];
  int32_t interpolation_mode;    /* v1.1 time up-samplers: 0 nearest, 1 trilinear                                 */
  /* the constructor arguments no shipped YAML sets (zero = the reference's default):                             */
  int32_t norm_type;             /* 0 "layernorm", 1 "groupnorm": torch.nn.GroupNorm(32, C, eps=1e-6) at every norm site, its
                                    statistics over the view the reference's call site passes (model_3dcausal.py:30-34;
                                    parameters at "<norm>.weight" / ".bias", no ".norm" level); never fused into an epilogue */
  int32_t fsq_num_codebooks;     /* FSQRegularizer num_codebooks (0 or 1 = one): indices carry the codebook axis last,
                                    [B][T'][H'][W'][c] (regularizers.py:131-133)                                  */
  int32_t fsq_dim;               /* FSQRegularizer dim (0 = len(levels) * num_codebooks): when it differs, z_channels = dim and the
                                    quantiser sits between project_in / project_out (nn.Linear along the channel axis,
                                    "regularization.project_in.weight" ..., regularizers.py:137-139,225,255)       */
} vt_model_config;
typedef struct vt_model vt_model;
int vt_model_config_size(void);    /* sizeof(vt_model_config) as compiled: lets a binding verify its struct mirror */
int vt_create(const vt_model_config* cfg, int32_t compute_dtype, vt_model** out);
int vt_destroy(vt_model* h);
int vt_load_weight(vt_model* h, const char* ref_key, const float* data_host, const int64_t* shape, int32_t ndim);
int vt_weight_count(vt_model* h);
const char* vt_weight_name(vt_model* h, int32_t i);
int vt_weight_shape(vt_model* h, int32_t i, int64_t* shape5, int32_t* ndim);   /* the shape vt_load_weight expects for key i */
int64_t vt_workspace_bytes(vt_model* h, int32_t B, int32_t T, int32_t H, int32_t W);
int vt_latent_dims(const vt_model* h, int32_t T, int32_t H, int32_t W, int32_t* out4);
int vt_encode(vt_model* h, const float* x, int32_t B, int32_t T, int32_t H, int32_t W, float* h_out, void* workspace,
              int64_t workspace_bytes, vt_stream stream);
int vt_regularize_kl(vt_model* h, const float* moments, const float* noise, float* z, float* kl_out, int32_t B, int32_t Tz,
                     int32_t Hz, int32_t Wz, vt_stream stream);
int vt_regularize_fsq(vt_model* h, const float* pre, float* z, int32_t* indices, int32_t B, int32_t Tz, int32_t Hz, int32_t Wz,
                      vt_stream stream);
int vt_indices_to_latent(vt_model* h, const int32_t* indices, float* z, int32_t B, int32_t Tz, int32_t Hz, int32_t Wz,
                         vt_stream stream);    /* FSQ: decode(indices, decode_from_indices=True) = this, then vt_decode */
int vt_decode(vt_model* h, const float* z, int32_t B, int32_t Tz, int32_t Hz, int32_t Wz, float* x_out, void* workspace,
              int64_t workspace_bytes, vt_stream stream);
int vt_reset_cache(vt_model* h);
int vt_prepare(vt_model* h);       /* pack + upload every weight now (blocking) instead of at first use */
int vt_regularize_fsq_aux(vt_model* h, const float* pre, int32_t B, int32_t Tz, int32_t Hz, int32_t Wz, float inv_temperature,
                          float* work, float* out3, vt_stream stream);
int32_t vt_tile_latent_frames(const vt_model* h, int32_t T, int32_t t_chunk_enc);
int64_t vt_tile_workspace_bytes(vt_model* h, int32_t B, int32_t T, int32_t H, int32_t W, int32_t t_chunk_enc, int32_t use_overlap);
int vt_tile_encode(vt_model* h, const float* x, int32_t B, int32_t T, int32_t H, int32_t W, int32_t t_chunk_enc, float* h_out,
                   void* workspace, int64_t workspace_bytes, vt_stream stream);
int vt_tile_decode(vt_model* h, const float* z, int32_t B, int32_t Tz, int32_t Hz, int32_t Wz, int32_t t_chunk_dec, int32_t use_overlap,
                   float* x_out, void* workspace, int64_t workspace_bytes, vt_stream stream);

/* ------------------------------------------------------------------------------------------
 * Video front / back end around model(x): the device halves of scripts/inference_reconstruct.py (the codec --
 * decord / torchvision.io.write_video -- stays with the caller).  B = 1 like the reference's DataLoader.
 * vt_frames_u8_to_ncthw: decoded frames uint8 [T][H0][W0][3] -> x fp32 [3][Tdst][H][W], frames t_off .. t_off+T:
 *   v/255, torchvision Resize(size, antialias=True) to (Hr, Wr) (= ATen's separable anti-aliased bilinear filter,
 *   horizontal pass then vertical pass in fp32; Hr == H0 and Wr == W0 is the identity), CenterCrop window
 *   [top, top+H) x [left, left+W) of the resized frame, Normalize(0.5, 0.5) = (v-0.5)/0.5
 *   (inference_reconstruct.py:39-45,70-73; vidtok/data/vidtok.py:180-188).  `work`: vt_frames_work_floats(T, H0, W) floats.
 * vt_ncthw_to_frames_u8: x fp32 [3][Tsrc][H][W] frames t0 .. t0+n -> out uint8 [n][H][Wtot][3] at column w_off:
 *   clamp(-1,1), (x+1)/2, *255, truncation (tensor_to_uint8, inference_reconstruct.py:76-80; w_off / Wtot place the
 *   input next to the reconstruction for --concate_input, :228-235).
 * vt_ncthw_copy_frames: dst[c][td0+k] = (clamp ? clamp(src[c][ts0+k], -1, 1) : src[c][ts0+k]), k < n, for C channels of
 *   [C][T*][HW] tensors: the --pad_gen_frames chaining (last f-1 generated frames prepended to the next clip, :213-221).
 * ------------------------------------------------------------------------------------------ */
int64_t vt_frames_work_floats(int32_t T, int32_t H0, int32_t W);
int vt_frames_u8_to_ncthw(const uint8_t* frames, int32_t T, int32_t H0, int32_t W0, int32_t Hr, int32_t Wr, int32_t top,
                          int32_t left, float* x, int32_t Tdst, int32_t t_off, int32_t H, int32_t W, float* work,
                          vt_stream stream);
int vt_ncthw_to_frames_u8(const float* x, int32_t Tsrc, int32_t t0, int32_t n, int32_t H, int32_t W, uint8_t* out,
                          int32_t Wtot, int32_t w_off, vt_stream stream);
int vt_ncthw_copy_frames(const float* src, float* dst, int32_t C, int32_t Ts, int32_t Td, int32_t ts0, int32_t td0,
                         int32_t n, int64_t HW, int32_t clamp, vt_stream stream);

/* copy `n` frames of `frame_elems` elements each: dst frame j <- src frame idx_host[j]
 * (cache maintenance of the v1.1 chunked path, model_3dcausal_v1_1.py:172-176,230-234);
 * src/dst are [B][Ts|Td][frame_elems] with the given batch strides, element size `esize`. */
int vt_gather_frames(const void* src, void* dst, int32_t esize, int32_t B, int64_t frame_elems,
                     int64_t src_bstride, int64_t dst_bstride, const int32_t* idx_host, int32_t n,
                     vt_stream stream);

/* nn.Linear along the channel axis of an NCTHW fp32 tensor viewed as [B][Cin][S] -> [B][Cout][S]:
 * FSQRegularizer.project_in / project_out when dim != len(levels) (vidtok/modules/regularizers.py:137-139,225,255).
 * w [Cout][Cin] row-major, bias [Cout] or NULL. */
int vt_channel_linear(const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t Cin,
                      int32_t Cout, int64_t S, vt_stream stream);

/* ------------------------------------------------------------------------------------------
 * evaluation metrics of the reference's eval loop: per frame
 *   out = clamp(y,-1,1); x,out -> (.+1)/2                          scripts/inference_evaluate.py:175-176
 *   psnr = -10 log10(mean_{c,h,w}(x-out)^2 + 1e-8)                  vidtok/modules/util.py:146-154
 *   ssim = mean_c mean_{h,w} SSIM map, 11x11 Gaussian sigma 1.5      vidtok/modules/util.py:157-222
 * x, y NCTHW fp32 [B][C][T][H][W]; psnr, ssim [B][T]; work: vt_eval_work_floats() floats of scratch.
 * raw = 1: x, y are the model's input / output in [-1,1], the post-processing above is applied inside;
 * raw = 0: x, y are already [0,1] images (the argument convention of compute_psnr / compute_ssim).
 * (The reference averages these per-frame values over 16-frame splits: a mean of means of equal size.)
 * ---------------------------------------------------------------------------------------- */
int64_t vt_eval_work_floats(int32_t B, int32_t T);
int vt_eval_psnr_ssim(const float* x, const float* y, float* psnr, float* ssim, float* work, int32_t B, int32_t C,
                      int32_t T, int32_t H, int32_t W, int32_t raw, vt_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* VIDTOK_AMD_H */
