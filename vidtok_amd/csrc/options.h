// Process-wide tuning / test switches of libvidtok_amd (vt_set_option / vt_get_option, include/vidtok_amd.h).
// Kernel selection is a function of the descriptor and of THIS table only: the table is filled once from the
// environment (VT_<NAME> of the option, so shell A/B runs keep working) and changed afterwards only through
// vt_set_option -- no launch path reads the environment.
#pragma once

enum VtOpt {
  OPT_CONV_BUF = 0,        // 1: gather through buffer descriptors (buffer_load ... lds); 0: 64-bit pointers (global_load_lds)
  OPT_CONV_TINNER,         // 1: temporal convs walk the tiles frames-innermost
  OPT_CONV_LDSEPI,         // 1: 128 x 128 tile: epilogue transposed through the LDS (needed by its fused LayerNorm)
  OPT_CONV_WS,             // weight-stationary persistent kernel (conv_ws2.hip) for the 3x3 128 -> 128 convolutions: 0 off
  OPT_CONV_NARROW,         // 1: conv3d_narrow_kernel for Cout <= 4
  OPT_CONV_TILE,           // 0 auto, 128 / 256: force the tile where legal
  OPT_CONV_TILE_MIN,       // fewest 256 x 256 tiles for which the 8-wave tile is chosen
  OPT_CONV_FUSE_LN,        // 1: LayerNorm in the 128 x 128 tile's epilogue (Cout = 128)
  OPT_CONV_FUSE_LN256,     // 1: LayerNorm in the 8-wave tile's epilogue (Cout = 256)
  OPT_TBLOCK_FUSED,        // 0: vt_temporal_block_supported answers no (the host keeps the blocks on the unfused operators)
  OPT_TBLOCK_PROF_MODE,    // vt_temporal_block_profile: bit 0 GEMMs skipped, bit 1 row units skipped, bit 4 no stores (wrong results)
  OPT_CONV_DEEP,           // 1: 128 x 128 tile on a 4-slot ring (three K steps in flight) when a launch has no more tiles than the device has CUs
  OPT_WS_PROF_MODE,        // vt_conv_profile on conv_ws2.hip: bit 0 = row slots skipped, bit 1 = LDS-DMA requests skipped (wrong results)
  OPT_ATTN_FLASH,          // 1: vt_flash_attention_supported answers yes where the kernel applies (0: the hosts keep the GEMM -> softmax -> GEMM operators)
  OPT_CONV_SPLITK,         // 1: 3-tap convolutions on few pixels PER CLIP run split over the tap planes when the caller gives scratch (another summation order; decided from one clip's geometry, never from B)
  OPT_CONV_TSKIP,          // 1: a tile whose leading time taps read only the zero frames in front of the clip (causal padding, tmode ZERO) starts its K walk behind them
  OPT_CONV_IN8,            // 1: conv_in8_kernel for the encoder's conv_in (bf16, 3 x 3 x 3, 8 stored input channels -> 128: halo patch by LDS-DMA, register-stationary weights); 0: the general path of the implicit-GEMM kernel (the same bits)
  OPT_CONV_TUP_LN,         // 1: the consumer's LayerNorm behind a v1.0 time up-sampler is emitted by its two parity launches (alpha-mix + interleaved frames + LayerNorm in the bf16 LDS epilogue of the 8-wave tile) instead of a separate pass
  OPT_CONV_NT_MB,          // > 0: outputs of at least this many MiB leave the LDS epilogues as streaming (nt) stores; 0 = plain stores everywhere
  OPT_COUNT
};

// current value of an option (atomic load; first call reads the environment)
int vt_opt(int id);
