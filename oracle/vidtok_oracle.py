"""TEST INFRASTRUCTURE ONLY -- CPU fp32 oracle for the VidTok causal encode/decode path.

This file restates, in plain functional PyTorch (fp32, NCTHW, CPU), the algorithm of the reference
microsoft/VidTok hot path.  It is the checker the parity tests, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` use; NOTHING in the product package `vidtok_amd/` imports it.

PARITY PINNING: the reference ships no tests or golden vectors, so the oracle is
pinned against the reference ITSELF: `tests/test_oracle_vs_reference.py` runs the unmodified reference
(imported from /root/reference via oracle/refload.py, build container only) against this file on
seeded inputs, and `scripts/make_golden.py` stores reference outputs as fixtures under
tests/golden/ which `tests/test_oracle_golden.py` replays anywhere (including the GPU box).

Every function cites the reference lines it follows (R = /root/reference/vidtok).  The restatement is
deliberately structured differently from the reference (no nn.Modules, no einops, every convolution is
one F.conv3d on the 5-D tensor, weights are read from a flat state_dict by key) so that it is an
independent statement of the arithmetic rather than a copy of the code.
"""
import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------------
# elementary operators
# --------------------------------------------------------------------------------------------------
def silu(x):
    """R/modules/model_3dcausal.py:26-27  x * sigmoid(x)."""
    return x * torch.sigmoid(x)


def layernorm_c(x, w, b, eps=1e-6):
    """Per-position LayerNorm over the channel dim of an NCTHW tensor.
    R/modules/model_3dcausal.py:62-80 (wrapper permutes to channels-last, nn.LayerNorm(C, eps=1e-6))."""
    y = F.layer_norm(x.movedim(1, -1), (x.shape[1],), w, b, eps)
    return y.movedim(-1, 1)


def norm_c(sd, prefix, x, site="clip"):
    """Normalize() of the reference at one call site, R/modules/model_3dcausal.py:30-34: the LayerNorm wrapper (keys
    `<prefix>.norm.weight`) or, with `norm_type: groupnorm`, torch.nn.GroupNorm(32, C, eps=1e-6) (keys
    `<prefix>.weight`).  GroupNorm normalises over the spatial axes of the VIEW the call site passes, `site`:
    "frame" = "(b t) c h w" (spatial ResnetBlock :14-19; in the causal family also the 3-D blocks :402-413, the attention
    norm :129-133 and norm_out :664-666), "pos" = "(b t) c s" with s = 1 (causal temporal blocks, :476-487),
    "pixel" = "(b h w) c t" (non-causal temporal blocks, model_3dnoncausal.py:228-236), "clip" = the 5-D tensor
    (non-causal 3-D blocks, attention norm, norm_out)."""
    if (prefix + ".norm.weight") in sd:
        return layernorm_c(x, sd[prefix + ".norm.weight"], sd[prefix + ".norm.bias"])
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    B, C, T, H, W = x.shape
    if site == "frame":
        v = F.group_norm(x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W), 32, w, b, 1e-6)
        return v.reshape(B, T, C, H, W).permute(0, 2, 1, 3, 4)
    if site == "pixel":
        v = F.group_norm(x.permute(0, 3, 4, 1, 2).reshape(B * H * W, C, T), 32, w, b, 1e-6)
        return v.reshape(B, H, W, C, T).permute(0, 3, 4, 1, 2)
    if site == "pos":
        v = F.group_norm(x.permute(0, 2, 3, 4, 1).reshape(-1, C, 1), 32, w, b, 1e-6)
        return v.reshape(B, T, H, W, C).permute(0, 4, 1, 2, 3)
    return F.group_norm(x, 32, w, b, 1e-6)


class ChunkState:
    """Chunk-to-chunk state of the v1.1 models (R/modules/model_3dcausal_v1_1.py:155-157, 212-214,
    286-287, 323-324 and R/models/autoencoder_v1_1.py:202-216)."""

    def __init__(self):
        self.first = True
        self.cache: Dict[str, torch.Tensor] = {}
        self.offset_rules = []  # list of (prefix, offset); the LAST matching rule wins

    def reset(self, prefix):
        for k in [k for k in self.cache if k.startswith(prefix)]:
            del self.cache[k]

    def offset(self, name):
        off = 0
        for prefix, o in self.offset_rules:
            if name.startswith(prefix):
                off = o
        return off


def causal_conv(sd, name, x, version, state: Optional[ChunkState], stride=(1, 1, 1)):
    """Causal convolution with weights sd[name+'.weight'] of rank 3 (Conv1d over T), 4 (Conv2d over H,W,
    per frame -- not causal, symmetric spatial zero pad (k-1)//2) or 5 (Conv3d).

    v1.0: front zero padding in time of k_t-1+(1-s_t) frames, spatial zero padding
          (R/modules/model_3dcausal.py:144-159, 162-197).
    v1.1: the time padding is the first frame repeated (first chunk) or the tail of the cached padded
          input of the previous chunk; cache = padded input minus `cache_offset` trailing frames
          (R/modules/model_3dcausal_v1_1.py:159-178, 216-236)."""
    w = sd[name + ".weight"]
    b = sd.get(name + ".bias")
    if w.dim() == 3:      # Conv1d over time
        w5 = w[:, :, :, None, None]
    elif w.dim() == 4:    # Conv2d per frame
        w5 = w[:, :, None, :, :]
    else:
        w5 = w
    kt, kh, kw = w5.shape[2:]
    causal = w.dim() != 4
    tpad = (kt - 1) + (1 - stride[0]) if causal else 0
    if tpad > 0 or (causal and version == "v1_1"):
        if version == "v1_0":
            x = F.pad(x, (0, 0, 0, 0, tpad, 0))
        else:
            key = name + "#cache"
            if state.first:
                front = x[:, :, :1].repeat(1, 1, tpad, 1, 1)
            else:
                front = state.cache[key][:, :, -tpad:] if tpad > 0 else x[:, :, 0:0]
            x = torch.cat([front, x], dim=2)
            off = state.offset(name)
            state.cache[key] = (x if off == 0 else x[:, :, :-off]).clone()
    if causal:
        hp, wp = (kh - 1) + (1 - stride[1]), (kw - 1) + (1 - stride[2])
        x = F.pad(x, (wp // 2, wp - wp // 2, hp // 2, hp - hp // 2))
    return F.conv3d(x, w5, b, stride=stride)


def conv2d_frames(sd, name, x, stride=1, pad=(1, 1, 1, 1)):
    """nn.Conv2d applied to every frame (the '(b t) c h w' view of the reference,
    R/modules/model_3dcausal.py:14-19); `pad` = (left, right, top, bottom) zeros."""
    w = sd[name + ".weight"][:, :, None]
    x = F.pad(x, pad)
    return F.conv3d(x, w, sd.get(name + ".bias"), stride=(1, stride, stride))


# --------------------------------------------------------------------------------------------------
# blocks
# --------------------------------------------------------------------------------------------------
def resnet_block_2d(sd, p, x):
    """ResnetBlock._forward, R/modules/model_3dcausal.py:317-337."""
    h = silu(norm_c(sd, p + ".norm1", x, "frame"))
    h = conv2d_frames(sd, p + ".conv1", h)
    h = silu(norm_c(sd, p + ".norm2", h, "frame"))
    h = conv2d_frames(sd, p + ".conv2", h)
    if (p + ".nin_shortcut.weight") in sd:
        x = conv2d_frames(sd, p + ".nin_shortcut", x, pad=(0, 0, 0, 0))
    return x + h


def resnet_block_causal(sd, p, x, version, state):
    """ResnetCausalBlock (3-D convs) and ResnetCausalBlock1D (temporal convs): same dataflow,
    R/modules/model_3dcausal.py:400-424 and 473-499.  The 1-D block's LayerNorm sees
    '(b t) c s' with s = 1, i.e. again per-position over C."""
    site = "pos" if sd[p + ".conv1.conv.weight"].dim() == 3 else "frame"   # temporal block: single positions; 3-D: frames
    h = silu(norm_c(sd, p + ".norm1", x, site))
    h = causal_conv(sd, p + ".conv1.conv", h, version, state)
    h = silu(norm_c(sd, p + ".norm2", h, site))
    h = causal_conv(sd, p + ".conv2.conv", h, version, state)
    if (p + ".nin_shortcut.conv.weight") in sd:
        x = causal_conv(sd, p + ".nin_shortcut.conv", x, version, state)
    return x + h


def attn_block(sd, p, x, version, state):
    """AttnBlockWrapper.attention + AttnBlock._forward, R/modules/model_3dcausal.py:114-118, 129-141:
    batch b, 'heads' = frames t, sequence h*w, head_dim c, scale c^-0.5, no mask."""
    B, C, T, H, W = x.shape
    hn = norm_c(sd, p + ".norm", x, "frame")
    q = causal_conv(sd, p + ".q.conv", hn, version, state)
    k = causal_conv(sd, p + ".k.conv", hn, version, state)
    v = causal_conv(sd, p + ".v.conv", hn, version, state)
    q, k, v = (t.permute(0, 2, 3, 4, 1).reshape(B, T, H * W, C) for t in (q, k, v))
    att = torch.softmax(q @ k.transpose(-1, -2) * (C ** -0.5), dim=-1)
    o = (att @ v).reshape(B, T, H, W, C).permute(0, 4, 1, 2, 3)
    return x + causal_conv(sd, p + ".proj_out.conv", o, version, state)


def time_downsample(sd, p, x, version, state):
    """TimeDownsampleResCausal2x.forward, R/modules/model_3dcausal.py:247-252 (v1.0: zero front frame),
    R/modules/model_3dcausal_v1_1.py:289-302 (replicate on the first chunk, cached last frame after)."""
    alpha = torch.sigmoid(sd[p + ".mix_factor"])
    if version == "v1_0":
        xp = F.pad(x, (0, 0, 0, 0, 1, 0))
    else:
        key = p + "#pool"
        xp = torch.cat([x[:, :, :1] if state.first else state.cache[key], x], dim=2)
        state.cache[key] = xp[:, :, -1:].clone()
    x1 = F.avg_pool3d(xp, (3, 1, 1), stride=(2, 1, 1))
    x2 = causal_conv(sd, p + ".conv.conv", x, version, state, stride=(2, 1, 1))
    return alpha * x1 + (1 - alpha) * x2


def _interp_t(x, mode):
    return F.interpolate(x.float(), scale_factor=[2.0, 1.0, 1.0], mode=mode)


def time_upsample(sd, p, x, version, state, mode, n):
    """TimeUpsampleResCausal2x.forward, R/modules/model_3dcausal.py:267-273 (nearest) and
    R/modules/model_3dcausal_v1_1.py:325-343 (trilinear with an n-frame cache; n = num_temp_upsample)."""
    alpha = torch.sigmoid(sd[p + ".mix_factor"])
    if version == "v1_0" or mode != "trilinear":
        x = _interp_t(x, "nearest")
    elif not state.first:
        key = p + "#interp"
        x = torch.cat([state.cache[key], x], dim=2)
        state.cache[key] = x[:, :, -2 * n:-n].clone()
        x = _interp_t(x, mode)[:, :, 2 * n:]
    else:
        state.cache[p + "#interp"] = x[:, :, -n:].clone()
        head, tail = x[:, :, :n], x[:, :, n:]
        x = _interp_t(head, mode)
        if tail.shape[2] > 0:
            x = torch.cat([x, _interp_t(tail, mode)], dim=2)
    x_ = causal_conv(sd, p + ".conv.conv", x, version, state)
    return alpha * x + (1 - alpha) * x_


# --------------------------------------------------------------------------------------------------
# encoder / decoder
# --------------------------------------------------------------------------------------------------
def _cfg(params, key, default):
    v = params.get(key)
    return default if v is None else v


def encoder_forward(sd, params, x, version="v1_0", state: Optional[ChunkState] = None, prefix="encoder"):
    """EncoderCausal3DPadding.forward + EncoderCausal3D.forward,
    R/modules/model_3dcausal.py:685-689, 631-671 (v1.1: model_3dcausal_v1_1.py:755-760)."""
    state = state or ChunkState()
    f = _cfg(params, "time_downsample_factor", 4)
    nres = len(params["ch_mult"])
    spatial_ds = _cfg(params, "spatial_ds", list(range(0, nres - 1)))
    tempo_ds = _cfg(params, "tempo_ds", [nres - 2, nres - 3])
    T = x.shape[2]
    if T % f != 0:
        npad = f - 1 if version == "v1_0" else f - T % f
        pad_mode = params.get("init_pad_mode", "replicate")          # pad_at_dim, R/modules/model_3dcausal.py:37-43,680,688
        if pad_mode == "replicate":
            x = torch.cat([x[:, :, :1].repeat(1, 1, npad, 1, 1), x], dim=2)
        elif pad_mode == "constant":
            x = torch.cat([torch.zeros_like(x[:, :, :1]).repeat(1, 1, npad, 1, 1), x], dim=2)
        else:                                                          # reflect: x[npad], ..., x[1] in front
            assert pad_mode == "reflect"
            x = torch.cat([x[:, :, 1:npad + 1].flip(2), x], dim=2)
    with_conv = params.get("resamp_with_conv", True)
    h = causal_conv(sd, f"{prefix}.conv_in.conv", x, version, state)
    for lvl in range(nres):
        for blk in range(params["num_res_blocks"]):
            h = resnet_block_2d(sd, f"{prefix}.down.{lvl}.block.{blk}", h)
            h = resnet_block_causal(sd, f"{prefix}.down_temporal.{lvl}.block.{blk}", h, version, state)
        if lvl in spatial_ds:
            if with_conv:
                h = conv2d_frames(sd, f"{prefix}.down.{lvl}.downsample.conv", h, stride=2, pad=(0, 1, 0, 1))
            else:                                                      # Downsample(with_conv=False), :228-229: avg_pool2d(2, 2) per frame
                h = F.avg_pool3d(h, kernel_size=(1, 2, 2), stride=(1, 2, 2))
            if lvl in tempo_ds:
                h = time_downsample(sd, f"{prefix}.down_temporal.{lvl}.downsample", h, version, state)
    h = resnet_block_causal(sd, f"{prefix}.mid.block_1", h, version, state)
    h = attn_block(sd, f"{prefix}.mid.attn_1", h, version, state)
    h = resnet_block_causal(sd, f"{prefix}.mid.block_2", h, version, state)
    h = silu(norm_c(sd, f"{prefix}.norm_out", h, "frame"))
    return causal_conv(sd, f"{prefix}.conv_out.conv", h, version, state)


def decoder_forward(sd, params, z, version="v1_0", state: Optional[ChunkState] = None, prefix="decoder"):
    """DecoderCausal3D.forward + DecoderCausal3DPadding.forward, R/modules/model_3dcausal.py:828-870,
    883-885 (v1.1 returns every frame: model_3dcausal_v1_1.py:957-959)."""
    state = state or ChunkState()
    f = _cfg(params, "time_downsample_factor", 4)
    nres = len(params["ch_mult"])
    spatial_us = _cfg(params, "spatial_us", list(range(1, nres)))
    tempo_us = _cfg(params, "tempo_us", [1, 2])
    mode = params.get("interpolation_mode", "nearest") if version == "v1_1" else "nearest"
    # num_temp_upsample doubles with every temporal up-sampler met going down the levels
    # (R/modules/model_3dcausal_v1_1.py:856, 880-881)
    n_of, n = {}, 1
    for lvl in reversed(range(nres)):
        if lvl in tempo_us:
            n_of[lvl] = n
            n *= 2
    h = causal_conv(sd, f"{prefix}.conv_in.conv", z, version, state)
    h = resnet_block_causal(sd, f"{prefix}.mid.block_1", h, version, state)
    h = attn_block(sd, f"{prefix}.mid.attn_1", h, version, state)
    h = resnet_block_causal(sd, f"{prefix}.mid.block_2", h, version, state)
    for lvl in reversed(range(nres)):
        for blk in range(params["num_res_blocks"] + 1):
            h = resnet_block_2d(sd, f"{prefix}.up.{lvl}.block.{blk}", h)
            h = resnet_block_causal(sd, f"{prefix}.up_temporal.{lvl}.block.{blk}", h, version, state)
        if lvl in spatial_us:
            h = F.interpolate(h, scale_factor=[1.0, 2.0, 2.0], mode="nearest")  # Upsample, :208-212
            if params.get("resamp_with_conv", True):
                h = conv2d_frames(sd, f"{prefix}.up.{lvl}.upsample.conv", h)
            if lvl in tempo_us:
                h = time_upsample(sd, f"{prefix}.up_temporal.{lvl}.upsample", h, version, state, mode, n_of[lvl])
    if not params.get("give_pre_end", False):                            # :862-869
        h = silu(norm_c(sd, f"{prefix}.norm_out", h, "frame"))
        h = causal_conv(sd, f"{prefix}.conv_out.conv", h, version, state)
        if params.get("tanh_out", False):
            h = torch.tanh(h)
    return h[:, :, f - 1:] if version == "v1_0" else h


# --------------------------------------------------------------------------------------------------
# non-causal family (R/modules/model_3dnoncausal.py): centred temporal windows, zeros outside the clip
# --------------------------------------------------------------------------------------------------
def conv_centered(sd, name, x, stride=(1, 1, 1), pad=None):
    """nn.Conv3d / nn.Conv1d / nn.Conv2d of the non-causal modules as ONE F.conv3d on the 5-D tensor.  A Conv1d
    weight [Co,Ci,k] acts along T ("(b h w) c t" view, model_3dcausal.py:14-23), a Conv2d weight on every frame;
    `pad` = (t_front, t_back, h, w) zeros, default k//2 on every axis the kernel extends over."""
    w = sd[name + ".weight"]
    if w.dim() == 3:
        w = w[:, :, :, None, None]
    elif w.dim() == 4:
        w = w[:, :, None]
    kt, kh, kw = w.shape[2:]
    tf, tb, ph, pw = pad if pad is not None else (kt // 2, kt // 2, kh // 2, kw // 2)
    x = F.pad(x, (pw, pw, ph, ph, tf, tb))
    return F.conv3d(x, w, sd.get(name + ".bias"), stride=stride)


def resnet_block_centered(sd, p, x):
    """ResnetBlock1D._forward / ResnetNoncausalBlock._forward, R/modules/model_3dnoncausal.py:228-248, 291-311
    (the channel-changing shortcut is never instantiated: in_channels == out_channels at every call site)."""
    site = "pixel" if sd[p + ".conv1.weight"].dim() == 3 else "clip"
    h = silu(norm_c(sd, p + ".norm1", x, site))
    h = conv_centered(sd, p + ".conv1", h)
    h = silu(norm_c(sd, p + ".norm2", h, site))
    return x + conv_centered(sd, p + ".conv2", h)


def attn_block_nc(sd, p, x):
    """AttnBlockWrapper of the non-causal file (1x1x1 Conv3d projections), R/modules/model_3dnoncausal.py:17-34."""
    B, C, T, H, W = x.shape
    hn = norm_c(sd, p + ".norm", x, "clip")
    q, k, v = (conv_centered(sd, f"{p}.{n}", hn).permute(0, 2, 3, 4, 1).reshape(B, T, H * W, C) for n in "qkv")
    att = torch.softmax(q @ k.transpose(-1, -2) * (C ** -0.5), dim=-1)
    o = (att @ v).reshape(B, T, H, W, C).permute(0, 4, 1, 2, 3)
    return x + conv_centered(sd, p + ".proj_out", o)


def time_downsample_nc(sd, p, x):
    """TimeDownsampleRes2x.forward, R/modules/model_3dnoncausal.py:84-90: one zero frame AFTER the clip."""
    alpha = torch.sigmoid(sd[p + ".mix_factor"])
    xp = F.pad(x, (0, 0, 0, 0, 0, 1))
    x1 = F.avg_pool3d(xp, (3, 1, 1), stride=(2, 1, 1))
    x2 = conv_centered(sd, p + ".conv", x, stride=(2, 1, 1), pad=(0, 1, 1, 1))
    return alpha * x1 + (1 - alpha) * x2


def time_upsample_nc(sd, p, x):
    """TimeUpsampleRes2x.forward, R/modules/model_3dnoncausal.py:105-115: every frame twice, then a centred conv."""
    alpha = torch.sigmoid(sd[p + ".mix_factor"])
    x = x.repeat_interleave(2, dim=2)
    return alpha * x + (1 - alpha) * conv_centered(sd, p + ".conv", x)


def encoder3d_forward(sd, params, x, prefix="encoder"):
    """Encoder3D.forward, R/modules/model_3dnoncausal.py:446-482."""
    nres = len(params["ch_mult"])
    tempo_ds = [nres - 2, nres - 3]
    h = conv_centered(sd, f"{prefix}.conv_in", x)
    for lvl in range(nres):
        for blk in range(params["num_res_blocks"]):
            h = resnet_block_2d(sd, f"{prefix}.down.{lvl}.block.{blk}", h)
            h = resnet_block_centered(sd, f"{prefix}.down_temporal.{lvl}.block.{blk}", h)
        if lvl != nres - 1:
            h = conv2d_frames(sd, f"{prefix}.down.{lvl}.downsample.conv", h, stride=2, pad=(0, 1, 0, 1))
            if lvl in tempo_ds:
                h = time_downsample_nc(sd, f"{prefix}.down_temporal.{lvl}.downsample", h)
    h = resnet_block_centered(sd, f"{prefix}.mid.block_1", h)
    h = attn_block_nc(sd, f"{prefix}.mid.attn_1", h)
    h = resnet_block_centered(sd, f"{prefix}.mid.block_2", h)
    h = silu(norm_c(sd, f"{prefix}.norm_out", h, "clip"))
    return conv_centered(sd, f"{prefix}.conv_out", h)


def decoder3d_forward(sd, params, z, prefix="decoder"):
    """Decoder3D.forward, R/modules/model_3dnoncausal.py:618-651 (tempo_us = [1, 2], every frame returned)."""
    nres = len(params["ch_mult"])
    h = conv_centered(sd, f"{prefix}.conv_in", z)
    h = resnet_block_centered(sd, f"{prefix}.mid.block_1", h)
    h = attn_block_nc(sd, f"{prefix}.mid.attn_1", h)
    h = resnet_block_centered(sd, f"{prefix}.mid.block_2", h)
    for lvl in reversed(range(nres)):
        for blk in range(params["num_res_blocks"] + 1):
            h = resnet_block_2d(sd, f"{prefix}.up.{lvl}.block.{blk}", h)
            h = resnet_block_centered(sd, f"{prefix}.up_temporal.{lvl}.block.{blk}", h)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=[1.0, 2.0, 2.0], mode="nearest")
            h = conv2d_frames(sd, f"{prefix}.up.{lvl}.upsample.conv", h)
            if lvl in (1, 2):
                h = time_upsample_nc(sd, f"{prefix}.up_temporal.{lvl}.upsample", h)
    h = silu(norm_c(sd, f"{prefix}.norm_out", h, "clip"))
    return conv_centered(sd, f"{prefix}.conv_out", h)


# --------------------------------------------------------------------------------------------------
# regularizers
# --------------------------------------------------------------------------------------------------
def kl_regularize(h, sample=True, noise=None):
    """DiagonalGaussianRegularizer.forward, R/modules/regularizers.py:82-92 with
    DiagonalGaussianDistribution, R/modules/distributions.py:6-28.  When `noise` is None and
    `sample`, the noise is drawn exactly like the reference: torch.randn(mean.shape) on the CPU
    default generator."""
    mean, logvar = torch.chunk(h, 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    std, var = torch.exp(0.5 * logvar), torch.exp(logvar)
    if sample:
        if noise is None:
            noise = torch.randn(mean.shape)
        z = mean + std * noise.to(mean.device)
    else:
        z = mean
    kl = 0.5 * torch.sum(mean * mean + var - 1.0 - logvar)  # sum over dims [1,2,3] then all: same total
    return z, {"kl_loss": kl / h.shape[0]}


def fsq_constants(levels):
    """R/modules/regularizers.py:153-158, 163 -- identical tensor expressions, fp32."""
    lv = torch.tensor(levels, dtype=torch.int32)
    half_l = (lv - 1) * (1 + 1e-3) / 2
    offset = torch.where(lv % 2 == 0, 0.5, 0.0)
    shift = (offset / half_l).atanh()
    half_w = lv // 2
    basis = torch.cumprod(torch.tensor([1] + list(levels[:-1])), dim=0, dtype=torch.int32)
    return lv, half_l, offset, shift, half_w, basis


def fsq_regularize(h, levels, num_codebooks=1, keep_num_codebooks_dim=None, scale=None, entropy_loss_weight=0.0,
                   entropy_loss_annealing_steps=0, entropy_loss_annealing_factor=1.0, commitment_loss_weight=0.0,
                   diversity_gamma=1.0, inv_temperature=100.0, n_steps=0, with_aux=True):
    """FSQRegularizer.forward behind project_in, R/modules/regularizers.py:206-268 (bound :153, quantize :160,
    codes_to_indices :174): h [b, c*d, ...] holds `num_codebooks` = c groups of d = len(levels) channels
    ("b n (c d) -> b n c d", :227), each quantised on its own; indices [b, ...] or, with keep_num_codebooks_dim
    (forced for c > 1, :131-133), [b, ..., c]."""
    lv, half_l, offset, shift, half_w, basis = fsq_constants(levels)
    c, d = int(num_codebooks), len(levels)
    keep = (c > 1) if keep_num_codebooks_dim is None else bool(keep_num_codebooks_dim)
    assert not (c > 1 and not keep)
    zf = h.float().movedim(1, -1)                      # b ... (c d)
    assert zf.shape[-1] == c * d
    zf = zf.reshape(zf.shape[:-1] + (c, d))             # b ... c d
    bounded = (zf + shift).tanh() * half_l - offset
    codes = bounded.round() / half_w                    # round_ste == round in value
    indices = ((codes * half_w + half_w) * basis).sum(dim=-1).to(torch.int32)      # b ... c
    aux = torch.tensor(0.0)
    if with_aux and (entropy_loss_weight > 0 or commitment_loss_weight > 0):
        # with the codebook axis kept the reference's implicit codebook is 1-D and its einsum raises (:143-146,191-192,234)
        assert not keep, "the reference cannot compute the FSQ aux loss with keep_num_codebooks_dim"
        J = int(torch.prod(lv))
        allidx = torch.arange(J)[:, None]
        codebook = ((allidx // basis) % lv - half_w) / half_w            # indices_to_codes :186-187,170-172
        flat = zf.reshape(-1, d)
        ent_sum, avg = 0.0, torch.zeros(J)
        for s in range(0, flat.shape[0], 2048):                           # chunked: [n, J] is large
            logits = (2.0 * flat[s:s + 2048] @ codebook.t()) * inv_temperature
            prob = logits.softmax(dim=-1)
            ent_sum = ent_sum + (-prob * prob.clamp(min=1e-5).log()).sum()
            avg = avg + prob.sum(dim=0)
        per_sample_entropy = ent_sum / flat.shape[0]
        avg = avg / flat.shape[0]
        codebook_entropy = (-avg * avg.clamp(min=1e-5).log()).sum()
        commit = ((zf - codes) ** 2).mean()
        if n_steps >= entropy_loss_annealing_steps:
            w = entropy_loss_weight
        else:
            start = entropy_loss_annealing_factor * entropy_loss_weight
            w = start - (n_steps / entropy_loss_annealing_steps) * (start - entropy_loss_weight)
        aux = (per_sample_entropy - diversity_gamma * codebook_entropy) * w + commit * commitment_loss_weight
    codes = codes.reshape(codes.shape[:-2] + (c * d,))
    if not keep:
        indices = indices[..., 0]
    return codes.movedim(-1, 1), {"indices": indices, "aux_loss": aux}


def fsq_indices_to_codes(indices, levels, keep_num_codebooks_dim=False):
    """FSQRegularizer.indices_to_codes (video form), R/modules/regularizers.py:180-198: indices [b, ...] or, with
    keep_num_codebooks_dim, [b, ..., c] -> codes [b, d, ...] / [b, c*d, ...]."""
    lv, _, _, _, half_w, basis = fsq_constants(levels)
    codes = ((indices[..., None] // basis) % lv - half_w) / half_w
    if keep_num_codebooks_dim:
        codes = codes.reshape(codes.shape[:-2] + (-1,))                   # "... c d -> ... (c d)"
    return codes.movedim(-1, 1)


# --------------------------------------------------------------------------------------------------
# engine
# --------------------------------------------------------------------------------------------------
class OracleEngine:
    """AutoencodingEngine of the reference, R/models/autoencoder.py:197-229 and (v1.1, tiling)
    R/models/autoencoder_v1_1.py:212-342, over a flat state_dict."""

    def __init__(self, model_params: dict, state_dict: Dict[str, torch.Tensor], version: str = None):
        self.enc_params = dict(model_params["encoder_config"]["params"])
        dec = model_params["decoder_config"]["params"]
        self.dec_params = dict(self.enc_params if isinstance(dec, str) else dec)
        enc_target = model_params["encoder_config"]["target"]
        self.noncausal = "noncausal" in enc_target.lower() or enc_target.endswith("Encoder3D")
        self.version = version or ("v1_1" if ("v1_1" in enc_target or "V11" in enc_target) else "v1_0")
        reg = model_params["regularizer_config"]
        self.reg_kind = "fsq" if "FSQ" in reg["target"] else "kl"
        self.reg_params = dict(reg.get("params") or {})
        self.sd = {k: v.detach().float() for k, v in state_dict.items()}
        self.sample = True
        self.use_tiling = bool(model_params.get("use_tiling", False))
        self.t_chunk_enc = int(model_params.get("t_chunk_enc", 16))
        self.use_overlap = False
        self.f = int(self.enc_params.get("time_downsample_factor", 4))
        self.with_aux = True
        self._enc_state, self._dec_state = ChunkState(), ChunkState()
        self._dec_offsets = []

    @property
    def t_chunk_dec(self):
        return self.t_chunk_enc // self.f

    # -- regularizer --------------------------------------------------------------------------------
    def _project(self, name, x):
        """nn.Linear along the channel axis (FSQ project_in / project_out, R/modules/regularizers.py:137-139,225,255)"""
        w, b = self.sd[f"regularization.{name}.weight"], self.sd[f"regularization.{name}.bias"]
        return (torch.einsum("oc,bc...->bo...", w, x) + b.reshape((1, -1) + (1,) * (x.dim() - 2))).contiguous()

    def regularize(self, h):
        if self.reg_kind == "kl":
            return kl_regularize(h, sample=self.sample)
        proj = "regularization.project_in.weight" in self.sd      # dim != len(levels)
        if proj:
            h = self._project("project_in", h)
        z, log = fsq_regularize(h, self.reg_params["levels"],
                                **{k: v for k, v in self.reg_params.items() if k not in ("levels", "dim")},
                                with_aux=self.with_aux)
        return (self._project("project_out", z) if proj else z), log

    # -- chunks (R/models/autoencoder_v1_1.py:218-228) ---------------------------------------------
    def chunks(self, t, decoder_mode=False):
        step = self.t_chunk_dec if decoder_mode else self.t_chunk_enc
        out, start = [[0, 1]], 1
        while start < t:
            end = min(t, start + step)
            out.append([start, end])
            start = end
        return out

    @torch.no_grad()
    def pre_quant(self, x):
        """encoder output before the regulariser (un-tiled)"""
        if self.noncausal:
            return encoder3d_forward(self.sd, self.enc_params, x)
        return encoder_forward(self.sd, self.enc_params, x, self.version)

    @torch.no_grad()
    def encode(self, x):
        if self.noncausal:
            return self.regularize(encoder3d_forward(self.sd, self.enc_params, x))
        if self.version == "v1_0":
            return self.regularize(encoder_forward(self.sd, self.enc_params, x, "v1_0"))
        st = self._enc_state
        st.reset("encoder")
        st.first = True
        if not self.use_tiling:
            return self.regularize(encoder_forward(self.sd, self.enc_params, x, "v1_1", st))
        zs, logs = [], []
        for i, (s, e) in enumerate(self.chunks(x.shape[2])):     # tile_encode :244-264
            st.first = i == 0
            z, log = self.regularize(encoder_forward(self.sd, self.enc_params, x[:, :, s:e], "v1_1", st))
            zs.append(z)
            logs.append(log)
        z = torch.cat(zs, dim=2)
        if "kl_loss" in logs[0]:
            return z, {"kl_loss": torch.stack([d["kl_loss"] for d in logs]).mean()}
        return z, {"aux_loss": torch.stack([d["aux_loss"] for d in logs]).mean(),
                   "indices": torch.cat([d["indices"] for d in logs], dim=1)}

    def indices_to_latent(self, idx):
        nc = int(self.reg_params.get("num_codebooks", 1))
        keep = self.reg_params.get("keep_num_codebooks_dim")
        z = fsq_indices_to_codes(idx, self.reg_params["levels"], (nc > 1) if keep is None else bool(keep))
        return self._project("project_out", z) if "regularization.project_out.weight" in self.sd else z

    def _overlap_rules(self):
        """R/models/autoencoder_v1_1.py:307-320: sub-trees named by module path; later rules override."""
        f = self.f
        rules = [("decoder", 1)]
        if f == 4:
            rules += [("decoder.up_temporal.2.upsample", 2), ("decoder.up_temporal.1", 2),
                      ("decoder.up_temporal.1.upsample", 4), ("decoder.up_temporal.0", 4), ("decoder.conv_out", 4)]
        elif f == 2:
            rules += [("decoder.up_temporal.2.upsample", 2), ("decoder.up_temporal.1", 2),
                      ("decoder.up_temporal.0", 2), ("decoder.conv_out", 2)]
        else:
            rules += [("decoder.up_temporal.3.upsample", 2), ("decoder.up_temporal.2", 2),
                      ("decoder.up_temporal.2.upsample", 4), ("decoder.up_temporal.1", 4),
                      ("decoder.up_temporal.1.upsample", 8), ("decoder.up_temporal.0", 8), ("decoder.conv_out", 8)]
        return rules

    @torch.no_grad()
    def decode(self, z, decode_from_indices=False):
        if decode_from_indices:
            z = self.indices_to_latent(z)
        if self.noncausal:
            return decoder3d_forward(self.sd, self.dec_params, z)
        if self.version == "v1_0":
            return decoder_forward(self.sd, self.dec_params, z, "v1_0")
        st = self._dec_state
        st.reset("decoder")
        st.first = True
        if not self.use_tiling:
            return decoder_forward(self.sd, self.dec_params, z, "v1_1", st)
        if self.use_overlap:                                        # offsets persist once set (:307-320)
            st.offset_rules = self._overlap_rules()
        n, out = z.shape[2], []
        for i, (s, e) in enumerate(self.chunks(n, decoder_mode=True)):   # tile_decode :302-331
            st.first = i == 0
            look = self.use_overlap and e + 1 <= n
            c = decoder_forward(self.sd, self.dec_params, z[:, :, s:e + 1] if look else z[:, :, s:e], "v1_1", st)
            out.append(c[:, :, :-self.f] if look else c)
        return torch.cat(out, dim=2)

    @torch.no_grad()
    def forward(self, x):
        z, log = self.encode(x)
        dec = self.decode(z)
        if self.version == "v1_1" and dec.shape[2] != x.shape[2]:
            dec = dec[:, :, -x.shape[2]:]
        return z, dec, log

    __call__ = forward
