"""Host-side mirror of the reference's causal encoder / decoder modules.

Class names, constructor keywords, attributes read from outside and -- through the choice of
parameter containers -- every `state_dict` key and shape are those of the reference
(vidtok/modules/model_3dcausal.py and model_3dcausal_v1_1.py), so published
checkpoints load unchanged.  The forward passes are NOT the reference's: activations stay in one
NDHWC tensor end to end, the ~300 einops rearranges of the reference disappear, padding /
up-sampling / residual adds / alpha mixes are folded into the HIP convolution kernel (ops.conv)
and LayerNorm+SiLU is one HIP kernel.  torch.nn.Conv*/LayerNorm objects appear only as parameter
containers (initialisation + key names); their forward() is never called.

`version` selects the reference variant: "v1_0" = zero causal padding, nearest time up-sampling,
decoder drops 3 frames; "v1_1" = first-frame-replicate / cached causal padding, trilinear time
up-sampling, chunk-to-chunk caches (`causal_cache`, `is_first_chunk`, `cache_offset` attributes
exactly as model_3dcausal_v1_1.py:155-157,212-214 so an engine can drive them).
"""
import functools
import os

import torch
import torch.nn as nn

from . import lib as L
from . import ops
from .ops import ConvGeom
from .packing import (PackedCache, space_upsample_parity_mix, space_upsample_parity_weights, time_upsample_parity_mix,
                      time_upsample_parity_weights)


def _check_norm(norm_type):
    if norm_type not in ("layernorm", "groupnorm"):
        raise NotImplementedError(f"norm_type={norm_type!r}: the reference knows 'layernorm' and 'groupnorm'")


# which view of the activation a norm's call site hands to it in the reference (matters for GroupNorm only, whose
# statistics run over that view's spatial axes): "(b t) c h w", "(b h w) c t", "b c t h w", or "(b t) c s" with s = 1.
# The causal family normalises per frame everywhere (model_3dcausal.py:402-413, 129-133, 664-666) except in its
# temporal blocks, which see single positions (:476-487); the non-causal family uses the pixel and clip views
# (model_3dnoncausal.py:228-236, 26, 477).
SITE_FRAME, SITE_PIXEL, SITE_CLIP, SITE_POS = L.VT_GN_FRAME, L.VT_GN_PIXEL, L.VT_GN_CLIP, ops.GN_POS


class Normed:
    """An activation together with the LayerNorm(+SiLU) its consumer applies first, already produced by the conv
    that wrote the activation (`ln=` of ops.conv: one pass over the tensor instead of write + read + write)."""

    __slots__ = ("y", "n", "norm", "silu")

    def __init__(self, y, n, norm, silu):
        self.y, self.n, self.norm, self.silu = y, n, norm, silu


def plain(x):
    return x.y if isinstance(x, Normed) else x


_EMIT_NEXT_NORM = os.environ.get("VIDTOK_AMD_EMIT_NEXT_NORM", "1") != "0"   # A/B switch (0: consumers run their own norm)
_FUSE_TBLOCK = os.environ.get("VIDTOK_AMD_FUSE_TBLOCK", "1") != "0"         # A/B switch (0: temporal blocks as two convs)


def _emit(next_norm):
    """kwargs for the last conv of a block: also emit the consumer's norm; `next_norm` = (LayerNorm, silu) or None"""
    if next_norm is None or not _EMIT_NEXT_NORM or not next_norm[0].fusable:
        return {}
    return dict(ln=next_norm[0].fused(next_norm[1]), ln_keep_y=True)


def _wrap(out, next_norm):
    if next_norm is None or not _EMIT_NEXT_NORM or not next_norm[0].fusable:
        return out
    return Normed(out[0], out[1], next_norm[0], next_norm[1])


class LayerNorm(nn.Module):
    """Parameter holder for the channels-last LayerNorm wrapper (model_3dcausal.py:62-80)."""

    def __init__(self, num_channels, eps=1e-6):
        super().__init__()
        self.norm = nn.LayerNorm(num_channels, eps=eps, elementwise_affine=True)
        self._cache = None

    def affine(self):
        w, b = self.norm.weight, self.norm.bias
        key = (w._version, b._version, w.device, w.data_ptr())
        if self._cache is None or self._cache[0] != key:
            self._cache = (key, w.detach().float().contiguous(), b.detach().float().contiguous())
        return self._cache[1], self._cache[2]

    fusable = True     # per-position statistics: the producing conv can emit this norm from its epilogue

    def apply_ndhwc(self, x, silu, dt, site=None):
        if isinstance(x, Normed):
            if x.norm is self and x.silu == silu:
                return x.n            # the producing conv already applied this norm
            x = x.y
        g, b = self.affine()
        return ops.layernorm_act(x, g, b, silu=silu, eps=self.norm.eps, out_dtype=dt)

    def fused(self, silu):
        """(gamma, beta, eps, silu) for the `ln=` argument of ops.conv: this norm applied by the producing conv."""
        g, b = self.affine()
        return (g, b, self.norm.eps, silu)

    def after(self, conv, silu, dt, site=None):
        """norm(conv(...)) where only the normalised tensor is needed: `conv(**kw)` runs the convolution"""
        return conv(ln=self.fused(silu), ln_keep_y=False)


class GroupNorm32(nn.GroupNorm):
    """torch.nn.GroupNorm(32, C, eps=1e-6, affine=True) parameter holder -- `norm_type: groupnorm` of Normalize()
    (model_3dcausal.py:30-32; state_dict keys `...norm1.weight / .bias`, no `.norm` level).  Its statistics span the
    spatial axes of the call site's view, so it is never fused into a conv epilogue (vt_groupnorm_act)."""

    fusable = False

    def __init__(self, num_channels):
        super().__init__(num_groups=32, num_channels=num_channels, eps=1e-6, affine=True)

    def apply_ndhwc(self, x, silu, dt, site=SITE_FRAME):
        x = plain(x)
        return ops.groupnorm_act(x, self.weight.detach().float().contiguous(), self.bias.detach().float().contiguous(),
                                 scope=site, silu=silu, eps=self.eps, out_dtype=dt, c=self.num_channels)

    def after(self, conv, silu, dt, site=SITE_FRAME):
        return self.apply_ndhwc(conv(), silu, dt, site)


def Normalize(in_channels, norm_type="layernorm"):
    _check_norm(norm_type)
    return GroupNorm32(in_channels) if norm_type == "groupnorm" else LayerNorm(in_channels, eps=1e-6)


def _persistent_cache(module, shape, like):
    """see _CausalState._persistent (the time resamplers carry a cache without being causal convolutions)"""
    return _CausalState._persistent(module, shape, like)


class _CausalState:
    """v1.1 chunk-to-chunk state shared by the causal convs (model_3dcausal_v1_1.py:155-178)."""

    def _init_state(self):
        self.is_first_chunk = True
        self.causal_cache = None
        self.cache_offset = 0

    def _tmode_and_cache(self, version, time_pad):
        if version == "v1_0" or time_pad == 0:
            return L.VT_TPAD_ZERO, None
        if self.is_first_chunk:
            return L.VT_TPAD_REPLICATE, None
        if self.causal_cache is None or self.causal_cache.shape[1] < time_pad:
            raise RuntimeError("causal cache missing: run the first chunk with is_first_chunk=True")
        return L.VT_TPAD_CACHE, self.causal_cache

    def _persistent(self, shape, like):
        """The cache lives in one buffer per (module, shape), allocated on first use and rewritten in place afterwards:
        a chunk of a given kind reads and writes the same addresses every time, which is what lets a captured chunk
        (vidtok_amd/graphs.py) be replayed.  `causal_cache` (the reference's attribute, reset to None between clips) is
        bound to the buffer after each update; `_cache_bufs` survives the reset."""
        bufs = self.__dict__.setdefault("_cache_bufs", {})
        key = (tuple(shape), like.dtype, like.device)
        buf = bufs.get(key)
        if buf is None:
            buf = bufs[key] = torch.empty(tuple(shape), dtype=like.dtype, device=like.device)
        return buf

    def _update_cache(self, x, time_pad):
        """Keep the last `time_pad` frames of padded[:len-cache_offset] where
        padded = [pad frames (x[0] repeated | previous cache), x]  (model_3dcausal_v1_1.py:172-176)."""
        if time_pad == 0:
            return
        T, off, P = x.shape[1], self.cache_offset, time_pad
        src_x, src_c = [], []  # (dst slot, frame index)
        for j in range(P):
            q = T - off + j            # index into the padded sequence
            if q >= P:
                src_x.append((j, q - P))
            elif q < 0:
                raise RuntimeError("chunk shorter than cache_offset")
            elif self.is_first_chunk:
                src_x.append((j, 0))
            else:
                src_c.append((j, q))
        # slots taken from the old cache come first (q grows with j), the ones from x after them.  The old frames move
        # inside the buffer they are read from: through a temporary (launches are stream-ordered, so the buffer is
        # rewritten only after the copy -- and after the convolution that read it -- has finished)
        kept = ops.gather_frames(self.causal_cache, [i for _, i in src_c]) if src_c else None
        buf = self._persistent((x.shape[0], P) + tuple(x.shape[2:]), x)
        if kept is not None:
            ops.gather_frames(kept, list(range(len(src_c))), out=buf, out_t0=src_c[0][0])
        if src_x:
            ops.gather_frames(x, [i for _, i in src_x], out=buf, out_t0=src_x[0][0])
        self.causal_cache = buf


class CausalConv3d(nn.Module, _CausalState):
    """model_3dcausal.py:162-197 / model_3dcausal_v1_1.py:181-236.  `conv` holds the parameters."""

    def __init__(self, chan_in, chan_out, kernel_size, stride=1, version="v1_0"):
        super().__init__()
        ks = kernel_size if isinstance(kernel_size, tuple) else (kernel_size,) * 3
        st = stride if isinstance(stride, tuple) else (stride,) * 3
        assert ks[1] % 2 == 1 and ks[2] % 2 == 1
        self.conv = nn.Conv3d(chan_in, chan_out, ks, stride=st)
        self.ks, self.strides = ks, st
        self.time_pad = (ks[0] - 1) + (1 - st[0])
        self.chan_out = chan_out
        self.version = version
        self._pack = PackedCache()
        self._init_state()

    def geom(self, ups_t=0):
        kt, kh, kw = self.ks
        hp, wp = (kh - 1) + (1 - self.strides[1]), (kw - 1) + (1 - self.strides[2])
        return ConvGeom(kt=kt, kh=kh, kw=kw, st=self.strides[0], sh=self.strides[1], sw=self.strides[2],
                        pt=self.time_pad, ph=hp // 2, pw=wp // 2, ph_hi=hp - hp // 2, pw_hi=wp - wp // 2,
                        ups_t=ups_t)

    def run(self, x, dt, *, ups_t=0, **kw):
        w, b = self._pack.get(self.conv.weight, self.conv.bias, dt, cin_stored=x.shape[-1])
        tmode, cache = self._tmode_and_cache(self.version, self.time_pad)
        y = ops.conv(x, w, b, self.geom(ups_t), cout=self.chan_out, tmode=tmode, cache=cache, **kw)
        if self.version == "v1_1":
            self._update_cache(x, self.time_pad)
        return y


class CausalConv1d(nn.Module, _CausalState):
    """model_3dcausal.py:144-159 / model_3dcausal_v1_1.py:144-178 -- k taps along T only."""

    def __init__(self, chan_in, chan_out, kernel_size, stride=1, version="v1_0"):
        super().__init__()
        self.conv = nn.Conv1d(chan_in, chan_out, kernel_size, stride=stride)
        self.k, self.stride = kernel_size, stride
        self.time_pad = (kernel_size - 1) + (1 - stride)
        self.chan_out = chan_out
        self.version = version
        self._pack = PackedCache()
        self._init_state()

    def run(self, x, dt, **kw):
        w, b = self._pack.get(self.conv.weight, self.conv.bias, dt, cin_stored=x.shape[-1])
        tmode, cache = self._tmode_and_cache(self.version, self.time_pad)
        g = ConvGeom(kt=self.k, st=self.stride, pt=self.time_pad)
        y = ops.conv(x, w, b, g, cout=self.chan_out, tmode=tmode, cache=cache, **kw)
        if self.version == "v1_1":
            self._update_cache(x, self.time_pad)
        return y


class _Conv2dHolder:
    """Runs an nn.Conv2d parameter set as a per-frame spatial conv on NDHWC."""

    @staticmethod
    def run(conv: nn.Conv2d, pack: PackedCache, x, dt, geom: ConvGeom, **kw):
        w, b = pack.get(conv.weight, conv.bias, dt, cin_stored=x.shape[-1])
        return ops.conv(x, w, b, geom, cout=conv.out_channels, **kw)


_G3x3 = ConvGeom(kh=3, kw=3, ph=1, pw=1, ph_hi=1, pw_hi=1)
_G1x1 = ConvGeom()


class Upsample(nn.Module):
    """nearest x2 + conv3x3 (model_3dcausal.py:200-212); the up-sampling is folded into the gather."""

    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        self.in_channels = in_channels
        self._ident = {}
        if not with_conv:       # nearest x2 only (model_3dcausal.py:208-212): no parameters, no `conv` key
            return
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)
        # up(x)[Y][X] = x[Y>>1][X>>1]: an output pixel of parity (py, px) sees a 2x2 window of x, so the 3x3 conv over
        # the up-sampled frame is four 2x2 convs over x with pre-summed taps (4/9 of the MACs), each writing its
        # parity class of the output: rows (a-1, a) for py = 0, (a, a+1) for py = 1, likewise for columns
        self._parity = [(py, px, PackedCache(functools.partial(space_upsample_parity_weights, py=py, px=px),
                                             mix=functools.partial(space_upsample_parity_mix, py=py, px=px)),
                         ConvGeom(kh=2, kw=2, ph=1 - py, pw=1 - px, ph_hi=py, pw_hi=px))
                        for py in (0, 1) for px in (0, 1)]

    def run(self, x, dt, next_norm=None):
        x = plain(x)
        B, T, H, W, C = x.shape
        if not self.with_conv:
            # no shipped config uses this: the gather's folded x2 up-sampling under a 1x1 identity convolution copies
            # every input pixel to its four output pixels exactly (1.0 * x plus zeros)
            key = (dt, x.device, C)
            if key not in self._ident:
                self._ident[key] = torch.eye(C, dtype=torch.float32, device=x.device).to(dt).contiguous()
            return ops.conv(x, self._ident[key], None, ConvGeom(ups_s=1), cout=self.in_channels, ldy=C)
        cout = self.conv.out_channels
        ld = ops.pad_channels(cout)
        y = (torch.empty if ld == cout else torch.zeros)((B, T, 2 * H, 2 * W, ld), dtype=dt, device=x.device)
        for py, px, pack, g in self._parity:
            w, b = pack.get(self.conv.weight, self.conv.bias, dt, cin_stored=C)
            ops.conv(x, w, b, g, cout=cout, out=y, out_s=(py, px))
        return y


class Downsample(nn.Module):
    """F.pad(0,1,0,1) + conv3x3 stride 2 (model_3dcausal.py:215-230)."""

    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        self.in_channels = in_channels
        self._pool_w = {}
        if not with_conv:       # avg_pool2d(2, 2) (model_3dcausal.py:228-229): no parameters
            return
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=0)
        self._pack = PackedCache()

    def run(self, x, dt, next_norm=None):
        if not self.with_conv:
            # no shipped config uses this: 2x2 average pooling as a stride-2 2x2 convolution with 0.25 * identity taps
            # (products exact, the four-term sum accumulated in fp32)
            xp = plain(x)
            C = xp.shape[-1]
            key = (dt, xp.device, C)
            if key not in self._pool_w:
                w = (0.25 * torch.eye(C, dtype=torch.float32, device=xp.device)).repeat(1, 4)     # [C, 4 taps * C]
                self._pool_w[key] = w.to(dt).contiguous()
            y = ops.conv(xp, self._pool_w[key], None, ConvGeom(kh=2, kw=2, sh=2, sw=2), cout=self.in_channels, ldy=C,
                         **_emit(next_norm))
            return _wrap(y, next_norm)
        g = ConvGeom(kh=3, kw=3, sh=2, sw=2, ph=0, pw=0, ph_hi=1, pw_hi=1)
        return _wrap(_Conv2dHolder.run(self.conv, self._pack, plain(x), dt, g, **_emit(next_norm)), next_norm)


class TimeDownsampleResCausal2x(nn.Module):
    """alpha*avgpool3(stride 2) + (1-alpha)*causal conv3d stride (2,1,1)
    (model_3dcausal.py:233-252, v1.1 model_3dcausal_v1_1.py:272-302)."""

    def __init__(self, in_channels, out_channels, mix_factor: float = 2.0, version="v1_0"):
        super().__init__()
        self.conv = CausalConv3d(in_channels, out_channels, 3, stride=(2, 1, 1), version=version)
        self.mix_factor = nn.Parameter(torch.Tensor([mix_factor]))
        self.version = version
        self.is_first_chunk = True
        self.causal_cache = None

    def run(self, x, dt, next_norm=None):
        x = plain(x)
        if self.version == "v1_0":
            x1 = ops.time_avgpool3s2(x, L.VT_TPAD_ZERO)
        else:
            if self.is_first_chunk:
                x1 = ops.time_avgpool3s2(x, L.VT_TPAD_REPLICATE)
            else:
                x1 = ops.time_avgpool3s2(x, L.VT_TPAD_CACHE, cache=self.causal_cache)
            self.causal_cache = ops.gather_frames(x, [x.shape[1] - 1], out=_persistent_cache(self, (x.shape[0], 1) + tuple(x.shape[2:]), x))
        return _wrap(self.conv.run(x, dt, res=x1, res_mode=L.VT_RES_MIX, mix_factor=self.mix_factor.detach(),
                                   **_emit(next_norm)), next_norm)


class TimeUpsampleResCausal2x(nn.Module):
    """alpha*up(x) + (1-alpha)*causal conv3d(up(x)); up = nearest (v1.0, folded into the conv's
    gather and into the mix operand's time index) or trilinear with a frame cache (v1.1)
    (model_3dcausal.py:255-273, model_3dcausal_v1_1.py:305-343)."""

    def __init__(self, in_channels, out_channels, mix_factor: float = 2.0, interpolation_mode="nearest",
                 num_temp_upsample=1, version="v1_0"):
        super().__init__()
        self.conv = CausalConv3d(in_channels, out_channels, 3, version=version)
        self.mix_factor = nn.Parameter(torch.Tensor([mix_factor]))
        self.interpolation_mode = interpolation_mode
        self.num_temp_upsample = num_temp_upsample
        self.enable_cached = interpolation_mode == "trilinear"
        self.version = version
        self.is_first_chunk = True
        self.causal_cache = None
        self._parity_packs = (PackedCache(functools.partial(time_upsample_parity_weights, early=True), mix=functools.partial(time_upsample_parity_mix, early=True)),
                              PackedCache(functools.partial(time_upsample_parity_weights, early=False), mix=functools.partial(time_upsample_parity_mix, early=False)))

    def _interp_v11(self, x):
        n, T = self.num_temp_upsample, x.shape[1]
        if not self.enable_cached:
            return ops.gather_frames(x, [t // 2 for t in range(2 * T)])
        if not self.is_first_chunk:
            nc = self.causal_cache.shape[1]
            Tc = nc + T
            if Tc - 2 * n >= nc:
                # the frames to keep all lie in x: [cache | x] is never assembled -- the interpolation reads both parts in
                # place and leaves out the frames the previous chunk delivered; then the cache buffer is rewritten
                up = ops.time_lerp2x_cat(self.causal_cache, x, 2 * n)
                keep = list(range(Tc - 2 * n - nc, Tc - n - nc))
                self.causal_cache = ops.gather_frames(x, keep, out=_persistent_cache(self, (x.shape[0], len(keep)) + tuple(x.shape[2:]), x))
                return up
            xc = torch.empty((x.shape[0], Tc) + tuple(x.shape[2:]), dtype=x.dtype, device=x.device)   # [cache | x]
            ops.gather_frames(self.causal_cache, list(range(nc)), out=xc, out_t0=0)
            ops.gather_frames(x, list(range(T)), out=xc, out_t0=nc)
            keep = list(range(max(0, Tc - 2 * n), Tc - n))       # (xc is a copy: the cache buffer is free to be rewritten)
            self.causal_cache = ops.gather_frames(xc, keep, out=_persistent_cache(self, (x.shape[0], len(keep)) + tuple(x.shape[2:]), x))
            up = ops.time_lerp2x(xc)
            return ops.gather_frames(up, list(range(2 * n, 2 * Tc)))
        keep = list(range(max(0, T - n), T))
        self.causal_cache = ops.gather_frames(x, keep, out=_persistent_cache(self, (x.shape[0], len(keep)) + tuple(x.shape[2:]), x))
        hn = min(n, T)
        up = torch.empty((x.shape[0], 2 * T) + tuple(x.shape[2:]), dtype=x.dtype, device=x.device)
        ops.time_lerp2x(ops.gather_frames(x, list(range(0, hn))), out=up, out_t0=0)              # head: its own interpolation
        if T > n:
            ops.time_lerp2x(ops.gather_frames(x, list(range(n, T))), out=up, out_t0=2 * hn)    # tail
        return up

    def run(self, x, dt, next_norm=None):
        x = plain(x)
        mf = self.mix_factor.detach()
        if self.version == "v1_0":
            # up(x)[t] = x[t >> 1]: the 3 temporal taps of an output frame hit 2 input frames, so even / odd output
            # frames are two k=2 causal convs over x with pre-summed weights (2/3 of the MACs of the 27-tap form):
            #   o[2j] = (W0+W1) x[j-1] + W2 x[j]      o[2j+1] = W0 x[j-1] + (W1+W2) x[j]
            # each launch writes its frames of the interleaved output; the mix operand up(x)[2j+p] is x[j].
            B, T, H, W, C = x.shape
            ld = ops.pad_channels(self.conv.chan_out)
            y = (torch.empty if ld == self.conv.chan_out else torch.zeros)((B, 2 * T, H, W, ld), dtype=dt, device=x.device)
            g = ConvGeom(kt=2, kh=3, kw=3, pt=1, ph=1, pw=1, ph_hi=1, pw_hi=1)
            # the consumer's LayerNorm from the two launches' epilogues where they can take it (vt_conv_plan decides: bf16, Cout = 256
            # on the 8-wave tile, option conv_tup_ln), else the consumer runs its own pass
            emit = dict(_emit(next_norm))
            n = None

            def alloc_n():           # the LayerNorm twin, allocated by ops.conv only once vt_conv_plan says the launch emits it (pad lanes as y's)
                return (torch.empty if ld == self.conv.chan_out else torch.zeros)(y.shape, dtype=dt, device=x.device)

            for par, pack in enumerate(self._parity_packs):
                w, b = pack.get(self.conv.conv.weight, self.conv.conv.bias, dt, cin_stored=C)
                r = ops.conv(x, w, b, g, cout=self.conv.chan_out, tmode=L.VT_TPAD_ZERO, res=x, res_mode=L.VT_RES_MIX,
                             mix_factor=mf, out=y, out_t=(2, par), **(dict(emit, ln_out=(alloc_n if n is None else n), ln_optional=True) if emit else {}))
                if emit and not isinstance(r, tuple):
                    emit, n = {}, None
                elif emit:
                    n = r[1]
            return y if n is None else Normed(y, n, next_norm[0], next_norm[1])
        xi = self._interp_v11(x)
        return _wrap(self.conv.run(xi, dt, res=xi, res_mode=L.VT_RES_MIX, mix_factor=mf, **_emit(next_norm)), next_norm)


class ResnetBlock(nn.Module):
    """Per-frame spatial block LN-SiLU-conv3x3-LN-SiLU-conv3x3 (+1x1 shortcut) + x
    (model_3dcausal.py:276-337)."""

    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=0,
                 use_checkpoint=False, norm_type="layernorm"):
        super().__init__()
        assert temb_channels == 0 and not conv_shortcut
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = Normalize(in_channels, norm_type)
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.norm2 = Normalize(out_channels, norm_type)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if in_channels != out_channels:
            self.nin_shortcut = nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)
        self._p1, self._p2, self._p3 = PackedCache(), PackedCache(), PackedCache()

    def first_norm(self):
        return (self.norm1, True)

    def run(self, x, dt, next_norm=None):
        h = self.norm1.apply_ndhwc(x, True, dt, SITE_FRAME)
        x = plain(x)
        # conv1's result is only ever seen through norm2 + SiLU: with LayerNorm the conv emits that directly
        h = self.norm2.after(lambda **kw: _Conv2dHolder.run(self.conv1, self._p1, h, dt, _G3x3, **kw), True, dt, SITE_FRAME)
        if self.in_channels != self.out_channels:
            x = _Conv2dHolder.run(self.nin_shortcut, self._p3, x, dt, _G1x1)
        return _wrap(_Conv2dHolder.run(self.conv2, self._p2, h, dt, _G3x3, res=x, res_mode=L.VT_RES_ADD,
                                       **_emit(next_norm)), next_norm)


class ResnetCausalBlock(nn.Module):
    """3-D causal residual block of the mid section (model_3dcausal.py:340-424)."""

    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=0,
                 use_checkpoint=False, norm_type="layernorm", version="v1_0"):
        super().__init__()
        assert temb_channels == 0 and not conv_shortcut
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = Normalize(in_channels, norm_type)
        self.conv1 = CausalConv3d(in_channels, out_channels, 3, version=version)
        self.norm2 = Normalize(out_channels, norm_type)
        self.conv2 = CausalConv3d(out_channels, out_channels, 3, version=version)
        if in_channels != out_channels:
            self.nin_shortcut = CausalConv3d(in_channels, out_channels, 1, version=version)

    def first_norm(self):
        return (self.norm1, True)

    def run(self, x, dt, next_norm=None):
        h = self.norm1.apply_ndhwc(x, True, dt, SITE_FRAME)
        x = plain(x)
        h = self.norm2.after(lambda **kw: self.conv1.run(h, dt, **kw), True, dt, SITE_FRAME)
        if self.in_channels != self.out_channels:
            x = self.nin_shortcut.run(x, dt)
        return _wrap(self.conv2.run(h, dt, res=x, res_mode=L.VT_RES_ADD, **_emit(next_norm)), next_norm)


class ResnetCausalBlock1D(nn.Module):
    """Temporal residual block: per-position LN-SiLU-causal conv1d x2 + x; conv2 is zero-initialised
    (model_3dcausal.py:427-499).  On NDHWC the "(b h w) c t" view of the reference is just a conv
    whose taps run along T."""

    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=0,
                 zero_init=False, use_checkpoint=False, norm_type="layernorm", version="v1_0"):
        super().__init__()
        assert temb_channels == 0 and not conv_shortcut
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = Normalize(in_channels, norm_type)
        self.conv1 = CausalConv1d(in_channels, out_channels, 3, version=version)
        self.norm2 = Normalize(out_channels, norm_type)
        self.conv2 = CausalConv1d(out_channels, out_channels, 3, version=version)
        if in_channels != out_channels:
            self.nin_shortcut = CausalConv1d(in_channels, out_channels, 1, version=version)
        if zero_init:
            self.conv2.conv.weight.data.zero_()
            self.conv2.conv.bias.data.zero_()

    # v1.1 blocks keep chunk-to-chunk caches of their convolutions' INPUTS, which a fused launch never materialises.  An
    # engine that is NOT tiling sets `allow_fused` (AutoencodingEngineV11._set_fused_temporal): no chunk follows, nothing
    # to keep.  A tiling engine leaves it False and the launch keeps the chunk state itself (ops.temporal_block `caches`);
    # v1.0 has no caches.
    allow_fused = False

    def _fusable(self, dt):
        """may run as ONE launch (ops.temporal_block): LayerNorm variant, C -> C, bf16"""
        if not _FUSE_TBLOCK or dt not in ops.H16 or self.in_channels != self.out_channels:
            return False
        if not (self.norm1.fusable and self.norm2.fusable) or self.norm1.norm.eps != self.norm2.norm.eps:
            return False
        if self.in_channels != 128:          # the kernel's LayerNorm statistics span exactly 128 REAL channels
            return False
        if self.conv1.version == "v1_0" or (self.allow_fused and self.conv1.is_first_chunk):
            return True
        # a chunk of a tiled v1.1 pass: both convolutions at the same point of the chunk schedule, and -- past the first
        # chunk -- both caches there
        c1, c2 = self.conv1, self.conv2
        if c1.cache_offset != c2.cache_offset or c1.is_first_chunk != c2.is_first_chunk:
            return False
        return c1.is_first_chunk or (c1.causal_cache is not None and c2.causal_cache is not None)

    def _chunk_state(self, xp):
        """(tmode, caches, cache_offset) of a fused launch: (zero | replicate, None, 0) where no chunk state lives, else the
        two persistent cache buffers (allocated on the first chunk, the convolutions' `causal_cache` afterwards)"""
        c1, c2 = self.conv1, self.conv2
        if c1.version == "v1_0":
            return L.VT_TPAD_ZERO, None, 0
        if self.allow_fused and c1.is_first_chunk:
            return L.VT_TPAD_REPLICATE, None, 0
        shape = (xp.shape[0], 2) + tuple(xp.shape[2:])
        if c1.is_first_chunk:
            return L.VT_TPAD_REPLICATE, (c1._persistent(shape, xp), c2._persistent(shape, xp)), c1.cache_offset
        ok = all(t.shape == shape and t.dtype == xp.dtype and t.is_contiguous() for t in (c1.causal_cache, c2.causal_cache))
        return L.VT_TPAD_CACHE, ((c1.causal_cache, c2.causal_cache) if ok else None), c1.cache_offset

    def first_norm(self, dt=None):
        # a fused block normalises x itself: its producer must not spend a write on LayerNorm1(x).  Whether the block then
        # really fuses also depends on the activation (vt_temporal_block_supported: frames past the cache offset, pixels per
        # frame), which the producer's caller does not have; where it does not, run() spends its own LayerNorm pass.  Remembering
        # run()'s decision here was tried and taken back: a producer's epilogue normalises the fp32 accumulators, a
        # separate pass the stored bf16 rows, so the first pass of a shape would differ in bits from every later one.
        if dt is not None and self._fusable(dt):
            return None
        return (self.norm1, True)

    def run(self, x, dt, next_norm=None):
        xp = plain(x)
        fusable = self._fusable(dt)
        tmode, caches, off = self._chunk_state(xp) if fusable else (L.VT_TPAD_ZERO, None, 0)
        if fusable and ops.temporal_block_supported(xp, tmode, self.in_channels, caches, off):
            c = xp.shape[-1]
            w1, b1 = self.conv1._pack.get(self.conv1.conv.weight, self.conv1.conv.bias, dt, cin_stored=c)
            w2, b2 = self.conv2._pack.get(self.conv2.conv.weight, self.conv2.conv.bias, dt, cin_stored=c)
            nxt = None
            if next_norm is not None and _EMIT_NEXT_NORM and next_norm[0].fusable and next_norm[0].norm.eps == self.norm1.norm.eps:
                g, b = next_norm[0].affine()
                nxt = (g, b, next_norm[1])
            out = ops.temporal_block(xp, w1, b1, w2, b2, self.norm1.affine(), self.norm2.affine(), tmode=tmode,
                                     eps=self.norm1.norm.eps, next_ln=nxt, keep_y=True, caches=caches, cache_offset=off)
            if caches is not None:          # the launch has rewritten the chunk state in place
                self.conv1.causal_cache, self.conv2.causal_cache = caches
            return out if nxt is None else Normed(out[0], out[1], next_norm[0], next_norm[1])
        h = self.norm1.apply_ndhwc(x, True, dt, SITE_POS)
        x = xp
        h = self.norm2.after(lambda **kw: self.conv1.run(h, dt, **kw), True, dt, SITE_POS)
        if self.in_channels != self.out_channels:
            x = self.nin_shortcut.run(x, dt)
        return _wrap(self.conv2.run(h, dt, res=x, res_mode=L.VT_RES_ADD, **_emit(next_norm)), next_norm)


class AttnBlockWrapper(nn.Module):
    """Per-frame spatial self-attention (model_3dcausal.py:83-141): LN, q/k/v 1x1x1, softmax(QK^T/sqrt(C))V
    with "heads" = frames, proj_out, + x.  Q K^T and P V run on the MFMA GEMM kernel (ops.gemm_nt),
    the softmax on a wave-shuffle kernel; V^T is produced directly by swapping the GEMM operands
    (W_v as the row operand), and v's bias is added after P V (rows of P sum to 1)."""

    def __init__(self, in_channels, use_checkpoint=False, norm_type="layernorm", version="v1_0"):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels, norm_type)
        self.q = CausalConv3d(in_channels, in_channels, 1, version=version)
        self.k = CausalConv3d(in_channels, in_channels, 1, version=version)
        self.v = CausalConv3d(in_channels, in_channels, 1, version=version)
        self.proj_out = CausalConv3d(in_channels, in_channels, 1, version=version)
        self._v_rows = PackedCache(pin_native=True)     # W_v as the ROW operand of a GEMM against activations: plain rows in every mode

    def first_norm(self):
        return (self.norm, False)

    def run(self, x, dt, next_norm=None):
        hn = self.norm.apply_ndhwc(x, False, dt, SITE_FRAME)
        x = plain(x)
        B, T, H, W, Cc = x.shape
        S, Z = H * W, B * T
        q = self.q.run(hn, dt).view(Z, S, Cc)
        k = self.k.run(hn, dt).view(Z, S, Cc)
        wv, bv = self._v_rows.get(self.v.conv.weight, self.v.conv.bias, dt, cin_stored=Cc)
        Sp = ops.pad_channels(S)        # K-contiguous operands need 16-byte rows: pad S with zero columns
        vT = ops.gemm_nt(wv.view(1, Cc, Cc), hn.view(Z, S, Cc), ld_out=Sp)                     # [Z, C, Sp]
        if ops.flash_attention_supported(q, vT):
            # one launch, online softmax: no [Z, S, S] scores in memory (1 GiB per latent frame at 1024 x 1024 input)
            o = ops.flash_attention(q, k, vT, bv, float(Cc) ** -0.5).view(B, T, H, W, Cc)
        else:
            s = ops.gemm_nt(q, k, out_dtype=torch.float32)                                     # [Z, S, S]
            p = ops.softmax_rows(s, float(Cc) ** -0.5, dt, ld_out=Sp)                          # [Z, S, Sp]
            o = ops.gemm_nt(p, vT, bias=bv).view(B, T, H, W, Cc)
        return _wrap(self.proj_out.run(o, dt, res=x, res_mode=L.VT_RES_ADD, **_emit(next_norm)), next_norm)


def first_norm_of(stage, dt=None):
    """(LayerNorm, silu) a stage wants applied to its input by its producer, or None (resamplers, fused temporal blocks)"""
    if not hasattr(stage, "first_norm"):
        return None
    if isinstance(stage, ResnetCausalBlock1D):
        return stage.first_norm(dt)
    return stage.first_norm()


def run_stages(stages, h, dt, last_norm=None, first=None, switch=None):
    """Run the blocks in order; every block is told which norm its consumer starts with, so the conv that writes the
    activation can emit that norm as well (ops.conv `ln=`).  `h` may already come with stages[0]'s norm (`first`).
    `switch` = (i, dtype): stages[i:] (and `last_norm`) run in that storage / arithmetic type -- the activation is
    converted once, un-normalised, in front of stages[i] (i == len(stages): in front of `last_norm`'s consumer)."""
    if first is not None and not isinstance(h, Normed):
        h = _wrap(h, first)
    for i, stage in enumerate(stages):
        if switch is not None and i == switch[0]:
            h, dt = plain(h).to(switch[1]), switch[1]
        if switch is not None and i + 1 == switch[0]:
            nxt = None                               # the consumer works in another type: it runs its own norm
        else:
            nxt = first_norm_of(stages[i + 1], dt) if i + 1 < len(stages) else last_norm
        h = stage.run(h, dt, next_norm=nxt)
    if switch is not None and switch[0] == len(stages):
        h = plain(h).to(switch[1])
    return h


def _level_module():
    m = nn.Module()
    m.block = nn.ModuleList()
    m.attn = nn.ModuleList()
    return m


class EncoderCausal3DPadding(nn.Module):
    """EncoderCausal3D + EncoderCausal3DPadding of the reference (model_3dcausal.py:502-689,
    v1.1 model_3dcausal_v1_1.py:745-760).  forward(x NCTHW fp32) -> h NCTHW fp32."""

    version = "v1_0"

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), spatial_ds=None, tempo_ds=None, num_res_blocks,
                 dropout=0.0, resamp_with_conv=True, in_channels, z_channels, double_z=True,
                 norm_type="layernorm", **ignore_kwargs):
        super().__init__()
        _check_norm(norm_type)
        v = self.version
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.in_channels = in_channels
        self.norm_type = norm_type
        self.fix_encoder = ignore_kwargs.get("fix_encoder", False)
        self.is_causal = True
        self.time_downsample_factor = ignore_kwargs.get("time_downsample_factor", 4)
        self.init_pad_mode = ignore_kwargs.get("init_pad_mode", "replicate")
        assert self.init_pad_mode in ("constant", "replicate", "reflect")          # pad_at_dim, model_3dcausal.py:37-43
        self.time_padding = self.time_downsample_factor - 1
        self.out_channels = 2 * z_channels if double_z else z_channels
        self.compute_dtype = torch.float32
        # mixed precision (AutoencodingEngine.set_compute_dtype(..., encoder_tail=)): the levels from `tail_level` on, the
        # mid section and conv_out run in `tail_dtype`; tail_level == num_resolutions: mid + conv_out only
        self.tail_dtype, self.tail_level = None, None

        self.conv_in = CausalConv3d(in_channels, ch, 3, version=v)
        in_ch_mult = (1,) + tuple(ch_mult)
        self.spatial_ds = list(range(0, self.num_resolutions - 1)) if spatial_ds is None else list(spatial_ds)
        self.tempo_ds = ([self.num_resolutions - 2, self.num_resolutions - 3] if tempo_ds is None else list(tempo_ds))
        self.down, self.down_temporal = nn.ModuleList(), nn.ModuleList()
        block_in = ch
        for i_level in range(self.num_resolutions):
            block_in = ch * in_ch_mult[i_level]
            block_out = ch * ch_mult[i_level]
            down, down_t = _level_module(), _level_module()
            for _ in range(num_res_blocks):
                down.block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, norm_type=norm_type))
                down_t.block.append(ResnetCausalBlock1D(in_channels=block_out, out_channels=block_out,
                                                        zero_init=True, norm_type=norm_type, version=v))
                block_in = block_out
            if i_level in self.spatial_ds:
                down.downsample = Downsample(block_in, resamp_with_conv)
                if i_level in self.tempo_ds:
                    down_t.downsample = TimeDownsampleResCausal2x(block_in, block_in, version=v)
            self.down.append(down)
            self.down_temporal.append(down_t)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetCausalBlock(in_channels=block_in, out_channels=block_in, norm_type=norm_type, version=v)
        self.mid.attn_1 = AttnBlockWrapper(block_in, norm_type=norm_type, version=v)
        self.mid.block_2 = ResnetCausalBlock(in_channels=block_in, out_channels=block_in, norm_type=norm_type, version=v)
        self.norm_out = Normalize(block_in, norm_type)
        self.conv_out = CausalConv3d(block_in, self.out_channels, 3, version=v)
        if self.fix_encoder:
            for p in self.parameters():
                p.requires_grad = False

    def _front_pad(self, T):
        f = self.time_downsample_factor
        if T % f == 0:
            return 0
        return self.time_padding if self.version == "v1_0" else f - T % f

    @torch.no_grad()
    def forward(self, x):
        assert x.dim() == 5, "input should be 5D tensor, but got {}D tensor".format(x.dim())
        dt = self.compute_dtype
        npad = self._front_pad(x.shape[2])
        if npad and self.init_pad_mode != "replicate":
            # no shipped config uses these modes: the padded clip is assembled frame by frame, then converted unpadded
            xc = x.contiguous().float()
            xp = torch.zeros((x.shape[0], x.shape[1], x.shape[2] + npad) + tuple(x.shape[3:]), dtype=torch.float32, device=x.device)
            ops.ncthw_copy_frames(xc, xp, 0, npad, x.shape[2])
            if self.init_pad_mode == "reflect":                       # x[npad], ..., x[1] in front
                for i in range(npad):
                    ops.ncthw_copy_frames(xc, xp, npad - i, i, 1)
            h = ops.ncthw_to_ndhwc(xp, dt, tpad=0)
        else:
            h = ops.ncthw_to_ndhwc(x.contiguous().float(), dt, tpad=npad)
        stages, level_start = [], []
        for i_level in range(self.num_resolutions):
            level_start.append(len(stages))
            for i_block in range(self.num_res_blocks):
                stages += [self.down[i_level].block[i_block], self.down_temporal[i_level].block[i_block]]
            if i_level in self.spatial_ds:
                stages.append(self.down[i_level].downsample)
                if i_level in self.tempo_ds:
                    stages.append(self.down_temporal[i_level].downsample)
        level_start.append(len(stages))
        stages += [self.mid.block_1, self.mid.attn_1, self.mid.block_2]
        switch = None
        if self.tail_dtype is not None and self.tail_dtype != dt:
            switch = (level_start[self.tail_level], self.tail_dtype)
        h = run_stages(stages, self.conv_in.run(h, dt, **_emit(first_norm_of(stages[0], dt))), dt,
                       last_norm=(self.norm_out, True), first=first_norm_of(stages[0], dt), switch=switch)
        if switch is not None:
            dt = switch[1]
        h = self.norm_out.apply_ndhwc(h, True, dt, SITE_FRAME)
        return self.conv_out.run(h, dt, out_layout=L.VT_NCTHW)


class DecoderCausal3DPadding(nn.Module):
    """DecoderCausal3D + DecoderCausal3DPadding (model_3dcausal.py:692-885, v1.1
    model_3dcausal_v1_1.py:767-959).  forward(z NCTHW fp32) -> x_hat NCTHW fp32."""

    version = "v1_0"

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), spatial_us=None, tempo_us=None, num_res_blocks,
                 dropout=0.0, resamp_with_conv=True, in_channels, z_channels, give_pre_end=False, tanh_out=False,
                 norm_type="layernorm", **ignorekwargs):
        super().__init__()
        _check_norm(norm_type)
        self.give_pre_end, self.tanh_out = give_pre_end, tanh_out     # model_3dcausal.py:862-869; no shipped config sets them
        v = self.version
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.in_channels = in_channels
        self.out_ch = out_ch
        self.norm_type = norm_type
        self.fix_decoder = ignorekwargs.get("fix_decoder", False)
        self.interpolation_mode = ignorekwargs.get("interpolation_mode", "nearest") if v == "v1_1" else "nearest"
        assert self.interpolation_mode in ["nearest", "trilinear"]
        self.time_downsample_factor = ignorekwargs.get("time_downsample_factor", 4)
        self.time_padding = self.time_downsample_factor - 1
        self.compute_dtype = torch.float32

        block_in = ch * ch_mult[self.num_resolutions - 1]
        self.conv_in = CausalConv3d(z_channels, block_in, 3, version=v)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetCausalBlock(in_channels=block_in, out_channels=block_in, norm_type=norm_type, version=v)
        self.mid.attn_1 = AttnBlockWrapper(block_in, norm_type=norm_type, version=v)
        self.mid.block_2 = ResnetCausalBlock(in_channels=block_in, out_channels=block_in, norm_type=norm_type, version=v)

        self.spatial_us = list(range(1, self.num_resolutions)) if spatial_us is None else list(spatial_us)
        self.tempo_us = [1, 2] if tempo_us is None else list(tempo_us)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            up = _level_module()
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                up.block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, norm_type=norm_type))
                block_in = block_out
            if i_level in self.spatial_us:
                up.upsample = Upsample(block_in, resamp_with_conv)
            self.up.insert(0, up)
        self.up_temporal = nn.ModuleList()
        num_temp_upsample = 1
        for i_level in reversed(range(self.num_resolutions)):
            up_t = _level_module()
            c = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                up_t.block.append(ResnetCausalBlock1D(in_channels=c, out_channels=c, zero_init=True,
                                                      norm_type=norm_type, version=v))
            if i_level in self.tempo_us:
                up_t.upsample = TimeUpsampleResCausal2x(c, c, interpolation_mode=self.interpolation_mode,
                                                        num_temp_upsample=num_temp_upsample, version=v)
                num_temp_upsample *= 2
            self.up_temporal.insert(0, up_t)
        self.norm_out = Normalize(block_in, norm_type)
        self.conv_out = CausalConv3d(block_in, out_ch, 3, version=v)
        if self.fix_decoder:
            for p in self.parameters():
                p.requires_grad = False

    def get_last_layer(self, **kwargs):
        return self.conv_out.conv.weight

    @torch.no_grad()
    def forward(self, z):
        dt = self.compute_dtype
        h = ops.ncthw_to_ndhwc(z.contiguous().float(), dt)
        stages = [self.mid.block_1, self.mid.attn_1, self.mid.block_2]
        for i_level in reversed(range(self.num_resolutions)):
            for i_block in range(self.num_res_blocks + 1):
                stages += [self.up[i_level].block[i_block], self.up_temporal[i_level].block[i_block]]
            if i_level in self.spatial_us:
                stages.append(self.up[i_level].upsample)
                if i_level in self.tempo_us:
                    stages.append(self.up_temporal[i_level].upsample)
        h = run_stages(stages, self.conv_in.run(h, dt, **_emit(first_norm_of(stages[0], dt))), dt,
                       last_norm=(self.norm_out, True), first=first_norm_of(stages[0], dt))
        trim = self.time_padding if self.version == "v1_0" else 0
        if self.give_pre_end:
            hp = plain(h)
            return ops.ndhwc_to_ncthw(hp, self.up[0].block[-1].out_channels, ttrim=trim)
        h = self.norm_out.apply_ndhwc(h, True, dt, SITE_FRAME)
        y = self.conv_out.run(h, dt, out_layout=L.VT_NCTHW, t_trim=trim)
        return ops.tanh_(y) if self.tanh_out else y


class EncoderCausal3DPaddingV11(EncoderCausal3DPadding):
    version = "v1_1"


class DecoderCausal3DPaddingV11(DecoderCausal3DPadding):
    version = "v1_1"
