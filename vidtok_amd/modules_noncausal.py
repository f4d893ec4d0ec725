"""Host-side mirror of the reference's NON-causal tokenizer (vidtok/modules/model_3dnoncausal.py): same module tree
and state_dict keys (published non-causal checkpoints load unchanged), every operator on the HIP kernels of
libvidtok_amd.so.  SURVEY.md section 8(f) rank 2: "same kernels with symmetric time padding and plain Conv3d".

What differs from the causal family (vidtok_amd/modules.py):
  * every temporal tap window is centred: Conv3d / Conv1d with padding 1 read one frame before and one after, both
    zero outside the clip -- `ConvGeom(pt=1, pt_hi=1)`, the kernel returns 0 for taps beyond either end;
  * TimeDownsampleRes2x pads ONE zero frame after the clip (avg-pool and the stride-2 conv both see [x, 0]);
  * TimeUpsampleRes2x repeats every frame (no first-frame special case), and the decoder drops nothing;
  * the encoder pads nothing in front: T must be a multiple of the temporal compression factor.
The spatial blocks (ResnetBlock, Upsample, Downsample) are the same classes as in the causal family.
"""
import functools

import torch
import torch.nn as nn

from . import lib as L
from . import ops
from .modules import (SITE_CLIP, SITE_PIXEL, Downsample, Normalize, Normed, ResnetBlock, Upsample, _check_norm, _emit,
                      _level_module, _wrap, first_norm_of, plain, run_stages)
from .ops import ConvGeom
from .packing import PackedCache, time_upsample_parity_mix, time_upsample_parity_weights


class _Conv3dSym:
    """Runs a plain nn.Conv3d / nn.Conv1d parameter set with `padding = k // 2` on NDHWC."""

    @staticmethod
    def geom(conv, ups_t=0):
        if isinstance(conv, nn.Conv1d):
            (kt,), (st,), (pt,) = conv.kernel_size, conv.stride, conv.padding
            return ConvGeom(kt=kt, st=st, pt=pt, pt_hi=pt, ups_t=ups_t)
        kt, kh, kw = conv.kernel_size
        st, sh, sw = conv.stride
        pt, ph, pw = conv.padding
        return ConvGeom(kt=kt, kh=kh, kw=kw, st=st, sh=sh, sw=sw, pt=pt, ph=ph, pw=pw, pt_hi=pt, ph_hi=ph, pw_hi=pw,
                        ups_t=ups_t)

    @staticmethod
    def run(conv, pack: PackedCache, x, dt, geom=None, **kw):
        w, b = pack.get(conv.weight, conv.bias, dt, cin_stored=x.shape[-1])
        return ops.conv(x, w, b, geom or _Conv3dSym.geom(conv), cout=conv.out_channels, **kw)


class TimeDownsampleRes2x(nn.Module):
    """alpha * avgpool3(stride 2) + (1 - alpha) * conv3d stride (2,1,1), both over [x, 0]
    (model_3dnoncausal.py:70-90)."""

    def __init__(self, in_channels, out_channels, mix_factor: float = 2.0):
        super().__init__()
        self.conv = nn.Conv3d(in_channels, out_channels, 3, stride=(2, 1, 1), padding=(0, 1, 1))
        self.mix_factor = nn.Parameter(torch.Tensor([mix_factor]))
        self._pack = PackedCache()

    def run(self, x, dt, next_norm=None):
        x = plain(x)
        x1 = ops.time_avgpool3s2(x, L.VT_TPAD_ZERO_BACK)
        g = ConvGeom(kt=3, kh=3, kw=3, st=2, pt=0, pt_hi=1, ph=1, pw=1, ph_hi=1, pw_hi=1)
        return _wrap(_Conv3dSym.run(self.conv, self._pack, x, dt, g, res=x1, res_mode=L.VT_RES_MIX,
                                    mix_factor=self.mix_factor.detach(), **_emit(next_norm)), next_norm)


class TimeUpsampleRes2x(nn.Module):
    """alpha * up(x) + (1 - alpha) * conv3d(up(x)), up = every frame twice (model_3dnoncausal.py:93-115); the
    repetition is folded into the conv's gather and into the time index of the mix operand."""

    def __init__(self, in_channels, out_channels, mix_factor: float = 2.0):
        super().__init__()
        self.conv = nn.Conv3d(in_channels, out_channels, 3, padding=1)
        self.mix_factor = nn.Parameter(torch.Tensor([mix_factor]))
        # centred window over up(x)[t] = x[t >> 1]:  o[2j] = W0 x[j-1] + (W1+W2) x[j],  o[2j+1] = (W0+W1) x[j] + W2 x[j+1]
        self._parity = ((PackedCache(functools.partial(time_upsample_parity_weights, early=False), mix=functools.partial(time_upsample_parity_mix, early=False)),
                         ConvGeom(kt=2, kh=3, kw=3, pt=1, pt_hi=0, ph=1, pw=1, ph_hi=1, pw_hi=1)),
                        (PackedCache(functools.partial(time_upsample_parity_weights, early=True), mix=functools.partial(time_upsample_parity_mix, early=True)),
                         ConvGeom(kt=2, kh=3, kw=3, pt=0, pt_hi=1, ph=1, pw=1, ph_hi=1, pw_hi=1)))

    def run(self, x, dt, next_norm=None):
        x = plain(x)
        B, T, H, W, C = x.shape
        cout = self.conv.out_channels
        ld = ops.pad_channels(cout)
        y = (torch.empty if ld == cout else torch.zeros)((B, 2 * T, H, W, ld), dtype=dt, device=x.device)
        # the consumer's LayerNorm from the two launches' epilogues where they can take it (modules.TimeUpsampleResCausal2x.run)
        emit = dict(_emit(next_norm))
        n = None

        def alloc_n():               # allocated by ops.conv only once vt_conv_plan says the launch emits the LayerNorm (pad lanes as y's)
            return (torch.empty if ld == cout else torch.zeros)(y.shape, dtype=dt, device=x.device)

        for par, (pack, g) in enumerate(self._parity):      # two k=2 convs with pre-summed taps: 2/3 of the MACs
            w, b = pack.get(self.conv.weight, self.conv.bias, dt, cin_stored=C)
            r = ops.conv(x, w, b, g, cout=cout, res=x, res_mode=L.VT_RES_MIX, mix_factor=self.mix_factor.detach(),
                         out=y, out_t=(2, par), **(dict(emit, ln_out=(alloc_n if n is None else n), ln_optional=True) if emit else {}))
            if emit and not isinstance(r, tuple):
                emit, n = {}, None
            elif emit:
                n = r[1]
        return y if n is None else Normed(y, n, next_norm[0], next_norm[1])


class _ResnetSym(nn.Module):
    """LN-SiLU-conv-LN-SiLU-conv + x with centred convs; `make_conv(cin, cout, k)` builds the conv type."""

    site = SITE_CLIP   # view the reference hands to this block's norms (GroupNorm statistics)

    def __init__(self, make_conv, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=0,
                 zero_init=False, use_checkpoint=False, norm_type="layernorm"):
        super().__init__()
        assert temb_channels == 0 and not conv_shortcut
        out_channels = in_channels if out_channels is None else out_channels
        if in_channels != out_channels:
            # never instantiated by a config; the 3-D variant's 1x1x1 shortcut carries padding=1 in the reference
            # (model_3dnoncausal.py:281), which would not even preserve the shape
            raise NotImplementedError("non-causal temporal / 3-D blocks always keep the channel count")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = Normalize(in_channels, norm_type)
        self.conv1 = make_conv(in_channels, out_channels)
        self.norm2 = Normalize(out_channels, norm_type)
        self.conv2 = make_conv(out_channels, out_channels)
        if zero_init:
            self.conv2.weight.data.zero_()
            self.conv2.bias.data.zero_()
        self._p1, self._p2 = PackedCache(), PackedCache()

    def first_norm(self):
        return (self.norm1, True)

    def run(self, x, dt, next_norm=None):
        h = self.norm1.apply_ndhwc(x, True, dt, self.site)
        x = plain(x)
        h = self.norm2.after(lambda **kw: _Conv3dSym.run(self.conv1, self._p1, h, dt, **kw), True, dt, self.site)
        return _wrap(_Conv3dSym.run(self.conv2, self._p2, h, dt, res=x, res_mode=L.VT_RES_ADD, **_emit(next_norm)),
                     next_norm)


class ResnetBlock1D(_ResnetSym):
    """Temporal block on the "(b h w) c t" view of the reference = taps along T on NDHWC; conv2 zero-initialised
    (model_3dnoncausal.py:182-248)."""

    site = SITE_PIXEL

    def __init__(self, **kw):
        super().__init__(lambda ci, co: nn.Conv1d(ci, co, kernel_size=3, stride=1, padding=1), **kw)


class ResnetNoncausalBlock(_ResnetSym):
    """3-D block of the mid section (model_3dnoncausal.py:251-311)."""

    def __init__(self, **kw):
        kw.pop("zero_init", None)
        super().__init__(lambda ci, co: nn.Conv3d(ci, co, kernel_size=3, stride=1, padding=1), **kw)


class AttnBlockWrapper(nn.Module):
    """Per-frame spatial self-attention with plain 1x1x1 Conv3d projections (model_3dnoncausal.py:17-34, AttnBlock of
    model_3dcausal.py:83-118); same GEMM / softmax kernels as the causal wrapper."""

    def __init__(self, in_channels, use_checkpoint=False, norm_type="layernorm"):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels, norm_type)
        self.q = nn.Conv3d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.k = nn.Conv3d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.v = nn.Conv3d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.proj_out = nn.Conv3d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self._pq, self._pk, self._pv, self._po = PackedCache(), PackedCache(), PackedCache(pin_native=True), PackedCache()

    def first_norm(self):
        return (self.norm, False)

    def run(self, x, dt, next_norm=None):
        hn = self.norm.apply_ndhwc(x, False, dt, SITE_CLIP)
        x = plain(x)
        B, T, H, W, Cc = x.shape
        S, Z = H * W, B * T
        q = _Conv3dSym.run(self.q, self._pq, hn, dt).view(Z, S, Cc)
        k = _Conv3dSym.run(self.k, self._pk, hn, dt).view(Z, S, Cc)
        wv, bv = self._pv.get(self.v.weight, self.v.bias, dt, cin_stored=Cc)
        Sp = ops.pad_channels(S)
        vT = ops.gemm_nt(wv.view(1, Cc, Cc), hn.view(Z, S, Cc), ld_out=Sp)
        if ops.flash_attention_supported(q, vT):
            o = ops.flash_attention(q, k, vT, bv, float(Cc) ** -0.5).view(B, T, H, W, Cc)
        else:
            s = ops.gemm_nt(q, k, out_dtype=torch.float32)
            p = ops.softmax_rows(s, float(Cc) ** -0.5, dt, ld_out=Sp)
            o = ops.gemm_nt(p, vT, bias=bv).view(B, T, H, W, Cc)
        return _wrap(_Conv3dSym.run(self.proj_out, self._po, o, dt, res=x, res_mode=L.VT_RES_ADD, **_emit(next_norm)),
                     next_norm)


class Encoder3D(nn.Module):
    """model_3dnoncausal.py:314-482.  forward(x NCTHW fp32, T a multiple of the temporal factor) -> h NCTHW fp32."""

    def __init__(self, *, ch, out_ch=8, ch_mult=(1, 2, 4, 8), num_res_blocks, dropout=0.0, resamp_with_conv=True,
                 in_channels, z_channels, double_z=True, norm_type="layernorm", **ignore_kwargs):
        super().__init__()
        _check_norm(norm_type)
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.in_channels = in_channels
        self.norm_type = norm_type
        self.fix_encoder = ignore_kwargs.get("fix_encoder", False)
        self.time_downsample_factor = ignore_kwargs.get("time_downsample_factor", 4)
        self.tempo_ds = [self.num_resolutions - 2, self.num_resolutions - 3]
        self.is_causal = False
        self.out_channels = 2 * z_channels if double_z else z_channels
        self.compute_dtype = torch.float32

        self.conv_in = nn.Conv3d(in_channels, ch, kernel_size=3, stride=1, padding=1)
        in_ch_mult = (1,) + tuple(ch_mult)
        self.down, self.down_temporal = nn.ModuleList(), nn.ModuleList()
        block_in = ch
        for i_level in range(self.num_resolutions):
            block_in = ch * in_ch_mult[i_level]
            block_out = ch * ch_mult[i_level]
            down, down_t = _level_module(), _level_module()
            for _ in range(num_res_blocks):
                down.block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, norm_type=norm_type))
                down_t.block.append(ResnetBlock1D(in_channels=block_out, out_channels=block_out, zero_init=True,
                                                  norm_type=norm_type))
                block_in = block_out
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
                if i_level in self.tempo_ds:
                    down_t.downsample = TimeDownsampleRes2x(block_in, block_in)
            self.down.append(down)
            self.down_temporal.append(down_t)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetNoncausalBlock(in_channels=block_in, out_channels=block_in, norm_type=norm_type)
        self.mid.attn_1 = AttnBlockWrapper(block_in, norm_type=norm_type)
        self.mid.block_2 = ResnetNoncausalBlock(in_channels=block_in, out_channels=block_in, norm_type=norm_type)
        self.norm_out = Normalize(block_in, norm_type)
        self.conv_out = nn.Conv3d(block_in, self.out_channels, kernel_size=3, stride=1, padding=1)
        self._pin, self._pout = PackedCache(), PackedCache()
        if self.fix_encoder:
            for p in self.parameters():
                p.requires_grad = False

    @torch.no_grad()
    def forward(self, x):
        assert x.dim() == 5, "input should be 5D tensor, but got {}D tensor".format(x.dim())
        if x.shape[1] == 4 and self.conv_in.in_channels == 3:
            raise ValueError("Mismatched number of input channels")
        dt = self.compute_dtype
        h = ops.ncthw_to_ndhwc(x.contiguous().float(), dt)
        stages = []
        for i_level in range(self.num_resolutions):
            for i_block in range(self.num_res_blocks):
                stages += [self.down[i_level].block[i_block], self.down_temporal[i_level].block[i_block]]
            if i_level != self.num_resolutions - 1:
                stages.append(self.down[i_level].downsample)
                if i_level in self.tempo_ds:
                    stages.append(self.down_temporal[i_level].downsample)
        stages += [self.mid.block_1, self.mid.attn_1, self.mid.block_2]
        first = first_norm_of(stages[0])
        h = run_stages(stages, _Conv3dSym.run(self.conv_in, self._pin, h, dt, **_emit(first)), dt,
                       last_norm=(self.norm_out, True), first=first)
        h = self.norm_out.apply_ndhwc(h, True, dt, SITE_CLIP)
        return _Conv3dSym.run(self.conv_out, self._pout, h, dt, out_layout=L.VT_NCTHW)


class Decoder3D(nn.Module):
    """model_3dnoncausal.py:485-651.  forward(z NCTHW fp32) -> x_hat NCTHW fp32 (every frame is returned)."""

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, dropout=0.0, resamp_with_conv=True,
                 in_channels=8, z_channels, give_pre_end=False, tanh_out=False, norm_type="layernorm", **ignorekwargs):
        super().__init__()
        _check_norm(norm_type)
        if give_pre_end or tanh_out:
            raise NotImplementedError("give_pre_end / tanh_out are never set by a VidTok config")
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.in_channels = in_channels
        self.out_ch = out_ch
        self.norm_type = norm_type
        self.fix_decoder = ignorekwargs.get("fix_decoder", False)
        self.tempo_us = [1, 2]
        self.compute_dtype = torch.float32

        block_in = ch * ch_mult[self.num_resolutions - 1]
        self.conv_in = nn.Conv3d(z_channels, block_in, kernel_size=3, stride=1, padding=1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetNoncausalBlock(in_channels=block_in, out_channels=block_in, norm_type=norm_type)
        self.mid.attn_1 = AttnBlockWrapper(block_in, norm_type=norm_type)
        self.mid.block_2 = ResnetNoncausalBlock(in_channels=block_in, out_channels=block_in, norm_type=norm_type)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            up = _level_module()
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                up.block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, norm_type=norm_type))
                block_in = block_out
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
            self.up.insert(0, up)
        self.up_temporal = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            up_t = _level_module()
            c = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                up_t.block.append(ResnetBlock1D(in_channels=c, out_channels=c, zero_init=True, norm_type=norm_type))
            if i_level in self.tempo_us:
                up_t.upsample = TimeUpsampleRes2x(c, c)
            self.up_temporal.insert(0, up_t)
        self.norm_out = Normalize(block_in, norm_type)
        self.conv_out = nn.Conv3d(block_in, out_ch, kernel_size=3, stride=1, padding=1)
        self._pin, self._pout = PackedCache(), PackedCache()
        if self.fix_decoder:
            for p in self.parameters():
                p.requires_grad = False

    def get_last_layer(self, **kwargs):
        return self.conv_out.weight

    @torch.no_grad()
    def forward(self, z):
        dt = self.compute_dtype
        h = ops.ncthw_to_ndhwc(z.contiguous().float(), dt)
        stages = [self.mid.block_1, self.mid.attn_1, self.mid.block_2]
        for i_level in reversed(range(self.num_resolutions)):
            for i_block in range(self.num_res_blocks + 1):
                stages += [self.up[i_level].block[i_block], self.up_temporal[i_level].block[i_block]]
            if i_level != 0:
                stages.append(self.up[i_level].upsample)
                if i_level in self.tempo_us:
                    stages.append(self.up_temporal[i_level].upsample)
        first = first_norm_of(stages[0])
        h = run_stages(stages, _Conv3dSym.run(self.conv_in, self._pin, h, dt, **_emit(first)), dt,
                       last_norm=(self.norm_out, True), first=first)
        h = self.norm_out.apply_ndhwc(h, True, dt, SITE_CLIP)
        return _Conv3dSym.run(self.conv_out, self._pout, h, dt, out_layout=L.VT_NCTHW)
