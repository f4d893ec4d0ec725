"""TEST INFRASTRUCTURE ONLY -- plain PyTorch fp32 statements of the operator contracts of
include/vidtok_amd.h, with the same Python signatures as vidtok_amd.ops.

Two uses:
  * `-m gpu` numerics tests: every HIP kernel is compared with its function here on the same inputs;
  * `-m "not gpu"` host-logic tests: `patch_ops(monkeypatch)` swaps these in for vidtok_amd.ops so the
    orchestration in vidtok_amd.modules / engine (padding bookkeeping, caches, tiling, state_dict
    packing) is exercised end-to-end against the oracle on a machine without a GPU.
The product never imports this file and has no switch to select it.
"""
import torch
import torch.nn.functional as F

from vidtok_amd import lib as L
from vidtok_amd.ops import ConvGeom, pad_channels  # noqa: F401  (dataclass / helper only)


def _round_like(t, dtype):
    return t.to(dtype)


def _conv_rows(x, w, bias, geom, cout, tmode, cache):
    """the convolution proper on plain fp32 rows: padded NCTHW input, one F.conv3d"""
    Cin = x.shape[-1]
    xp = x.float().permute(0, 4, 1, 2, 3)  # NCTHW
    if geom.ups_t:
        xp = xp.repeat_interleave(2, dim=2)
    if geom.ups_s:
        xp = xp.repeat_interleave(2, dim=3).repeat_interleave(2, dim=4)
    if geom.pt > 0:
        if tmode == L.VT_TPAD_ZERO:
            front = torch.zeros_like(xp[:, :, :1]).repeat(1, 1, geom.pt, 1, 1)
        elif tmode == L.VT_TPAD_REPLICATE:
            front = xp[:, :, :1].repeat(1, 1, geom.pt, 1, 1)
        else:
            assert not geom.ups_t and not geom.ups_s
            front = cache.float().permute(0, 4, 1, 2, 3)[:, :, -geom.pt:]
        xp = torch.cat([front, xp], dim=2)
    xp = F.pad(xp, (geom.pw, geom.pw_hi, geom.ph, geom.ph_hi, 0, geom.pt_hi))
    w5 = w.float().reshape(cout, geom.kt, geom.kh, geom.kw, Cin).permute(0, 4, 1, 2, 3)
    y = F.conv3d(xp, w5, None if bias is None else bias.float()[:cout], stride=(geom.st, geom.sh, geom.sw))
    return y


def conv(x, w, bias, geom, *, cout, out_dtype=None, tmode=L.VT_TPAD_ZERO, cache=None, res=None,
         res_mode=L.VT_RES_NONE, res_tshift=0, mix_factor=None, out_layout=L.VT_NDHWC, t_trim=0, ldy=None,
         ln=None, ln_keep_y=True, out=None, ln_out=None, out_t=None, out_s=None, ln_optional=False):
    if ln_optional:                # "emit this LayerNorm if the launch's epilogue can": the host statement answers no (option conv_tup_ln off)
        ln, ln_out = None, None
    if out_s is not None:          # this launch fills pixels (2ho+py, 2wo+px) of a preallocated tensor
        yv = conv(x, w, bias, geom, cout=cout, out_dtype=out_dtype, tmode=tmode, cache=cache, res=res, res_mode=res_mode,
                  res_tshift=res_tshift, mix_factor=mix_factor, ldy=out.shape[4])
        out[:, :, out_s[0]::2, out_s[1]::2] = yv
        return out
    if out_t is not None:          # this launch fills frames to*mul + off of preallocated tensors
        mul, off = out_t
        res_ = conv(x, w, bias, geom, cout=cout, out_dtype=out_dtype, tmode=tmode, cache=cache, res=res, res_mode=res_mode,
                    res_tshift=res_tshift, mix_factor=mix_factor, ldy=out.shape[4], ln=ln, ln_keep_y=True if ln else ln_keep_y)
        yv, nv = res_ if ln is not None else (res_, None)
        out[:, off::mul] = yv
        if ln is None:
            return out
        ln_out[:, off::mul] = nv
        return (out, ln_out) if ln_keep_y else ln_out
    B, Ti, Hi, Wi, Cin = x.shape
    out_dtype = out_dtype or x.dtype
    if w.dtype == torch.int32:        # packing.SPLIT3_DTYPE
        # VT_BF16X3 (include/vidtok_amd.h): rows of [hi 16 x bf16 | lo 16 x bf16] per 16 k; in fp32 terms the contract is
        # conv(x_lo, w_hi) + conv(x_hi, w_lo) + conv(x_hi, w_hi) with x_hi = bf16(x), x_lo = bf16(x - x_hi)
        K = geom.kt * geom.kh * geom.kw * Cin
        planes = w.contiguous().view(torch.bfloat16).reshape(cout, -1, 2, 16).float()
        w_hi, w_lo = planes[:, :, 0].reshape(cout, -1)[:, :K], planes[:, :, 1].reshape(cout, -1)[:, :K]

        def split(t):
            if t is None:
                return None, None
            hi = t.to(torch.bfloat16).float()
            return hi, (t.float() - hi).to(torch.bfloat16).float()

        (x_hi, x_lo), (c_hi, c_lo) = split(x), split(cache)
        y = _conv_rows(x_lo, w_hi, None, geom, cout, tmode, c_lo) + _conv_rows(x_hi, w_lo, None, geom, cout, tmode, c_hi)
        y = y + _conv_rows(x_hi, w_hi, bias, geom, cout, tmode, c_hi)
    else:
        y = _conv_rows(x, w, bias, geom, cout, tmode, cache)
    To = y.shape[2]
    if res_mode != L.VT_RES_NONE:
        r = res.float().permute(0, 4, 1, 2, 3)[:, :cout]
        tidx = torch.arange(To, device=y.device) >> res_tshift
        r = r[:, :, tidx]
        if res_mode == L.VT_RES_ADD:
            y = r + y
        else:
            a = torch.sigmoid(mix_factor.float())
            y = a * r + (1 - a) * y
    if out_layout == L.VT_NCTHW:
        return y[:, :, t_trim:].contiguous()
    ldy = ldy or pad_channels(cout)
    out = torch.zeros((B, To, y.shape[3], y.shape[4], ldy), dtype=out_dtype, device=x.device)
    out[..., :cout] = y.permute(0, 2, 3, 4, 1).to(out_dtype)
    if ln is None:
        return out
    gamma, beta, eps, silu = ln          # statistics of the fp32 result (the fused kernel normalises before rounding)
    nrm = F.layer_norm(y.permute(0, 2, 3, 4, 1), (cout,), gamma.float()[:cout], beta.float()[:cout], eps)
    if silu:
        nrm = nrm * torch.sigmoid(nrm)
    n = torch.zeros_like(out)
    n[..., :cout] = nrm.to(out_dtype)
    return (out, n) if ln_keep_y else n


def gemm_nt(a, b, *, out_dtype=None, bias=None, ld_out=None):
    out_dtype = out_dtype or a.dtype
    y = torch.matmul(a.float(), b.float().transpose(-1, -2))
    if bias is not None:
        y = y + bias.float()[: y.shape[-1]]
    y = y.to(out_dtype)
    if ld_out and ld_out > y.shape[-1]:
        y = F.pad(y, (0, ld_out - y.shape[-1]))
    return y.contiguous()


def layernorm_act(x, gamma, beta, *, silu, eps=1e-6, out_dtype=None, c=None):
    out_dtype = out_dtype or x.dtype
    c = c or x.shape[-1]
    y = F.layer_norm(x.float()[..., :c], (c,), gamma.float(), beta.float(), eps)
    if silu:
        y = y * torch.sigmoid(y)
    out = torch.zeros(x.shape, dtype=out_dtype, device=x.device)
    out[..., :c] = y.to(out_dtype)
    return out


def groupnorm_act(x, gamma, beta, *, scope, silu, eps=1e-6, out_dtype=None, c=None, groups=32):
    out_dtype = out_dtype or x.dtype
    if scope == 3:                   # GN_POS: "(b t) c s" with s = 1
        shp = x.shape
        return groupnorm_act(x.reshape(1, 1, -1, 1, shp[-1]), gamma, beta, scope=L.VT_GN_PIXEL, silu=silu, eps=eps,
                             out_dtype=out_dtype, c=c, groups=groups).reshape(shp)
    B, T, H, W, ld = x.shape
    c = c or ld
    xc = x.float()[..., :c]
    if scope == L.VT_GN_FRAME:       # "(b t) c h w"
        v = F.group_norm(xc.reshape(B * T, H, W, c).permute(0, 3, 1, 2), groups, gamma.float()[:c], beta.float()[:c], eps)
        v = v.permute(0, 2, 3, 1).reshape(B, T, H, W, c)
    elif scope == L.VT_GN_PIXEL:     # "(b h w) c t"
        v = F.group_norm(xc.permute(0, 2, 3, 4, 1).reshape(B * H * W, c, T), groups, gamma.float()[:c], beta.float()[:c], eps)
        v = v.reshape(B, H, W, c, T).permute(0, 4, 1, 2, 3)
    else:                            # "b c t h w"
        v = F.group_norm(xc.permute(0, 4, 1, 2, 3), groups, gamma.float()[:c], beta.float()[:c], eps).permute(0, 2, 3, 4, 1)
    if silu:
        v = v * torch.sigmoid(v)
    out = torch.zeros(x.shape, dtype=out_dtype, device=x.device)
    out[..., :c] = v.to(out_dtype)
    return out


def flash_attention_supported(q, vT):
    return False          # the CPU statement of the attention block is the operator sequence (gemm_nt -> softmax_rows -> gemm_nt)


def flash_attention(q, k, vT, bias_v, scale):
    """softmax(scale * q k^T) v + bias_v, fp32 arithmetic on the stored values; q, k [Z, S, C], vT [Z, C, ld] (keys contiguous)"""
    S = q.shape[1]
    p = torch.softmax(torch.matmul(q.float(), k.float().transpose(1, 2)) * scale, dim=-1)
    o = torch.matmul(p, vT.float()[:, :, :S].transpose(1, 2))
    if bias_v is not None:
        o = o + bias_v.float()[: o.shape[-1]]
    return o.to(q.dtype)


def softmax_rows(s, scale, out_dtype, ld_out=None):
    p = torch.softmax(s.float() * scale, dim=-1).to(out_dtype)
    if ld_out and ld_out > p.shape[-1]:
        p = F.pad(p, (0, ld_out - p.shape[-1]))
    return p.contiguous()


def ncthw_to_ndhwc(x, dtype, tpad=0, ld=None):
    B, C, T, H, W = x.shape
    ld = ld or pad_channels(C)
    if tpad:
        x = torch.cat([x[:, :, :1].repeat(1, 1, tpad, 1, 1), x], dim=2)
    out = torch.zeros((B, T + tpad, H, W, ld), dtype=dtype, device=x.device)
    out[..., :C] = x.permute(0, 2, 3, 4, 1).to(dtype)
    return out


def ndhwc_to_ncthw(x, c, ttrim=0):
    return x[:, ttrim:, :, :, :c].float().permute(0, 4, 1, 2, 3).contiguous()


def time_avgpool3s2(x, tmode=L.VT_TPAD_ZERO, cache=None):
    xf = x.float()
    if tmode == L.VT_TPAD_ZERO_BACK:
        xp = torch.cat([xf, torch.zeros_like(xf[:, :1])], dim=1)
        To = x.shape[1] // 2
        return ((xp[:, 0:2 * To:2] + xp[:, 1:2 * To + 1:2] + xp[:, 2:2 * To + 2:2]) / 3.0).to(x.dtype)
    if tmode == L.VT_TPAD_ZERO:
        front = torch.zeros_like(xf[:, :1])
    elif tmode == L.VT_TPAD_REPLICATE:
        front = xf[:, :1]
    else:
        front = cache.float().reshape(xf[:, :1].shape)
    xp = torch.cat([front, xf], dim=1)
    To = x.shape[1] // 2
    y = (xp[:, 0:2 * To:2] + xp[:, 1:2 * To + 1:2] + xp[:, 2:2 * To + 2:2]) / 3.0
    return y.to(x.dtype)


def time_lerp2x(x, out=None, out_t0=0):
    y = F.interpolate(x.float().permute(0, 4, 1, 2, 3), scale_factor=[2.0, 1.0, 1.0], mode="trilinear")
    y = y.permute(0, 2, 3, 4, 1).to(x.dtype).contiguous()
    if out is None:
        return y
    out[:, out_t0:out_t0 + y.shape[1]] = y
    return out


def time_lerp2x_cat(head, x, skip):
    """torch.cat + F.interpolate + slice (reference model_3dcausal_v1_1.py:331-341)"""
    return time_lerp2x(torch.cat([head, x], dim=1))[:, skip:].contiguous()


def gather_frames(src, idx, out=None, out_t0=0):
    g = src[:, list(idx)].contiguous()
    if out is None:
        return g
    out[:, out_t0:out_t0 + len(idx)] = g
    return out


def kl_sample(h, noise):
    mean, logvar = torch.chunk(h, 2, dim=1)
    logvar = logvar.clamp(-30.0, 20.0)
    z = mean if noise is None else mean + torch.exp(0.5 * logvar) * noise
    kl = 0.5 * torch.sum(mean * mean + torch.exp(logvar) - 1.0 - logvar) / h.shape[0]
    return z.contiguous(), kl


def _fsq_consts(levels):
    lv = torch.tensor(levels, dtype=torch.int32)
    half_l = (lv - 1) * (1 + 1e-3) / 2
    offset = torch.where(lv % 2 == 0, 0.5, 0.0)
    shift = (offset / half_l).atanh()
    basis = torch.cumprod(torch.tensor([1] + list(levels[:-1])), dim=0, dtype=torch.int32)
    return lv, half_l, offset, shift, lv // 2, basis


def fsq_quantize(h, levels, num_codebooks=1):
    lv, half_l, offset, shift, half_w, basis = (t.to(h.device) for t in _fsq_consts(levels))
    c, d = int(num_codebooks), len(levels)
    zf = h.float().movedim(1, -1)
    zf = zf.reshape(zf.shape[:-1] + (c, d))                      # "b n (c d) -> b n c d" (regularizers.py:227)
    codes = ((zf + shift).tanh() * half_l - offset).round() / half_w
    idx = ((codes * half_w + half_w) * basis).sum(-1).to(torch.int32)         # [b, ..., c]
    codes = codes.reshape(codes.shape[:-2] + (c * d,))
    return codes.movedim(-1, 1).contiguous(), (idx if c > 1 else idx[..., 0])


def fsq_indices_to_codes(idx, levels, num_codebooks=1):
    lv, _, _, _, half_w, basis = (t.to(idx.device) for t in _fsq_consts(levels))
    c = int(num_codebooks)
    if c == 1:
        idx = idx[..., None]
    codes = ((idx[..., None] // basis) % lv - half_w) / half_w                # [b, ..., c, d]
    codes = codes.reshape(codes.shape[:-2] + (c * len(levels),))
    return codes.movedim(-1, 1).contiguous().float()


def temporal_block_supported(x, tmode, c=None, caches=None, cache_offset=0):
    return False        # the CPU host-logic tests run the blocks unfused (the fused launch is a GPU kernel)


def temporal_block(x, w1, b1, w2, b2, norm1, norm2, *, tmode=L.VT_TPAD_ZERO, eps=1e-6, next_ln=None, keep_y=True):
    """contract of vt_temporal_block: the unfused sequence on the same operators (intermediates rounded to the
    storage dtype exactly where the unfused HIP path stores them)"""
    c = x.shape[-1]
    g = ConvGeom(kt=3, pt=2)
    h = layernorm_act(x, norm1[0], norm1[1], silu=True, eps=eps)
    h = conv(h, w1, b1, g, cout=c, tmode=tmode, ln=(norm2[0], norm2[1], eps, True), ln_keep_y=False)
    kw = {} if next_ln is None else dict(ln=(next_ln[0], next_ln[1], eps, next_ln[2]), ln_keep_y=keep_y)
    return conv(h, w2, b2, g, cout=c, tmode=tmode, res=x, res_mode=L.VT_RES_ADD, **kw)


def tanh_(x):
    return x.tanh_()


def entropy(avg):
    return (-avg * avg.clamp(min=1e-5).log()).sum()


def fsq_aux_loss(stats3, codebook_entropy, diversity_gamma, entropy_weight, commitment_weight):
    ce = stats3[1] if codebook_entropy is None else codebook_entropy.reshape(())
    return (stats3[0] - diversity_gamma * ce) * entropy_weight + stats3[2] * commitment_weight


def fsq_aux_stats(h, levels, inv_temperature=100.0, return_avg=False):
    lv, _, _, _, half_w, basis = (t.to(h.device) for t in _fsq_consts(levels))
    codes, _ = fsq_quantize(h, levels)
    zf = h.float().movedim(1, -1).reshape(-1, len(levels))
    J = int(torch.prod(lv))
    cb = ((torch.arange(J, device=h.device)[:, None] // basis) % lv - half_w) / half_w
    ent, avg = 0.0, torch.zeros(J, device=h.device)
    for s in range(0, zf.shape[0], 1024):
        p = ((2.0 * zf[s:s + 1024] @ cb.t()) * inv_temperature).softmax(-1)
        ent = ent + (-p * p.clamp(min=1e-5).log()).sum()
        avg = avg + p.sum(0)
    avg = avg / zf.shape[0]
    cbe = (-avg * avg.clamp(min=1e-5).log()).sum()
    commit = ((h.float() - codes) ** 2).mean()
    st = torch.stack([ent / zf.shape[0], cbe, commit])
    return (st, avg) if return_avg else st


def fsq_consts(levels):
    _, half_l, offset, shift, _, basis = _fsq_consts(levels)
    return half_l.tolist(), offset.tolist(), shift.tolist(), [float(b) for b in basis]


def channel_linear(x, w, bias):
    y = torch.einsum("oc,bc...->bo...", w.float(), x.float())
    if bias is not None:
        y = y + bias.float().reshape((1, -1) + (1,) * (x.dim() - 2))
    return y.contiguous()


def eval_psnr_ssim(x, y, raw=True):
    """contract of vt_eval_psnr_ssim: per-frame PSNR / SSIM, [B,T] each (oracle/metrics_oracle.py is the restatement of
    the reference; this is the operator contract the host mirror is tested against on CPU)"""
    from oracle import metrics_oracle as M

    if raw:
        x, y = M.postprocess(x, y)
    return M.psnr_frames(x, y), M.ssim_frames(x, y)


def frames_u8_to_ncthw(frames, resized_hw, crop_top_left, out_hw, out=None, t_off=0):
    """contract of vt_frames_u8_to_ncthw, stated with the oracle's pieces (oracle/video_io_oracle.py)"""
    from oracle import video_io_oracle as V

    x = frames.permute(0, 3, 1, 2).float() / 255.0
    if tuple(resized_hw) != tuple(x.shape[-2:]):
        x = V.resize_aa(x, *resized_hw)
    (top, left), (h, w) = crop_top_left, out_hw
    x = ((x[..., top:top + h, left:left + w] - 0.5) / 0.5).permute(1, 0, 2, 3)
    if out is None:
        return x.unsqueeze(0).contiguous()
    out[0, :, t_off:t_off + x.shape[1]] = x
    return out


def ncthw_to_frames_u8(x, t0=0, n=None, out=None, w_off=0):
    from oracle import video_io_oracle as V

    n = x.shape[2] - t0 if n is None else n
    fr = torch.from_numpy(V.tensor_to_uint8(x[0, :, t0:t0 + n])).permute(1, 2, 3, 0)   # [n, H, W, 3]
    if out is None:
        return fr.contiguous()
    out[:n, :, w_off:w_off + fr.shape[2]] = fr
    return out


def ncthw_copy_frames(src, dst, ts0, td0, n, clamp=False):
    v = src[:, :, ts0:ts0 + n]
    dst[:, :, td0:td0 + n] = v.clamp(-1, 1) if clamp else v
    return dst


ALL = ["conv", "gemm_nt", "layernorm_act", "softmax_rows", "ncthw_to_ndhwc", "ndhwc_to_ncthw", "time_avgpool3s2",
       "time_lerp2x", "time_lerp2x_cat", "gather_frames", "kl_sample", "fsq_quantize", "fsq_indices_to_codes", "fsq_aux_stats", "fsq_aux_loss", "entropy", "tanh_", "temporal_block", "temporal_block_supported", "frames_u8_to_ncthw", "ncthw_to_frames_u8", "ncthw_copy_frames",
       "eval_psnr_ssim", "channel_linear", "groupnorm_act"]


def patch_ops(monkeypatch):
    """Swap the reference statements in for the HIP operators (CPU host-logic tests only)."""
    import vidtok_amd.ops as ops

    g = globals()
    for name in ALL:
        monkeypatch.setattr(ops, name, g[name])
