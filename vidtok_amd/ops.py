"""Torch-tensor facing wrappers of the C-ABI operators (include/vidtok_amd.h).

PyTorch is plumbing here: tensors provide device memory and the current HIP stream; every
arithmetic operation of the path is one of the HIP kernels in vidtok_amd/csrc.  All wrappers
raise if the tensors are not on a GPU -- there is no CPU implementation in the product.

Activation tensors are NDHWC: shape [B, T, H, W, C], contiguous, float32, bfloat16 or float16.
"""
import ctypes as C
from dataclasses import dataclass
from typing import Optional

import torch

from . import lib as L

_DT = {torch.float32: L.VT_F32, torch.bfloat16: L.VT_BF16, torch.float16: L.VT_F16}
H16 = (torch.bfloat16, torch.float16)      # the 16-bit storage types: every kernel written for one takes the other
CH_ALIGN = 8  # channel padding granule: 16 B of bf16 (and a multiple of the 4-float fp32 granule)


# Optional launch record for bench.py's roofline leg: when set to a list, every launch of an MFMA kernel (vt_conv,
# vt_temporal_block) appends (descriptor, tensors it points into -- kept alive --, (pixels, Cout, K)), so the
# matrix-core launches of a step can be replayed on their own (replay_convs).  None in normal use.
CONV_RECORD = None


def _conv_launch(lib, d, what, keep=()):
    L.check(lib.vt_conv(C.byref(d), _stream()), what)
    if CONV_RECORD is not None:
        CONV_RECORD.append((d, keep, (d.B * d.To * d.Ho * d.Wo * max(1, d.nbatch), d.Cout, d.KT * d.KH * d.KW * d.Cin)))


def conv_plan(d):
    """vt_conv_plan(d) -> dict(tile=(BM, BN), waves, workgroups, ln_fused, launches, kernel="igemm" | "ws2" | "narrow" | "in8",
    lds_epilogue, deep_ring)"""
    out = (C.c_int32 * 8)()
    L.check(L.load().vt_conv_plan(C.byref(d), out), "vt_conv_plan")
    return dict(tile=(out[0], out[1]), waves=out[2], workgroups=out[3], ln_fused=bool(out[4]), launches=out[5],
                kernel={2: "narrow", 3: "ws2", 4: "in8"}.get(out[6], "igemm"), lds_epilogue=out[7] == 1, deep_ring=out[7] == 2)


def replay_convs(record, conv_kernel_only=True):
    """Re-issue recorded vt_conv launches on the current stream (same descriptors, same tensors).  With
    conv_kernel_only a descriptor whose LayerNorm is not produced by the conv epilogue is replayed without it (that
    LayerNorm is a separate vt_layernorm_act launch, not conv kernel time)."""
    lib = L.load()
    for d, _keep, _label in record:
        if isinstance(d, L.TBlockDesc):
            L.check(lib.vt_temporal_block(C.byref(d), _stream()), "vt_temporal_block(replay)")
            continue
        if isinstance(d, tuple):                 # ("flash", q, k, vT, bias, o, scale): vt_flash_attention
            _, q, k, vT, bias, o, scale = d
            L.check(lib.vt_flash_attention(_ptr(q), _ptr(k), _ptr(vT), _ptr(bias), _ptr(o), _DT[q.dtype], q.shape[0], q.shape[1], q.shape[2],
                                           vT.shape[2], scale, _stream()), "vt_flash_attention(replay)")
            continue
        if conv_kernel_only and d.ln_mode != 0 and not conv_plan(d)["ln_fused"]:
            d2 = L.ConvDesc()
            C.memmove(C.byref(d2), C.byref(d), C.sizeof(d))
            d2.ln_mode = 0
            d = d2
        L.check(lib.vt_conv(C.byref(d), _stream()), "vt_conv(replay)")


def pad_channels(c: int) -> int:
    return (c + CH_ALIGN - 1) // CH_ALIGN * CH_ALIGN


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise L.VtError(f"{name}: tensor is on {t.device}; vidtok_amd runs on the GPU only (no CPU fallback)")
    if not t.is_contiguous():
        raise L.VtError(f"{name}: tensor must be contiguous")


@dataclass(frozen=True)
class ConvGeom:
    """Static geometry of one convolution (see vt_conv in include/vidtok_amd.h)."""

    kt: int = 1
    kh: int = 1
    kw: int = 1
    st: int = 1
    sh: int = 1
    sw: int = 1
    pt: int = 0        # causal front pad in time (taps that fall before frame 0)
    ph: int = 0        # top / left pad
    pw: int = 0
    ph_hi: int = 0     # bottom / right pad (zeros)
    pw_hi: int = 0
    pt_hi: int = 0     # zero frames after the last one (non-causal convs; the kernel reads 0 beyond the end)
    ups_t: int = 0     # nearest x2 folded into the gather
    ups_s: int = 0

    def out_dims(self, Ti, Hi, Wi):
        Tv, Hv, Wv = Ti << self.ups_t, Hi << self.ups_s, Wi << self.ups_s
        To = (Tv + self.pt + self.pt_hi - self.kt) // self.st + 1
        Ho = (Hv + self.ph + self.ph_hi - self.kh) // self.sh + 1
        Wo = (Wv + self.pw + self.pw_hi - self.kw) // self.sw + 1
        return To, Ho, Wo


def conv(x, w, bias, geom: ConvGeom, *, cout: int, out_dtype=None, tmode=L.VT_TPAD_ZERO, cache=None,
         res=None, res_mode=L.VT_RES_NONE, res_tshift=0, mix_factor=None, out_layout=L.VT_NDHWC,
         t_trim=0, ldy=None, ln=None, ln_keep_y=True, out=None, ln_out=None, out_t=None, out_s=None, ln_optional=False):
    """y = conv(x) (+bias) (+res | alpha-mix); x [B,Ti,Hi,Wi,Cin], w packed [cout, ldw].
    ln = (gamma, beta, eps, silu) additionally returns n = [SiLU](LayerNorm(y)): (y, n), or just n with
    ln_keep_y=False (y is then scratch: the fused kernel never writes it).
    out_t = (mul, off) with out (and ln_out) preallocated [B, To*mul, Ho, Wo, ld]: this launch fills the frames
    to*mul + off (the parity classes of a time up-sampler).  out_s = (py, px) with out [B, To, 2Ho, 2Wo, ld]: this
    launch fills the pixels (2ho+py, 2wo+px) (the parity classes of a spatial up-sampler)."""
    lib = L.load()
    _chk(x, "conv.x"); _chk(w, "conv.w")
    B, Ti, Hi, Wi, Cin = x.shape
    x3 = w.dtype == torch.int32       # split-bf16 weight planes (packing.pack_split3 / SPLIT3_DTYPE): fp32 storage, bf16 MFMA
    assert (w.dtype == x.dtype or (x3 and x.dtype == torch.float32)) and x.dtype in _DT, (w.dtype, x.dtype)
    out_dtype = out_dtype or x.dtype
    To, Ho, Wo = geom.out_dims(Ti, Hi, Wi)
    assert To > 0 and Ho > 0 and Wo > 0, (To, Ho, Wo)
    mul, off = out_t or (1, 0)
    if out is not None:
        assert out_layout == L.VT_NDHWC and out.is_contiguous() and out.dtype == out_dtype
        smul = 2 if out_s is not None else 1
        assert tuple(out.shape[:4]) == (B, To * mul, Ho * smul, Wo * smul), (out.shape, (B, To * mul, Ho, Wo))
        y, ldy = out, out.shape[4]
    elif out_layout == L.VT_NCTHW:
        y = torch.empty((B, cout, To - t_trim, Ho, Wo), dtype=torch.float32, device=x.device)
        out_dtype = torch.float32
        ldy = cout
    else:
        assert mul == 1
        ldy = ldy or pad_channels(cout)
        if ldy != cout:  # keep the pad lanes defined (they feed the next conv's zero weights)
            y = torch.zeros((B, To, Ho, Wo, ldy), dtype=out_dtype, device=x.device)
        else:
            y = torch.empty((B, To, Ho, Wo, ldy), dtype=out_dtype, device=x.device)
    d = L.ConvDesc()
    d.x, d.w, d.bias, d.y = x.data_ptr(), w.data_ptr(), (bias.data_ptr() if bias is not None else None), y.data_ptr()
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() >= cout
    d.B, d.Ti, d.Hi, d.Wi, d.Cin = B, Ti, Hi, Wi, Cin
    d.To, d.Ho, d.Wo, d.Cout = To, Ho, Wo, cout
    d.ldw, d.ldy = w.shape[1], ldy
    assert not x3 or out_dtype == torch.float32
    d.KT, d.KH, d.KW = geom.kt, geom.kh, geom.kw
    d.st, d.sh, d.sw = geom.st, geom.sh, geom.sw
    d.pt, d.ph, d.pw = geom.pt, geom.ph, geom.pw
    d.tmode = tmode
    if cache is not None:
        _chk(cache, "conv.cache")
        assert cache.dtype == x.dtype and cache.shape[0] == B and tuple(cache.shape[2:]) == (Hi, Wi, Cin), cache.shape
        d.cache, d.ncache = cache.data_ptr(), cache.shape[1]
    d.ups_t, d.ups_s = geom.ups_t, geom.ups_s
    d.res_mode = res_mode
    if res_mode != L.VT_RES_NONE:
        _chk(res, "conv.res")
        assert res.dtype == out_dtype, (res.dtype, out_dtype)
        assert res.shape[0] == B and tuple(res.shape[2:4]) == (Ho, Wo) and res.shape[4] >= cout, (res.shape, y.shape)
        d.res, d.res_tshift, d.Tr, d.ldr = res.data_ptr(), res_tshift, res.shape[1], res.shape[4]
        if res_mode == L.VT_RES_MIX:
            assert mix_factor is not None and mix_factor.dtype == torch.float32 and mix_factor.is_cuda
            d.mix_factor = mix_factor.data_ptr()
    d.out_layout, d.t_trim = out_layout, t_trim
    d.dtype, d.out_dtype = (L.VT_BF16X3 if x3 else _DT[x.dtype]), _DT[out_dtype]
    d.nbatch = 1
    d.yt_mul, d.yt_off = mul, off
    if out_s is not None:
        assert out is not None and mul == 1
        d.ys_mul, d.ys_oh, d.ys_ow = 2, int(out_s[0]), int(out_s[1])
    n = None
    if ln is not None:
        gamma, beta, eps, silu = ln
        assert out_layout == L.VT_NDHWC and gamma.dtype == torch.float32 and beta.dtype == torch.float32
        assert gamma.numel() >= cout and beta.numel() >= cout and gamma.is_cuda and beta.is_cuda
        d.ln_gamma, d.ln_beta, d.ln_out = gamma.data_ptr(), beta.data_ptr(), y.data_ptr()      # (ln_out: a placeholder until the plan is known)
        d.ln_mode, d.ln_keep_y, d.ldn, d.ln_eps = (2 if silu else 1), int(bool(ln_keep_y)), ldy, float(eps)
        fused = True
        if ln_optional:              # ask before allocating the twin tensor (a launch that refuses left a full-size buffer behind)
            try:
                fused = conv_plan(d)["ln_fused"]
            except L.VtError:        # "LayerNorm of an interleaved output is only available fused": this launch cannot take it
                fused = False
        if fused:
            if callable(ln_out):     # the caller's allocator, run only now that the launch is known to emit the LayerNorm
                ln_out = ln_out()
            if ln_out is not None:
                assert ln_out.shape == y.shape and ln_out.dtype == out_dtype and ln_out.is_contiguous()
                n = ln_out
            else:                    # pad lanes defined like y's (they meet zero weights in the consumer)
                n = (torch.zeros if ldy != cout else torch.empty)(y.shape, dtype=out_dtype, device=x.device)
            d.ln_out = n.data_ptr()
        else:
            # the LayerNorm of an interleaved output exists only inside an epilogue; this launch's epilogue does not take it
            # (shape, arithmetic, option conv_tup_ln): run without, the caller's consumer normalises y itself
            assert ln_keep_y
            d.ln_gamma = d.ln_beta = d.ln_out = None
            d.ln_mode = 0
            ln, n = None, None
    work = None
    if x.dtype in H16 and (geom.kt == 3 or geom.kh == 3):      # split-K over tap planes (small-M launches): the library says how much scratch
        nb = lib.vt_conv_work_bytes(C.byref(d))
        if nb > 0:
            work = torch.empty((nb,), dtype=torch.uint8, device=x.device)
            d.work, d.work_bytes = work.data_ptr(), nb
    _conv_launch(lib, d, "vt_conv", (x, w, bias, y, res, cache, mix_factor, n, ln, work))
    if ln is None:
        return y
    return (y, n) if ln_keep_y else n


def _tblock_desc(x, tmode, c=None, caches=None, cache_offset=0):
    d = L.TBlockDesc()
    B, T, H, W, ld = x.shape
    d.dtype, d.C, d.ld, d.B, d.T, d.HW, d.tmode = _DT.get(x.dtype, -1), (ld if c is None else c), ld, B, T, H * W, tmode
    if caches is not None:
        for t in caches:
            _chk(t, "tblock.cache")
            assert tuple(t.shape) == (B, 2, H, W, ld) and t.dtype == x.dtype, (tuple(t.shape), tuple(x.shape))
        d.cache1, d.cache2, d.cache_offset = caches[0].data_ptr(), caches[1].data_ptr(), int(cache_offset)
    return d


def temporal_block_supported(x, tmode, c=None, caches=None, cache_offset=0) -> bool:
    """True if vt_temporal_block covers a block of `c` real channels (default: the stored count) on this activation
    (bf16, C = ld = 128 -- a block whose channels are PADDED to 128 is not covered: the statistics span C --, HW % 64 == 0;
    with chunk state -- `caches`, required by tmode VT_TPAD_CACHE -- the clip must keep T - cache_offset >= 3 frames):
    the fused launch for ResnetCausalBlock1D (reference model_3dcausal.py:473-499, v1.1 chunks model_3dcausal_v1_1.py:159-178)."""
    if not x.is_cuda or x.dtype not in _DT:
        return False
    return bool(L.load().vt_temporal_block_supported(C.byref(_tblock_desc(x, tmode, c, caches, cache_offset))))


def temporal_block(x, w1, b1, w2, b2, norm1, norm2, *, tmode=L.VT_TPAD_ZERO, eps=1e-6, next_ln=None, keep_y=True, profile_out=None, c=None,
                   caches=None, cache_offset=0):
    """y = x + conv2(SiLU(LN2(conv1(SiLU(LN1(x)))))) with causal k=3 temporal convs, one launch (vt_temporal_block).
    norm1 / norm2 = (gamma, beta) fp32; w packed [C, 3C].  next_ln = (gamma, beta, silu) additionally returns
    n = [SiLU](LayerNorm(y)): (y, n), or just n with keep_y=False.  caches = (cache1, cache2), [B, 2, H, W, C] each: the
    chunk state of the two convolutions (their inputs at frames -2, -1), read with tmode VT_TPAD_CACHE and rewritten IN
    PLACE with this clip's frames T - cache_offset - 2, T - cache_offset - 1."""
    lib = L.load()
    _chk(x, "tblock.x"); _chk(w1, "tblock.w1"); _chk(w2, "tblock.w2")
    d = _tblock_desc(x, tmode, c, caches, cache_offset)
    y = torch.empty_like(x) if keep_y else None
    n = torch.empty_like(x) if next_ln is not None else None
    d.x, d.y, d.n_out = x.data_ptr(), (y.data_ptr() if keep_y else None), (n.data_ptr() if n is not None else None)
    d.w1, d.w2 = w1.data_ptr(), w2.data_ptr()
    d.b1 = b1.data_ptr() if b1 is not None else None
    d.b2 = b2.data_ptr() if b2 is not None else None
    for t in (norm1[0], norm1[1], norm2[0], norm2[1]):
        assert t.dtype == torch.float32 and t.is_cuda and t.numel() >= x.shape[-1]
    d.norm1_gamma, d.norm1_beta, d.norm2_gamma, d.norm2_beta = (t.data_ptr() for t in (norm1[0], norm1[1], norm2[0], norm2[1]))
    if next_ln is not None:
        assert next_ln[0].dtype == torch.float32 and next_ln[1].dtype == torch.float32
        d.next_gamma, d.next_beta, d.ln_next_mode = next_ln[0].data_ptr(), next_ln[1].data_ptr(), (2 if next_ln[2] else 1)
    d.keep_y, d.eps = int(bool(keep_y)), float(eps)
    if profile_out is not None:       # measurement aid: cycle stamps of workgroup 0 (scripts/tblock_profile.py)
        L.check(lib.vt_temporal_block_profile(C.byref(d), profile_out.data_ptr(), _stream()), "vt_temporal_block_profile")
        return (y, n)
    L.check(lib.vt_temporal_block(C.byref(d), _stream()), "vt_temporal_block")
    if CONV_RECORD is not None:   # two K = 3C convolutions: label (pixels, C, 6C) carries their FLOPs
        CONV_RECORD.append((d, (x, w1, b1, w2, b2, norm1, norm2, next_ln, y, n, caches), (d.B * d.T * d.HW, d.C, 6 * d.C)))
    if next_ln is None:
        return y
    return (y, n) if keep_y else n


def gemm_nt(a, b, *, out_dtype=None, bias=None, ld_out=None):
    """Batched C[z] = A[z] @ B[z]^T : a [Z or 1, M, K], b [Z, N, K] -> [Z, M, N] (the two matmuls of
    scaled_dot_product_attention, reference model_3dcausal.py:140) on the vt_conv kernel.
    a with leading dim 1 is broadcast over Z (stride 0).  ld_out > N returns [Z, M, ld_out] with the
    columns N.. zero (so the result can be the K-contiguous operand of a following gemm_nt)."""
    lib = L.load()
    _chk(a, "gemm.a"); _chk(b, "gemm.b")
    Za, M, K = a.shape
    Z, N, Kb = b.shape
    assert Za in (1, Z) and K == Kb and a.dtype == b.dtype
    out_dtype = out_dtype or a.dtype
    ldo = ld_out or N
    assert ldo >= N
    y = (torch.empty if ldo == N else torch.zeros)((Z, M, ldo), dtype=out_dtype, device=a.device)
    d = L.ConvDesc()
    d.x, d.w, d.y = a.data_ptr(), b.data_ptr(), y.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.B, d.Ti, d.Hi, d.Wi, d.Cin = 1, 1, 1, M, K
    d.To, d.Ho, d.Wo, d.Cout = 1, 1, M, N
    d.ldw, d.ldy = K, ldo
    d.KT = d.KH = d.KW = 1
    d.st = d.sh = d.sw = 1
    d.dtype, d.out_dtype = _DT[a.dtype], _DT[out_dtype]
    d.nbatch = Z
    d.xs_z, d.ws_z, d.ys_z = (M * K if Za == Z else 0), N * K, M * ldo
    _conv_launch(lib, d, "vt_conv(gemm)", (a, b, bias, y))
    return y


def flash_attention_supported(q, vT) -> bool:
    """q [Z, S, C], vT [Z, C, ld]: does vt_flash_attention cover this attention (bf16, C = 512, S % 64 == 0; option attn_flash)"""
    if not q.is_cuda or q.dtype not in _DT:
        return False
    return bool(L.load().vt_flash_attention_supported(_DT[q.dtype], q.shape[1], q.shape[2], vT.shape[2]))


def flash_attention(q, k, vT, bias_v, scale: float):
    """o[z] = softmax(scale * q[z] k[z]^T) v[z] + bias_v in one launch, nothing S x S in memory (vt_flash_attention);
    q, k [Z, S, C], vT = V transposed [Z, C, ld] (keys contiguous) -> o [Z, S, C]"""
    lib = L.load()
    _chk(q, "attn.q"); _chk(k, "attn.k"); _chk(vT, "attn.vT")
    Z, S, Cc = q.shape
    assert k.shape == q.shape and k.dtype == q.dtype == vT.dtype and vT.shape[:2] == (Z, Cc)
    o = torch.empty_like(q)
    if bias_v is not None:
        assert bias_v.dtype == torch.float32 and bias_v.numel() >= Cc and bias_v.is_cuda
    L.check(lib.vt_flash_attention(_ptr(q), _ptr(k), _ptr(vT), _ptr(bias_v), _ptr(o), _DT[q.dtype], Z, S, Cc, vT.shape[2], float(scale), _stream()),
            "vt_flash_attention")
    if CONV_RECORD is not None:   # both products of the attention: label (rows, keys, 2 C) carries their FLOPs
        CONV_RECORD.append((("flash", q, k, vT, bias_v, o, float(scale)), (), (Z * S, S, 2 * Cc)))
    return o


# Optional record of the vt_layernorm_act launches of a step (bench.py's per-class HBM roofline): tuples
# (x, y, gamma, beta, M, c, eps, silu) with the tensors kept alive; None in normal use.
LN_RECORD = None


def layernorm_act(x, gamma, beta, *, silu: bool, eps: float = 1e-6, out_dtype=None, c: int = None):
    """Per-position LayerNorm over the last dim (+SiLU)."""
    lib = L.load()
    _chk(x, "layernorm.x")
    ld = x.shape[-1]
    c = c or ld
    M = x.numel() // ld
    out_dtype = out_dtype or x.dtype
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device) if c == ld else torch.zeros(
        x.shape, dtype=out_dtype, device=x.device)
    assert gamma.dtype == torch.float32 and beta.dtype == torch.float32 and gamma.numel() == c
    L.check(lib.vt_layernorm_act(_ptr(x), _DT[x.dtype], ld, _ptr(y), _DT[out_dtype], ld, _ptr(gamma), _ptr(beta),
                                 M, c, float(eps), int(bool(silu)), _stream()), "vt_layernorm_act")
    if LN_RECORD is not None:
        LN_RECORD.append((x, y, gamma, beta, M, c, float(eps), bool(silu)))
    return y


def replay_layernorms(record):
    """Re-issue recorded vt_layernorm_act launches on the current stream (same tensors)."""
    lib = L.load()
    for x, y, gamma, beta, M, c, eps, silu in record:
        ld = x.shape[-1]
        L.check(lib.vt_layernorm_act(_ptr(x), _DT[x.dtype], ld, _ptr(y), _DT[y.dtype], ld, _ptr(gamma), _ptr(beta), M, c, eps,
                                     int(silu), _stream()), "vt_layernorm_act(replay)")


def launch_bytes(d):
    """Algorithmic HBM bytes of one recorded MFMA-kernel launch: the input once, the weights once, the result(s) once,
    the residual once (what a perfectly cached launch would move)."""
    if isinstance(d, tuple):                     # vt_flash_attention: q, k, V^T in, o out
        _, q, k, vT, _b, o, _s = d
        return (q.numel() + k.numel() + vT.numel() + o.numel()) * q.element_size()
    if isinstance(d, L.TBlockDesc):
        es = 2
        px = d.B * d.T * d.HW
        return px * d.ld * es * (1 + (1 if d.keep_y else 0) + (1 if d.ln_next_mode else 0)) + 2 * d.C * 3 * d.C * es
    es = 2 if d.dtype in (L.VT_BF16, L.VT_F16) else 4
    eo = 4 if d.out_dtype == L.VT_F32 else 2
    nb = max(1, d.nbatch)
    M = d.B * d.To * d.Ho * d.Wo * nb
    plan = conv_plan(d)
    nbytes = d.B * d.Ti * d.Hi * d.Wi * d.Cin * es * (nb if d.xs_z or nb == 1 else 1) + d.Cout * d.ldw * es * nb
    if not (d.ln_mode != 0 and plan["ln_fused"] and not d.ln_keep_y):
        nbytes += M * d.Cout * eo
    if d.res_mode != L.VT_RES_NONE:
        nbytes += M * d.Cout * eo
    if d.ln_mode != 0 and plan["ln_fused"]:
        nbytes += M * d.Cout * eo
    return nbytes


def launch_class(d):
    """(class name, algorithmic HBM bytes) of a recorded MFMA-kernel launch whose FLOP per byte sits below the chip's
    ridge (2.5 PFLOP/s over 8 TB/s = 312): the launches bench.py prices against the HBM roofline.  None for the
    matrix-bound ones."""
    if isinstance(d, tuple):
        return None                              # attention: 0.1 % of the FLOPs, matrix-shaped
    if isinstance(d, L.TBlockDesc):
        return "temporal block fused (C=128)", launch_bytes(d)
    taps = d.KT * d.KH * d.KW
    if conv_plan(d)["kernel"] == "narrow":
        name = "conv_out 3x3x3 128->3 (conv3d_narrow_kernel)"
    elif taps == 1 and d.nbatch <= 1:
        name = "1x1 convolutions (nin_shortcut, attention projections)"
    elif d.Cin <= 8:
        name = "conv_in 3x3x3 (3 -> C)"
    elif d.KH == 1 and d.KW == 1 and d.KT == 3 and d.Cin <= 256:
        name = f"temporal k3 convolutions, unfused (C={d.Cin})"
    else:
        return None
    return name, launch_bytes(d)


GN_POS = 3   # host-level scope: statistics per position over the C/groups channels of a group (see groupnorm_act)


def groupnorm_act(x, gamma, beta, *, scope, silu, eps=1e-6, out_dtype=None, c=None, groups=32):
    """torch.nn.GroupNorm(groups, C)(+SiLU) on NDHWC x [B,T,H,W,ld]; `scope` = L.VT_GN_FRAME / PIXEL / CLIP says which
    view of the tensor the reference's call site normalises (see include/vidtok_amd.h); GN_POS = the causal temporal
    blocks, whose "(b t) c s" view has s = 1 (model_3dcausal.py:476-487)."""
    lib = L.load()
    _chk(x, "groupnorm.x")
    if scope == GN_POS:     # per position over C/groups only: the PIXEL domain of a one-frame view
        shp = x.shape
        return groupnorm_act(x.reshape(1, 1, -1, 1, shp[-1]), gamma, beta, scope=L.VT_GN_PIXEL, silu=silu, eps=eps,
                             out_dtype=out_dtype, c=c, groups=groups).reshape(shp)
    B, T, H, W, ld = x.shape
    c = c or ld
    out_dtype = out_dtype or x.dtype
    y = (torch.zeros if c != ld else torch.empty)(x.shape, dtype=out_dtype, device=x.device)
    nbytes = lib.vt_groupnorm_work_bytes(B, T, groups, scope)
    work = torch.empty((max(nbytes, 8) // 8,), dtype=torch.float64, device=x.device)
    L.check(lib.vt_groupnorm_act(_ptr(x), _DT[x.dtype], ld, _ptr(y), _DT[out_dtype], ld, _ptr(gamma), _ptr(beta), B, T, H * W,
                                 c, groups, scope, float(eps), int(bool(silu)), _ptr(work), _stream()), "vt_groupnorm_act")
    return y


def tanh_(x):
    """x = tanh(x) in place (fp32)"""
    _chk(x, "tanh.x")
    assert x.dtype == torch.float32
    L.check(L.load().vt_tanh_inplace(_ptr(x), x.numel(), _stream()), "vt_tanh_inplace")
    return x


def softmax_rows(s, scale: float, out_dtype, ld_out=None):
    """softmax(scale*s) over the last dim; ld_out > cols pads the output rows with zeros."""
    lib = L.load()
    _chk(s, "softmax.s")
    assert s.dtype == torch.float32
    cols = s.shape[-1]
    rows = s.numel() // cols
    ldo = ld_out or cols
    p = (torch.empty if ldo == cols else torch.zeros)(tuple(s.shape[:-1]) + (ldo,), dtype=out_dtype, device=s.device)
    L.check(lib.vt_softmax_rows(_ptr(s), _ptr(p), _DT[out_dtype], rows, cols, ldo, float(scale), _stream()),
            "vt_softmax_rows")
    return p


def ncthw_to_ndhwc(x, dtype, tpad: int = 0, ld: int = None):
    lib = L.load()
    _chk(x, "ncthw_to_ndhwc.x")
    assert x.dtype == torch.float32 and x.dim() == 5
    B, Cc, T, H, W = x.shape
    ld = ld or pad_channels(Cc)
    y = torch.empty((B, T + tpad, H, W, ld), dtype=dtype, device=x.device)
    L.check(lib.vt_ncthw_to_ndhwc(_ptr(x), _ptr(y), _DT[dtype], B, Cc, T, H, W, ld, tpad, _stream()),
            "vt_ncthw_to_ndhwc")
    return y


def ndhwc_to_ncthw(x, c: int, ttrim: int = 0):
    lib = L.load()
    _chk(x, "ndhwc_to_ncthw.x")
    B, T, H, W, ld = x.shape
    y = torch.empty((B, c, T - ttrim, H, W), dtype=torch.float32, device=x.device)
    L.check(lib.vt_ndhwc_to_ncthw(_ptr(x), _DT[x.dtype], _ptr(y), B, c, T, H, W, ld, ttrim, _stream()),
            "vt_ndhwc_to_ncthw")
    return y


def time_avgpool3s2(x, tmode=L.VT_TPAD_ZERO, cache=None):
    lib = L.load()
    _chk(x, "avgpool.x")
    B, Ti, H, W, Cc = x.shape
    y = torch.empty((B, Ti // 2, H, W, Cc), dtype=x.dtype, device=x.device)
    if cache is not None:
        _chk(cache, "avgpool.cache")
        assert cache.dtype == x.dtype and cache.numel() == B * H * W * Cc
    L.check(lib.vt_time_avgpool3s2(_ptr(x), _ptr(cache), _ptr(y), _DT[x.dtype], B, Ti, H * W, Cc, tmode, _stream()),
            "vt_time_avgpool3s2")
    return y


def time_lerp2x(x, out=None, out_t0=0):
    """trilinear x2 along T (align_corners=False) of frames x [B, Ti, H, W, C]; `out` [B, Td, H, W, C] receives the 2 Ti
    frames at out_t0 (one launch per clip of the batch then: the kernel's output is contiguous per clip)."""
    lib = L.load()
    _chk(x, "lerp.x")
    B, Ti, H, W, Cc = x.shape
    if out is None:
        y = torch.empty((B, 2 * Ti, H, W, Cc), dtype=x.dtype, device=x.device)
        L.check(lib.vt_time_lerp2x(_ptr(x), _ptr(y), _DT[x.dtype], B, Ti, H * W * Cc, _stream()), "vt_time_lerp2x")
        return y
    assert out.is_contiguous() and out.dtype == x.dtype and out.shape[0] == B and tuple(out.shape[2:]) == (H, W, Cc)
    assert 0 <= out_t0 and out_t0 + 2 * Ti <= out.shape[1]
    fr = H * W * Cc
    for b in range(B):
        xp = C.c_void_p(x.data_ptr() + b * Ti * fr * x.element_size())
        yp = C.c_void_p(out.data_ptr() + (b * out.shape[1] + out_t0) * fr * out.element_size())
        L.check(lib.vt_time_lerp2x(xp, yp, _DT[x.dtype], 1, Ti, fr, _stream()), "vt_time_lerp2x")
    return out


def time_lerp2x_cat(head, x, skip):
    """time_lerp2x of [head | x] along T (head [B, nh, H, W, C], x [B, T, H, W, C]) without its first `skip` frames -- the
    sequence is never assembled: [B, 2 (nh + T) - skip, H, W, C]"""
    lib = L.load()
    _chk(x, "lerp.x")
    _chk(head, "lerp.head")
    B, T, H, W, Cc = x.shape
    nh = head.shape[1]
    assert head.dtype == x.dtype and head.shape[0] == B and tuple(head.shape[2:]) == (H, W, Cc) and 0 <= skip < 2 * (nh + T)
    y = torch.empty((B, 2 * (nh + T) - skip, H, W, Cc), dtype=x.dtype, device=x.device)
    L.check(lib.vt_time_lerp2x_cat(_ptr(head), nh, _ptr(x), _ptr(y), _DT[x.dtype], B, T, skip, H * W * Cc, _stream()), "vt_time_lerp2x_cat")
    return y


def gather_frames(src, idx, out=None, out_t0=0):
    """dst[:, out_t0 + j] = src[:, idx[j]] along dim 1 of an NDHWC tensor (or any [B, T, ...] tensor: frames are copied
    as bytes).  v1.1 cache maintenance and chunk assembly; `out` [B, Td, ...] is filled in place."""
    lib = L.load()
    _chk(src, "gather.src")
    B, Ts = src.shape[:2]
    n = len(idx)
    assert n >= 1 and all(0 <= i < Ts for i in idx), (idx, Ts)
    frame = src[0, 0].numel()
    if out is None:
        out = torch.empty((B, n) + tuple(src.shape[2:]), dtype=src.dtype, device=src.device)
        out_t0 = 0
    assert out.is_contiguous() and out.dtype == src.dtype and out.shape[0] == B and tuple(out.shape[2:]) == tuple(src.shape[2:])
    Td = out.shape[1]
    assert 0 <= out_t0 and out_t0 + n <= Td
    for j0 in range(0, n, 128):                       # the C-ABI takes up to 128 frame indices per call
        part = list(idx[j0:j0 + 128])
        arr = (C.c_int32 * len(part))(*part)
        dptr = C.c_void_p(out.data_ptr() + (out_t0 + j0) * frame * out.element_size())
        L.check(lib.vt_gather_frames(_ptr(src), dptr, src.element_size(), B, frame, Ts * frame, Td * frame, arr,
                                     len(part), _stream()), "vt_gather_frames")
    return out


def pack_conv_weight(weight, dtype, cin_stored=None, mix=None, split3=False):
    """vt_pack_conv_weight: weight fp32 [Cout, Cin, *k] on the GPU (a reference parameter) -> packed rows [Cout, ldw] for
    vt_conv: k = tap * cin_stored + c, in `dtype` (float32 / bfloat16 / float16) or, with split3, the int32-typed split-bf16 container.
    mix = [[m0, m1, m2, m3], ...] per OUTPUT tap (-1 = absent): pre-summed taps of an up-sampler's parity class."""
    lib = L.load()
    _chk(weight, "pack.weight")
    assert weight.dtype == torch.float32 and weight.dim() >= 3
    cout, cin = weight.shape[:2]
    taps_in = weight[0, 0].numel()
    taps_out = taps_in if mix is None else len(mix)
    cin_p = cin_stored or pad_channels(cin)
    K = taps_out * cin_p
    if split3:
        assert dtype == torch.float32
        ldw = (K + 31) // 32 * 32
        out = torch.empty((cout, ldw), dtype=torch.int32, device=weight.device)
        mode = L.VT_BF16X3
    else:
        ldw = K
        out = torch.empty((cout, ldw), dtype=dtype, device=weight.device)
        mode = _DT[dtype]
    arr = None
    if mix is not None:
        flat = [int(v) for row in mix for v in (list(row) + [-1, -1, -1, -1])[:4]]
        arr = (C.c_int32 * len(flat))(*flat)
    L.check(lib.vt_pack_conv_weight(_ptr(weight), _ptr(out), mode, cout, cin, cin_p, taps_in, taps_out, arr, ldw, _stream()), "vt_pack_conv_weight")
    return out


def _levels_arr(levels):
    return (C.c_int32 * len(levels))(*[int(v) for v in levels])


def kl_sample(h, noise):
    """h [B, 2*zc, T, H, W] fp32 NCTHW -> (z [B, zc, T, H, W], kl 0-dim)."""
    lib = L.load()
    _chk(h, "kl.h")
    assert h.dtype == torch.float32
    B, c2 = h.shape[:2]
    zc = c2 // 2
    S = h[0, 0].numel()
    z = torch.empty((B, zc) + tuple(h.shape[2:]), dtype=torch.float32, device=h.device)
    kl = torch.empty((), dtype=torch.float32, device=h.device)
    if noise is not None:
        _chk(noise, "kl.noise")
        assert noise.shape == z.shape and noise.dtype == torch.float32
    L.check(lib.vt_kl_sample(_ptr(h), _ptr(noise), _ptr(z), _ptr(kl), B, zc, S, _stream()), "vt_kl_sample")
    return z, kl


def fsq_quantize(h, levels, num_codebooks: int = 1):
    """h [B, c*D, ...] fp32 (c = num_codebooks groups of D = len(levels) channels) -> (codes [B, c*D, ...], indices int32
    [B, ...] or, for c > 1, [B, ..., c]: the codebook axis last, as the reference keeps it)"""
    lib = L.load()
    _chk(h, "fsq.h")
    assert h.dtype == torch.float32
    B, D, c = h.shape[0], len(levels), int(num_codebooks)
    assert h.shape[1] == c * D
    S = h[0, 0].numel()
    z = torch.empty_like(h)
    idx = torch.empty((B,) + tuple(h.shape[2:]) + ((c,) if c > 1 else ()), dtype=torch.int32, device=h.device)
    L.check(lib.vt_fsq_quantize_cb(_ptr(h), _ptr(z), _ptr(idx), _levels_arr(levels), D, B, c, S, _stream()),
            "vt_fsq_quantize_cb")
    return z, idx


def fsq_indices_to_codes(idx, levels, num_codebooks: int = 1):
    """indices int32 [B, ...] (c > 1: [B, ..., c]) -> codes [B, c*D, ...]"""
    lib = L.load()
    _chk(idx, "fsq.indices")
    assert idx.dtype == torch.int32
    B, D, c = idx.shape[0], len(levels), int(num_codebooks)
    sp = tuple(idx.shape[1:-1]) if c > 1 else tuple(idx.shape[1:])
    assert c == 1 or idx.shape[-1] == c
    S = 1
    for v in sp:
        S *= int(v)
    z = torch.empty((B, c * D) + sp, dtype=torch.float32, device=idx.device)
    L.check(lib.vt_fsq_indices_to_codes_cb(_ptr(idx), _ptr(z), _levels_arr(levels), D, B, c, S, _stream()),
            "vt_fsq_indices_to_codes_cb")
    return z


def channel_linear(x, w, bias):
    """nn.Linear along dim 1 of an fp32 [B, Cin, ...] tensor -> [B, Cout, ...] (FSQ project_in / project_out)."""
    lib = L.load()
    _chk(x, "linear.x"); _chk(w, "linear.w")
    assert x.dtype == torch.float32 and w.dtype == torch.float32 and w.shape[1] == x.shape[1]
    B, Cin = x.shape[0], x.shape[1]
    S = x[0, 0].numel()
    y = torch.empty((B, w.shape[0]) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    L.check(lib.vt_channel_linear(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), B, Cin, w.shape[0], S, _stream()),
            "vt_channel_linear")
    return y


def fsq_aux_stats(h, levels, inv_temperature: float = 100.0, return_avg: bool = False):
    """-> fp32 tensor [3]: per-sample entropy, codebook entropy, commitment loss; with return_avg also the batch-mean
    code distribution avg_prob [prod(levels)] the codebook entropy was taken of."""
    lib = L.load()
    _chk(h, "fsq.h")
    B, D = h.shape[:2]
    S = h[0, 0].numel()
    arr = _levels_arr(levels)
    nwork = lib.vt_fsq_aux_work_floats(arr, D, B, S)
    work = torch.empty((nwork,), dtype=torch.float32, device=h.device)
    out = torch.empty((3,), dtype=torch.float32, device=h.device)
    avg = None
    if return_avg:
        J = 1
        for v in levels:
            J *= int(v)
        avg = torch.empty((J,), dtype=torch.float32, device=h.device)
    L.check(lib.vt_fsq_aux_stats_avg(_ptr(h), arr, D, B, S, float(inv_temperature), _ptr(work), _ptr(out), _ptr(avg),
                                     _stream()), "vt_fsq_aux_stats_avg")
    return (out, avg) if return_avg else out


def entropy(avg):
    """sum_j -avg_j log(max(avg_j, 1e-5)) of a distribution on the device (0-dim fp32)."""
    lib = L.load()
    _chk(avg, "entropy.avg")
    assert avg.dtype == torch.float32
    out = torch.empty((1,), dtype=torch.float32, device=avg.device)
    L.check(lib.vt_entropy(_ptr(avg), avg.numel(), _ptr(out), _stream()), "vt_entropy")
    return out[0]


def fsq_aux_loss(stats3, codebook_entropy, diversity_gamma: float, entropy_weight: float, commitment_weight: float):
    """(stats3[0] - gamma * ce) * entropy_weight + stats3[2] * commitment_weight as a fresh 0-dim fp32 tensor (vt_fsq_aux_loss);
    ce = `codebook_entropy` (0-dim / [1] device tensor) or, None, stats3[1]"""
    lib = L.load()
    _chk(stats3, "fsq.stats")
    assert stats3.dtype == torch.float32 and stats3.numel() >= 3
    out = torch.empty((1,), dtype=torch.float32, device=stats3.device)
    ce = None
    if codebook_entropy is not None:
        ce = codebook_entropy.reshape(1)
        assert ce.dtype == torch.float32 and ce.is_cuda
    L.check(lib.vt_fsq_aux_loss(_ptr(stats3), _ptr(ce), float(diversity_gamma), float(entropy_weight), float(commitment_weight), _ptr(out),
                                _stream()), "vt_fsq_aux_loss")
    return out[0]


def fsq_consts(levels):
    """Host-only: (half_l, offset, shift, basis) lists as the kernels use them."""
    lib = L.load()
    D = len(levels)
    out = (C.c_float * (4 * D))()
    L.check(lib.vt_fsq_consts(_levels_arr(levels), D, out), "vt_fsq_consts")
    v = list(out)
    return v[:D], v[D:2 * D], v[2 * D:3 * D], v[3 * D:]


def eval_psnr_ssim(x, y, raw=True):
    """Per-frame PSNR / SSIM of a reconstruction y against the input x (both NCTHW fp32 in [-1,1]); with raw=True the
    clamp and (.+1)/2 post-processing of the reference's eval loop are fused in, with raw=False x, y are [0,1] images.  Returns (psnr [B,T], ssim [B,T])."""
    lib = L.load()
    _chk(x, "eval.x"); _chk(y, "eval.y")
    assert x.dtype == torch.float32 and y.dtype == torch.float32 and x.shape == y.shape and x.dim() == 5
    B, Cc, T, H, W = x.shape
    psnr = torch.empty((B, T), dtype=torch.float32, device=x.device)
    ssim = torch.empty((B, T), dtype=torch.float32, device=x.device)
    work = torch.empty((lib.vt_eval_work_floats(B, T),), dtype=torch.float32, device=x.device)
    L.check(lib.vt_eval_psnr_ssim(_ptr(x), _ptr(y), _ptr(psnr), _ptr(ssim), _ptr(work), B, Cc, T, H, W, int(bool(raw)),
                                  _stream()), "vt_eval_psnr_ssim")
    return psnr, ssim


# ---- video front / back end (device halves of scripts/inference_reconstruct.py) ---------------------------------------
def frames_u8_to_ncthw(frames, resized_hw, crop_top_left, out_hw, out=None, t_off=0):
    """frames uint8 [T, H0, W0, 3] -> fp32 [1, 3, T, h, w] in [-1, 1]: /255, anti-aliased bilinear resize to
    `resized_hw`, crop window at `crop_top_left` of size `out_hw`, (v-0.5)/0.5.  `out` [1, 3, Tdst, h, w] + `t_off`
    write the frames into a larger clip buffer (--pad_gen_frames chaining)."""
    lib = L.load()
    _chk(frames, "frames")
    assert frames.dtype == torch.uint8 and frames.dim() == 4 and frames.shape[3] == 3
    T, H0, W0, _ = frames.shape
    (Hr, Wr), (top, left), (H, W) = resized_hw, crop_top_left, out_hw
    if out is None:
        out = torch.empty((1, 3, T, H, W), dtype=torch.float32, device=frames.device)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.shape[0] == 1 and out.shape[1] == 3 and tuple(out.shape[3:]) == (H, W)
    work = torch.empty((lib.vt_frames_work_floats(T, H0, W),), dtype=torch.float32, device=frames.device)
    L.check(lib.vt_frames_u8_to_ncthw(_ptr(frames), T, H0, W0, Hr, Wr, top, left, _ptr(out), out.shape[2], t_off, H, W, _ptr(work),
                                      _stream()), "vt_frames_u8_to_ncthw")
    return out


def ncthw_to_frames_u8(x, t0=0, n=None, out=None, w_off=0):
    """x fp32 [1, 3, T, H, W] frames t0 .. t0+n -> uint8 [n, H, Wtot, 3] (clamp, (x+1)/2, *255, truncation) at column w_off"""
    lib = L.load()
    _chk(x, "x")
    assert x.dtype == torch.float32 and x.dim() == 5 and x.shape[0] == 1 and x.shape[1] == 3
    T, H, W = x.shape[2:]
    n = T - t0 if n is None else n
    if out is None:
        out = torch.empty((n, H, W, 3), dtype=torch.uint8, device=x.device)
    assert out.dtype == torch.uint8 and out.is_contiguous() and out.shape[0] >= n and out.shape[1] == H and out.shape[3] == 3
    L.check(lib.vt_ncthw_to_frames_u8(_ptr(x), T, t0, n, H, W, _ptr(out), out.shape[2], w_off, _stream()), "vt_ncthw_to_frames_u8")
    return out


def ncthw_copy_frames(src, dst, ts0, td0, n, clamp=False):
    """dst[:, :, td0:td0+n] = src[:, :, ts0:ts0+n] (optionally clamped to [-1, 1]) for fp32 [B, C, T, H, W] tensors of equal
    B and C (the (b, c) planes are the kernel's "channels")"""
    lib = L.load()
    _chk(src, "src"); _chk(dst, "dst")
    assert src.dtype == torch.float32 and dst.dtype == torch.float32 and src.dim() == 5 and dst.dim() == 5
    assert tuple(src.shape[:2]) == tuple(dst.shape[:2]) and tuple(src.shape[3:]) == tuple(dst.shape[3:])
    L.check(lib.vt_ncthw_copy_frames(_ptr(src), _ptr(dst), src.shape[0] * src.shape[1], src.shape[2], dst.shape[2], ts0, td0, n,
                                     src.shape[3] * src.shape[4], int(bool(clamp)), _stream()), "vt_ncthw_copy_frames")
    return dst
