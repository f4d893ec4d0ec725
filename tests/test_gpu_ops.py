"""`-m gpu`: every HIP operator of libvidtok_amd.so against the plain PyTorch fp32 statement of its
contract (tests/torch_ops_ref.py) on the same seeded inputs.  fp32 arithmetic must agree to fp32
round-off (fmaf-chain MFMA), bf16 / fp16 to that type's round-off of the output; integer results bit-exactly.

Tiers: `-m gpu` runs the shipped paths (every kernel and every selectable option on both sides, at least once per
arithmetic); `-m "gpu and variants"` adds the exhaustive cross products (every case x every arithmetic x every gather
form) that earlier rounds ran by default -- the full matrix is for a changed kernel, not for every round."""
import math

import pytest
import torch

import torch_ops_ref as R
from util import rel_err
from vidtok_amd import lib as L
from vidtok_amd import ops
from vidtok_amd.ops import ConvGeom
from vidtok_amd.packing import pack_conv_weight, pack_split3

pytestmark = pytest.mark.gpu
DEV = "cuda"
# "x3" = vt_conv's split-bf16 arithmetic (VT_BF16X3): fp32 tensors, weights as bf16 hi / lo planes, three bf16 MFMAs per
# product -- checked against the same fp32 statement as the fp32 kernels (measured 2e-6 .. 6e-6 of the output's max norm)
X3 = "x3"
TOL = {torch.float32: 2e-5, torch.bfloat16: 1.2e-2, torch.float16: 1.5e-3, X3: 4e-5}
DTYPES = [torch.float32, torch.bfloat16]
DTYPES3 = DTYPES + [X3]
IDS3 = ["f32", "bf16", "x3"]
F16 = torch.float16
DTYPES4 = DTYPES3 + [F16]          # every arithmetic of vt_conv
IDS4 = IDS3 + ["f16"]
H16 = [torch.bfloat16, F16]        # the 16-bit storage types: one kernel source, two instantiations
H16_IDS = ["bf16", "f16"]
variants = pytest.mark.variants    # exhaustive cross products: `-m "gpu and variants"`


def _tier(values, ids, main):
    """pytest params of `values`: those in `main` run with `-m gpu`, the others only with `-m "gpu and variants"`"""
    return [pytest.param(v, id=i, marks=() if v in main else (variants,)) for v, i in zip(values, ids)]


def _rand(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(DEV)


def _act(B, T, H, W, c_real, dtype, seed):
    """NDHWC activation with channels padded to a multiple of 8 (pad lanes zero)."""
    cp = ops.pad_channels(c_real)
    x = torch.zeros((B, T, H, W, cp), dtype=dtype)
    g = torch.Generator().manual_seed(seed)
    x[..., :c_real] = torch.randn((B, T, H, W, c_real), generator=g).to(dtype)
    return x.to(DEV)


def _cpu(kw):
    return {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in kw.items()}


G3 = dict(kh=3, kw=3, ph=1, pw=1, ph_hi=1, pw_hi=1)
G333 = dict(kt=3, kh=3, kw=3, pt=2, ph=1, pw=1, ph_hi=1, pw_hi=1)

# name, (B,T,H,W), Cin, Cout, weight kernel dims, geom, extras
CONV_CASES = [
    ("conv2d_3x3_128_128", (2, 3, 16, 16), 128, 128, (3, 3), ConvGeom(**G3), {}),
    ("conv2d_3x3_256_128_res", (1, 2, 16, 24), 256, 128, (3, 3), ConvGeom(**G3), dict(res="add")),
    ("conv2d_3x3_64_192_ragged", (1, 3, 5, 7), 64, 192, (3, 3), ConvGeom(**G3), dict(res="add")),
    ("nin_1x1_128_256", (1, 2, 16, 16), 128, 256, (1, 1), ConvGeom(), {}),
    ("nin_1x1_64_256_one_kstep", (1, 2, 16, 16), 64, 256, (1, 1), ConvGeom(), dict(res="add")),
    ("downsample_s2", (1, 2, 16, 16), 128, 128, (3, 3), ConvGeom(kh=3, kw=3, sh=2, sw=2, ph_hi=1, pw_hi=1), {}),
    ("upsample_fold", (1, 2, 8, 8), 128, 128, (3, 3), ConvGeom(ups_s=1, **G3), {}),
    ("temporal_k3_res", (2, 6, 8, 8), 128, 128, (3,), ConvGeom(kt=3, pt=2), dict(res="add")),
    ("temporal_k3_512", (1, 5, 4, 4), 512, 512, (3,), ConvGeom(kt=3, pt=2), {}),
    ("conv3d_333_256", (1, 5, 8, 8), 256, 256, (3, 3, 3), ConvGeom(**G333), dict(res="add")),
    ("timedown_s2_mix", (1, 6, 8, 8), 128, 128, (3, 3, 3),
     ConvGeom(kt=3, kh=3, kw=3, st=2, pt=1, ph=1, pw=1, ph_hi=1, pw_hi=1), dict(res="mix")),
    ("timeup_fold_mix", (1, 3, 8, 8), 128, 128, (3, 3, 3), ConvGeom(ups_t=1, **G333), dict(res="mix_up")),
    ("conv_in_3_128", (1, 5, 16, 16), 3, 128, (3, 3, 3), ConvGeom(**G333), {}),
    ("dec_conv_in_4_512", (2, 3, 4, 4), 4, 512, (3, 3, 3), ConvGeom(**G333), {}),
    ("conv_out_128_3_ncthw_trim", (1, 8, 16, 16), 128, 3, (3, 3, 3), ConvGeom(**G333), dict(ncthw=3)),
    ("enc_conv_out_512_8_ncthw", (2, 3, 4, 4), 512, 8, (3, 3, 3), ConvGeom(**G333), dict(ncthw=0)),
    ("enc_conv_out_512_32_ncthw", (1, 3, 4, 4), 512, 32, (3, 3, 3), ConvGeom(**G333), dict(ncthw=0)),
    ("narrow_ndhwc_128_5", (1, 2, 8, 8), 128, 5, (3, 3), ConvGeom(**G3), {}),
    ("cout64_tile", (1, 2, 8, 8), 128, 64, (3, 3), ConvGeom(**G3), {}),
    ("v11_replicate_3d", (1, 4, 8, 8), 128, 128, (3, 3, 3), ConvGeom(**G333), dict(tmode="replicate")),
    ("v11_cache_3d", (2, 4, 8, 8), 128, 128, (3, 3, 3), ConvGeom(**G333), dict(tmode="cache")),
    ("v11_cache_1d", (1, 4, 8, 8), 256, 256, (3,), ConvGeom(kt=3, pt=2), dict(tmode="cache", res="add")),
    ("v11_cache_s2", (1, 4, 8, 8), 128, 128, (3, 3, 3),
     ConvGeom(kt=3, kh=3, kw=3, st=2, pt=1, ph=1, pw=1, ph_hi=1, pw_hi=1), dict(tmode="cache")),
    # non-causal family: centred temporal windows (zeros after the clip as well), back-padded stride-2 conv
    ("nc_conv3d_sym", (1, 4, 8, 8), 128, 128, (3, 3, 3),
     ConvGeom(kt=3, kh=3, kw=3, pt=1, pt_hi=1, ph=1, pw=1, ph_hi=1, pw_hi=1), dict(res="add")),
    ("nc_conv1d_sym_512", (1, 5, 4, 4), 512, 512, (3,), ConvGeom(kt=3, pt=1, pt_hi=1), {}),
    ("nc_timedown_back_mix", (1, 6, 8, 8), 128, 128, (3, 3, 3),
     ConvGeom(kt=3, kh=3, kw=3, st=2, pt=0, pt_hi=1, ph=1, pw=1, ph_hi=1, pw_hi=1), dict(res="mix")),
    ("nc_timeup_sym_mix", (1, 3, 8, 8), 128, 128, (3, 3, 3),
     ConvGeom(kt=3, kh=3, kw=3, pt=1, pt_hi=1, ph=1, pw=1, ph_hi=1, pw_hi=1, ups_t=1), dict(res="mix_up")),
    # frames-innermost tile order (Ho*Wo a multiple of the pixel tile) and the LDS epilogue on several full tiles
    ("temporal_k3_tinner", (2, 5, 16, 16), 128, 128, (3,), ConvGeom(kt=3, pt=2), dict(res="add")),
    ("conv3d_333_tinner_256", (1, 4, 16, 16), 256, 256, (3, 3, 3), ConvGeom(**G333), {}),
    # LayerNorm (+SiLU) of the result: fused in the epilogue (Cout = 128, full tiles) ...
    ("conv2d_ln_fused", (2, 3, 16, 16), 128, 128, (3, 3), ConvGeom(**G3), dict(ln="only")),
    ("conv2d_ln_fused_res_keep", (1, 2, 16, 16), 256, 128, (3, 3), ConvGeom(**G3), dict(res="add", ln="keep")),
    ("temporal_ln_fused", (1, 6, 16, 8), 128, 128, (3,), ConvGeom(kt=3, pt=2), dict(ln="only")),
    # ... or conv + vt_layernorm_act behind the same call (ragged pixel count / other channel counts)
    ("conv2d_ln_fallback_ragged", (1, 3, 5, 7), 128, 128, (3, 3), ConvGeom(**G3), dict(ln="only")),
    ("conv2d_ln_fallback_256", (1, 2, 8, 8), 128, 256, (3, 3), ConvGeom(**G3), dict(ln="keep")),
    ("conv3d_ln_fallback_512", (1, 3, 4, 4), 512, 512, (3, 3, 3), ConvGeom(**G333), dict(ln="only")),
    # Cout = 256 on full 256-pixel tiles: fused in the 8-wave tile's LDS-transposed epilogue when that tile is chosen
    # (test_conv_forced_256_tile / test_conv_large), conv + vt_layernorm_act on the 128 tile (test_conv)
    ("conv2d_ln256_only", (1, 2, 16, 16), 256, 256, (3, 3), ConvGeom(**G3), dict(ln="only")),
    ("conv2d_ln256_res_keep", (2, 2, 16, 16), 128, 256, (3, 3), ConvGeom(**G3), dict(res="add", ln="keep")),
    ("temporal_ln256_only", (1, 4, 16, 16), 256, 256, (3,), ConvGeom(kt=3, pt=2), dict(ln="only")),
]


@pytest.mark.parametrize("dtype", DTYPES4, ids=IDS4)
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv(case, dtype):
    _check_conv(case, dtype)


# The 8-wave 256x256 instantiation is chosen only for Cout % 256 == 0 with >= 128 tiles (option conv_tile_min): every
# Cout % 256 == 0 case above is replayed with option conv_tile = 256 (forces that tile however few tiles there are: ragged
# pixel tiles, every padding / cache / residual mode), and the cases below reach it -- and the frames-innermost
# order, the parity interleave, cache mode -- at sizes where the dispatcher picks it by itself (BASELINE-sized layers).
BIG256 = [c for c in CONV_CASES if c[3] % 256 == 0]
CONV_CASES_LARGE = [
    ("L_conv3d_333_256_m98k", (1, 6, 128, 128), 256, 256, (3, 3, 3), ConvGeom(**G333), dict(res="add")),
    ("L_conv2d_3x3_256_256", (2, 3, 128, 128), 256, 256, (3, 3), ConvGeom(**G3), dict(res="add")),
    ("L_temporal_k3_512", (2, 12, 64, 64), 512, 512, (3,), ConvGeom(kt=3, pt=2), dict(res="add")),
    ("L_timeup_parity_kt2_256", (1, 6, 128, 128), 256, 256, (2, 3, 3),
     ConvGeom(kt=2, kh=3, kw=3, pt=1, ph=1, pw=1, ph_hi=1, pw_hi=1), dict(res="mix")),
    ("L_v11_cache_3d_256", (1, 6, 128, 128), 256, 256, (3, 3, 3), ConvGeom(**G333), dict(tmode="cache")),
    ("L_v11_cache_1d_128", (1, 8, 128, 128), 128, 128, (3,), ConvGeom(kt=3, pt=2), dict(tmode="cache", res="add")),
    ("L_conv2d_128_128_ln", (1, 4, 256, 256), 128, 128, (3, 3), ConvGeom(**G3), dict(res="add", ln="keep")),
    ("L_conv2d_256_256_ln", (2, 3, 128, 128), 256, 256, (3, 3), ConvGeom(**G3), dict(res="add", ln="keep")),
    ("L_temporal_k3_256_ln_only", (1, 8, 128, 128), 256, 256, (3,), ConvGeom(kt=3, pt=2), dict(ln="only")),
]


@pytest.mark.parametrize("dtype", DTYPES4, ids=IDS4)
@pytest.mark.parametrize("case", BIG256, ids=[c[0] for c in BIG256])
def test_conv_forced_256_tile(case, dtype, vt_opts):
    vt_opts(conv_tile=256)
    plan = _check_conv(case, dtype)
    assert plan["tile"] == (256, 256)
    (B, T, H, W), cout = case[1], case[3]
    if "ln" in case[6] and cout == 256 and (B * T * H * W) % 256 == 0:
        assert plan["ln_fused"] and plan["launches"] == 1       # conv_epilogue_lds256


@pytest.mark.parametrize("dtype", DTYPES4, ids=IDS4)
@pytest.mark.parametrize("case", CONV_CASES_LARGE, ids=[c[0] for c in CONV_CASES_LARGE])
def test_conv_large(case, dtype):
    plan = _check_conv(case, dtype)
    small = (64, 128) if plan["kernel"] == "ws2" else (128, 128)        # conv_ws2.hip walks 4 x 16-pixel tiles
    assert plan["tile"] == ((256, 256) if case[3] % 256 == 0 else small) and plan["workgroups"] >= 384
    if "ln" in case[6]:
        assert plan["ln_fused"]


# The weight-stationary persistent kernel (conv_ws2.hip) takes 16-bit 3x3 / Cin = Cout = 128 / frames tiling by 8 x 16:
# single-tile frames (every halo side out of range at once), tile rows / columns with one border, many tiles per
# workgroup (double-buffered patch pipeline), + residual, LayerNorm fused with and without keeping y.  Each case also
# runs with option conv_ws = 0 so the tile-per-workgroup kernel keeps its coverage of the same shapes.
WS_CASES = [
    ("ws_single_tile", (1, 1, 8, 16), 128, 128, (3, 3), ConvGeom(**G3), {}),
    ("ws_two_tiles_one_wg", (1, 1, 16, 16), 128, 128, (3, 3), ConvGeom(**G3), dict(res="add", ln="keep")),
    ("ws_more_tiles_than_cus", (2, 5, 128, 128), 128, 128, (3, 3), ConvGeom(**G3), dict(res="add", ln="only")),
    ("ws_one_tile_column", (1, 2, 24, 16), 128, 128, (3, 3), ConvGeom(**G3), dict(res="add")),
    ("ws_one_tile_row", (1, 2, 8, 48), 128, 128, (3, 3), ConvGeom(**G3), dict(res="add", ln="keep")),
    ("ws_3x3_tiles_frames", (2, 3, 24, 48), 128, 128, (3, 3), ConvGeom(**G3), dict(ln="only")),
    ("ws_many_tiles_res_ln", (1, 5, 128, 128), 128, 128, (3, 3), ConvGeom(**G3), dict(res="add", ln="keep")),
    ("ws_many_tiles_plain", (1, 3, 256, 256), 128, 128, (3, 3), ConvGeom(**G3), {}),
    ("ws_no_bias_path", (1, 2, 16, 32), 128, 128, (3, 3), ConvGeom(**G3), dict(res="add", nobias=True)),
]


@pytest.mark.parametrize("dtype", H16, ids=H16_IDS)
@pytest.mark.parametrize("ws", ["2", "0"], ids=["ws2", "igemm"])
@pytest.mark.parametrize("case", WS_CASES, ids=[c[0] for c in WS_CASES])
def test_conv_weight_stationary(case, ws, dtype, vt_opts):
    vt_opts(conv_ws=ws)
    plan = _check_conv(case, dtype)
    assert plan["kernel"] == {"0": "igemm", "2": "ws2"}[ws]
    if "ln" in case[6]:
        assert plan["ln_fused"]


GC = dict(kt=3, kh=3, kw=3, pt=1, pt_hi=1, ph=1, pw=1, ph_hi=1, pw_hi=1)    # centred in time (non-causal family)
NARROW_CASES = [   # conv3d_narrow_kernel: bf16 -> fp32 NCTHW, Cin 128, Cout <= 4, 3x3x3
    ("nw_conv_out_trim", (1, 8, 16, 16), 128, 3, (3, 3, 3), ConvGeom(**G333), dict(ncthw=3)),
    ("nw_ragged_rows_cols", (2, 5, 20, 37), 128, 3, (3, 3, 3), ConvGeom(**G333), dict(ncthw=0)),      # H % 8, W % 14 != 0
    ("nw_one_window", (1, 4, 8, 10), 128, 3, (3, 3, 3), ConvGeom(**G333), dict(ncthw=1)),            # W < 14, 3 idle waves
    ("nw_single_frame", (1, 1, 8, 16), 128, 3, (3, 3, 3), ConvGeom(**G333), dict(ncthw=0)),
    ("nw_replicate", (1, 6, 16, 30), 128, 3, (3, 3, 3), ConvGeom(**G333), dict(ncthw=0, tmode="replicate")),
    ("nw_cache", (2, 4, 16, 16), 128, 3, (3, 3, 3), ConvGeom(**G333), dict(ncthw=0, tmode="cache")),
    ("nw_centred", (1, 5, 16, 16), 128, 3, (3, 3, 3), ConvGeom(**GC), dict(ncthw=0)),
    ("nw_cout4_nobias", (1, 4, 8, 16), 128, 4, (3, 3, 3), ConvGeom(**G333), dict(ncthw=0, nobias=True)),
    ("nw_cout1", (1, 4, 8, 16), 128, 1, (3, 3, 3), ConvGeom(**G333), dict(ncthw=0)),
    ("nw_time_segments", (1, 20, 64, 64), 128, 3, (3, 3, 3), ConvGeom(**G333), dict(ncthw=3)),       # 17 frames in 2 segments
    ("nw_segments_cache", (1, 14, 32, 32), 128, 3, (3, 3, 3), ConvGeom(**G333), dict(ncthw=0, tmode="cache")),
    ("nw_full_frame", (1, 5, 256, 256), 128, 3, (3, 3, 3), ConvGeom(**G333), dict(ncthw=0)),
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, X3, F16], ids=["bf16", "x3", "f16"])
@pytest.mark.parametrize("nw", _tier(["1", "0"], ["narrow", "igemm"], main=["1"]))
@pytest.mark.parametrize("case", NARROW_CASES, ids=[c[0] for c in NARROW_CASES])
def test_conv_narrow_output(case, nw, dtype, vt_opts):
    """bf16, and the split-bf16 form (fp32 x, weight planes): two passes of the same kernel, hi plane -> y, lo plane onto y"""
    vt_opts(conv_narrow=nw)
    plan = _check_conv(case, dtype)
    assert plan["kernel"] == ("narrow" if nw == "1" else "igemm")
    assert plan["launches"] == (2 if (nw == "1" and dtype == X3) else 1)


def test_conv_narrow_is_tiling_independent():
    """same clip alone and inside a batch, same frames in one time segment or two: bit-identical planes"""
    name, (B, T, H, W), cin, cout, kdims, geom, ex = NARROW_CASES[9]
    x = _act(2, T, H, W, cin, torch.bfloat16, 1)
    wt = torch.randn((cout, cin) + tuple(kdims), generator=torch.Generator().manual_seed(2)) / math.sqrt(cin * 27)
    w = pack_conv_weight(wt, torch.bfloat16, cin_stored=cin).to(DEV)
    bias = _rand((cout,), torch.float32, 3, 0.1)
    both = ops.conv(x, w, bias, geom, cout=cout, out_layout=L.VT_NCTHW, t_trim=3)
    one = ops.conv(x[1:].contiguous(), w, bias, geom, cout=cout, out_layout=L.VT_NCTHW, t_trim=3)
    assert torch.equal(both[1:], one)
    late = ops.conv(x[1:].contiguous(), w, bias, geom, cout=cout, out_layout=L.VT_NCTHW, t_trim=12)   # 8 frames: one segment
    assert torch.equal(late, one[:, :, 9:])


TBLOCK_CASES = [  # (B, T, H, W), tmode, next norm (None | silu flag), keep_y
    ((1, 5, 8, 8), L.VT_TPAD_ZERO, None, True),
    ((2, 7, 16, 16), L.VT_TPAD_ZERO, True, True),
    ((1, 1, 8, 16), L.VT_TPAD_ZERO, False, True),            # single frame: both missing taps skipped
    ((1, 2, 8, 8), L.VT_TPAD_REPLICATE, True, False),        # only the next norm is written
    ((2, 6, 16, 8), L.VT_TPAD_REPLICATE, None, True),
    ((1, 20, 128, 128), L.VT_TPAD_ZERO, True, True),         # 256 columns: every CU walks a full 20-frame column
    ((3, 9, 64, 48), L.VT_TPAD_ZERO, True, True),            # 144 columns: uneven split over the workgroups
]


@pytest.mark.parametrize("dt", H16, ids=H16_IDS)
@pytest.mark.parametrize("shape,tmode,nxt,keep", TBLOCK_CASES, ids=[f"{c[0]}-{c[1]}-{c[2]}-{c[3]}" for c in TBLOCK_CASES])
def test_temporal_block_fused(shape, tmode, nxt, keep, dt):
    """vt_temporal_block (one launch) vs the unfused operator sequence it replaces, stated in torch on the host"""
    B, T, H, W = shape
    C_ = 128
    x = _act(B, T, H, W, C_, dt, 1)
    g = torch.Generator().manual_seed(2)
    ws = [pack_conv_weight(torch.randn((C_, C_, 3), generator=g) / math.sqrt(3 * C_), dt, cin_stored=C_).to(DEV) for _ in range(2)]
    bs = [_rand((C_,), torch.float32, 3 + i, 0.1) for i in range(2)]
    norms = [(_rand((C_,), torch.float32, 10 + i, 0.3) + 1.0, _rand((C_,), torch.float32, 20 + i, 0.2)) for i in range(3)]
    next_ln = None if nxt is None else (norms[2][0], norms[2][1], nxt)
    assert ops.temporal_block_supported(x, tmode)
    out = ops.temporal_block(x, ws[0], bs[0], ws[1], bs[1], norms[0], norms[1], tmode=tmode, next_ln=next_ln, keep_y=keep)
    torch.cuda.synchronize()
    c = lambda t: None if t is None else (tuple(u.cpu() if isinstance(u, torch.Tensor) else u for u in t) if isinstance(t, tuple) else t.cpu())
    ref = R.temporal_block(x.cpu(), ws[0].cpu(), bs[0].cpu(), ws[1].cpu(), bs[1].cpu(), c(norms[0]), c(norms[1]), tmode=tmode,
                           next_ln=c(next_ln), keep_y=keep)
    outs = out if isinstance(out, tuple) else (out,)
    refs = ref if isinstance(ref, tuple) else (ref,)
    assert len(outs) == len(refs)
    for o, r in zip(outs, refs):
        assert o.shape == r.shape and o.dtype == r.dtype and torch.isfinite(o.float()).all()
        e = rel_err(o, r)
        print(f"tblock {shape} tmode={tmode}: rel_err={e:.3e}")
        assert e < 2 * TOL[dt]
    # not covered: fp32, other channel counts, ragged pixel counts -> the host falls back to the unfused launches
    assert not ops.temporal_block_supported(_act(1, 2, 8, 8, 128, torch.float32, 1), tmode)
    assert not ops.temporal_block_supported(_act(1, 2, 8, 8, 256, dt, 1), tmode)
    assert not ops.temporal_block_supported(_act(1, 2, 5, 7, 128, dt, 1), tmode)
    assert not ops.temporal_block_supported(x, L.VT_TPAD_CACHE)          # cache mode without caches


@pytest.mark.parametrize("dt", H16, ids=H16_IDS)
@pytest.mark.parametrize("shape,off,cuts", [((2, 14, 16, 16), 0, (5, 9)), ((1, 26, 64, 64), 4, (9, 17)), ((3, 12, 24, 48), 2, (6,))],
                         ids=["two_cuts", "lookahead_4_256_columns", "lookahead_2_uneven_split"])
def test_temporal_block_chunked(shape, off, cuts, dt):
    """v1.1 tiling through the fused launch (VERDICT r2 #5a): a clip run as chunks -- first chunk with replicate padding,
    later ones from the chunk state the launch itself keeps (caches of BOTH convolutions' inputs, rewritten in place,
    `cache_offset` look-ahead frames recomputed by the next chunk: reference model_3dcausal_v1_1.py:159-178) -- must give
    the bits of the same clip run in one launch: the ring rows a chunk restores are the ring rows the previous one had."""
    B, T, H, W = shape
    C_ = 128
    x = _act(B, T, H, W, C_, dt, 1)
    g = torch.Generator().manual_seed(2)
    ws = [pack_conv_weight(torch.randn((C_, C_, 3), generator=g) / math.sqrt(3 * C_), dt, cin_stored=C_).to(DEV) for _ in range(2)]
    bs = [_rand((C_,), torch.float32, 3 + i, 0.1) for i in range(2)]
    norms = [(_rand((C_,), torch.float32, 10 + i, 0.3) + 1.0, _rand((C_,), torch.float32, 20 + i, 0.2)) for i in range(3)]
    nxt = (norms[2][0], norms[2][1], True)
    run = lambda xs, **kw: ops.temporal_block(xs.contiguous(), ws[0], bs[0], ws[1], bs[1], norms[0], norms[1], next_ln=nxt, **kw)
    whole = run(x, tmode=L.VT_TPAD_REPLICATE)
    caches = tuple(torch.full((B, 2, H, W, C_), float("nan"), dtype=dt, device=DEV) for _ in range(2))
    # chunk i covers frames [start_i, end_i); the next one starts `off` frames before end_i (they are computed twice)
    bounds, start = [], 0
    for cpos in list(cuts) + [T]:
        bounds.append((start, cpos))
        start = cpos - off
    for i, (t0, t1) in enumerate(bounds):
        tmode = L.VT_TPAD_REPLICATE if i == 0 else L.VT_TPAD_CACHE
        assert ops.temporal_block_supported(x[:, t0:t1].contiguous(), tmode, None, caches, off)
        part = run(x[:, t0:t1], tmode=tmode, caches=caches, cache_offset=off)
        torch.cuda.synchronize()
        for o, w_ in zip(part, whole):
            assert torch.equal(o, w_[:, t0:t1]), (i, t0, t1)
        assert all(torch.isfinite(c_.float()).all() for c_ in caches)
    # a clip too short to leave two frames behind its look-ahead is not covered (the host keeps such blocks unfused)
    assert not ops.temporal_block_supported(x[:, :off + 2].contiguous(), L.VT_TPAD_CACHE, None, caches, off)


@pytest.mark.parametrize("dt", H16, ids=H16_IDS)
def test_weight_stationary_kernels_are_split_independent(dt):
    """A pixel's bits must not depend on which workgroup / code path computed it: run-to-run equality and
    batch slice == batch of one for the persistent kernels (their row phases exist in a sliced and a plain version);
    the 3x3 kernel without LayerNorm is moreover bit-identical to the tile-per-workgroup kernel (same fp32 sums)."""
    C_ = 128
    B, T, H, W = 2, 5, 128, 128
    x, res = _act(B, T, H, W, C_, dt, 1), _act(B, T, H, W, C_, dt, 2)
    g = torch.Generator().manual_seed(3)
    w = pack_conv_weight(torch.randn((C_, C_, 3, 3), generator=g) / math.sqrt(9 * C_), dt, cin_stored=C_).to(DEV)
    bias = _rand((C_,), torch.float32, 4, 0.1)
    ln = (_rand((C_,), torch.float32, 5, 0.3) + 1.0, _rand((C_,), torch.float32, 6, 0.2), 1e-6, True)
    tup = lambda o: o if isinstance(o, tuple) else (o,)
    for kw in ({}, dict(res=res, res_mode=L.VT_RES_ADD), dict(res=res, res_mode=L.VT_RES_ADD, ln=ln), dict(ln=ln, ln_keep_y=False)):
        a = tup(ops.conv(x, w, bias, ConvGeom(**G3), cout=C_, **kw))
        b = tup(ops.conv(x, w, bias, ConvGeom(**G3), cout=C_, **kw))
        kw1 = dict(kw, res=res[1:2].contiguous()) if "res" in kw else kw
        one = tup(ops.conv(x[1:2].contiguous(), w, bias, ConvGeom(**G3), cout=C_, **kw1))
        assert all(torch.equal(u, v) for u, v in zip(a, b)) and all(torch.equal(u[1:2], v) for u, v in zip(a, one)), kw.keys()
        if "ln" not in kw:
            with L.options(conv_ws=0):
                ig = tup(ops.conv(x, w, bias, ConvGeom(**G3), cout=C_, **kw))
            assert torch.equal(a[0], ig[0])
    ws = [pack_conv_weight(torch.randn((C_, C_, 3), generator=g) / math.sqrt(3 * C_), dt, cin_stored=C_).to(DEV) for _ in range(2)]
    bs = [_rand((C_,), torch.float32, 7 + i, 0.1) for i in range(2)]
    nm = (ln[0], ln[1])
    for tmode in (L.VT_TPAD_ZERO, L.VT_TPAD_REPLICATE):
        for nxt in (None, (ln[0], ln[1], True)):
            a = tup(ops.temporal_block(x, ws[0], bs[0], ws[1], bs[1], nm, nm, tmode=tmode, next_ln=nxt))
            b = tup(ops.temporal_block(x, ws[0], bs[0], ws[1], bs[1], nm, nm, tmode=tmode, next_ln=nxt))
            one = tup(ops.temporal_block(x[1:2].contiguous(), ws[0], bs[0], ws[1], bs[1], nm, nm, tmode=tmode, next_ln=nxt))
            assert all(torch.equal(u, v) for u, v in zip(a, b)) and all(torch.equal(u[1:2], v) for u, v in zip(a, one)), (tmode, nxt is None)


def test_conv_weight_stationary_not_for_fp32_or_other_shapes():
    assert _check_conv(WS_CASES[0], torch.float32)["kernel"] == "igemm"
    assert _check_conv(NARROW_CASES[0], torch.float32)["kernel"] == "igemm"
    assert _check_conv(("ws_ragged_w", (1, 1, 8, 24), 128, 128, (3, 3), ConvGeom(**G3), {}), torch.bfloat16)["kernel"] == "igemm"


POINTER_CASES = [c for c in CONV_CASES if c[0] in ("conv2d_3x3_256_128_res", "conv2d_3x3_64_192_ragged", "upsample_fold",
                                                   "temporal_k3_512", "conv3d_333_256", "timeup_fold_mix", "conv_in_3_128",
                                                   "conv_out_128_3_ncthw_trim", "v11_replicate_3d", "nc_conv3d_sym",
                                                   "conv3d_333_tinner_256", "conv2d_ln_fused")] + CONV_CASES_LARGE[:3]


@pytest.mark.parametrize("tile", ["", "256"])
@pytest.mark.parametrize("dtype", _tier(DTYPES4, IDS4, main=[torch.bfloat16]))
@pytest.mark.parametrize("case", POINTER_CASES, ids=[c[0] for c in POINTER_CASES])
def test_conv_pointer_gather(case, dtype, tile, vt_opts):
    """conv_buf = 0: the 64-bit pointer form of the gather (what tensors >= 4 GiB and v1.1 cache mode use)"""
    if tile and case[3] % 256 != 0:
        pytest.skip("256 tile needs Cout % 256 == 0")
    vt_opts(conv_buf=0)
    if tile:
        vt_opts(conv_tile=tile)
    plan = _check_conv(case, dtype)
    assert not tile or plan["tile"] == (256, 256)


SCHED_CASES = [c for c in BIG256 if c[0] in ("nin_1x1_128_256", "nin_1x1_64_256_one_kstep", "temporal_k3_512", "conv3d_333_256", "v11_cache_1d", "nc_conv1d_sym_512",
                                             "conv3d_333_tinner_256", "conv2d_ln256_only", "conv2d_ln256_res_keep", "temporal_ln256_only")]


# Cache mode (v1.1 chunks after the first) gathers through buffer descriptors when a tile lies inside one output frame
# (Ho * Wo % tile rows == 0: a time tap then reads the cache or x for the whole tile and the kernel switches the descriptor
# per tap), through pointers otherwise.  Two clips (the cache's own batch stride), caches longer than the padding, a time
# stride of 2 (the down-sampler's convolution), frames of exactly one 256-row tile, LayerNorm emitted; both gather forms.
CACHE_BUF_CASES = [
    ("cb_3d_256_b2", (2, 5, 32, 32), 256, 256, (3, 3, 3), ConvGeom(**G333), dict(tmode="cache", res="add")),
    ("cb_3d_s2_256", (2, 6, 32, 32), 256, 256, (3, 3, 3),
     ConvGeom(kt=3, kh=3, kw=3, st=2, pt=1, ph=1, pw=1, ph_hi=1, pw_hi=1), dict(tmode="cache")),
    ("cb_1d_512_b2", (2, 4, 16, 16), 512, 512, (3,), ConvGeom(kt=3, pt=2), dict(tmode="cache", res="add")),
    ("cb_1d_256_ln", (2, 4, 16, 16), 256, 256, (3,), ConvGeom(kt=3, pt=2), dict(tmode="cache", res="add", ln="keep")),
    ("cb_3d_128_b2", (2, 3, 16, 16), 128, 128, (3, 3, 3), ConvGeom(**G333), dict(tmode="cache", ln="keep")),
    ("cb_3d_128_to_3", (1, 3, 16, 16), 128, 8, (3, 3, 3), ConvGeom(**G333), dict(tmode="cache")),
]


@pytest.mark.parametrize("dtype", _tier(DTYPES4, IDS4, main=[torch.bfloat16, X3]))
@pytest.mark.parametrize("gather", ["descriptors", "pointers"])
@pytest.mark.parametrize("case", CACHE_BUF_CASES, ids=[c[0] for c in CACHE_BUF_CASES])
def test_conv_cache_mode_gather_forms(case, gather, dtype, vt_opts):
    vt_opts(conv_buf=(0 if gather == "pointers" else 1))
    if case[3] % 256 == 0:
        vt_opts(conv_tile=256)
    plan = _check_conv(case, dtype)
    assert plan["tile"][0] in (128, 256)


# The 128 x 128 tile has two rings: two slots (two workgroups per CU cover each other's DMA latency) and, for launches with
# no more tiles than CUs -- every workgroup alone on its CU -- four slots with three K steps in flight (option conv_deep).
# Both instantiations on the shapes that pick the deep one by themselves: the deep 512-channel layers of a v1.1 chunk,
# cache mode in both gather forms, fused LayerNorm, ragged pixel / channel tiles, exactly 8 K steps (the shortest it takes).
DEEP_CASES = [
    ("deep_3d_512", (1, 3, 16, 16), 512, 512, (3, 3, 3), ConvGeom(**G333), dict(res="add")),
    ("deep_cache_1d_256", (2, 4, 16, 16), 256, 256, (3,), ConvGeom(kt=3, pt=2), dict(tmode="cache", res="add")),
    ("deep_cache_3d_ragged", (1, 3, 10, 10), 128, 128, (3, 3, 3), ConvGeom(**G333), dict(tmode="cache")),
    ("deep_ln128", (1, 2, 16, 16), 128, 128, (3, 3), ConvGeom(**G3), dict(res="add", ln="keep")),
    ("deep_ragged_192", (1, 3, 10, 10), 256, 192, (3, 3, 3), ConvGeom(**G333), dict(res="mix")),
    ("deep_1x1_512_8steps", (1, 2, 16, 16), 512, 256, (1, 1), ConvGeom(), {}),
    ("deep_replicate_s2", (1, 6, 16, 16), 256, 256, (3, 3, 3),
     ConvGeom(kt=3, kh=3, kw=3, st=2, pt=1, ph=1, pw=1, ph_hi=1, pw_hi=1), dict(tmode="replicate")),
]


@pytest.mark.parametrize("dtype", _tier(DTYPES4, IDS4, main=[torch.bfloat16]))
@pytest.mark.parametrize("ring", _tier(["deep", "deep_pointers", "two_slots"], ["deep", "deep_pointers", "two_slots"], main=["deep", "two_slots"]))
@pytest.mark.parametrize("case", DEEP_CASES, ids=[c[0] for c in DEEP_CASES])
def test_conv_128_tile_rings(case, ring, dtype, vt_opts):
    vt_opts(conv_tile=128, conv_ws=0, conv_deep=(0 if ring == "two_slots" else 1), conv_buf=(0 if ring == "deep_pointers" else 1))
    plan = _check_conv(case, dtype)
    steps = math.prod(case[4]) * case[2] // (64 if dtype in H16 else 32)
    assert plan["tile"] == (128, 128) and plan["deep_ring"] == (ring != "two_slots" and steps >= 8), (plan, steps)


# Split-K over the time taps (vt_conv_work_bytes): 3x3x3 convolutions on few pixels -- no more tiles than CUs -- run as three tap-plane
# launches-in-one into fp32 partials + a reduction that owns bias / residual / rounding.  The 512-channel mid blocks of a v1.1 chunk
# (M = 4 096), cache mode, a residual, the 8-wave tile (M = 20 480: the mid blocks of the benchmark batch), Cin = 256.
SPLITK_CASES = [
    ("sk_3d_512_m4096", (1, 4, 32, 32), 512, 512, (3, 3, 3), ConvGeom(**G333), dict(tmode="replicate")),
    ("sk_3d_512_cache_res", (1, 4, 32, 32), 512, 512, (3, 3, 3), ConvGeom(**G333), dict(tmode="cache", res="add")),
    ("sk_3d_512_m20480_8wave", (4, 5, 32, 32), 512, 512, (3, 3, 3), ConvGeom(**G333), dict(res="add")),
    ("sk_3d_256_m1024", (1, 4, 16, 16), 256, 256, (3, 3, 3), ConvGeom(**G333), {}),
    ("sk_3d_512_ln_after", (1, 5, 32, 32), 512, 512, (3, 3, 3), ConvGeom(**G333), dict(ln="only")),
    ("sk_2d_512_rows", (1, 4, 32, 32), 512, 512, (3, 3), ConvGeom(**G3), dict(res="add")),                  # no time taps: planes = the rows of the 3 x 3
    ("sk_2d_512_m5120", (1, 5, 32, 32), 512, 512, (3, 3), ConvGeom(**G3), {}),
]


@pytest.mark.parametrize("dtype", H16, ids=H16_IDS)
@pytest.mark.parametrize("split", [1, 0], ids=["split", "whole"])
@pytest.mark.parametrize("case", SPLITK_CASES, ids=[c[0] for c in SPLITK_CASES])
def test_conv_split_k(case, split, dtype, vt_opts):
    vt_opts(conv_splitk=split)
    plan = _check_conv(case, dtype)
    extra = 1 if "ln" in case[6] else 0                      # Cout = 512: the LayerNorm is its own launch either way
    assert plan["launches"] == (2 if split else 1) + extra, plan


def test_conv_split_k_needs_scratch_and_small_m(vt_opts):
    """no scratch in the descriptor -> the call runs whole; many pixels per clip (more tiles than CUs) -> no split is asked for"""
    import ctypes as C

    lib = L.load()
    x = _act(1, 4, 32, 32, 512, torch.bfloat16, 1)
    assert L.get_option("conv_splitk") == 1          # on by default: the decision is per clip, never per batch
    wt = torch.randn((512, 512, 3, 3, 3), generator=torch.Generator().manual_seed(2)) / math.sqrt(512 * 27)
    w = pack_conv_weight(wt, torch.bfloat16, cin_stored=512).to(DEV)
    ops.CONV_RECORD = []
    y = ops.conv(x, w, None, ConvGeom(**G333), cout=512)
    rec, ops.CONV_RECORD = ops.CONV_RECORD, None
    d = rec[0][0]
    assert d.work and lib.vt_conv_work_bytes(C.byref(d)) == 3 * 4096 * 512 * 4 == d.work_bytes
    d2 = L.ConvDesc()
    C.memmove(C.byref(d2), C.byref(d), C.sizeof(d))
    y2 = torch.empty_like(y)
    d2.y, d2.work, d2.work_bytes = y2.data_ptr(), None, 0
    L.check(lib.vt_conv(C.byref(d2), None), "vt_conv")
    torch.cuda.synchronize()
    assert ops.conv_plan(d2)["launches"] == 1 and rel_err(y2, y) < 1e-2 and not torch.equal(y2, y)      # another summation order
    big = _act(4, 20, 64, 64, 512, torch.bfloat16, 1)                                                     # 327 680 pixels: 2 560 tiles
    ops.CONV_RECORD = []
    ops.conv(big, w, None, ConvGeom(**G333), cout=512)
    rec, ops.CONV_RECORD = ops.CONV_RECORD, None
    assert not rec[0][0].work and lib.vt_conv_work_bytes(C.byref(rec[0][0])) == 0


@pytest.mark.parametrize("coalesced", [True, False], ids=["lds_epilogue", "vector_epilogue"])
@pytest.mark.parametrize("case", [c for c in CONV_CASES_LARGE if c[3] % 256 == 0], ids=[c[0] for c in CONV_CASES_LARGE if c[3] % 256 == 0])
def test_conv_8wave_plain_epilogues(case, coalesced, vt_opts):
    """8-wave tile without LayerNorm: the LDS-transposed epilogue (16-bit full tiles; + residual / alpha-mix / interleaved
    output frames) and the MFMA-layout vector epilogue it replaces (option conv_ldsepi = 0), both against the reference"""
    vt_opts(conv_ldsepi=(1 if coalesced else 0), conv_fuse_ln256=(1 if coalesced else 0))
    plan = _check_conv(case, torch.bfloat16)
    if plan["tile"] == (256, 256) and not plan["ln_fused"]:
        name, (B, T, H, W), cin, cout, kdims, geom, ex = case
        To, Ho, Wo = geom.out_dims(T, H, W)
        full = (B * To * Ho * Wo) % 256 == 0 and "ncthw" not in ex and ex.get("res") != "mix_up"
        assert plan["lds_epilogue"] == (coalesced and full), (name, plan)


LN256_CASES = [c for c in CONV_CASES if c[0] in ("conv2d_ln256_only", "conv2d_ln256_res_keep", "temporal_ln256_only")] + \
              [c for c in CONV_CASES_LARGE if c[0] in ("L_conv2d_256_256_ln", "L_temporal_k3_256_ln_only")]


@pytest.mark.parametrize("dtype", DTYPES4, ids=IDS4)
@pytest.mark.parametrize("mode", ["unfused", "fused"])
@pytest.mark.parametrize("case", LN256_CASES, ids=[c[0] for c in LN256_CASES])
def test_conv_ln256_variants(case, mode, dtype, vt_opts):
    vt_opts(conv_tile=256, conv_fuse_ln256=(mode != "unfused"))
    plan = _check_conv(case, dtype)
    assert plan["tile"] == (256, 256) and plan["ln_fused"] == (mode != "unfused") and plan["launches"] == (2 if mode == "unfused" else 1)


LDSEPI_CASES = [c for c in CONV_CASES if c[0] in ("conv2d_3x3_128_128", "conv2d_3x3_256_128_res", "temporal_k3_tinner", "conv2d_ln_fused",
                                                  "conv2d_ln_fused_res_keep", "temporal_ln_fused")]


@pytest.mark.parametrize("dtype", _tier(DTYPES4, IDS4, main=[torch.float32, torch.bfloat16]))
@pytest.mark.parametrize("case", LDSEPI_CASES, ids=[c[0] for c in LDSEPI_CASES])
def test_conv_without_lds_epilogue(case, dtype, vt_opts):
    vt_opts(conv_ldsepi=0, conv_ws=0)
    plan = _check_conv(case, dtype)
    assert plan["tile"] == (128, 128) and not plan["ln_fused"]


# The host reference of a (case, arithmetic) does not depend on the options a test sets: the parametrisations that replay a case under
# several options (gather forms, rings, split / whole, fused / unfused) share one evaluation of the torch statement -- the CPU
# convolutions of the large cases are most of this file's run time.  A few most recent entries only (their outputs reach 100 MB).
_REF_CACHE = {}


def _ref_cached(key, fn):
    if key in _REF_CACHE:
        _REF_CACHE[key] = _REF_CACHE.pop(key)
        return _REF_CACHE[key]
    val = fn()
    _REF_CACHE[key] = val
    while len(_REF_CACHE) > 8:
        _REF_CACHE.pop(next(iter(_REF_CACHE)))
    return val


def _check_conv(case, dtype, keep_outputs=None):
    """one case against the host reference; returns vt_conv_plan of the launch (keep_outputs: a list that receives the device results)"""
    name, (B, T, H, W), cin, cout, kdims, geom, ex = case
    mode, dtype = dtype, (torch.float32 if dtype == X3 else dtype)      # x3: fp32 tensors, split weight planes
    x = _act(B, T, H, W, cin, dtype, 1)
    g = torch.Generator().manual_seed(2)
    fan = cin * math.prod(kdims)
    wt = torch.randn((cout, cin) + tuple(kdims), generator=g) / math.sqrt(fan)
    w_rows = pack_conv_weight(wt, dtype, cin_stored=x.shape[-1]).to(DEV)
    w = pack_split3(w_rows) if mode == X3 else w_rows
    bias = None if ex.get("nobias") else _rand((cout,), torch.float32, 3, 0.1)
    kw = {}
    To, Ho, Wo = geom.out_dims(T, H, W)
    out_dtype = dtype
    if "ncthw" in ex:
        kw.update(out_layout=L.VT_NCTHW, t_trim=ex["ncthw"])
        out_dtype = torch.float32
    if ex.get("res") == "add":
        kw.update(res=_act(B, To, Ho, Wo, cout, out_dtype, 4), res_mode=L.VT_RES_ADD)
    elif ex.get("res") == "mix":
        kw.update(res=_act(B, To, Ho, Wo, cout, out_dtype, 4), res_mode=L.VT_RES_MIX,
                  mix_factor=torch.tensor([0.37], device=DEV))
    elif ex.get("res") == "mix_up":
        kw.update(res=x, res_mode=L.VT_RES_MIX, res_tshift=1, mix_factor=torch.tensor([-0.6], device=DEV))
    if ex.get("tmode") == "replicate":
        kw.update(tmode=L.VT_TPAD_REPLICATE)
    elif ex.get("tmode") == "cache":
        kw.update(tmode=L.VT_TPAD_CACHE, cache=_act(B, geom.pt + 1, H, W, cin, dtype, 5))
    if "ln" in ex:
        gam, bet = _rand((cout,), torch.float32, 6, 0.5) + 1.0, _rand((cout,), torch.float32, 7, 0.2)
        keep = ex["ln"] == "keep"
        ops.CONV_RECORD = []
        out = ops.conv(x, w, bias, geom, cout=cout, ln=(gam, bet, 1e-6, True), ln_keep_y=keep, **kw)
        rec, ops.CONV_RECORD = ops.CONV_RECORD, None
        torch.cuda.synchronize()
        ref = _ref_cached((name, str(mode)), lambda: R.conv(x.cpu(), w_rows.cpu(), None if bias is None else bias.cpu(), geom, cout=cout,
                                                            ln=(gam.cpu(), bet.cpu(), 1e-6, True), ln_keep_y=keep, **_cpu(kw)))
        outs, refs = (out if keep else (out,)), (ref if keep else (ref,))
        if keep_outputs is not None:
            keep_outputs.extend(outs)
        for o, r in zip(outs, refs):
            assert o.shape == r.shape and o.dtype == r.dtype and torch.isfinite(o.float()).all()
            e = rel_err(o, r)
            # the normalised output amplifies the bf16 rounding of y when the library normalises the stored y
            assert e < 2 * TOL[mode], f"{name} {mode}: rel_err={e}"
        return ops.conv_plan(rec[0][0])
    ops.CONV_RECORD = []
    y = ops.conv(x, w, bias, geom, cout=cout, **kw)
    rec, ops.CONV_RECORD = ops.CONV_RECORD, None
    torch.cuda.synchronize()
    yr = _ref_cached((name, str(mode)), lambda: R.conv(x.cpu(), w_rows.cpu(), None if bias is None else bias.cpu(), geom, cout=cout, **_cpu(kw)))   # reference on the host
    assert y.shape == yr.shape and y.dtype == yr.dtype
    assert torch.isfinite(y.float()).all()
    if keep_outputs is not None:
        keep_outputs.append(y)
    e = rel_err(y, yr)
    print(f"{name} {mode}: rel_err={e:.3e}")
    assert e < TOL[mode], f"{name} {mode}: rel_err={e}"
    return ops.conv_plan(rec[0][0])


@pytest.mark.parametrize("dtype", DTYPES + [F16], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("cout,hw", [(128, (16, 16)), (256, (8, 8)), (512, (5, 7))])
def test_conv_output_frame_interleave(cout, hw, dtype):
    """yt_mul / yt_off: two k=2 launches fill the even / odd frames of one output (the parity convs of a time
    up-sampler), through every epilogue: LDS-transposed (Cout 128, full tiles), vector, scalar (ragged)."""
    B, T, (H, W), cin = 2, 3, hw, 128
    x = _act(B, T, H, W, cin, dtype, 1)
    g = torch.Generator().manual_seed(2)
    geom = ConvGeom(kt=2, kh=3, kw=3, pt=1, ph=1, pw=1, ph_hi=1, pw_hi=1)
    bias = _rand((cout,), torch.float32, 3, 0.1)
    res = _act(B, T, H, W, cout, dtype, 4)
    mf = torch.tensor([0.2], device=DEV)
    y = torch.full((B, 2 * T, H, W, cout), float("nan"), dtype=dtype, device=DEV)
    yr = torch.zeros((B, 2 * T, H, W, cout), dtype=dtype)
    for par in (0, 1):
        wt = torch.randn((cout, cin, 2, 3, 3), generator=g) / math.sqrt(cin * 18)
        w = pack_conv_weight(wt, dtype, cin_stored=x.shape[-1]).to(DEV)
        kw = dict(res=res, res_mode=L.VT_RES_MIX, mix_factor=mf)
        ops.conv(x, w, bias, geom, cout=cout, out=y, out_t=(2, par), **kw)
        R.conv(x.cpu(), w.cpu(), bias.cpu(), geom, cout=cout, out=yr, out_t=(2, par), **_cpu(kw))
    torch.cuda.synchronize()
    assert torch.isfinite(y.float()).all() and rel_err(y, yr) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES + [F16], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("cout,hw", [(128, (8, 16)), (256, (8, 8)), (512, (5, 7))])
def test_conv_output_pixel_interleave(cout, hw, dtype):
    """ys_mul = 2: four 2x2 launches fill the parity classes of a 2H x 2W output (spatial up-sampler)"""
    from vidtok_amd.packing import space_upsample_parity_weights

    B, T, (H, W), cin = 1, 2, hw, 128
    x = _act(B, T, H, W, cin, dtype, 1)
    g = torch.Generator().manual_seed(2)
    w3 = torch.randn((cout, cin, 3, 3), generator=g) / math.sqrt(cin * 9)
    bias = _rand((cout,), torch.float32, 3, 0.1)
    y = torch.full((B, T, 2 * H, 2 * W, cout), float("nan"), dtype=dtype, device=DEV)
    for py in (0, 1):
        for px in (0, 1):
            w = pack_conv_weight(space_upsample_parity_weights(w3, py, px), dtype, cin_stored=x.shape[-1]).to(DEV)
            geom = ConvGeom(kh=2, kw=2, ph=1 - py, pw=1 - px, ph_hi=py, pw_hi=px)
            ops.conv(x, w, bias, geom, cout=cout, out=y, out_s=(py, px))
    torch.cuda.synchronize()
    # against the plain statement: 3x3 conv over the nearest-x2 up-sampled frame
    w9 = pack_conv_weight(w3, dtype, cin_stored=x.shape[-1])
    yr = R.conv(x.cpu(), w9, bias.cpu(), ConvGeom(ups_s=1, **G3), cout=cout)
    assert torch.isfinite(y.float()).all() and rel_err(y, yr) < 1.5 * TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES + [F16], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("Z,M,N,K,bcast,use_bias", [(3, 80, 48, 128, False, False), (2, 512, 64, 512, True, True),
                                                      (5, 16, 16, 16, False, False), (2, 100, 512, 104, False, True),
                                                      (2, 1024, 1024, 512, False, False)])
def test_gemm_nt(Z, M, N, K, bcast, use_bias, dtype):
    a = _rand((1 if bcast else Z, M, K), dtype, 1, 1.0 / math.sqrt(K))
    b = _rand((Z, N, K), dtype, 2)
    bias = _rand((N,), torch.float32, 3) if use_bias else None
    for od in (dtype, torch.float32):
        y = ops.gemm_nt(a, b, out_dtype=od, bias=bias)
        yr = R.gemm_nt(a.cpu(), b.cpu(), out_dtype=od, bias=None if bias is None else bias.cpu())
        e = rel_err(y, yr)
        assert y.shape == (Z, M, N) and e < TOL[dtype], f"gemm {Z,M,N,K} {dtype}->{od}: {e}"


@pytest.mark.parametrize("Z,S,ld,use_bias", [(2, 64, 64, True), (3, 256, 256, False), (2, 1024, 1024, True), (1, 4096, 4096, True), (2, 128, 136, True)],
                         ids=["one_q_tile", "s256", "s1024_benchmark_size", "s4096_512x512_input", "padded_vT"])
@pytest.mark.parametrize("dt", H16, ids=H16_IDS)
def test_flash_attention(Z, S, ld, use_bias, dt):
    """vt_flash_attention (one launch, online softmax, nothing S x S in memory) against the fp32 statement of the attention and against
    the operator sequence it replaces (batched GEMM -> row softmax -> batched GEMM) on the same tensors: S = 1 024 is the attention of
    a 256 x 256 input, S = 4 096 of a 512 x 512 one"""
    C_ = 512
    q, k = _rand((Z, S, C_), dt, 1), _rand((Z, S, C_), dt, 2)
    v = _rand((Z, S, C_), dt, 3)
    vT = torch.zeros((Z, C_, ld), dtype=dt, device=DEV)
    vT[:, :, :S] = v.transpose(1, 2)
    bias = _rand((C_,), torch.float32, 4, 0.3) if use_bias else None
    scale = C_ ** -0.5
    assert ops.flash_attention_supported(q, vT)
    o = ops.flash_attention(q, k, vT, bias, scale)
    torch.cuda.synchronize()
    ref = R.flash_attention(q.cpu(), k.cpu(), vT.cpu(), None if bias is None else bias.cpu(), scale)
    e = rel_err(o, ref)
    assert o.shape == (Z, S, C_) and torch.isfinite(o.float()).all() and e < TOL[dt], f"flash attention Z={Z} S={S}: {e}"
    # the operator path on the same tensors (P normalised and rounded to bf16 before the second product: the two differ by roundings only)
    s_ = ops.gemm_nt(q, k, out_dtype=torch.float32)
    p_ = ops.softmax_rows(s_, scale, dt, ld_out=ld)
    o2 = ops.gemm_nt(p_, vT, bias=bias)
    e2 = rel_err(o, o2)
    print(f"flash attention Z={Z} S={S}: vs fp32 statement {e:.3e}, vs operator path {e2:.3e}")
    assert e2 < TOL[dt]


def test_flash_attention_declines_other_shapes(vt_opts):
    q = _rand((1, 96, 512), torch.bfloat16, 1)
    assert not ops.flash_attention_supported(q, torch.zeros((1, 512, 96), dtype=torch.bfloat16, device=DEV))          # S % 64
    q = _rand((1, 64, 256), torch.bfloat16, 1)
    assert not ops.flash_attention_supported(q, torch.zeros((1, 256, 64), dtype=torch.bfloat16, device=DEV))          # C != 512
    q = _rand((1, 64, 512), torch.float32, 1)
    assert not ops.flash_attention_supported(q, torch.zeros((1, 512, 64), dtype=torch.float32, device=DEV))           # fp32 storage
    q = _rand((1, 64, 512), torch.bfloat16, 1)
    vt_opts(attn_flash=0)
    assert not ops.flash_attention_supported(q, torch.zeros((1, 512, 64), dtype=torch.bfloat16, device=DEV))          # switched off


@pytest.mark.parametrize("C", [32, 64, 128, 256, 512, 1024])
@pytest.mark.parametrize("din,dout", [(torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16),
                                      (torch.float32, torch.bfloat16), (torch.bfloat16, torch.float32),
                                      (F16, F16), (torch.float32, F16), (F16, torch.float32)])
@pytest.mark.parametrize("silu", [True, False])
def test_layernorm_act(C, din, dout, silu):
    x = _rand((3, 7, 11, C), din, 1, 2.0) + 0.5
    gm, bt = 1 + 0.1 * _rand((C,), torch.float32, 2), 0.1 * _rand((C,), torch.float32, 3)
    y = ops.layernorm_act(x, gm, bt, silu=silu, out_dtype=dout)
    yr = R.layernorm_act(x.cpu(), gm.cpu(), bt.cpu(), silu=silu, out_dtype=dout)
    e = rel_err(y, yr)
    assert e < (2e-5 if dout == torch.float32 else 1e-2), f"LN C={C} {din}->{dout} silu={silu}: {e}"


@pytest.mark.parametrize("cols", [16, 100, 1024])
@pytest.mark.parametrize("dtype", DTYPES + [F16], ids=["f32", "bf16", "f16"])
def test_softmax_rows(cols, dtype):
    s = _rand((5, 33, cols), torch.float32, 1, 8.0)
    p = ops.softmax_rows(s, 0.0442, dtype)
    pr = R.softmax_rows(s.cpu(), 0.0442, dtype)
    assert rel_err(p, pr) < (1e-5 if dtype == torch.float32 else 1e-2)
    assert (p.float().sum(-1) - 1).abs().max() < (1e-5 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("dtype", DTYPES + [F16], ids=["f32", "bf16", "f16"])
def test_layout_roundtrip(dtype):
    x = _rand((2, 3, 5, 6, 7), torch.float32, 1)
    y = ops.ncthw_to_ndhwc(x, dtype, tpad=3)
    yr = R.ncthw_to_ndhwc(x.cpu(), dtype, tpad=3)
    assert torch.equal(y.cpu(), yr) and y.shape == (2, 8, 6, 7, 8)
    back = ops.ndhwc_to_ncthw(y, 3, ttrim=3)
    assert torch.equal(back, x.to(dtype).float())
    z = _rand((2, 16, 2, 4, 4), torch.float32, 2)
    assert torch.equal(ops.ndhwc_to_ncthw(ops.ncthw_to_ndhwc(z, dtype), 16), z.to(dtype).float())


@pytest.mark.parametrize("dtype", DTYPES + [F16], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("tmode", [L.VT_TPAD_ZERO, L.VT_TPAD_REPLICATE, L.VT_TPAD_CACHE, L.VT_TPAD_ZERO_BACK])
@pytest.mark.parametrize("Ti", [6, 9], ids=["T6", "T9odd"])
def test_time_avgpool(dtype, tmode, Ti):
    """Ti = 9: a v1.0 clip whose length is 2 or 3 mod 4 reaches the time down-sampler with an odd frame count
    (6 frames pad to 9); avg_pool3d floors, the last frame is dropped."""
    x = _act(2, Ti, 4, 4, 128, dtype, 1)
    cache = _act(2, 1, 4, 4, 128, dtype, 2) if tmode == L.VT_TPAD_CACHE else None
    y = ops.time_avgpool3s2(x, tmode, cache)
    yr = R.time_avgpool3s2(x.cpu(), tmode, None if cache is None else cache.cpu())
    assert y.shape == (2, Ti // 2, 4, 4, 128) and rel_err(y, yr) < (1e-6 if dtype == torch.float32 else 8e-3)


@pytest.mark.parametrize("dtype", DTYPES + [F16], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("Ti", [1, 2, 5])
def test_time_lerp2x(dtype, Ti):
    x = _act(2, Ti, 4, 4, 128, dtype, 1)
    y = ops.time_lerp2x(x)
    yr = R.time_lerp2x(x.cpu())
    assert y.shape == yr.shape and rel_err(y, yr) < (1e-6 if dtype == torch.float32 else 8e-3)
    out = torch.zeros((2, 2 * Ti + 3, 4, 4, 128), dtype=dtype, device=DEV)                  # in place at a frame offset
    ops.time_lerp2x(x, out=out, out_t0=2)
    assert torch.equal(out[:, 2:2 + 2 * Ti], y) and float(out[:, :2].float().abs().max()) == 0


@pytest.mark.parametrize("dtype", DTYPES + [F16], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("nh,T,skip", [(1, 4, 2), (2, 5, 4), (4, 8, 8), (2, 1, 0), (1, 3, 7)])
def test_time_lerp2x_cat(dtype, nh, T, skip):
    """interpolation of [head | x] without assembling it, minus the first `skip` frames (v1.1 chunks after the first:
    torch.cat + F.interpolate + slice, reference model_3dcausal_v1_1.py:331-341) = the bits of the assembled form"""
    head, x = _act(2, nh, 4, 4, 128, dtype, 1), _act(2, T, 4, 4, 128, dtype, 2)
    y = ops.time_lerp2x_cat(head, x, skip)
    full = ops.time_lerp2x(torch.cat([head, x], dim=1).contiguous())
    assert y.shape[1] == 2 * (nh + T) - skip and torch.equal(y, full[:, skip:])
    yr = R.time_lerp2x(torch.cat([head, x], dim=1).cpu())[:, skip:]
    assert rel_err(y, yr) < (1e-6 if dtype == torch.float32 else 8e-3)


def test_gather_frames():
    x = _act(2, 5, 4, 4, 128, torch.bfloat16, 1)
    idx = [4, 0, 0, 3, 2, 2]
    assert torch.equal(ops.gather_frames(x, idx), x[:, idx])
    xf = _act(1, 3, 2, 2, 8, torch.float32, 2)
    assert torch.equal(ops.gather_frames(xf, [2]), xf[:, 2:3])
    # into a preallocated destination at a frame offset (v1.1 cache / chunk assembly: no torch.cat)
    out = torch.full((2, 9, 4, 4, 128), 3.0, dtype=torch.bfloat16, device=DEV)
    ops.gather_frames(x, [1, 4], out=out, out_t0=3)
    ops.gather_frames(x, [0], out=out, out_t0=8)
    assert torch.equal(out[:, 3:5], x[:, [1, 4]]) and torch.equal(out[:, 8:9], x[:, 0:1])
    assert float(out[:, :3].float().min()) == 3.0 and float(out[:, 5:8].float().max()) == 3.0
    ii = torch.arange(2 * 3 * 5, dtype=torch.int32, device=DEV).reshape(2, 3, 5)          # any [B, T, ...] tensor: bytes
    oi = torch.zeros((2, 7, 5), dtype=torch.int32, device=DEV)
    ops.gather_frames(ii, [0, 1, 2], out=oi, out_t0=4)
    assert torch.equal(oi[:, 4:7], ii) and int(oi[:, :4].abs().max()) == 0


@pytest.mark.parametrize("with_noise", [True, False])
def test_kl_sample(with_noise):
    h = _rand((3, 8, 2, 4, 4), torch.float32, 1, 3.0)
    h[0, 4:, 0, 0, 0] = torch.tensor([50.0, -50.0, 0.0, 1.0], device=DEV)   # exercises the clamp
    noise = _rand((3, 4, 2, 4, 4), torch.float32, 2) if with_noise else None
    z, kl = ops.kl_sample(h, noise)
    zr, klr = R.kl_sample(h.cpu(), None if noise is None else noise.cpu())
    assert rel_err(z, zr) < 1e-6 and abs(float(kl) - float(klr)) < 1e-5 * abs(float(klr))


@pytest.mark.parametrize("levels", [[8, 8, 8, 8, 8], [8, 5, 5, 5], [7, 5, 5, 5, 5], [8] * 6])
def test_fsq_quantize_bit_exact(levels):
    D = len(levels)
    h = _rand((4, D, 5, 16, 16), torch.float32, 1, 1.5)
    z, idx = ops.fsq_quantize(h, levels)
    zr, idxr = R.fsq_quantize(h.cpu(), levels)          # CPU torch: the reference's own arithmetic
    n_bad = int((idx.cpu() != idxr).sum())
    print(f"fsq levels={levels}: {n_bad} / {idx.numel()} index mismatches vs CPU torch")
    assert idx.dtype == torch.int32 and n_bad <= 1      # only a 1-ulp tanh difference on a rounding boundary
    if n_bad == 0:
        assert torch.equal(z.cpu(), zr)
    # round trip: indices -> codes == quantised z, bit for bit (SURVEY.md section 8c identity)
    assert torch.equal(ops.fsq_indices_to_codes(idx, levels), z)


def test_fsq_indices_to_codes_exhaustive():
    levels = [8, 8, 8, 8, 8]
    allidx = torch.arange(32768, dtype=torch.int32, device=DEV).reshape(1, 8, 64, 64)
    codes = ops.fsq_indices_to_codes(allidx, levels)
    ref = R.fsq_indices_to_codes(allidx.cpu(), levels)
    assert torch.equal(codes.cpu(), ref)
    # quantising the exact code points returns the same indices (codes sit mid-bin after the bound)
    zin = torch.atanh(((codes * 4 + 0.5) / 3.5035).clamp(-0.9999, 0.9999)) - 0.14369
    assert torch.equal(ops.fsq_quantize(zin.contiguous(), levels)[1], allidx)


@pytest.mark.parametrize("levels,B", [([8, 8, 8, 8, 8], 2), ([8, 8, 8, 8], 1)])
def test_fsq_aux_stats(levels, B):
    h = _rand((B, len(levels), 2, 8, 8), torch.float32, 1, 0.7)
    st = ops.fsq_aux_stats(h, levels, 100.0).cpu()
    ref = R.fsq_aux_stats(h.cpu(), levels, 100.0)
    print("fsq aux", st.tolist(), ref.tolist())
    assert torch.allclose(st, ref, rtol=2e-4, atol=2e-5)


def test_fsq_aux_avg_and_entropy():
    """the batch-mean code distribution (what the reference all-reduces across ranks) and its entropy"""
    levels = [8, 8, 8, 5, 5]
    h = _rand((2, 5, 2, 8, 8), torch.float32, 1, 0.7)
    st, avg = ops.fsq_aux_stats(h, levels, 100.0, return_avg=True)
    str_, avgr = R.fsq_aux_stats(h.cpu(), levels, 100.0, return_avg=True)
    assert avg.shape == (8 * 8 * 8 * 5 * 5,) and torch.allclose(avg.cpu(), avgr, rtol=2e-4, atol=1e-7)
    assert abs(float(avg.sum()) - 1.0) < 1e-4
    assert torch.allclose(ops.entropy(avg).cpu(), st[1].cpu(), rtol=1e-5)
    assert torch.allclose(ops.entropy(avg).cpu(), R.entropy(avgr), rtol=2e-4)


@pytest.mark.parametrize("din,dout", [(torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16), (F16, F16)],
                         ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("scope", [L.VT_GN_FRAME, L.VT_GN_PIXEL, L.VT_GN_CLIP, ops.GN_POS], ids=["frame", "pixel", "clip", "pos"])
@pytest.mark.parametrize("C", [128, 512])
def test_groupnorm_act(C, scope, din, dout):
    x = _act(2, 5, 12, 10, C, din, 1) * 1.5 + 0.3
    gam, bet = _rand((C,), torch.float32, 2, 0.3) + 1.0, _rand((C,), torch.float32, 3, 0.2)
    y = ops.groupnorm_act(x, gam, bet, scope=scope, silu=True, out_dtype=dout)
    yr = R.groupnorm_act(x.cpu(), gam.cpu(), bet.cpu(), scope=scope, silu=True, out_dtype=dout)
    assert y.shape == yr.shape and y.dtype == yr.dtype
    assert rel_err(y, yr) < (2e-5 if dout == torch.float32 else 1.2e-2)
    y2 = ops.groupnorm_act(x, gam, bet, scope=scope, silu=False, out_dtype=dout)
    assert rel_err(y2, R.groupnorm_act(x.cpu(), gam.cpu(), bet.cpu(), scope=scope, silu=False, out_dtype=dout)) < (
        2e-5 if dout == torch.float32 else 1.2e-2)


def test_channel_linear():
    g = torch.Generator().manual_seed(3)
    x = torch.randn((2, 8, 3, 5, 7), generator=g).to(DEV)
    w, b = torch.randn((6, 8), generator=g).to(DEV), torch.randn((6,), generator=g).to(DEV)
    y = ops.channel_linear(x, w, b)
    assert y.shape == (2, 6, 3, 5, 7) and rel_err(y, R.channel_linear(x.cpu(), w.cpu(), b.cpu())) < 1e-6
    assert rel_err(ops.channel_linear(x, w, None), R.channel_linear(x.cpu(), w.cpu(), None)) < 1e-6


def test_library_is_the_hip_extension():
    lib = L.load()
    assert lib.vt_version() >= 100 and L.LIB_PATH.endswith("libvidtok_amd.so")
    with pytest.raises(L.VtError):
        ops.layernorm_act(torch.zeros(4, 128), torch.ones(128), torch.zeros(128), silu=True)  # CPU tensor


# Zero-padded time taps skipped per tile (option conv_tskip, default on): causal convolutions of the v1.0 models (tmode ZERO) whose
# tiles lie inside one output frame start their K walk behind the tap planes that read only the zero frames in front of the clip.
# The skipped products are exact zeros, so the result must be the SAME BITS as the full walk -- in every arithmetic, on both tiles,
# with stride 2 in time (one plane skipped for frame 0) and for the k = 2 parity convolutions of a time up-sampler.
TSKIP_CASES = [
    ("ts_temporal_k3_256", (2, 5, 16, 16), 256, 256, (3,), ConvGeom(kt=3, pt=2), dict(res="add")),
    ("ts_temporal_k3_512_ln", (1, 5, 16, 16), 256, 256, (3,), ConvGeom(kt=3, pt=2), dict(ln="keep")),
    ("ts_conv3d_333_512", (1, 5, 16, 16), 512, 512, (3, 3, 3), ConvGeom(**G333), dict(res="add")),
    ("ts_conv3d_333_128", (1, 4, 16, 16), 128, 128, (3, 3, 3), ConvGeom(**G333), {}),
    ("ts_timedown_s2_mix", (1, 6, 16, 16), 128, 128, (3, 3, 3),
     ConvGeom(kt=3, kh=3, kw=3, st=2, pt=1, ph=1, pw=1, ph_hi=1, pw_hi=1), dict(res="mix")),
    ("ts_parity_kt2_256", (1, 4, 32, 32), 256, 256, (2, 3, 3), ConvGeom(kt=2, kh=3, kw=3, pt=1, ph=1, pw=1, ph_hi=1, pw_hi=1), dict(res="mix")),
    ("ts_noncausal_sym", (1, 4, 16, 16), 128, 128, (3, 3, 3), ConvGeom(kt=3, kh=3, kw=3, pt=1, pt_hi=1, ph=1, pw=1, ph_hi=1, pw_hi=1), {}),
]


@pytest.mark.parametrize("tile", [0, 256], ids=["auto_tile", "tile256"])
@pytest.mark.parametrize("dtype", _tier(DTYPES4, IDS4, main=[torch.float32, torch.bfloat16, X3]))
@pytest.mark.parametrize("case", TSKIP_CASES, ids=[c[0] for c in TSKIP_CASES])
def test_conv_time_tap_skip_is_bit_exact(case, dtype, tile, vt_opts):
    if tile == 256 and case[3] % 256 != 0:
        pytest.skip("256 tile needs Cout % 256")
    outs = {}
    for skip in (1, 0):
        vt_opts(conv_tskip=skip, conv_tile=tile, conv_ws=0)
        keep = []
        _check_conv(case, dtype, keep_outputs=keep)
        outs[skip] = keep
    assert len(outs[1]) == len(outs[0]) >= 1
    for a, b in zip(outs[1], outs[0]):
        assert torch.equal(a, b), f"{case[0]}: the skipped walk differs from the full one"


def test_conv_split_k_does_not_depend_on_the_batch(vt_opts):
    """Split-K is decided from ONE clip's geometry (To, Ho, Wo, Cout, K), never from B: a clip's result is the same bits
    whether it is convolved alone or inside a batch of four (the round-4 rule looked at the launch's pixel
    count and tied a clip's bits to its batch)."""
    import ctypes as C

    lib = L.load()
    vt_opts(conv_splitk=1)
    wt = torch.randn((512, 512, 3, 3, 3), generator=torch.Generator().manual_seed(2)) / math.sqrt(512 * 27)
    w = pack_conv_weight(wt, torch.bfloat16, cin_stored=512).to(DEV)
    xb = _act(4, 5, 32, 32, 512, torch.bfloat16, 1)                      # the mid block of the benchmark batch: 4 x 5 120 pixels
    ys = {}
    for name, x in (("batch", xb), ("alone", xb[2:3].contiguous())):
        ops.CONV_RECORD = []
        ys[name] = ops.conv(x, w, None, ConvGeom(**G333), cout=512)
        rec, ops.CONV_RECORD = ops.CONV_RECORD, None
        assert rec[0][0].work and ops.conv_plan(rec[0][0])["launches"] == 2, name     # both split: 160 tiles per clip <= 256 CUs
        assert lib.vt_conv_work_bytes(C.byref(rec[0][0])) == 3 * x.shape[0] * 5120 * 512 * 4
    torch.cuda.synchronize()
    assert torch.equal(ys["batch"][2:3], ys["alone"])


@pytest.mark.parametrize("mode", ["f32", "bf16", "x3", "f16"])
def test_pack_conv_weight_on_device_equals_host_statements(mode):
    """vt_pack_conv_weight (the device packer PackedCache uses for GPU parameters: no ATen kernel in the process) against the torch
    statements of vidtok_amd/packing.py -- plain rows for 1-, 2- and 3-D kernels with channel padding, the pre-summed taps of the
    up-samplers' parity classes (time: early / late; space: the four (py, px)), the split-bf16 container: SAME BITS."""
    import functools

    from vidtok_amd import packing as P

    dtype = {"bf16": torch.bfloat16, "f16": F16}.get(mode, torch.float32)
    bits = torch.int16 if mode in ("bf16", "f16") else torch.int32
    g = torch.Generator().manual_seed(11)
    cases = [((16, 3, 3, 3, 3), 8, None, None), ((24, 20, 3), 24, None, None), ((8, 12, 3, 3), 16, None, None), ((40, 8, 1, 1, 1), 8, None, None)]
    for early in (True, False):
        cases.append(((16, 8, 3, 3, 3), 8, functools.partial(P.time_upsample_parity_weights, early=early), functools.partial(P.time_upsample_parity_mix, early=early)))
    for py in (0, 1):
        for px in (0, 1):
            cases.append(((16, 24, 3, 3), 24, functools.partial(P.space_upsample_parity_weights, py=py, px=px),
                          functools.partial(P.space_upsample_parity_mix, py=py, px=px)))
    for shape, cin_p, xf, mixf in cases:
        w = torch.randn(shape, generator=g)
        ref = P.pack_conv_weight(w if xf is None else xf(w), dtype, cin_p)
        if mode == "x3":
            ref = P.pack_split3(ref)
        got = ops.pack_conv_weight(w.to(DEV), dtype, cin_p, mix=None if mixf is None else mixf(tuple(shape[2:])), split3=mode == "x3")
        torch.cuda.synchronize()
        assert got.dtype == ref.dtype and got.shape == ref.shape, (shape, got.shape, ref.shape)
        assert torch.equal(got.cpu().view(bits), ref.view(bits)), shape
    # a PackedCache answers GPU parameters through the device packer and host parameters through the torch statements: same bits
    conv = torch.nn.Conv3d(8, 16, 3)
    pc_host, pc_dev = P.PackedCache(), P.PackedCache()
    wh, bh = pc_host.get(conv.weight, conv.bias, dtype, cin_stored=8)
    conv = conv.to(DEV)
    wd, bd = pc_dev.get(conv.weight, conv.bias, dtype, cin_stored=8)
    assert wd.is_cuda and torch.equal(wd.cpu().view(bits), wh.view(bits)) and torch.equal(bd.cpu(), bh)


# conv_in8_kernel (option conv_in8, default on): the encoder's conv_in -- CausalConv3d 3 -> 128, 3 x 3 x 3 on the 8 stored input
# channels, bf16 -- with register-stationary weights and fragments read from a halo patch in the LDS, through the SAME MFMA
# sequence and row arithmetic as the general path of the implicit-GEMM kernel: the same bits, with or without the consumer's LayerNorm, zero and
# replicate (v1.1) time padding, several tiles per workgroup.
IN8_CASES = [
    ("in8_t5_16x16", (1, 5, 16, 16), 3, 128, (3, 3, 3), ConvGeom(**G333), {}),
    ("in8_ln_keep", (2, 4, 16, 32), 3, 128, (3, 3, 3), ConvGeom(**G333), dict(ln="keep")),
    ("in8_ln_only_nobias", (1, 3, 32, 32), 3, 128, (3, 3, 3), ConvGeom(**G333), dict(ln="only", nobias=True)),
    ("in8_replicate_ln", (1, 5, 16, 16), 3, 128, (3, 3, 3), ConvGeom(**G333), dict(tmode="replicate", ln="keep")),
    ("in8_many_tiles", (2, 9, 64, 64), 3, 128, (3, 3, 3), ConvGeom(**G333), dict(ln="keep")),
    ("in8_one_frame", (1, 1, 16, 8), 3, 128, (3, 3, 3), ConvGeom(**G333), {}),
    ("in8_row_segments_256", (1, 3, 8, 256), 3, 128, (3, 3, 3), ConvGeom(**G333), dict(ln="keep")),        # Wo % 128 == 0: a tile = half a row
    ("in8_replicate_rows_64", (2, 3, 64, 64), 3, 128, (3, 3, 3), ConvGeom(**G333), dict(tmode="replicate", ln="only")),
]


@pytest.mark.parametrize("dt", H16, ids=H16_IDS)
@pytest.mark.parametrize("case", IN8_CASES, ids=[c[0] for c in IN8_CASES])
def test_conv_in8_kernel_equals_general_path(case, dt, vt_opts):
    outs = {}
    for on in (1, 0):
        vt_opts(conv_in8=on)
        keep = []
        plan = _check_conv(case, dt, keep_outputs=keep)
        assert plan["kernel"] == ("in8" if on else "igemm"), plan
        outs[on] = keep
    for a, b in zip(outs[1], outs[0]):
        assert torch.equal(a, b), f"{case[0]}: conv_in8_kernel differs from the general path"


def test_conv_in8_kernel_is_not_taken_elsewhere(vt_opts):
    """cache mode (a later chunk of a tiled v1.1 pass), fp32 / split-bf16, other channel counts, ragged pixel counts: the general path"""
    vt_opts(conv_in8=1)
    assert _check_conv(("in8_cache", (1, 4, 16, 16), 3, 128, (3, 3, 3), ConvGeom(**G333), dict(tmode="cache")), torch.bfloat16)["kernel"] == "igemm"
    assert _check_conv(("in8_f32", (1, 3, 16, 16), 3, 128, (3, 3, 3), ConvGeom(**G333), {}), torch.float32)["kernel"] == "igemm"
    assert _check_conv(("in8_x3", (1, 3, 16, 16), 3, 128, (3, 3, 3), ConvGeom(**G333), {}), X3)["kernel"] == "igemm"
    assert _check_conv(("in8_ragged", (1, 3, 10, 10), 3, 128, (3, 3, 3), ConvGeom(**G333), {}), torch.bfloat16)["kernel"] == "igemm"
    assert _check_conv(("in8_w24", (1, 2, 16, 24), 3, 128, (3, 3, 3), ConvGeom(**G333), {}), torch.bfloat16)["kernel"] == "igemm"      # 24 does not divide 128
    assert _check_conv(("in8_cout256", (1, 3, 16, 16), 3, 256, (3, 3, 3), ConvGeom(**G333), {}), torch.bfloat16)["kernel"] == "igemm"


@pytest.mark.parametrize("silu", [True, False], ids=["ln_silu", "ln"])
@pytest.mark.parametrize("shape", [(1, 3, 32, 32), (2, 2, 16, 32)], ids=["12_tiles", "4_tiles"])
def test_conv_interleaved_mix_with_layernorm(shape, silu, vt_opts):
    """option conv_tup_ln: the two parity launches of a v1.0 time up-sampler (k = 2 in time, alpha-mix against x, output frames
    interleaved) also emit the consumer's LayerNorm(+SiLU) from the 8-wave tile's bf16 LDS epilogue -- against the host statement
    (convolution, mix, LayerNorm of the unrounded rows); with the option off vt_conv_plan refuses and ops.conv runs without."""
    B, T, H, W = shape
    cin = cout = 256
    dtype = torch.bfloat16
    vt_opts(conv_tile=256)
    x = _act(B, T, H, W, cin, dtype, 1)
    g = torch.Generator().manual_seed(2)
    geom = ConvGeom(kt=2, kh=3, kw=3, pt=1, ph=1, pw=1, ph_hi=1, pw_hi=1)
    bias = _rand((cout,), torch.float32, 3, 0.1)
    mf = torch.tensor([0.2], device=DEV)
    gam, bet = _rand((cout,), torch.float32, 6, 0.5) + 1.0, _rand((cout,), torch.float32, 7, 0.2)
    y = torch.full((B, 2 * T, H, W, cout), float("nan"), dtype=dtype, device=DEV)
    n = torch.full_like(y, float("nan"))
    yr, nr = torch.zeros((B, 2 * T, H, W, cout), dtype=dtype), torch.zeros((B, 2 * T, H, W, cout), dtype=dtype)
    for par in (0, 1):
        wt = torch.randn((cout, cin, 2, 3, 3), generator=g) / math.sqrt(cin * 18)
        w = pack_conv_weight(wt, dtype, cin_stored=x.shape[-1]).to(DEV)
        kw = dict(res=x, res_mode=L.VT_RES_MIX, mix_factor=mf)
        r = ops.conv(x, w, bias, geom, cout=cout, out=y, out_t=(2, par), ln=(gam, bet, 1e-6, silu), ln_out=n, ln_optional=True, **kw)
        assert isinstance(r, tuple), "the 8-wave tile's epilogue takes alpha-mix + interleave + LayerNorm together"
        R.conv(x.cpu(), w.cpu(), bias.cpu(), geom, cout=cout, out=yr, out_t=(2, par), ln=(gam.cpu(), bet.cpu(), 1e-6, silu), ln_out=nr, **_cpu(kw))
    torch.cuda.synchronize()
    assert torch.isfinite(y.float()).all() and torch.isfinite(n.float()).all()
    assert rel_err(y, yr) < TOL[dtype] and rel_err(n, nr) < 2 * TOL[dtype], (rel_err(y, yr), rel_err(n, nr))
    vt_opts(conv_tup_ln=0)
    w = pack_conv_weight(torch.randn((cout, cin, 2, 3, 3), generator=g) / math.sqrt(cin * 18), dtype, cin_stored=x.shape[-1]).to(DEV)
    r = ops.conv(x, w, bias, geom, cout=cout, out=y, out_t=(2, 0), ln=(gam, bet, 1e-6, silu), ln_out=n, ln_optional=True, res=x, res_mode=L.VT_RES_MIX, mix_factor=mf)
    assert not isinstance(r, tuple)


@pytest.mark.parametrize("dt", list(H16) + [torch.float32], ids=list(H16_IDS) + ["f32"])
@pytest.mark.parametrize("cout,tile", [(128, 0), (256, 256)], ids=["lds128_epilogue", "lds256_epilogue"])
def test_conv_streaming_stores_same_bits(cout, tile, dt, vt_opts):
    """option conv_nt_mb: outputs at least that large leave the LDS epilogues as streaming (nt) stores -- a cache policy, not arithmetic:
    y and the fused LayerNorm are the bits of the plain stores (threshold 1 MiB vs off, on a tensor of 2 / 4 MiB)."""
    B, T, H, W = 1, 2, 64, 64
    cin = cout
    if tile:
        vt_opts(conv_tile=tile)
    x = _act(B, T, H, W, cin, dt, 1)
    res = _act(B, T, H, W, cout, dt, 4)
    g = torch.Generator().manual_seed(2)
    geom = ConvGeom(kh=3, kw=3, ph=1, pw=1, ph_hi=1, pw_hi=1)
    w = pack_conv_weight(torch.randn((cout, cin, 3, 3), generator=g) / math.sqrt(cin * 9), dt, cin_stored=cin).to(DEV)
    bias = _rand((cout,), torch.float32, 3, 0.1)
    ln = (_rand((cout,), torch.float32, 6, 0.5) + 1.0, _rand((cout,), torch.float32, 7, 0.2), 1e-6, True)
    outs = []
    for mb in (0, 1):
        vt_opts(conv_nt_mb=mb, conv_ws=0)
        y, n = ops.conv(x, w, bias, geom, cout=cout, res=res, res_mode=L.VT_RES_ADD, ln=ln, ln_keep_y=True)
        torch.cuda.synchronize()
        outs.append((y.clone(), n.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.isfinite(outs[1][1].float()).all()
