"""GPU micro-benchmark of the vt_conv kernel on the layer shapes of vidtok_kl_causal_488_4chn at
B=4, 17(20)x256x256.  `[VT_CONV_BUF=0|1] [VT_CONV_KWIN=0|1] python scripts/conv_microbench.py`
prints one line per layer class: ms, TFLOP/s.  Used for within-run A/B of kernel variants."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vidtok_amd import lib as L  # noqa: E402
from vidtok_amd import ops  # noqa: E402
from vidtok_amd.ops import ConvGeom  # noqa: E402

G3 = dict(kh=3, kw=3, ph=1, pw=1, ph_hi=1, pw_hi=1)
G333 = dict(kt=3, kh=3, kw=3, pt=2, ph=1, pw=1, ph_hi=1, pw_hi=1)
B = int(os.environ.get("MB_BATCH", "4"))
CASES = [  # name, (T,H,W), cin, cout, taps, geom, residual
    ("L0 spatial 3x3 128->128 @256^2", (20, 256, 256), 128, 128, 9, ConvGeom(**G3), True),
    ("L0 temporal k3 128->128 @256^2", (20, 256, 256), 128, 128, 3, ConvGeom(kt=3, pt=2), True),
    ("L1 spatial 3x3 256->256 @128^2", (20, 128, 128), 256, 256, 9, ConvGeom(**G3), True),
    ("L1 temporal k3 256->256 @128^2", (20, 128, 128), 256, 256, 3, ConvGeom(kt=3, pt=2), True),
    ("L2 spatial 3x3 512->512 @64^2 T10", (10, 64, 64), 512, 512, 9, ConvGeom(**G3), True),
    ("L3 spatial 3x3 512->512 @32^2 T5", (5, 32, 32), 512, 512, 9, ConvGeom(**G3), True),
    ("mid 3x3x3 512->512 @32^2 T5", (5, 32, 32), 512, 512, 27, ConvGeom(**G333), True),
    ("dec up_t.1 3x3x3 256->256 @256^2 (ups_t)", (10, 256, 256), 256, 256, 27, ConvGeom(ups_t=1, **G333), False),
    ("dec up.1 3x3 256->256 (ups_s) ->256^2 T10", (10, 128, 128), 256, 256, 9, ConvGeom(ups_s=1, **G3), False),
    ("dec up.0 3x3 256->128 @256^2", (20, 256, 256), 256, 128, 9, ConvGeom(**G3), False),
    ("nin 1x1 256->128 @256^2", (20, 256, 256), 256, 128, 1, ConvGeom(), False),
    ("conv_out 3x3x3 128->3 NCTHW", (20, 256, 256), 128, 3, 27, ConvGeom(**G333), False),
    ("conv_in 3x3x3 3->128", (20, 256, 256), 3, 128, 27, ConvGeom(**G333), False),
]


def main():
    dtype = torch.bfloat16 if os.environ.get("MB_DTYPE", "bf16") == "bf16" else torch.float32
    only = os.environ.get("MB_ONLY")
    print(f"buf={os.environ.get('VT_CONV_BUF', '1')} kwin={os.environ.get('VT_CONV_KWIN', '0')} dtype={dtype} B={B}")
    tot_ms = tot_fl = 0.0
    for name, (T, H, W), cin, cout, taps, geom, res in CASES:
        if only and only not in name:
            continue
        cp = ops.pad_channels(cin)
        torch.manual_seed(0)
        x = torch.randn((B, T, H, W, cp), device="cuda", dtype=dtype)
        if os.environ.get("MB_ZERO", "0") == "1":      # all-zero activations: same instructions, far fewer toggling bits (power / clock A/B)
            x.zero_()
        w = (torch.randn((cout, taps * cp), device="cuda") / math.sqrt(taps * cin)).to(dtype)
        bias = torch.randn((cout,), device="cuda")
        To, Ho, Wo = geom.out_dims(T, H, W)
        kw = {}
        if res:
            kw = dict(res=torch.randn((B, To, Ho, Wo, cout), device="cuda", dtype=dtype), res_mode=L.VT_RES_ADD)
        if "NCTHW" in name:
            kw = dict(out_layout=L.VT_NCTHW, t_trim=3)
        if os.environ.get("MB_LN", "0") == "1" and cout == 128 and "NCTHW" not in name:
            kw.update(ln=(torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda"), 1e-6, True), ln_keep_y=True)
        for _ in range(2):
            y = ops.conv(x, w, bias, geom, cout=cout, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        e0.record()
        for _ in range(n):
            y = ops.conv(x, w, bias, geom, cout=cout, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        fl = 2.0 * B * To * Ho * Wo * cout * taps * cin
        tot_ms += ms
        tot_fl += fl
        yf = (y[0] if isinstance(y, tuple) else y).float()
        print(f"  {name:46s} {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s   chk {yf.sum().item():+.6e} {yf.abs().sum().item():.6e}")
        del x, w, y, kw
    if tot_ms > 0:
        print(f"  total {tot_ms:.2f} ms, {tot_fl / tot_ms / 1e9:.1f} TFLOP/s aggregate")
    # fused temporal residual block of the widest level (vt_temporal_block) vs its two K=384 convolutions above
    if dtype == torch.bfloat16 and (not only or "tblock" in only or "L0" in only):
        T, H, W, Cc = 20, 256, 256, 128
        x = torch.randn((B, T, H, W, Cc), device="cuda", dtype=dtype)
        ws = [(torch.randn((Cc, 3 * Cc), device="cuda") / math.sqrt(3 * Cc)).to(dtype) for _ in range(2)]
        bs = [torch.randn((Cc,), device="cuda") for _ in range(2)]
        nm = (torch.ones(Cc, device="cuda"), torch.zeros(Cc, device="cuda"))
        for nxt in (None, (nm[0], nm[1], True)):
            if not ops.temporal_block_supported(x, L.VT_TPAD_ZERO):
                break
            for _ in range(2):
                ops.temporal_block(x, ws[0], bs[0], ws[1], bs[1], nm, nm, next_ln=nxt)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                out = ops.temporal_block(x, ws[0], bs[0], ws[1], bs[1], nm, nm, next_ln=nxt)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            fl = 2 * 2.0 * B * T * H * W * Cc * 3 * Cc
            nb = (2 if nxt is None else 3) * x.numel() * 2
            print(f"  tblock fused 128 @256^2 (next norm: {nxt is not None})        {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s  {nb / ms / 1e6:7.0f} GB/s")
        del x
    # LayerNorm+SiLU bandwidth on the two biggest activations
    for (T, H, W, C) in ((20, 256, 256, 128), (20, 128, 128, 256)):
        x = torch.randn((B, T, H, W, C), device="cuda", dtype=dtype)
        g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
        ops.layernorm_act(x, g, b, silu=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.layernorm_act(x, g, b, silu=True)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"  layernorm+silu {T}x{H}x{W}x{C}: {ms:.3f} ms  {2 * x.numel() * x.element_size() / ms / 1e6:.0f} GB/s")


if __name__ == "__main__":
    main()
