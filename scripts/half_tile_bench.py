"""A/B of the 128 x 256 half-tile form (options conv_half256, conv_half_stagger) against the 8-wave tile on the short-K layers of
the 256-channel level of vidtok_kl_causal_488_4chn at B = 4 (GPU):  python scripts/half_tile_bench.py"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vidtok_amd import lib as L  # noqa: E402
from vidtok_amd import ops  # noqa: E402
from vidtok_amd.ops import ConvGeom  # noqa: E402

G3 = dict(kh=3, kw=3, ph=1, pw=1, ph_hi=1, pw_hi=1)
CASES = [  # name, (B, T, H, W), channels, taps, geom, residual, LayerNorm ("only" | "keep" | None)
    ("temporal k3 256 @128^2 T20, LN only", (4, 20, 128, 128), 256, 3, ConvGeom(kt=3, pt=2), False, "only"),
    ("temporal k3 256 @128^2 T20, +res, LN keep", (4, 20, 128, 128), 256, 3, ConvGeom(kt=3, pt=2), True, "keep"),
    ("temporal k3 256 @128^2 T10, +res", (4, 10, 128, 128), 256, 3, ConvGeom(kt=3, pt=2), True, None),
    ("spatial 3x3 256 @128^2 T20, +res, LN keep", (4, 20, 128, 128), 256, 9, ConvGeom(**G3), True, "keep"),
    ("temporal k3 512 @64^2 T10, +res", (4, 10, 64, 64), 512, 3, ConvGeom(kt=3, pt=2), True, None),
]
SETTINGS = [(0, 0, 2), (1 << 20, 0, 0), (1 << 20, 900, 0), (1 << 20, 0, 2), (1 << 20, 600, 2), (1 << 20, 900, 2), (1 << 20, 1500, 2), (1 << 20, 2500, 2)]      # conv_half256, conv_half_stagger, conv_sched


def main():
    for name, (B, T, H, W), c, taps, geom, res, ln in CASES:
        torch.manual_seed(0)
        x = torch.randn((B, T, H, W, c), device="cuda", dtype=torch.bfloat16)
        w = (torch.randn((c, taps * c), device="cuda") / math.sqrt(taps * c)).to(torch.bfloat16)
        bias = torch.randn((c,), device="cuda")
        kw = {}
        if res:
            kw.update(res=torch.randn((B, T, H, W, c), device="cuda", dtype=torch.bfloat16), res_mode=L.VT_RES_ADD)
        if ln:
            kw.update(ln=(torch.ones(c, device="cuda"), torch.zeros(c, device="cuda"), 1e-6, True), ln_keep_y=(ln == "keep"))
        print(name)
        ref = None
        for half, stagger, sched in SETTINGS:
            with L.options(conv_half256=half, conv_half_stagger=stagger, conv_half_plain=1, conv_sched=sched):
                for _ in range(3):
                    y = ops.conv(x, w, bias, geom, cout=c, **kw)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = 10
                e0.record()
                for _ in range(n):
                    y = ops.conv(x, w, bias, geom, cout=c, **kw)
                e1.record()
                torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            ys = y if isinstance(y, tuple) else (y,)
            if ref is None:
                ref = ys
            same = all(torch.equal(a, b) for a, b in zip(ys, ref))
            fl = 2.0 * B * T * H * W * c * taps * c
            print(f"   half256={half:8d} stagger={stagger:4d} sched={sched}  {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s   equal to the 8-wave tile: {same}")


if __name__ == "__main__":
    main()
