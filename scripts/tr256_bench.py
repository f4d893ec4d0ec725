"""conv_tr256.hip (loader-fed 128 x 256 tile: four matrix + four loader waves) against the 8-wave 256 x 256 tile of conv_igemm_kernel.h on the
Cout % 256 == 0 layers of the benchmark step (vidtok_kl_causal_488_4chn, B = 4, 17 x 256 x 256), same tensors, HIP-event time per launch:
    python scripts/tr256_bench.py [reps] [mode]"""
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
# label, (B, T, H, W), Cin, Cout, kernel dims, geometry, residual, LayerNorm ("" | "only" | "keep")
CASES = [
    ("temporal k3 256 (K=768) conv1: LN+SiLU only", (4, 20, 128, 128), 256, 256, (3,), ConvGeom(kt=3, pt=2), False, "only"),
    ("temporal k3 256 (K=768) conv2: +res, y, LN+SiLU", (4, 20, 128, 128), 256, 256, (3,), ConvGeom(kt=3, pt=2), True, "keep"),
    ("temporal k3 256 (K=768) @64^2 conv2", (4, 10, 128, 128), 256, 256, (3,), ConvGeom(kt=3, pt=2), True, "keep"),
    ("3x3 256 (K=2304) conv1: LN+SiLU only", (4, 20, 128, 128), 256, 256, (3, 3), ConvGeom(**G3), False, "only"),
    ("3x3 256 (K=2304) conv2: +res, y, LN+SiLU", (4, 20, 128, 128), 256, 256, (3, 3), ConvGeom(**G3), True, "keep"),
    ("3x3 256 (K=2304) plain", (4, 10, 128, 128), 256, 256, (3, 3), ConvGeom(**G3), False, ""),
    ("parity 2x3x3 256 (K=4608) mix + LN", (4, 10, 256, 256), 256, 256, (2, 3, 3), ConvGeom(kt=2, kh=3, kw=3, pt=1, ph=1, pw=1, ph_hi=1, pw_hi=1), "mix", "keep"),
    ("3x3x3 256 (K=6912) +res", (4, 20, 64, 64), 256, 256, (3, 3, 3), ConvGeom(**G333), True, ""),
    ("1x1 256->256 (K=1024: Cin 1024? no: 4 taps) skip", None, 0, 0, None, None, False, ""),
    ("3x3 512 (K=4608) +res @64^2", (4, 10, 64, 64), 512, 512, (3, 3), ConvGeom(**G3), True, ""),
    ("parity 2x3x3 512 (K=9216) mix", (4, 5, 128, 128), 512, 512, (2, 3, 3), ConvGeom(kt=2, kh=3, kw=3, pt=1, ph=1, pw=1, ph_hi=1, pw_hi=1), "mix", ""),
]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # 1: rows behind the K loop, 2: rows overlapped with the next tile's K loop
    dev = "cuda:0"
    dt = torch.bfloat16
    print(f"{'layer':52s} {'8-wave tile':>22s} {'conv_tr256 = ' + str(mode):>22s}   ratio   bit-equal")
    for label, shape, cin, cout, kd, geom, res, ln in CASES:
        if shape is None:
            continue
        B, T, H, W = shape
        torch.manual_seed(0)
        x = torch.randn((B, T, H, W, cin), device=dev, dtype=dt)
        taps = math.prod(kd)
        w = (torch.randn((cout, taps * cin), device=dev) / math.sqrt(taps * cin)).to(dt)
        bias = torch.randn((cout,), device=dev)
        To, Ho, Wo = geom.out_dims(T, H, W)
        kw = {}
        if res == "mix":
            kw.update(res=torch.randn((B, To, Ho, Wo, cout), device=dev, dtype=dt), res_mode=L.VT_RES_MIX, mix_factor=torch.tensor([0.3], device=dev))
        elif res:
            kw.update(res=torch.randn((B, To, Ho, Wo, cout), device=dev, dtype=dt), res_mode=L.VT_RES_ADD)
        if ln:
            kw.update(ln=(torch.rand((cout,), device=dev) + 0.5, torch.randn((cout,), device=dev) * 0.1, 1e-6, True), ln_keep_y=(ln == "keep"))
        M = B * To * Ho * Wo
        out, ms = {}, {}
        for tr in (0, 1):
            L.set_option("conv_tr256", mode if tr else 0)
            run = lambda: ops.conv(x, w, bias, geom, cout=cout, **kw)       # noqa: E731
            for _ in range(3):
                o = run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                o = run()
            e1.record()
            torch.cuda.synchronize()
            ms[tr] = e0.elapsed_time(e1) / reps
            out[tr] = o if isinstance(o, tuple) else (o,)
        L.set_option("conv_tr256", 0)
        same = all(torch.equal(a, b) for a, b in zip(out[0], out[1]))
        tf = lambda t: 2.0 * M * cout * taps * cin / t / 1e9     # noqa: E731
        print(f"{label:52s} {ms[0]:7.3f} ms {tf(ms[0]):7.1f} TF/s {ms[1]:7.3f} ms {tf(ms[1]):7.1f} TF/s   {ms[0] / ms[1]:5.2f}   {same}")
    L.load().vt_reset_options()


if __name__ == "__main__":
    main()
