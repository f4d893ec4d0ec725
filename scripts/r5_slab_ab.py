"""what does re-streaming the weight slab cost in TIME?  The M = 327 680 x N = 512 x K = 9 216 parity convolution of the time
up-sampler (11.5x its algorithmic traffic: every 256 x 256 tile streams its own 4.7 MB slab) and the K = 4 608 3x3 at 64^2 (7.3x), replayed
through vt_conv_profile with the weight pieces of every K step turned into descriptor zero fills (ws_prof_mode bit 1: no weight bytes move at
all -- an UPPER bound on what any slab-sharing scheme could win, and a generous one: all-zero weights also lower the matrix pipe's power draw)
and, for scale, with the activation pieces as zero fills (bit 0) and with both.  Zero operands also lower the matrix pipe's power draw
(+24 % on all-zero activations), so the decisive mode is bit 6: every K step re-reads the FIRST 128 bytes of its weight rows --
live non-zero data that stays in the L2 -- i.e. the launch with a weight slab that costs nothing beyond the L2.  python scripts/r5_slab_ab.py"""
import ctypes as C
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vidtok_amd import lib as L  # noqa: E402
from vidtok_amd import ops  # noqa: E402
from vidtok_amd.ops import ConvGeom  # noqa: E402


def main():
    dev = "cuda:0"
    cases = [("time up-sampler parity conv 2x3x3 512->512 @128^2, M = 327 680, K = 9 216", (4, 5, 128, 128), 512, 512,
              ConvGeom(kt=2, kh=3, kw=3, pt=1, ph=1, pw=1, ph_hi=1, pw_hi=1), 18),
             ("3x3 512->512 @64^2, M = 163 840, K = 4 608", (4, 10, 64, 64), 512, 512, ConvGeom(kh=3, kw=3, ph=1, pw=1, ph_hi=1, pw_hi=1), 9)]
    lib = L.load()
    L.set_option("conv_tskip", 0)
    for label, (B, T, H, W), cin, cout, geom, taps in cases:
        torch.manual_seed(0)
        x = torch.randn((B, T, H, W, cin), device=dev, dtype=torch.bfloat16)
        w = (torch.randn((cout, taps * cin), device=dev) / math.sqrt(taps * cin)).to(torch.bfloat16)
        bias = torch.randn((cout,), device=dev)
        ops.CONV_RECORD = []
        ops.conv(x, w, bias, geom, cout=cout)
        rec, ops.CONV_RECORD = ops.CONV_RECORD, None
        d = rec[0][0]
        plan = ops.conv_plan(d)
        stamps = torch.zeros((8, 4, 8), dtype=torch.int64, device=dev)
        res = {}
        for rep in range(2):
            for pm, name in ((0, "as shipped"), (64, "weights: every K step reads the rows' first 128 B (live data, L2-resident)"), (2, "weight pieces = zero fills"),
                             (1, "activation pieces = zero fills"), (3, "no memory traffic in the K loop")):
                L.set_option("ws_prof_mode", pm)
                for _ in range(3):
                    L.check(lib.vt_conv_profile(C.byref(d), stamps.data_ptr(), None), "vt_conv_profile")
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    L.check(lib.vt_conv_profile(C.byref(d), stamps.data_ptr(), None), "vt_conv_profile")
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(name, []).append(e0.elapsed_time(e1) / 10)
        L.set_option("ws_prof_mode", 0)
        base = min(res["as shipped"])
        print(f"{label}: tile {plan['tile']}, {plan['workgroups']} workgroups")
        for name, v in res.items():
            print(f"    {name:78s} {min(v):7.3f} ms (runs {', '.join(f'{t:.3f}' for t in v)})  {100 * (min(v) / base - 1):+5.1f} %")
    lib.vt_reset_options()


if __name__ == "__main__":
    main()
