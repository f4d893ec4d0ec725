"""is the 2-11x fabric traffic of the activation gather costing TIME, or is the power envelope the ceiling?  Three arms on
the K = 9 216 parity convolution of the widest time up-sampler (10.9x its algorithmic traffic), on a 256-channel 3 x 3 layer of the benchmark
step (2.4x) and on the K = 768 temporal convolution at 256 channels (1.05x), replayed through vt_conv_profile:
  (i)   activations gathered from an L2-resident LIVE patch (ws_prof_mode bit 7: the first 128 KiB of x, non-zero data -- the matrix
        pipe's power draw stays what it is; wrong results),
  (ii)  as shipped,
  (iii) activation pieces = descriptor zero fills (bit 0: no bytes move, and all-zero operands lower the matrix pipe's power draw).
Run under `rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE` (scripts/collect_r06.sh) the same process also yields the shader clock of every
arm (GRBM_GUI_ACTIVE / duration): if (i) ~ (ii) in time AND clock, the gather's traffic costs nothing and the ceiling is power; if (i) << (ii)
the traffic costs time.  python scripts/activation_restream_ab.py [reps]"""
import ctypes as C
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vidtok_amd import lib as L  # noqa: E402
from vidtok_amd import ops  # noqa: E402
from vidtok_amd.ops import ConvGeom  # noqa: E402

ARMS = ((128, "(i)   activations from an L2-resident live patch"), (0, "(ii)  as shipped"), (1, "(iii) activation pieces = zero fills"),
        (64, "      weights from their rows' first 128 B (L2-resident, live)"), (192, "      both operands L2-resident, live"))


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = "cuda:0"
    cases = [("time up-sampler parity conv 2x3x3 512->512 @128^2, M = 327 680, K = 9 216", (4, 5, 128, 128), 512, 512,
              ConvGeom(kt=2, kh=3, kw=3, pt=1, ph=1, pw=1, ph_hi=1, pw_hi=1), 18),
             ("3x3 256->256 @128^2, M = 1 310 720, K = 2 304", (4, 20, 128, 128), 256, 256, ConvGeom(kh=3, kw=3, ph=1, pw=1, ph_hi=1, pw_hi=1), 9),
             ("3x3x3 256->256 @128^2, M = 1 310 720, K = 6 912", (4, 20, 128, 128), 256, 256,
              ConvGeom(kt=3, kh=3, kw=3, pt=2, ph=1, pw=1, ph_hi=1, pw_hi=1), 27)]
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
            for pm, name in ARMS:
                L.set_option("ws_prof_mode", pm)
                for _ in range(3):
                    L.check(lib.vt_conv_profile(C.byref(d), stamps.data_ptr(), None), "vt_conv_profile")
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    L.check(lib.vt_conv_profile(C.byref(d), stamps.data_ptr(), None), "vt_conv_profile")
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(name, []).append(e0.elapsed_time(e1) / reps)
        L.set_option("ws_prof_mode", 0)
        base = min(res[ARMS[1][1]])
        M = B * T * H * W
        print(f"{label}: tile {plan['tile']}, {plan['workgroups']} workgroups; launches per arm and repetition: 3 warm-up + {reps} timed, arms in the order below, twice")
        for name, v in res.items():
            print(f"    {name:66s} {min(v):7.3f} ms (runs {', '.join(f'{t:.3f}' for t in v)})  {100 * (min(v) / base - 1):+5.1f} %   "
                  f"{2.0 * M * cout * taps * cin / min(v) / 1e9:7.1f} TFLOP/s")
    lib.vt_reset_options()


if __name__ == "__main__":
    main()
