"""Where does an iteration of conv3x3_ws2_kernel spend its cycles?  Replays a benchmark-sized 3x3 128->128 convolution
(B=4, 20 frames, 256x256, + residual + LayerNorm+SiLU) through vt_conv_profile with option conv_ws = 2 and prints, per wave
of workgroup 0, the shader-clock ticks between the phase boundaries of iterations 8 and 9 (waves 0-3 = group 0: MFMA
phase then row slot; waves 4-7 = group 1: row slot + DMA requests then MFMA phase).  An iteration = one 4 x 16-pixel tile."""
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
    L.set_option("conv_ws", 2)
    B, T, H, W, C_ = 4, 20, 256, 256, 128
    torch.manual_seed(0)
    x = torch.randn((B, T, H, W, C_), device=dev, dtype=torch.bfloat16)
    res = torch.randn((B, T, H, W, C_), device=dev, dtype=torch.bfloat16)
    w = (torch.randn((C_, 9 * C_), device=dev) / math.sqrt(9 * C_)).to(torch.bfloat16)
    bias = torch.randn((C_,), device=dev)
    ln = (torch.ones(C_, device=dev), torch.zeros(C_, device=dev), 1e-6, True)
    for label, kw in (("plain", {}), ("+ residual + LayerNorm+SiLU, y kept", dict(res=res, res_mode=L.VT_RES_ADD, ln=ln, ln_keep_y=True))):
        ops.CONV_RECORD = []
        ops.conv(x, w, bias, ConvGeom(kh=3, kw=3, ph=1, pw=1, ph_hi=1, pw_hi=1), cout=C_, **kw)
        rec, ops.CONV_RECORD = ops.CONV_RECORD, None
        d = rec[0][0]
        for mode, mlabel in ((0, "as shipped"), (1, "row slots skipped"), (2, "LDS-DMA requests skipped"), (3, "row slots and requests skipped")):
            L.set_option("ws_prof_mode", mode)
            stamps = torch.zeros((8, 16), dtype=torch.int64, device=dev)
            for _ in range(2):
                L.check(L.load().vt_conv_profile(C.byref(d), stamps.data_ptr(), None), "vt_conv_profile")
            torch.cuda.synchronize()
            s = stamps.cpu()
            print(f"3x3 128->128 @256^2 {label}, {mlabel}: MFMA-bound time of an iteration = 144 MFMAs x 32 = 4 608 cycles per SIMD")
            for wv in range(8):
                for it in range(2):
                    t = [int(s[wv, 8 * it + k]) for k in range(8)]
                    if wv < 4:     # group 0 (split phase, round 6): stamps 0 [tail MFMAs] 6 [P writes] 1 [barrier] 2 [requests + rows + head's fragments] 5 [head MFMAs] 3 [barrier] 4
                        seg = [("tail MFMAs", t[6] - t[0]), ("partial sums -> LDS", t[1] - t[6]), ("barrier", t[2] - t[1]),
                               ("requests + rows [32,64) + wait + first fragments of the next tile", t[5] - t[2]), ("head MFMAs (beside group 1's phase)", t[3] - t[5]),
                               ("barrier", t[4] - t[3])]
                    else:          # group 1: 0 [requests + rows] 1 [barrier] 2 [partial sums + fragments] 5 [72 MFMAs] 6 [T writes] 7 [wait] 3 [barrier] 4
                        seg = [("requests + rows [0,32)", t[1] - t[0]), ("barrier", t[2] - t[1]), ("partial sums <- LDS + first fragments", t[5] - t[2]),
                               ("72 MFMAs", t[6] - t[5]), ("sums -> LDS", t[7] - t[6]), ("wait for my requests", t[3] - t[7]), ("barrier", t[4] - t[3])]
                    print(f"  wave {wv} iteration {8 + it}: total {t[4] - t[0]:6d} | " + " | ".join(f"{n} {v}" for n, v in seg))
        L.set_option("ws_prof_mode", 0)


if __name__ == "__main__":
    main()
