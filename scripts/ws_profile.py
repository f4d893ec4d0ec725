"""Where does a tile of conv3x3_ws128_kernel spend its cycles?  vt_conv_profile on the widest level's 3x3 convolution
(B=4, 20 frames, 256x256, 128 -> 128), plain and with residual + LayerNorm+SiLU emission: shader-clock stamps of workgroup
0's fourth tile, per wave: barrier A | 72 / 72 / 80 MFMAs with row-phase pieces | 64 bare MFMAs | vmcnt drain | barrier B
| accumulators -> T.  MFMA-bound time of a tile: 288 x 32 = 9 216 cycles."""
import ctypes as C
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vidtok_amd import lib as L  # noqa: E402
from vidtok_amd import ops  # noqa: E402
from vidtok_amd.ops import ConvGeom  # noqa: E402

NAMES = ["barrier A", "(set-up, residual rows)", "MFMA 0-71 + rows", "MFMA 72-143 + rows", "MFMA 144-223 + rows", "MFMA 224-287 bare",
         "vmcnt drain", "barrier B", "acc -> T"]


def main():
    dev = "cuda:0"
    B, T, H, W, C_ = 4, 20, 256, 256, 128
    torch.manual_seed(0)
    x = torch.randn((B, T, H, W, C_), device=dev, dtype=torch.bfloat16)
    w = (torch.randn((C_, 9 * C_), device=dev) / math.sqrt(9 * C_)).to(torch.bfloat16)
    bias = torch.randn((C_,), device=dev)
    res = torch.randn((B, T, H, W, C_), device=dev, dtype=torch.bfloat16)
    geom = ConvGeom(kh=3, kw=3, ph=1, pw=1, ph_hi=1, pw_hi=1)
    ln = (torch.ones(C_, device=dev), torch.zeros(C_, device=dev), 1e-6, True)
    for label, kw in [("plain", {}), ("+ residual, LayerNorm+SiLU emitted, y kept", dict(res=res, res_mode=L.VT_RES_ADD, ln=ln, ln_keep_y=True))]:
        ops.CONV_RECORD = []
        ops.conv(x, w, bias, geom, cout=C_, **kw)
        rec, ops.CONV_RECORD = ops.CONV_RECORD, None
        d = rec[0][0]
        assert ops.conv_plan(d)["kernel"] == "ws128"
        stamps = torch.zeros((4, 16), dtype=torch.int64, device=dev)
        lib = L.load()
        for _ in range(2):
            L.check(lib.vt_conv_profile(C.byref(d), stamps.data_ptr(), None), "vt_conv_profile")
        torch.cuda.synchronize()
        s = stamps.cpu()
        print(f"conv3x3_ws128_kernel, {label}:")
        for wv in range(4):
            dl = [int(s[wv, k + 1] - s[wv, k]) for k in range(9)]
            print(f"  wave {wv}: tile {int(s[wv, 9] - s[wv, 0]):6d} cycles | " + " | ".join(f"{n} {v}" for n, v in zip(NAMES, dl)))


if __name__ == "__main__":
    main()
