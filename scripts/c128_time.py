"""Launch times of the widest level's two weight-stationary kernels at the benchmark's shapes (B = 4, 20 frames, 256 x 256, C = 128):
the fused temporal block (vt_temporal_block: next norm LayerNorm+SiLU, y kept -- the shape four of the step's five launches have) and the
3 x 3 convolution (vt_conv -> conv3x3_ws2_kernel: plain; + residual + LayerNorm+SiLU with y kept; LayerNorm+SiLU only).  HIP-event time
per launch, best of three repetitions of ten.  For A/B runs of library variants: VIDTOK_AMD_LIB=ab_libs/libvidtok_amd_<name>.so."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vidtok_amd import lib as L  # noqa: E402
from vidtok_amd import ops  # noqa: E402
from vidtok_amd.ops import ConvGeom  # noqa: E402


def timed(fn, n=10, reps=3):
    best = 1e9
    for _ in range(2):
        fn()
    for _ in range(reps):
        t0 = torch.cuda.Event(enable_timing=True)
        t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(n):
            fn()
        t1.record()
        torch.cuda.synchronize()
        best = min(best, t0.elapsed_time(t1) / n)
    return best


def main():
    dev = "cuda:0"
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
    B, T, H, W, C_ = 4, 20, 256, 256, 128
    torch.manual_seed(0)
    x = torch.randn((B, T, H, W, C_), device=dev, dtype=dt)
    res = torch.randn((B, T, H, W, C_), device=dev, dtype=dt)
    ws = [(torch.randn((C_, 3 * C_), device=dev) / math.sqrt(3 * C_)).to(dt) for _ in range(2)]
    bs = [torch.randn((C_,), device=dev) * 0.1 for _ in range(2)]
    norms = [(torch.ones(C_, device=dev), torch.zeros(C_, device=dev)) for _ in range(3)]
    out = []
    for tm, nm in ((L.VT_TPAD_ZERO, "zero"), (L.VT_TPAD_REPLICATE, "replicate")):
        ms = timed(lambda: ops.temporal_block(x, ws[0], bs[0], ws[1], bs[1], norms[0], norms[1], tmode=tm, next_ln=(norms[2][0], norms[2][1], True), keep_y=True))
        fl = 2.0 * B * T * H * W * C_ * 3 * C_ * 2
        out.append(f"tblock_pair ({nm} padding, y + LN+SiLU)  {ms:.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s")
    w3 = (torch.randn((C_, 9 * C_), device=dev) / math.sqrt(9 * C_)).to(dt)
    bias = torch.randn((C_,), device=dev)
    ln = (torch.ones(C_, device=dev), torch.zeros(C_, device=dev), 1e-6, True)
    g = ConvGeom(kh=3, kw=3, ph=1, pw=1, ph_hi=1, pw_hi=1)
    for label, kw in (("plain", {}), ("+res, y, LN+SiLU", dict(res=res, res_mode=L.VT_RES_ADD, ln=ln, ln_keep_y=True)), ("LN+SiLU only", dict(ln=ln, ln_keep_y=False))):
        ms = timed(lambda: ops.conv(x, w3, bias, g, cout=C_, **kw))
        fl = 2.0 * B * T * H * W * C_ * 9 * C_
        out.append(f"conv3x3_ws2 ({label})  {ms:.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s")
    print(os.environ.get("VIDTOK_AMD_LIB", "shipped library").split("/")[-1] + ": " + " | ".join(out))


if __name__ == "__main__":
    main()
