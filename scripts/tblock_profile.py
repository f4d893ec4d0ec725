"""Where does a step of the fused temporal block spend its cycles?  Runs vt_temporal_block_profile on the benchmark's
widest level (B=4, 20 frames, 256x256, C=128) and prints, per wave of workgroup 0 and per measured step, the shader-clock
ticks between the phase boundaries (s_memtime stamps kept in the LDS; see tblock_ws128.hip), as shipped and with parts
of the work switched off (option tblock_prof_mode).  NB: the stamped instantiation spills (the stamps cost scalar
registers), so its absolute numbers are high; the launch times at the end are the shipped kernel.  Kept under profiles/."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vidtok_amd import lib as L  # noqa: E402
from vidtok_amd import ops  # noqa: E402

def main():
    dev = "cuda:0"
    B, T, H, W, C_ = 4, 20, 256, 256, 128
    torch.manual_seed(0)
    x = torch.randn((B, T, H, W, C_), device=dev, dtype=torch.bfloat16)
    ws = [(torch.randn((C_, 3 * C_), device=dev) / math.sqrt(3 * C_)).to(torch.bfloat16) for _ in range(2)]
    bs = [torch.randn((C_,), device=dev) * 0.1 for _ in range(2)]
    norms = [(torch.ones(C_, device=dev), torch.zeros(C_, device=dev)) for _ in range(3)]
    names = {0: "as shipped", 1: "GEMMs skipped", 2: "row units skipped", 16: "no stores"}
    for mode in (0, 1, 2, 16):
        L.set_option("tblock_prof_mode", mode)
        stamps = torch.zeros((8, 4, 8), dtype=torch.int64, device=dev)
        for _ in range(2):
            ops.temporal_block(x, ws[0], bs[0], ws[1], bs[1], norms[0], norms[1], tmode=L.VT_TPAD_ZERO,
                               next_ln=(norms[2][0], norms[2][1], True), keep_y=True, profile_out=stamps)
        torch.cuda.synchronize()
        s = stamps.cpu()
        print(f"{names[mode]}:")
        n0 = ["A: G1(k)", "L2(k-1) unit 3", "barrier + requests + T1 <- acc", "O0, L1_0, O1 + stores", "L1_1, L1_2", "barrier"]
        n1 = ["A: T2 <- acc", "L2(k-1) units 0-2", "barrier + requests + G2(k-1)", "O2 + stores", "L1_3, O3 + stores", "barrier"]
        for w in range(8):
            for st in range(4):
                d = [int(s[w, st, i + 1] - s[w, st, i]) for i in range(6)]
                per = int(s[w, st + 1, 0] - s[w, st, 0]) if st < 3 else sum(d)
                print(f"  group {w // 4} wave {w} step {8 + st}: period {per:6d} | " + " | ".join(f"{nm} {v}" for nm, v in zip(n0 if w < 4 else n1, d)))
    L.set_option("tblock_prof_mode", 0)
    for tm in (L.VT_TPAD_ZERO, L.VT_TPAD_REPLICATE):
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        for _ in range(2):
            t0.record()
            for _ in range(5):
                ops.temporal_block(x, ws[0], bs[0], ws[1], bs[1], norms[0], norms[1], tmode=tm, next_ln=(norms[2][0], norms[2][1], True), keep_y=True)
            t1.record(); torch.cuda.synchronize()
        print(f"tmode={tm}: {t0.elapsed_time(t1) / 5:.3f} ms per launch")


if __name__ == "__main__":
    main()
