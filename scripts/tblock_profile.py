"""Where does a step of tblock_split_kernel spend its cycles?  Runs vt_temporal_block_profile on the benchmark's widest
level (B=4, 20 frames, 256x256, C=128) and prints, per wave of workgroup 0 and per measured step, the shader-clock ticks
between the phase boundaries (s_memtime stamps kept in the LDS; see tblock_ws128.hip).  Output kept under profiles/."""
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
    stamps = torch.zeros((8, 4, 8), dtype=torch.int64, device=dev)
    for _ in range(2):
        ops.temporal_block(x, ws[0], bs[0], ws[1], bs[1], norms[0], norms[1], tmode=L.VT_TPAD_ZERO,
                           next_ln=(norms[2][0], norms[2][1], True), keep_y=True, profile_out=stamps)
    torch.cuda.synchronize()
    s = stamps.cpu()
    mn = ["A: G1(k)", "barrier", "T1 <- acc", "B: G2(k-1)", "barrier", "T2 <- acc"]
    vn = ["A: L2(k-1) rows", "barrier", "B: O(k-2) rows + stores", "B: L1(k+1) rows", "barrier", "-"]
    for w in range(8):
        names = mn if w < 4 else vn
        for st in range(4):
            d = [int(s[w, st, i + 1] - s[w, st, i]) for i in range(6)]
            per = int(s[w, st + 1, 0] - s[w, st, 0]) if st < 3 else sum(d)
            print(f"  {'matrix' if w < 4 else 'row   '} wave {w} step {8 + st}: period {per:6d} | " + " | ".join(f"{nm} {v}" for nm, v in zip(names, d)))


if __name__ == "__main__":
    main()
