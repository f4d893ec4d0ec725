"""Where does a step of tblock_ws128_kernel spend its cycles?  Runs vt_temporal_block_profile on the benchmark's widest
level (B=4, 20 frames, 256x256, C=128) and prints, per wave of workgroup 0 and per measured step, the shader-clock ticks
between the phase boundaries (s_memtime stamps kept in the LDS; see tblock_ws128.hip).  Output kept under profiles/."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vidtok_amd import lib as L  # noqa: E402
from vidtok_amd import ops  # noqa: E402

NAMES = ["A: GEMM1 + OUT job", "barrier (T free)", "acc -> T + barrier", "B: LN2 -> ring2", "wait x(v+1) [vmcnt]", "barrier (ring2)",
         "C: GEMM2 + LN1 job", "acc -> T", "barrier (T done)"]


def main():
    dev = "cuda:0"
    B, T, H, W, C_ = 4, 20, 256, 256, 128
    torch.manual_seed(0)
    x = torch.randn((B, T, H, W, C_), device=dev, dtype=torch.bfloat16)
    ws = [(torch.randn((C_, 3 * C_), device=dev) / math.sqrt(3 * C_)).to(torch.bfloat16) for _ in range(2)]
    bs = [torch.randn((C_,), device=dev) * 0.1 for _ in range(2)]
    norms = [(torch.ones(C_, device=dev), torch.zeros(C_, device=dev)) for _ in range(3)]
    v3 = os.environ.get("VT_TBLOCK_V3", "1") != "0"
    stamps = torch.zeros((8, 4, 8) if v3 else (4, 4, 16), dtype=torch.int64, device=dev)
    for _ in range(2):
        ops.temporal_block(x, ws[0], bs[0], ws[1], bs[1], norms[0], norms[1], tmode=L.VT_TPAD_ZERO,
                           next_ln=(norms[2][0], norms[2][1], True), keep_y=True, profile_out=stamps)
    torch.cuda.synchronize()
    s = stamps.cpu()
    if v3:
        mn = ["A: G1(k)", "barrier", "T1 <- acc", "B: G2(k-1)", "barrier", "T2 <- acc"]
        vn = ["A: L2(k-1) rows", "barrier", "B: O(k-2) rows + stores", "B: L1(k+1) rows", "barrier", "-"]
        for w in range(8):
            names = mn if w < 4 else vn
            for st in range(4):
                d = [int(s[w, st, i + 1] - s[w, st, i]) for i in range(6)]
                per = int(s[w, st + 1, 0] - s[w, st, 0]) if st < 3 else sum(d)
                print(f"  {'matrix' if w < 4 else 'row   '} wave {w} step {8 + st}: period {per:6d} | " + " | ".join(f"{nm} {v}" for nm, v in zip(names, d)))
        return
    for w in range(4):
        print(f"wave {w}:")
        for st in range(4):
            d = [int(s[w, st, k + 1] - s[w, st, k]) for k in range(9)]
            nxt = int(s[w, st + 1, 0] - s[w, st, 0]) if st < 3 else sum(d)
            print(f"  step {8 + st}: total {nxt:6d} ticks | " + " | ".join(f"{n.split(':')[0]} {v}" for n, v in zip(NAMES, d)))
    avg = [sum(int(s[w, st, k + 1] - s[w, st, k]) for w in range(4) for st in range(4)) / 16 for k in range(9)]
    tot = sum(avg)
    print("average over waves and steps (s_memtime ticks; 100 MHz constant clock on gfx9 -- ratios are what matters):")
    for n, v in zip(NAMES, avg):
        print(f"  {n:28s} {v:9.1f}  {100 * v / tot:5.1f} %")


if __name__ == "__main__":
    main()
