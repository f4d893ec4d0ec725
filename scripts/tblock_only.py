"""Ten launches of the fused temporal block at the benchmark's shape (for counter passes: rocprofv3 --pmc ... -- python scripts/tblock_only.py)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vidtok_amd import lib as L  # noqa: E402
from vidtok_amd import ops  # noqa: E402

dev = "cuda:0"
B, T, H, W, C_ = 4, 20, 256, 256, 128
torch.manual_seed(0)
x = torch.randn((B, T, H, W, C_), device=dev, dtype=torch.bfloat16)
ws = [(torch.randn((C_, 3 * C_), device=dev) / math.sqrt(3 * C_)).to(torch.bfloat16) for _ in range(2)]
bs = [torch.randn((C_,), device=dev) * 0.1 for _ in range(2)]
norms = [(torch.ones(C_, device=dev), torch.zeros(C_, device=dev)) for _ in range(3)]
for _ in range(10):
    ops.temporal_block(x, ws[0], bs[0], ws[1], bs[1], norms[0], norms[1], tmode=L.VT_TPAD_ZERO, next_ln=(norms[2][0], norms[2][1], True), keep_y=True)
torch.cuda.synchronize()
