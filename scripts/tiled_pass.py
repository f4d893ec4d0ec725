"""One warm-up and N timed temporally tiled v1.1 passes (BASELINE.json configs[4]: 1 clip of 129x256x256, 16-frame chunks,
decoder look-ahead), for `rocprofv3 --kernel-trace --stats -- python scripts/tiled_pass.py` (kernel table of the tiled path)."""
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import build_model  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = "cuda:0"
    xl = (torch.rand((1, 3, 129, 256, 256), generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
    m, _, _ = build_model("vidtok_v1_1/vidtok_kl_causal_488_16chn_v1_1", seed=22, device=dev, dtype=torch.bfloat16)
    m.use_tiling, m.t_chunk_enc, m.use_overlap = True, 16, True
    m(xl)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        m(xl)
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / n
    print(f"tiled 129x256x256 pass, eager launches: {ms:.1f} ms ({129 / ms * 1e3:.1f} frames/s)")


if __name__ == "__main__":
    main()
