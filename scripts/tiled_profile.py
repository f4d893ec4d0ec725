"""Where does a temporally tiled v1.1 pass (BASELINE.json configs[4]: 1 clip 129x256x256, chunks of 16 frames, decoder
look-ahead) spend its time?  Runs one warm pass under torch's profiler-free event timing per (pixels, Cout, K) launch group,
the same grouping as bench.py --breakdown.  Output kept under profiles/."""
import os
import sys
from collections import OrderedDict

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import build_model  # noqa: E402
from vidtok_amd import ops  # noqa: E402


def main():
    dev = "cuda:0"
    g = torch.Generator().manual_seed(0)
    xl = (torch.rand((1, 3, 129, 256, 256), generator=g) * 2 - 1).to(dev)
    m, _, _ = build_model("vidtok_v1_1/vidtok_kl_causal_488_16chn_v1_1", seed=22, device=dev, dtype=torch.bfloat16)
    m.use_tiling, m.t_chunk_enc, m.use_overlap = True, 16, True
    m(xl)
    ops.CONV_RECORD = []
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); m(xl); t1.record(); torch.cuda.synchronize()
    rec, ops.CONV_RECORD = ops.CONV_RECORD, None
    total = t0.elapsed_time(t1)
    groups = OrderedDict()
    for item in rec:
        groups.setdefault(item[2], []).append(item)
    rows = []
    for label, items in groups.items():
        for _ in range(2):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); ops.replay_convs(items); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        M, N, K = label
        rows.append((ms, M, N, K, len(items), 2.0 * M * N * K * len(items) / ms / 1e9))
    rows.sort(reverse=True)
    mf = sum(r[0] for r in rows)
    print(f"tiled 129x256x256 pass: {total:.1f} ms ({129 / total * 1e3:.1f} frames/s), {len(rec)} MFMA-kernel launches, {mf:.1f} ms when replayed alone by group")
    for ms, M, N, K, n, tf in rows[:28]:
        print(f"  M={M:8d} N={N:4d} K={K:6d} x{n:4d} {ms:8.3f} ms {tf:8.1f} TFLOP/s {100 * ms / mf:5.1f}%")


if __name__ == "__main__":
    main()
