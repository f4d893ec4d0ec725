"""Round-5 A/B (one box, one process, alternating): the bench step under conv_in8 = 0 / 1.  python scripts/r5_ab2.py [reps]"""
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import vidtok_amd  # noqa: E402
from vidtok_amd import lib as L  # noqa: E402
from vidtok_amd import ops  # noqa: E402
from bench import randomize_weights  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = "cuda:0"
    m = vidtok_amd.load_model_from_config(os.path.join(ROOT, "configs", "vidtok_kl_causal_488_4chn.yaml"), verbose=False)
    randomize_weights(m, 0)
    m = m.to(dev).eval().set_compute_dtype(torch.bfloat16)
    x = (torch.rand((4, 3, 17, 256, 256), generator=torch.Generator().manual_seed(1234)) * 2 - 1).to(dev)
    # the conv_in launch alone
    ops.CONV_RECORD = []
    m(x)
    rec, ops.CONV_RECORD = ops.CONV_RECORD, None
    cin = [r for r in rec if not isinstance(r[0], (tuple, L.TBlockDesc)) and r[0].Cin == 8 and r[0].Cout == 128]
    for r in range(reps):
        for on in (0, 1):
            L.set_option("conv_in8", on)
            m.enable_graphs(True)
            for _ in range(3):
                m(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                m(x)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 10 * 1e3
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ops.replay_convs(cin)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                ops.replay_convs(cin)
            e1.record()
            torch.cuda.synchronize()
            print(f"[ab2] rep {r} conv_in8={on}: step {ms:7.3f} ms = {68 / ms * 1e3:7.1f} frames/s | conv_in launch alone ({ops.conv_plan(cin[0][0])['kernel']}): "
                  f"{e0.elapsed_time(e1) / 10:.3f} ms", flush=True)
    L.load().vt_reset_options()


if __name__ == "__main__":
    main()
