"""Round-5 A/B (one box, one process, alternating): the bench step under conv_tup_ln = 0 / 1 (the consumer's LayerNorm behind the v1.0 time
up-samplers emitted by their parity launches instead of a separate pass).  python scripts/r5_ab3.py [reps]"""
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
    m.regularization.sample = False
    x = (torch.rand((4, 3, 17, 256, 256), generator=torch.Generator().manual_seed(1234)) * 2 - 1).to(dev)
    outs = {}
    for r in range(reps):
        for on in (0, 1):
            L.set_option("conv_tup_ln", on)
            m.enable_graphs(False)
            ops.LN_RECORD = []
            outs[on] = m(x)[1]
            nln, ops.LN_RECORD = len(ops.LN_RECORD), None
            m.enable_graphs(True)
            for _ in range(3):
                m(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                m(x)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 10 * 1e3
            print(f"[ab3] rep {r} conv_tup_ln={on}: step {ms:7.3f} ms = {68 / ms * 1e3:7.1f} frames/s, standalone LayerNorm passes per step: {nln}", flush=True)
    d = (outs[1].float() - outs[0].float()).abs().max().item() / outs[0].float().abs().max().item()
    print(f"[ab3] reconstruction with vs without: max rel diff {d:.3e} (the fused norm reads the fp32 rows, the separate pass the stored bf16 rows)")
    L.load().vt_reset_options()


if __name__ == "__main__":
    main()
