"""Round-5 A/B on one box, one process: the bench step (configs[1], B=4, bf16, graph replay) and the tiled configs[4] pass under option
sets (vt_set_option), alternating.  python scripts/r5_ab.py [reps]"""
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vidtok_amd  # noqa: E402
from vidtok_amd import lib as L  # noqa: E402
from bench import randomize_weights  # noqa: E402


def timed(fn, n):
    fn(); fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    dev = "cuda:0"
    sets = [("base(tskip=0,splitk=0)", dict(conv_tskip=0, conv_splitk=0)), ("tskip", dict(conv_tskip=1, conv_splitk=0)),
            ("tskip+splitk", dict(conv_tskip=1, conv_splitk=1))]
    m = vidtok_amd.load_model_from_config(os.path.join(ROOT, "configs", "vidtok_kl_causal_488_4chn.yaml"), verbose=False)
    randomize_weights(m, 0)
    m = m.to(dev).eval().set_compute_dtype(torch.bfloat16)
    x = (torch.rand((4, 3, 17, 256, 256), generator=torch.Generator().manual_seed(1234)) * 2 - 1).to(dev)
    m2 = vidtok_amd.load_model_from_config(os.path.join(ROOT, "configs", "vidtok_v1_1/vidtok_kl_causal_488_16chn_v1_1.yaml"), verbose=False)
    randomize_weights(m2, 0)
    m2 = m2.to(dev).eval().set_compute_dtype(torch.bfloat16)
    m2.use_tiling, m2.t_chunk_enc, m2.use_overlap = True, 16, True
    xl = (torch.rand((1, 3, 129, 256, 256), generator=torch.Generator().manual_seed(2)) * 2 - 1).to(dev)
    for r in range(reps):
        for name, opts in sets:
            for k, v in opts.items():
                L.set_option(k, v)
            m.enable_graphs(True)
            ms = timed(lambda: m(x), 10)
            m2.enable_graphs(True)
            ms2 = timed(lambda: m2(xl), 3)
            print(f"[ab] rep {r} {name:24s} configs[1] B=4: {ms:7.3f} ms/step = {4 * 17 / ms * 1e3:7.1f} frames/s | configs[4] tiled: {ms2:7.2f} ms = "
                  f"{129 / ms2 * 1e3:6.1f} frames/s", flush=True)
    L.load().vt_reset_options()


if __name__ == "__main__":
    main()
