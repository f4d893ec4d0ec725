"""Round-5 experiment: the bench batch (B=4) as ONE graph-replayed pass vs TWO half batches (B=2) on two streams (two engine
copies, each with its own graph cache), so that the small-M launches of one half (deep levels: 160 tiles on 256 CUs) run beside
the wide launches of the other.  python scripts/r5_streams.py"""
import copy
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import vidtok_amd  # noqa: E402
from bench import randomize_weights  # noqa: E402


def main():
    dev = "cuda:0"
    m = vidtok_amd.load_model_from_config(os.path.join(ROOT, "configs", "vidtok_kl_causal_488_4chn.yaml"), verbose=False)
    randomize_weights(m, 0)
    m = m.to(dev).eval().set_compute_dtype(torch.bfloat16)
    x = (torch.rand((4, 3, 17, 256, 256), generator=torch.Generator().manual_seed(1234)) * 2 - 1).to(dev)
    m.enable_graphs(True)

    def timed(fn, n=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    ms1 = timed(lambda: m(x))
    print(f"[streams] one pass B=4: {ms1:.3f} ms = {68 / ms1 * 1e3:.1f} frames/s", flush=True)
    for parts in (2, 4):
        models = [m] + [copy.deepcopy(m) for _ in range(parts - 1)]
        for mm in models:
            mm.enable_graphs(True)
        streams = [torch.cuda.Stream() for _ in range(parts)]
        xs = [c.contiguous() for c in x.chunk(parts)]

        def run():
            cur = torch.cuda.current_stream()
            outs = []
            for mm, st, xc in zip(models, streams, xs):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    outs.append(mm(xc))
            for st in streams:
                cur.wait_stream(st)
            return outs

        ms = timed(run)
        print(f"[streams] {parts} parts of B={4 // parts} on {parts} streams: {ms:.3f} ms = {68 / ms * 1e3:.1f} frames/s", flush=True)
        ref = m(x)[1]
        got = torch.cat([o[1] for o in run()], 0)
        torch.cuda.synchronize()
        print(f"[streams]   reconstruction equal to the one-pass result (host noise differs per call -> compare shapes only): {got.shape == ref.shape}")


if __name__ == "__main__":
    main()
