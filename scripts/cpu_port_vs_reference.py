"""Build container only (needs /root/reference): wall time of the UNMODIFIED reference forward against the oracle
port (oracle/vidtok_oracle.py) on the same seeded model and clip, same thread count -- the evidence behind
bench.py's `cpu_baseline.kind = "port"` (the reference is Python and cannot travel to the GPU box).

    python scripts/cpu_port_vs_reference.py [--threads 8] [--res 64 128] > profiles/rNN_cpu_port_vs_reference.txt
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.refload import load_reference_model, randomize_weights  # noqa: E402
from oracle.vidtok_oracle import OracleEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=min(8, os.cpu_count() or 1))
    ap.add_argument("--res", type=int, nargs="+", default=[64, 128])
    ap.add_argument("--config", default="vidtok_kl_causal_488_4chn")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    ref, c = load_reference_model(args.config)
    randomize_weights(ref)
    ora = OracleEngine(c["model"]["params"], ref.state_dict())
    print(f"# {args.config}, fp32, 1 clip of 17 frames, {args.threads} threads of {os.cpu_count()}, torch {torch.__version__}")
    print("# res  reference_s  oracle_port_s  reference/port  max|dec_ref - dec_port| / max|dec_ref|")
    with torch.no_grad():
        ref(torch.rand(1, 3, 5, 32, 32))
        ora(torch.rand(1, 3, 5, 32, 32))
        for res in args.res:
            x = torch.rand(1, 3, 17, res, res, generator=torch.Generator().manual_seed(res)) * 2 - 1
            torch.manual_seed(3)
            t0 = time.perf_counter()
            _, dr, _ = ref(x)
            t_ref = time.perf_counter() - t0
            torch.manual_seed(3)
            t0 = time.perf_counter()
            _, do, _ = ora(x)
            t_ora = time.perf_counter() - t0
            err = ((dr - do).abs().max() / dr.abs().max()).item()
            print(f"{res:4d}  {t_ref:10.2f}  {t_ora:12.2f}  {t_ref / t_ora:13.3f}  {err:.2e}", flush=True)


if __name__ == "__main__":
    main()
