"""Where the split-bf16 mode ("bf16x3") sits between the fp32 kernels and the CPU oracle (GPU box):
    python scripts/x3_accuracy.py [config] [T] [S] [B]
prints, for the pre-quantisation encoder output h (what an FSQ code is rounded from), the reconstruction and the FSQ codes,
the distance of the fp32 kernels and of the split-bf16 kernels from the oracle, and how close h sits to a rounding boundary
where a code differs."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import build_model, build_oracle, rel_err  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "vidtok_v1_1/vidtok_fsq_causal_888_32768_v1_1"
    T, S, B = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((2, 17), (3, 64), (4, 1)))
    seed = int(os.environ.get("SEED", "21"))
    x = torch.rand((B, 3, T, S, S), generator=torch.Generator().manual_seed(9)) * 2 - 1
    outs = {}
    for mode in (torch.float32, "bf16x3", torch.bfloat16):
        model, cfg, sd = build_model(name, seed=seed, device="cuda", dtype=mode)
        torch.manual_seed(4)
        h = model.encoder(x.cuda())
        torch.manual_seed(4)
        z, dec, log = model(x.cuda())
        outs[str(mode)] = (h.cpu(), z.cpu(), dec.cpu(), log.get("indices", None))
    ora = build_oracle(cfg, sd)
    torch.set_num_threads(16)
    h0 = ora.pre_quant(x)
    torch.manual_seed(4)
    z0, dec0, log0 = ora(x)
    for mode, (h, z, dec, idx) in outs.items():
        msg = f"{mode:16s} h rel {rel_err(h, h0):.3e}  z rel {rel_err(z, z0):.3e}  recon rel {rel_err(dec, dec0):.3e}"
        if idx is not None:
            bad = idx.cpu() != log0["indices"]
            msg += f"  codes differing {int(bad.sum())} of {bad.numel()}"
        print(msg)


if __name__ == "__main__":
    main()
