"""How many of a bf16 pass's FSQ code flips come from the small deep layers of the encoder?  (VERDICT r2, weak #1.)
vidtok_fsq_causal_488_32768, B=4 clips of 17x256x256 (BASELINE.json configs[2]), seeded weights as in smoke(): integer codes
of the bf16 kernels with the encoder levels from `tail_level` on (+ mid + conv_out) in fp32, against the codes of the fp32
kernels (which are the CPU oracle's, 5 120 of 5 120 -- smoke()), and what the encoder pass costs in each mode."""
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import build_model  # noqa: E402


def main():
    name = "vidtok_fsq_causal_488_32768"
    x = (torch.rand(4, 3, 17, 256, 256, generator=torch.Generator().manual_seed(5)) * 2 - 1).to("cuda:0")
    model, cfg, sd = build_model(name, seed=7, device="cuda:0", dtype=torch.float32)
    model.regularization.compute_aux_loss = False

    def run(dtype, tail, level):
        model.set_compute_dtype(dtype, encoder_tail=tail, tail_level=level)
        for _ in range(2):
            z, log = model.encode(x, return_reg_log=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            z, log = model.encode(x, return_reg_log=True)
        e1.record()
        torch.cuda.synchronize()
        return log["indices"].clone(), z.clone(), e0.elapsed_time(e1) / 3

    ref, zref, ms32 = run(torch.float32, None, None)
    n = ref.numel()
    print(f"{name} B=4 17x256x256, {n} codes; encoder in fp32: {ms32:.2f} ms")
    L = model.encoder.num_resolutions
    for label, tail, level in [("bf16 throughout", None, None), (f"mid + conv_out in fp32", torch.float32, L),
                               ("level 3 (32^2, T=5) on in fp32", torch.float32, 3), ("level 2 (64^2, T=10) on in fp32", torch.float32, 2),
                               ("level 1 (128^2, T=20) on in fp32", torch.float32, 1), ("level 0 on in fp32 (conv_in alone in bf16)", torch.float32, 0)]:
        idx, z, ms = run(torch.bfloat16, tail, level)
        same = int((idx == ref).sum())
        rel = float((z - zref).norm() / zref.norm())
        print(f"  {label:46s} codes {same}/{n} = {same / n:.4f}   z rel {rel:.2e}   encoder {ms:7.2f} ms", flush=True)
    # the other direction: which part of the error is the wide levels'?  fp32 wide levels, bf16 tail
    for label, level in [("fp32 levels 0-2, bf16 from level 3 on", 3), ("fp32 levels 0-1, bf16 from level 2 on", 2), ("fp32 level 0, bf16 from level 1 on", 1)]:
        model.set_compute_dtype(torch.float32, encoder_tail=torch.bfloat16, tail_level=level)
        z, log = model.encode(x, return_reg_log=True)
        same = int((log["indices"] == ref).sum())
        print(f"  {label:46s} codes {same}/{n} = {same / n:.4f}   z rel {float((z - zref).norm() / zref.norm()):.2e}", flush=True)


if __name__ == "__main__":
    main()
