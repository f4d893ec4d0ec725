"""Build-container helper: per-frame PSNR / SSIM of the UNMODIFIED reference (vidtok/modules/util.py compute_psnr,
compute_ssim after the post-processing of scripts/inference_evaluate.py:175-176) on seeded inputs ->
tests/golden/metrics.safetensors.  Inputs are regenerated from the seeds in tests/golden_cases.METRIC_CASES, only the
reference's outputs are stored.  Re-run: `python scripts/make_golden_metrics.py`."""
import os
import sys

import torch
from safetensors.torch import save_file

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from golden_cases import METRIC_CASES, make_metric_inputs  # noqa: E402
from oracle.refload import _install_stubs  # noqa: E402
from util import GOLDEN_DIR  # noqa: E402


def main():
    _install_stubs()
    from vidtok.modules.util import compute_psnr, compute_ssim

    out = {}
    for case in METRIC_CASES:
        x, y = make_metric_inputs(case)
        y = y.clamp(-1, 1)
        xp, yp = (x + 1) / 2, (y + 1) / 2
        B, C, T, H, W = x.shape
        ps = torch.empty(B, T)
        ss = torch.empty(B, T)
        for b in range(B):
            for t in range(T):
                ps[b, t] = compute_psnr(xp[b:b + 1, :, t:t + 1], yp[b:b + 1, :, t:t + 1])
                ss[b, t] = compute_ssim(xp[b:b + 1, :, t:t + 1], yp[b:b + 1, :, t:t + 1])
        # the whole-clip call of the eval loop is the mean of the per-frame values
        assert abs(float(compute_psnr(xp, yp)) - float(ps.mean())) < 1e-4
        assert abs(float(compute_ssim(xp, yp)) - float(ss.mean())) < 1e-5
        out[case["name"] + ".psnr"], out[case["name"] + ".ssim"] = ps, ss
        print(case["name"], ps.flatten()[:3].tolist(), ss.flatten()[:3].tolist())
    save_file(out, os.path.join(GOLDEN_DIR, "metrics.safetensors"))


if __name__ == "__main__":
    main()
