"""Evaluation metrics: the oracle against the reference and its committed outputs, the
host mirror's conventions on CPU (operator emulated), and -- `-m gpu` -- the HIP kernel against golden and oracle."""
import os

import pytest
import torch
from safetensors.torch import load_file

import torch_ops_ref
from golden_cases import METRIC_CASES, make_metric_inputs
from oracle import metrics_oracle as M
from oracle.refload import reference_available
from util import GOLDEN_DIR

# fp32 accumulation order differs between the separable HIP filter, torch's conv2d and the reference's box: PSNR in dB
# and SSIM in [0,1] agree to a few 1e-6 relative; the tolerance written here is what the tests enforce.
PSNR_TOL, SSIM_TOL = 2e-4, 2e-5
IDS = [c["name"] for c in METRIC_CASES]


def _gold(case):
    g = load_file(os.path.join(GOLDEN_DIR, "metrics.safetensors"))
    return g[case["name"] + ".psnr"], g[case["name"] + ".ssim"]


@pytest.mark.parametrize("case", METRIC_CASES, ids=IDS)
def test_oracle_reproduces_reference_golden(case):
    x, y = make_metric_inputs(case)
    ps, ss = M.eval_psnr_ssim(x, y)
    gp, gs = _gold(case)
    assert (ps - gp).abs().max() < PSNR_TOL and (ss - gs).abs().max() < SSIM_TOL


@pytest.mark.reference
@pytest.mark.skipif(not reference_available(), reason="needs /root/reference")
def test_oracle_matches_reference_functions():
    from oracle.refload import _install_stubs

    _install_stubs()
    from vidtok.modules.util import compute_psnr, compute_ssim

    torch.manual_seed(0)
    x = torch.rand(2, 3, 4, 48, 40)
    y = (x + 0.1 * torch.randn_like(x)).clamp(0, 1)
    assert abs(float(compute_psnr(x, y)) - float(M.psnr_frames(x, y).mean())) < PSNR_TOL
    assert abs(float(compute_ssim(x, y)) - float(M.ssim_frames(x, y).mean())) < SSIM_TOL
    with pytest.raises(ValueError):
        M.ssim_frames(x[..., :10, :], y[..., :10, :])


def test_host_mirror_conventions(monkeypatch):
    """compute_psnr / compute_ssim keep the reference's convention: [0,1] images, 4-D or 5-D, scalar mean."""
    torch_ops_ref.patch_ops(monkeypatch)
    from vidtok_amd import metrics

    torch.manual_seed(1)
    x = torch.rand(1, 3, 2, 32, 32)
    y = (x + 0.05 * torch.randn_like(x)).clamp(0, 1)
    p5, s5 = metrics.compute_psnr(x, y), metrics.compute_ssim(x, y)
    x4, y4 = x[0].transpose(0, 1), y[0].transpose(0, 1)          # (t, c, h, w) frames as a batch
    assert p5.dim() == 0 and abs(float(p5) - float(metrics.compute_psnr(x4, y4))) < 1e-6
    assert abs(float(s5) - float(metrics.compute_ssim(x4, y4))) < 1e-6
    assert float(metrics.compute_ssim(x, x)) == pytest.approx(1.0, abs=1e-6)
    assert float(metrics.compute_psnr(x, x)) == pytest.approx(80.0, abs=1e-3)       # -10 log10(1e-8)


def test_eval_refuses_cpu_tensors():
    from vidtok_amd import ops

    with pytest.raises(Exception):
        ops.eval_psnr_ssim(torch.zeros(1, 3, 1, 16, 16), torch.zeros(1, 3, 1, 16, 16))


@pytest.mark.gpu
@pytest.mark.parametrize("case", METRIC_CASES, ids=IDS)
def test_hip_metrics_match_reference_golden(case):
    from vidtok_amd import ops

    x, y = make_metric_inputs(case)
    ps, ss = ops.eval_psnr_ssim(x.cuda(), y.cuda())
    gp, gs = _gold(case)
    assert (ps.cpu() - gp).abs().max() < PSNR_TOL and (ss.cpu() - gs).abs().max() < SSIM_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 3, 17, 256, 256), (2, 3, 3, 100, 77), (1, 3, 1, 720, 1280), (1, 1, 2, 11, 50)])
def test_hip_metrics_match_oracle(shape):
    from vidtok_amd import metrics, ops

    g = torch.Generator().manual_seed(sum(shape))
    x = torch.rand(shape, generator=g) * 2 - 1
    y = x + 0.1 * torch.randn(shape, generator=g)
    ps, ss = ops.eval_psnr_ssim(x.cuda(), y.cuda())
    ops_, oss = M.eval_psnr_ssim(x, y)
    assert (ps.cpu() - ops_).abs().max() < PSNR_TOL and (ss.cpu() - oss).abs().max() < SSIM_TOL
    # [0,1] convention + properties: identical clips -> SSIM 1, PSNR 80 dB (the 1e-8 floor)
    xp, yp = M.postprocess(x, y)
    assert abs(float(metrics.compute_psnr(xp.cuda(), yp.cuda())) - float(ops_.mean())) < PSNR_TOL
    assert float(metrics.compute_ssim(xp.cuda(), xp.cuda())) == pytest.approx(1.0, abs=1e-6)
    assert float(metrics.compute_psnr(xp.cuda(), xp.cuda())) == pytest.approx(80.0, abs=1e-3)


@pytest.mark.gpu
def test_hip_metrics_bad_size_fails_loudly():
    from vidtok_amd import ops

    with pytest.raises(RuntimeError, match="Kernel size"):
        ops.eval_psnr_ssim(torch.zeros(1, 3, 1, 10, 32).cuda(), torch.zeros(1, 3, 1, 10, 32).cuda())


@pytest.mark.gpu
def test_evaluate_clip_matches_oracle_metrics():
    """model(x) + fused post-processing + per-frame metrics: the per-batch body of the reference's eval loop"""
    from util import build_model
    from vidtok_amd import metrics

    model, _, _ = build_model("vidtok_kl_causal_488_4chn", seed=3, device="cuda")
    x = torch.rand((1, 3, 5, 64, 64), generator=torch.Generator().manual_seed(4)) * 2 - 1
    torch.manual_seed(0)
    xrec, psnr, ssim = metrics.evaluate_clip(model, x.cuda())
    assert xrec.shape == x.shape and psnr.shape == ssim.shape == (1, 5)
    ops_, oss = M.eval_psnr_ssim(x, xrec.cpu())
    assert (psnr.cpu() - ops_).abs().max() < PSNR_TOL and (ssim.cpu() - oss).abs().max() < SSIM_TOL
