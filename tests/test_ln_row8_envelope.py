"""The one-pass LayerNorm + SiLU row arithmetic of the fused 16-bit sites (`ln_row8`, vidtok_amd/csrc/common.h), restated in fp32 torch
operation by operation, against the float64 statement of LayerNorm -> x * sigmoid(x): what the single-pass variance (E[x^2] - mean^2) and
the folded exponent cost in accuracy, and where they stop being harmless.  CPU only: the envelope of the formula, not a kernel test (those
are tests/test_gpu_ops.py::test_conv*ln*, test_temporal_block_fused, and every 16-bit end-to-end gate)."""
import math

import pytest
import torch

NEG_LOG2E = -1.4426950408889634


def ln_row8_fp32(x, gamma, beta, eps, silu):
    """x: [rows, C] fp32 (the fp32 row a site holds before rounding to 16 bits)."""
    C = x.shape[1]
    x = x.float()
    s = x.sum(dim=1, keepdim=True, dtype=torch.float32)
    q = (x * x).sum(dim=1, keepdim=True, dtype=torch.float32)
    mean = s * (1.0 / C)
    ex2 = q * (1.0 / C)
    var = torch.clamp(ex2 - mean * mean, min=0.0)
    rstd = torch.rsqrt(var + eps)
    nm = -mean * rstd
    t = x * rstd + nm
    if not silu:
        return t * gamma + beta
    a = t * (gamma * NEG_LOG2E) + beta * NEG_LOG2E            # = -log2(e) u
    den = torch.exp2(a) * NEG_LOG2E + NEG_LOG2E               # = -log2(e) (1 + 2^a)
    return a / den


def ln_f64(x, gamma, beta, eps, silu):
    u = torch.nn.functional.layer_norm(x.double(), (x.shape[1],), gamma.double(), beta.double(), eps)
    return u * torch.sigmoid(u) if silu else u


@pytest.mark.parametrize("silu", [True, False], ids=["ln_silu", "ln"])
@pytest.mark.parametrize("C", [128, 256])
@pytest.mark.parametrize("mean,std,bound", [(0.0, 1.0, 2e-6), (3.0, 1.0, 1.2e-5), (30.0, 1.0, 8e-4), (-8.0, 0.25, 1e-3)],
                         ids=["centred", "mean_3_sigma", "mean_30_sigma", "mean_32_sigma_small_scale"])
def test_one_pass_rows_stay_inside_a_16_bit_ulp(C, mean, std, bound, silu):
    """Rows as the network produces them (|mean| up to a few standard deviations) come out 1e-6 ... 1e-5 from the float64 statement -- two
    to three orders below half a bf16 ulp (2^-10 ~ 1e-3 of an O(1) output) and one to two below half an fp16 ulp (2^-13 ~ 1.2e-4).  The error
    grows like (mean / sigma)^2 * 6e-7 (the cancellation in E[x^2] - mean^2): at |mean| = 30 sigma it is 5e-4 ... 8e-4 -- still under half a
    bf16 ulp, several fp16 ulps -- which is where the single pass stops being harmless; the measured end-to-end distances of the 16-bit modes
    did not move when it replaced the two-pass form (fp16: z 7.1e-4 -> 7.3e-4, reconstruction 2.7e-3 -> 2.6e-3 of the fp32 oracle)."""
    g = torch.Generator().manual_seed(C + int(mean * 10))
    x = torch.randn((512, C), generator=g) * std + mean
    gamma = torch.rand((C,), generator=g) + 0.5
    beta = torch.randn((C,), generator=g) * 0.2
    got = ln_row8_fp32(x, gamma, beta, 1e-6, silu).double()
    ref = ln_f64(x, gamma, beta, 1e-6, silu)
    err = (got - ref).abs().max().item()
    print(f"C={C} mean={mean} std={std} silu={silu}: max abs err {err:.2e}")
    assert err < bound <= 1e-3


def test_degenerate_rows_stay_finite():
    """All channels (nearly) equal: the single-pass variance cancels to a few ulps of mean^2 and may come out negative -- clamped at 0, so the
    row is normalised by rsqrt(eps) like the two-pass form would do for a truly constant row: finite, bounded by |x - mean| / sqrt(eps)."""
    C = 128
    x = torch.full((4, C), 7.25)
    x[1] += torch.linspace(-1e-4, 1e-4, C)
    x[2] = 0.0
    x[3] = -1e4
    out = ln_row8_fp32(x, torch.ones(C), torch.zeros(C), 1e-6, True)
    assert torch.isfinite(out).all()
    assert out[0].abs().max() < 1e-2 and out[2].abs().max() == 0.0


def test_folded_silu_equals_u_sigmoid_u():
    """a * rcp(fma(2^a, -log2e, -log2e)) with a = -log2e u is u * sigmoid(u): the algebra, in float64, over the range an affine LayerNorm reaches."""
    u = torch.linspace(-30.0, 30.0, 20001, dtype=torch.float64)
    a = u * NEG_LOG2E
    got = a / (torch.exp2(a) * NEG_LOG2E + NEG_LOG2E)
    ref = u * torch.sigmoid(u)
    assert (got - ref).abs().max().item() < 1e-12 * max(1.0, math.fabs(30.0))
