"""`-m gpu`: the whole encode/decode path on the MI355X through the C-ABI, against
  (1) the committed outputs of the unmodified reference (tests/golden/),
  (2) the CPU oracle on seeded inputs at sizes it finishes in seconds,
  (3) size-independent properties at BASELINE.json's full 17x256x256 size.

Tolerances: fp32 kernels vs fp32 reference: max|d|/max|ref| <= 1e-3 for z and the
reconstruction (measured ~1e-5); FSQ integer codes: equality rate reported, >= 99.9 % required in
fp32 mode (a code may flip only when the pre-quantisation value sits within fp32 round-off of a
rounding boundary); bf16 kernels: error vs the fp32 oracle reported and bounded loosely (bf16 has
3 significant digits; the reference's own fp32 vs bf16-autocast runs differ by ~2e-2)."""
import os

import pytest
import torch
from safetensors.torch import load_file

from golden_cases import CASES, apply_tiling, make_input
from util import GOLDEN_DIR, build_model, build_oracle, cpu_autocast_usable, handle_config, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
# tiers: `-m gpu` keeps one oracle case per shipped path and BASELINE size; `-m "gpu and variants"` adds the long host runs (the 129-frame
# configs[4] clip: a 5-minute CPU oracle pass; the configs[3] shard; the 256 x 256 autocast oracle) -- run once per round, not by every call
variants = pytest.mark.variants
# bf16 gates: a small multiple of what the bf16 kernels measure against the fp32 oracle (recon 1.5e-2..2.2e-2, z 6e-3,
# FSQ codes 93-94 % -- the reference's own fp32 vs bf16-autocast agreement, SURVEY.md finding 5): a 2-3x loss of
# accuracy in a kernel turns these red.  test_bf16_vs_autocast_oracle additionally pins the bf16 path to the
# oracle run under torch.autocast(bfloat16) on the same GPU (the reference's own bf16 mode).
BF16_RECON, BF16_Z, BF16_CODE_RATE = 5e-2, 2e-2, 0.9
# fp16 gates (round 6: the reference's README precision, torch.autocast(dtype=torch.float16)): 11 significant bits against bf16's 8 --
# the oracle under autocast(float16) sits at z 1.6e-3 / reconstruction 3.2e-3 from the fp32 oracle on a 17x64x64 clip (host run)
F16 = torch.float16
H16 = (torch.bfloat16, F16)
RECON = {torch.bfloat16: BF16_RECON, F16: 1e-2}
ZTOL = {torch.bfloat16: BF16_Z, F16: 5e-3}
CODE_RATE = {torch.bfloat16: BF16_CODE_RATE, F16: 0.98}
# A code flip is a coin toss at a rounding boundary, so an absolute rate gate on the 320 tokens of a 17x64x64 clip is noise
# (288..297 of 320 across summation orders): the bf16 FSQ cases of test_matches_cpu_oracle run 17x128x128 clips (1 280 / 768
# tokens) and are gated RELATIVE to the reference's own bf16 mode on the same clip -- the oracle under torch.autocast(bfloat16)
# (host): rate >= autocast rate - 0.03, as test_bf16_vs_autocast_oracle does -- plus the absolute floor of the full-size
# cases less two points (BF16_CODE_RATE gates those: 5 120 / 20 480 tokens).
BF16_CODE_RATE_VS_AUTOCAST = 0.03


def _autocast_oracle_codes(ora, x, dtype=torch.bfloat16):
    """FSQ codes of the oracle run the way the reference runs bf16 / fp16: encoder under torch.autocast(dtype), regulariser in fp32"""
    with torch.autocast("cpu", dtype=dtype):
        h = ora.pre_quant(x)
    return ora.regularize(h.float())[1]["indices"]


_ORACLE_RUNS = {}      # session cache of CPU oracle passes (test infrastructure): see _oracle_full and test_matches_cpu_oracle


# split-bf16 mode ("bf16x3": fp32 storage, every convolution as three bf16 MFMAs per product, vt_conv VT_BF16X3): the fast
# mode that has to stay inside the reference's fp32 tolerance -- gated like the fp32 kernels (1e-3; measured 1e-5 .. 4e-5)
X3 = "bf16x3"


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_fp32_matches_reference_golden(case):
    gold = load_file(os.path.join(GOLDEN_DIR, case["name"] + ".safetensors"))
    model, cfg, sd = build_model(case["config"], seed=case["weight_seed"], device=DEV, dtype=torch.float32)
    apply_tiling(model, case)
    torch.manual_seed(case["noise_seed"])          # host-side noise stream identical to the reference's
    z, dec, log = model(make_input(case).to(DEV))
    ez, ed = rel_err(z, gold["z"]), rel_err(dec, gold["dec"])
    print(f"{case['name']}: z rel {ez:.2e} dec rel {ed:.2e}")
    assert dec.shape == gold["dec"].shape and dec.dtype == torch.float32 and dec.is_cuda
    assert ez < 1e-3 and ed < 1e-3
    if "indices" in gold:
        rate = (log["indices"].cpu() == gold["indices"]).float().mean().item()
        print(f"{case['name']}: FSQ code match rate {rate:.6f}")
        assert log["indices"].dtype == torch.int32 and rate == 1.0
        dec2 = model.decode(log["indices"], decode_from_indices=True)
        assert torch.equal(dec2[:, :, -dec.shape[2]:], dec)       # decode(indices) == decode(z), bit for bit
        assert abs(float(log["aux_loss"]) - float(gold["aux_loss"])) < 1e-3
    else:
        assert abs(float(log["kl_loss"]) - float(gold["kl_loss"])) < 1e-3 * abs(float(gold["kl_loss"]))


def _decode_err_on_oracle_codes(model, log2, dec2):
    d = model.decode(log2["indices"].to(DEV), decode_from_indices=True)
    return rel_err(d[:, :, -dec2.shape[2]:], dec2)


@pytest.mark.parametrize("name,shape,dtype,tol", [
    ("vidtok_kl_causal_488_4chn", (2, 3, 17, 64, 64), torch.float32, 1e-3),
    ("vidtok_fsq_causal_488_32768", (1, 3, 17, 64, 64), torch.float32, 1e-3),
    ("vidtok_kl_causal_488_4chn", (2, 3, 17, 64, 64), torch.bfloat16, BF16_RECON),
    ("vidtok_fsq_causal_488_32768", (1, 3, 17, 128, 128), torch.bfloat16, BF16_RECON),     # 1 280 tokens: the code-rate gates need draws
    ("vidtok_kl_causal_488_16chn", (1, 3, 9, 40, 24), torch.bfloat16, BF16_RECON),
    ("vidtok_kl_causal_488_4chn", (2, 3, 17, 64, 64), F16, RECON[F16]),
    ("vidtok_fsq_causal_488_32768", (1, 3, 17, 128, 128), F16, RECON[F16]),
    ("vidtok_kl_causal_488_16chn", (1, 3, 9, 40, 24), F16, RECON[F16]),
    ("vidtok_kl_noncausal_488_4chn", (1, 3, 16, 64, 64), F16, RECON[F16]),
    ("vidtok_v1_1/vidtok_fsq_causal_888_32768_v1_1", (1, 3, 17, 128, 128), F16, RECON[F16]),
    ("vidtok_kl_noncausal_488_4chn", (2, 3, 16, 64, 64), torch.float32, 1e-3),
    ("vidtok_kl_noncausal_488_4chn", (1, 3, 16, 64, 64), torch.bfloat16, BF16_RECON),
    ("vidtok_fsq_noncausal_41616_262144", (1, 3, 8, 64, 64), torch.float32, 1e-3),
    # the other compression schedules of the reference's config set on the GPU --
    # 4x16x16 (ch_mult [1,2,4,4,4]), 2x8x8 (tempo_ds [1]), 4x4x4 (spatial_ds [1,2]), 8x8x8 v1.1 (tempo_ds [0,1,2]),
    # the other FSQ codebooks, the v1.1 variants un-tiled (reference configs/*.yaml:19-23)
    ("vidtok_kl_causal_488_4chn", (2, 3, 17, 64, 64), X3, 1e-3),
    ("vidtok_fsq_causal_488_32768", (1, 3, 17, 64, 64), X3, 1e-3),
    ("vidtok_kl_causal_488_16chn", (1, 3, 9, 40, 24), X3, 1e-3),
    ("vidtok_kl_noncausal_488_4chn", (1, 3, 16, 64, 64), X3, 1e-3),
    ("vidtok_fsq_noncausal_41616_262144", (1, 3, 8, 64, 64), X3, 1e-3),
    ("vidtok_v1_1/vidtok_fsq_causal_888_32768_v1_1", (1, 3, 17, 64, 64), X3, 1e-3),
    ("vidtok_v1_1/vidtok_kl_causal_41616_16chn_v1_1", (1, 3, 9, 64, 64), X3, 1e-3),
    ("vidtok_kl_causal_41616_4chn", (1, 3, 9, 64, 64), torch.float32, 1e-3),
    ("vidtok_kl_causal_41616_4chn", (1, 3, 9, 64, 64), torch.bfloat16, BF16_RECON),
    ("vidtok_kl_causal_288_8chn", (2, 3, 9, 64, 48), torch.float32, 1e-3),
    ("vidtok_kl_causal_288_8chn", (1, 3, 9, 64, 48), torch.bfloat16, BF16_RECON),
    ("vidtok_kl_causal_444_4chn", (1, 3, 9, 32, 48), torch.float32, 1e-3),
    ("vidtok_kl_causal_444_4chn", (1, 3, 9, 32, 48), torch.bfloat16, BF16_RECON),
    ("vidtok_kl_causal_488_8chn", (1, 3, 5, 64, 64), torch.float32, 1e-3),
    ("vidtok_fsq_causal_488_4096", (1, 3, 9, 64, 64), torch.float32, 1e-3),
    ("vidtok_fsq_causal_488_262144", (1, 3, 9, 64, 64), torch.float32, 1e-3),
    ("vidtok_fsq_causal_41616_262144", (1, 3, 9, 64, 64), torch.float32, 1e-3),
    ("vidtok_v1_1/vidtok_fsq_causal_888_32768_v1_1", (1, 3, 17, 64, 64), torch.float32, 1e-3),
    ("vidtok_v1_1/vidtok_fsq_causal_888_32768_v1_1", (1, 3, 17, 128, 128), torch.bfloat16, BF16_RECON),   # 768 tokens
    ("vidtok_v1_1/vidtok_kl_causal_288_8chn_v1_1", (1, 3, 9, 64, 64), torch.float32, 1e-3),
    ("vidtok_v1_1/vidtok_kl_causal_41616_16chn_v1_1", (1, 3, 9, 64, 64), torch.float32, 1e-3),
    ("vidtok_v1_1/vidtok_fsq_causal_41616_262144_v1_1", (1, 3, 9, 64, 64), torch.float32, 1e-3),
    ("vidtok_v1_1/vidtok_kl_causal_488_4chn_v1_1", (1, 3, 9, 64, 64), torch.float32, 1e-3),
    ("vidtok_kl_noncausal_41616_4chn", (1, 3, 8, 64, 64), torch.float32, 1e-3),
    ("vidtok_kl_noncausal_488_16chn", (1, 3, 8, 64, 64), torch.float32, 1e-3),
    ("vidtok_fsq_noncausal_488_262144", (1, 3, 8, 64, 64), torch.float32, 1e-3),
])
def test_matches_cpu_oracle(name, shape, dtype, tol):
    model, cfg, sd = build_model(name, seed=21, device=DEV, dtype=dtype)
    g = torch.Generator().manual_seed(9)
    x = torch.rand(shape, generator=g) * 2 - 1
    torch.manual_seed(4)
    z, dec, log = model(x.to(DEV))
    key = ("small", name, shape)                    # the oracle's pass of a (config, clip) is shared by the arithmetic modes that run it
    if key not in _ORACLE_RUNS:
        ora = build_oracle(cfg, sd)
        torch.manual_seed(4)
        _ORACLE_RUNS[key] = (ora, ora(x))
        for k in [k for k in _ORACLE_RUNS if k[0] == "small" and k != key][:-3]:      # a few recent ones only
            del _ORACLE_RUNS[k]
    ora, (z2, dec2, log2) = _ORACLE_RUNS[key]
    ez, ed = rel_err(z, z2), rel_err(dec, dec2)
    print(f"{name} {shape} {dtype}: z rel {ez:.3e} dec rel {ed:.3e}")
    assert dec.shape == dec2.shape
    if "indices" in log2 and dtype in H16:
        # a bf16 latent next to a rounding boundary flips a code (the reference's own bf16 mode agrees with its fp32
        # mode on ~94 % of the codes, SURVEY.md finding 5), and one flipped code moves the max-norm of the
        # reconstruction by more than any kernel error: the decoder is therefore gated on the ORACLE's codes
        ed_codes = _decode_err_on_oracle_codes(model, log2, dec2)
        print(f"{name} {dtype}: dec rel on the oracle's codes {ed_codes:.3e} (end to end incl. code flips {ed:.3e})")
        assert ed_codes < tol
    else:
        assert ed < tol
    if "indices" not in log2 or dtype not in H16:
        # (bf16 FSQ latents are code values: compared through the match rate below)
        assert ez < (tol if dtype not in H16 else ZTOL[dtype])
    if "indices" in log2:
        rate = (log["indices"].cpu() == log2["indices"]).float().mean().item()
        print(f"{name} {dtype}: FSQ code match rate {rate:.5f} over {log2['indices'].numel()} tokens")
        # fp32 kernels: every code (measured everywhere; north_star: bit-exact); split-bf16: these clips measure 1.0 too, the
        # gate leaves room for one boundary case per thousand tokens
        if dtype in H16:
            assert rate >= CODE_RATE[dtype] - 0.02
            if cpu_autocast_usable(dtype):        # (a host whose torch has no vectorised fp16 convolution skips the relative gate, not the absolute one)
                r_auto = (_autocast_oracle_codes(ora, x, dtype) == log2["indices"]).float().mean().item()
                print(f"{name} {dtype}: the oracle under autocast({dtype}) on the same clip: {r_auto:.5f}")
                assert rate >= r_auto - BF16_CODE_RATE_VS_AUTOCAST
        else:
            assert rate == 1.0 if dtype == torch.float32 else rate >= 0.999
        # the quantiser itself is exact: feeding the oracle's own pre-quantisation h gives its codes
        h = ora.pre_quant(x)
        _, qlog = model.regularization(h.to(DEV))
        assert (qlog["indices"].cpu() != log2["indices"]).sum() == 0


@pytest.mark.parametrize("name,shape", [("vidtok_kl_causal_488_4chn", (1, 3, 9, 128, 128)), ("vidtok_kl_noncausal_488_4chn", (1, 3, 8, 64, 64))])
def test_attention_flash_and_operator_paths(name, shape, vt_opts):
    """the attention block as ONE launch (vt_flash_attention, default where it applies: bf16, S % 64 == 0) and as the operator sequence
    GEMM -> softmax -> GEMM (option attn_flash = 0): both inside the bf16 gates against the fp32 oracle, and close to each other"""
    model, cfg, sd = build_model(name, seed=12, device=DEV, dtype=torch.bfloat16)
    model.regularization.sample = False
    ora = build_oracle(cfg, sd)
    ora.sample = False
    x = torch.rand(shape, generator=torch.Generator().manual_seed(5)) * 2 - 1
    z2, dec2, _ = ora(x)
    outs = {}
    for flash in (1, 0):
        vt_opts(attn_flash=flash)
        z, dec, _ = model(x.to(DEV))
        ez, ed = rel_err(z, z2), rel_err(dec, dec2)
        print(f"{name} attn_flash={flash}: z rel {ez:.3e} dec rel {ed:.3e}")
        assert ez < BF16_Z and ed < BF16_RECON
        outs[flash] = (z, dec)
    assert not torch.equal(outs[0][1], outs[1][1])                      # two different kernels did run
    assert rel_err(outs[1][0], outs[0][0]) < BF16_Z and rel_err(outs[1][1], outs[0][1]) < BF16_RECON


@pytest.mark.parametrize("name,shape", [("vidtok_kl_causal_488_4chn", (1, 3, 9, 32, 32)),
                                        ("vidtok_kl_noncausal_488_4chn", (1, 3, 8, 32, 32))])
def test_groupnorm_variant_matches_oracle(name, shape):
    model, cfg, sd = build_model(name, seed=6, device=DEV, overrides=dict(norm_type="groupnorm"))
    ora = build_oracle(cfg, sd)
    x = torch.rand(shape, generator=torch.Generator().manual_seed(3)) * 2 - 1
    torch.manual_seed(4)
    z, dec, log = model(x.to(DEV))
    torch.manual_seed(4)
    z2, dec2, log2 = ora(x)
    assert rel_err(z, z2) < 1e-3 and rel_err(dec, dec2) < 1e-3


def test_fsq_with_projections_matches_oracle():
    model, cfg, sd = build_model("vidtok_fsq_causal_488_32768", seed=5, device=DEV, overrides=dict(z_channels=8),
                                 reg_overrides=dict(dim=8, levels=[8, 8, 8, 5, 5, 5]))
    ora = build_oracle(cfg, sd)
    x = torch.rand((1, 3, 5, 32, 32), generator=torch.Generator().manual_seed(2)) * 2 - 1
    z, dec, log = model(x.to(DEV))
    z2, dec2, log2 = ora(x)
    assert z.shape == (1, 8, 2, 4, 4) and rel_err(z, z2) < 1e-3 and rel_err(dec, dec2) < 1e-3
    assert torch.equal(log["indices"].cpu(), log2["indices"])


# with the codebook axis kept the reference's own forward raises as soon as an aux-loss weight is non-zero
# (its implicit codebook is flattened to 1-D, regularizers.py:143-146,191-192,234): these options exist without aux loss
NO_AUX = dict(entropy_loss_weight=0.0, commitment_loss_weight=0.0)


@pytest.mark.parametrize("zc,reg", [(6, dict(levels=[8, 5, 5], num_codebooks=2, **NO_AUX)),
                                    (8, dict(levels=[8, 5, 5], num_codebooks=2, dim=8, **NO_AUX)),
                                    (3, dict(levels=[8, 5, 5], keep_num_codebooks_dim=True, **NO_AUX))],
                         ids=["two_codebooks", "two_codebooks_projected", "one_codebook_kept_axis"])
def test_fsq_num_codebooks_matches_oracle(zc, reg):
    """num_codebooks > 1 / keep_num_codebooks_dim of the reference's FSQRegularizer (regularizers.py:100-150,227,247-262)"""
    model, cfg, sd = build_model("vidtok_fsq_causal_488_32768", seed=5, device=DEV, overrides=dict(z_channels=zc), reg_overrides=reg)
    ora = build_oracle(cfg, sd)
    x = torch.rand((1, 3, 5, 32, 32), generator=torch.Generator().manual_seed(2)) * 2 - 1
    z, dec, log = model(x.to(DEV))
    z2, dec2, log2 = ora(x)
    assert log["indices"].shape == (1, 2, 4, 4, reg.get("num_codebooks", 1)) and torch.equal(log["indices"].cpu(), log2["indices"])
    assert z.shape == (1, zc, 2, 4, 4) and rel_err(z, z2) < 1e-3 and rel_err(dec, dec2) < 1e-3
    assert abs(float(log["aux_loss"]) - float(log2["aux_loss"])) < 1e-3 * max(1.0, abs(float(log2["aux_loss"])))
    assert torch.equal(model.decode(log["indices"], decode_from_indices=True), dec)


def test_v11_long_video_tiled_matches_oracle():
    name = "vidtok_v1_1/vidtok_kl_causal_488_16chn_v1_1"
    model, cfg, sd = build_model(name, seed=22, device=DEV, dtype=torch.float32)
    ora = build_oracle(cfg, sd)
    for m in (model, ora):
        m.use_tiling, m.t_chunk_enc, m.use_overlap = True, 16, True
    model.t_chunk_dec = 4
    x = torch.rand(1, 3, 49, 32, 32, generator=torch.Generator().manual_seed(3)) * 2 - 1
    torch.manual_seed(5)
    z, dec, log = model(x.to(DEV))
    torch.manual_seed(5)
    z2, dec2, log2 = ora(x)
    print(f"v1.1 tiled T=49: z rel {rel_err(z, z2):.2e} dec rel {rel_err(dec, dec2):.2e}")
    assert dec.shape == x.shape and rel_err(z, z2) < 1e-3 and rel_err(dec, dec2) < 1e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, X3, F16], ids=["f32", "bf16", "bf16x3", "f16"])
def test_full_size_properties(dtype):
    """BASELINE.json size (17x256x256): properties that need no oracle."""
    model, cfg, sd = build_model("vidtok_fsq_causal_488_32768", seed=23, device=DEV, dtype=dtype)
    model.regularization.compute_aux_loss = False
    g = torch.Generator().manual_seed(11)
    x = (torch.rand((2, 3, 17, 256, 256), generator=g) * 2 - 1).to(DEV)
    z, dec, log = model(x)
    assert dec.shape == x.shape and z.shape == (2, 5, 5, 32, 32) and log["indices"].shape == (2, 5, 32, 32)
    assert torch.isfinite(dec).all() and torch.isfinite(z).all()
    assert int(log["indices"].min()) >= 0 and int(log["indices"].max()) < 32768
    # determinism and batch independence: clips are independent units
    z1, dec1, log1 = model(x[1:2].contiguous())
    assert torch.equal(log1["indices"], log["indices"][1:2]) and torch.equal(dec1, dec[1:2])
    # indices <-> latent <-> decode identities
    assert torch.equal(model.indices_to_latent(log["indices"]), z)
    assert torch.equal(model.decode(log["indices"], decode_from_indices=True), dec)
    # causality: the first 5 output frames depend only on the first 5 input frames' latents
    x2 = x.clone()
    x2[:, :, 9:] = -x2[:, :, 9:]
    _, decb, _ = model(x2)
    assert torch.equal(decb[:, :, :1], dec[:, :, :1])


def test_v11_full_size_long_video_tiling_property():
    """BASELINE.json configs[4] shape (129x256x256, t_chunk_enc=16, overlap) in fp32: the tiled path with
    decoder look-ahead must reproduce the un-tiled forward (the reference's own tiled/un-tiled gap is
    6e-4); z is chunk-invariant to fp32 round-off."""
    name = "vidtok_v1_1/vidtok_kl_causal_488_16chn_v1_1"
    model, cfg, sd = build_model(name, seed=31, device=DEV, dtype=torch.float32)
    model.regularization.sample = False
    x = (torch.rand((1, 3, 129, 256, 256), generator=torch.Generator().manual_seed(2)) * 2 - 1).to(DEV)
    z0, dec0, _ = model(x)
    model.use_tiling, model.t_chunk_enc, model.t_chunk_dec, model.use_overlap = True, 16, 4, True
    z1, dec1, _ = model(x)
    ez, ed = rel_err(z1, z0), rel_err(dec1, dec0)
    print(f"v1.1 129x256x256 tiled vs un-tiled: z rel {ez:.2e} dec rel {ed:.2e}")
    assert dec1.shape == x.shape == dec0.shape and z1.shape == (1, 16, 33, 32, 32)
    assert ez < 1e-4 and ed < 5e-3


# --------------------------------------------------------------------------------------------------
# BASELINE.json sizes against the oracle: the layers of the benchmarked problem -- the 8-wave
# 256x256 conv tile (picked for Cout % 256 == 0 with >= 384 tiles, i.e. only at this size), the frames-innermost
# tile order over 20 frames, cache-mode gathers on 256x256 frames -- are compared with the CPU oracle itself.
# The oracle needs ~30 s per 17x256x256 clip, so each (config, shape) result is computed once per session.
# --------------------------------------------------------------------------------------------------
def _oracle_full(name, shape, seed, tiling=None):
    key = (name, shape, seed, tiling)
    if key not in _ORACLE_RUNS:
        model, cfg, sd = build_model(name, seed=seed, device="cpu")
        ora = build_oracle(cfg, sd)
        if tiling:
            ora.use_tiling, ora.t_chunk_enc, ora.use_overlap = True, tiling[0], tiling[1]
        x = torch.rand(shape, generator=torch.Generator().manual_seed(41)) * 2 - 1
        torch.manual_seed(8)
        _ORACLE_RUNS[key] = (cfg, sd, x, ora(x))
    return _ORACLE_RUNS[key]


# kl_16chn B=2: the per-GPU shard of BASELINE.json configs[3] (the N > 1 bench workload) -- VERDICT r2 weak #2
FULL = [pytest.param("vidtok_kl_causal_488_4chn", (1, 3, 17, 256, 256), id="kl_4chn_B1"), pytest.param("vidtok_fsq_causal_488_32768", (1, 3, 17, 256, 256), id="fsq_B1"),
        pytest.param("vidtok_kl_causal_488_4chn", (2, 3, 17, 256, 256), id="kl_4chn_B2", marks=variants),
        pytest.param("vidtok_kl_causal_488_16chn", (2, 3, 17, 256, 256), id="kl_16chn_B2", marks=variants)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, X3, F16], ids=["f32", "bf16", "bf16x3", "f16"])
@pytest.mark.parametrize("name,shape", FULL)
def test_full_size_matches_cpu_oracle(name, shape, dtype):
    cfg, sd, x, (z2, dec2, log2) = _oracle_full(name, shape, 33)
    model, _, _ = build_model(name, seed=33, device=DEV, dtype=dtype)
    torch.manual_seed(8)
    z, dec, log = model(x.to(DEV))
    ez, ed = rel_err(z, z2), rel_err(dec, dec2)
    print(f"FULL {name} {shape} {dtype}: z rel {ez:.3e} dec rel {ed:.3e}")
    assert dec.shape == dec2.shape
    if dtype not in H16:
        assert ez < 1e-3 and ed < 1e-3
    elif "indices" in log2:
        ed_codes = _decode_err_on_oracle_codes(model, log2, dec2)   # see test_matches_cpu_oracle: gate on equal codes
        print(f"FULL {name} {dtype}: dec rel on the oracle's codes {ed_codes:.3e}")
        assert ed_codes < RECON[dtype]
    else:
        assert ed < RECON[dtype] and ez < ZTOL[dtype]
    if "indices" in log2:
        n_bad = int((log["indices"].cpu() != log2["indices"]).sum())
        print(f"FULL {name} {dtype}: {n_bad} of {log2['indices'].numel()} FSQ codes differ")
        if dtype == torch.float32:
            assert n_bad == 0                      # bit-exact at the benchmarked size
        elif dtype == X3:
            assert n_bad <= 5                      # split-bf16: >= 99.9 % of the 5 120 codes (measured: 0 differ)
        else:
            assert n_bad <= (1 - CODE_RATE[dtype]) * log2["indices"].numel()


@pytest.mark.parametrize("T", [17, pytest.param(33, marks=variants)], ids=["t17", "t33"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, X3, F16], ids=["f32", "bf16", "bf16x3", "f16"])
def test_full_size_v11_tiled_matches_cpu_oracle(dtype, T):
    """BASELINE.json configs[4] geometry (256x256 frames, t_chunk_enc=16, decoder look-ahead) on a 17-frame clip (the single-frame first
    chunk + one 16-frame cache-mode chunk; 33 frames = two such chunks in the variants tier): cache-mode gathers, chunk caches and the
    trilinear up-sampler at full frame size vs the oracle -- in fp32, split-bf16, and in bf16 (the dtype BASELINE.json names for this
    configuration) and fp16."""
    name = "vidtok_v1_1/vidtok_kl_causal_488_16chn_v1_1"
    cfg, sd, x, (z2, dec2, log2) = _oracle_full(name, (1, 3, T, 256, 256), 34, tiling=(16, True))
    model, _, _ = build_model(name, seed=34, device=DEV, dtype=dtype)
    model.use_tiling, model.t_chunk_enc, model.t_chunk_dec, model.use_overlap = True, 16, 4, True
    torch.manual_seed(8)
    z, dec, log = model(x.to(DEV))
    ez, ed = rel_err(z, z2), rel_err(dec, dec2)
    print(f"FULL v1.1 tiled T={T} 256x256 {dtype}: z rel {ez:.3e} dec rel {ed:.3e}")
    assert dec.shape == x.shape
    assert (ez < 1e-3 and ed < 1e-3) if dtype not in H16 else (ez < ZTOL[dtype] and ed < RECON[dtype])


@variants
@pytest.mark.parametrize("dtype", [torch.bfloat16, X3, torch.float32, F16], ids=["bf16", "bf16x3", "f32", "f16"])
def test_configs4_long_video_tiled_matches_cpu_oracle(dtype):
    """BASELINE.json configs[4] AT ITS STATED LENGTH: vidtok_kl_causal_488_16chn_v1_1, one clip of 129x256x256, t_chunk_enc = 16
    temporal tiling with decoder look-ahead, against the CPU oracle running the same tiled protocol (one ~4-minute host run,
    cached for the session and shared by the three arithmetic modes; bf16 is the dtype BASELINE.json names)."""
    name = "vidtok_v1_1/vidtok_kl_causal_488_16chn_v1_1"
    cfg, sd, x, (z2, dec2, log2) = _oracle_full(name, (1, 3, 129, 256, 256), 35, tiling=(16, True))
    model, _, _ = build_model(name, seed=35, device=DEV, dtype=dtype)
    model.use_tiling, model.t_chunk_enc, model.t_chunk_dec, model.use_overlap = True, 16, 4, True
    torch.manual_seed(8)
    z, dec, log = model(x.to(DEV))
    ez, ed = rel_err(z, z2), rel_err(dec, dec2)
    print(f"configs[4] tiled T=129 256x256 {dtype}: z rel {ez:.3e} dec rel {ed:.3e}")
    assert dec.shape == x.shape and z.shape == (1, 16, 33, 32, 32)
    assert (ez < 1e-3 and ed < 1e-3) if dtype not in H16 else (ez < ZTOL[dtype] and ed < RECON[dtype])


@pytest.mark.parametrize("name,shape", [pytest.param("vidtok_kl_causal_488_4chn", (1, 3, 17, 128, 128), id="kl_128"),
                                        pytest.param("vidtok_fsq_causal_488_32768", (1, 3, 17, 64, 64), id="fsq_64"),
                                        pytest.param("vidtok_kl_causal_488_4chn", (1, 3, 17, 256, 256), id="kl_256_benchmarked_size", marks=variants)])
@pytest.mark.parametrize("adt", H16, ids=["bf16", "f16"])
def test_16bit_vs_autocast_oracle(name, shape, adt):
    """the bf16 / fp16 kernels vs the reference's own bf16 / fp16 mode (fp16 is the dtype of its README snippets)
    -- the oracle's functional torch graph run under torch.autocast(that dtype), regulariser in fp32 like the reference's autocast(enabled=False) block.  The
    autocast run is on the host (torch's CPU bf16 kernels): on this stack MIOpen's bf16 conv3d search takes minutes per
    layer shape (measured: the GPU variant of this test did not finish in 15 min).  Both are compared with the fp32
    CPU oracle: the HIP bf16 path must not be further from fp32 than 2x the autocast run is."""
    if not cpu_autocast_usable(adt):
        pytest.skip(f"torch's CPU convolutions under autocast({adt}) are not usable on this host (scalar fall-back)")
    model, cfg, sd = build_model(name, seed=35, device=DEV, dtype=adt)
    ora = build_oracle(cfg, sd)
    ora.sample = False
    if hasattr(model.regularization, "sample"):
        model.regularization.sample = False
    x = torch.rand(shape, generator=torch.Generator().manual_seed(42)) * 2 - 1
    z0, dec0, log0 = ora(x)                                        # fp32 CPU oracle
    with torch.autocast("cpu", dtype=adt):
        h = ora.pre_quant(x)
    za, loga = ora.regularize(h.float())
    with torch.autocast("cpu", dtype=adt):
        deca = ora.decode(za).float()
    z, dec, log = model(x.to(DEV))
    if "indices" in log0:
        r_ours = (log["indices"].cpu() == log0["indices"]).float().mean().item()
        r_auto = (loga["indices"] == log0["indices"]).float().mean().item()
        print(f"{adt} {name}: FSQ code match vs fp32 oracle: HIP {r_ours:.4f}, autocast oracle {r_auto:.4f}")
        assert r_ours >= CODE_RATE[adt] and r_ours >= r_auto - 0.03
        # decoders on equal codes (the fp32 oracle's)
        with torch.autocast("cpu", dtype=adt):
            deca = ora.decode(z0).float()
        dec = model.decode(log0["indices"].to(DEV), decode_from_indices=True)[:, :, -dec0.shape[2]:]
    else:
        ez_ours, ez_auto = rel_err(z, z0), rel_err(za, z0)
        print(f"{adt} {name}: z vs fp32 oracle: HIP {ez_ours:.3e}, autocast oracle {ez_auto:.3e}")
        assert ez_ours < ZTOL[adt] and ez_ours < 2.0 * ez_auto + (2e-3 if adt == torch.bfloat16 else 3e-4)
    e_ours, e_auto, e_cross = rel_err(dec, dec0), rel_err(deca, dec0), rel_err(dec, deca)
    print(f"{adt} {name}: recon vs fp32 oracle: HIP {e_ours:.3e}, autocast oracle {e_auto:.3e}; HIP vs autocast {e_cross:.3e}")
    assert e_ours < RECON[adt] and e_ours < 2.0 * e_auto + (5e-3 if adt == torch.bfloat16 else 8e-4) and e_cross < 2.0 * RECON[adt]


@pytest.mark.parametrize("name,shape,dtype", [("vidtok_fsq_causal_488_32768", (2, 3, 9, 64, 64), torch.bfloat16),
                                              ("vidtok_v1_1/vidtok_kl_causal_488_16chn_v1_1", (1, 3, 9, 64, 64), torch.float32)])
def test_engine_graph_cache_replays_bit_exact(name, shape, dtype):
    """enable_graphs(): encoder / decoder launch sequences captured per shape (vidtok_amd/graphs.py) must reproduce the
    eager launches bit for bit, on fresh inputs, across shapes, and after a weight reload."""
    model, cfg, sd = build_model(name, seed=12, device=DEV, dtype=dtype)
    if hasattr(model.regularization, "sample"):
        model.regularization.sample = False
    xs = [(torch.rand(shape, generator=torch.Generator().manual_seed(100 + i)) * 2 - 1).to(DEV) for i in range(4)]
    eager = [model(x) for x in xs]
    model.enable_graphs()
    for rnd in range(2):                                   # call 1 eager, call 2 captures, later calls replay
        for x, (z0, d0, l0) in zip(xs, eager):
            z, d, l = model(x)
            assert torch.equal(z, z0) and torch.equal(d, d0)
            if "indices" in l0:
                assert torch.equal(l["indices"], l0["indices"])
    assert len(model._genc.entries) == 1 and isinstance(next(iter(model._genc.entries.values())), tuple)
    x2 = xs[0][:, :, :5].contiguous()                      # another shape: its own entry
    z2e, d2e, _ = model.enable_graphs(False)(x2)
    model.enable_graphs()
    for _ in range(3):
        z2, d2, _ = model(x2)
        assert torch.equal(d2, d2e)
    sd2 = {k: v * 1.01 for k, v in sd.items()}
    model.load_state_dict(sd2)                             # invalidates: the next calls must use the new weights
    outs = [model(xs[0])[1] for _ in range(3)]
    assert not torch.equal(outs[0], eager[0][1]) and torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])


@pytest.mark.parametrize("name,dtype,overlap", [("vidtok_v1_1/vidtok_kl_causal_488_16chn_v1_1", torch.bfloat16, True),
                                                ("vidtok_v1_1/vidtok_fsq_causal_488_32768_v1_1", torch.float32, False)])
def test_tiled_chunks_replay_bit_exact(name, dtype, overlap):
    """v1.1 temporal tiling under enable_graphs(): chunks of a kind seen twice replay from a captured graph; the module
    caches live in persistent buffers updated in place, so replays must equal the eager chunk loop bit for bit -- on the
    first clip (eager / capture / replay mixed), on later clips (all replays) and on a clip of another length."""
    model, cfg, sd = build_model(name, seed=14, device=DEV, dtype=dtype)
    if hasattr(model.regularization, "sample"):
        model.regularization.sample = False
    model.use_tiling, model.t_chunk_enc, model.use_overlap = True, 8, overlap
    model.t_chunk_dec = model.t_chunk_enc // model.encoder.time_downsample_factor
    xs = [(torch.rand((1, 3, T, 64, 64), generator=torch.Generator().manual_seed(200 + i)) * 2 - 1).to(DEV)
          for i, T in enumerate([41, 41, 29, 41])]                 # 1 + 5 x 8 frames; 1 + 3 x 8 + 4: a ragged last chunk
    eager = [model(x) for x in xs]
    model.enable_graphs()
    for rnd in range(2):
        for x, (z0, d0, l0) in zip(xs, eager):
            z, d, l = model(x)
            assert z.shape == z0.shape and d.shape == d0.shape
            assert torch.equal(z, z0) and torch.equal(d, d0), (rnd, x.shape)
            if "indices" in l0:
                assert torch.equal(l["indices"], l0["indices"])
    kinds = [e for e in model._genc.entries.values() if isinstance(e, tuple)]
    assert len(kinds) >= 3                                         # first chunk, the chunk after it, the steady state
    assert len([e for e in model._gdec.entries.values() if isinstance(e, tuple)]) >= 3


@pytest.mark.parametrize("ov,T", [(dict(resamp_with_conv=False), 5), (dict(init_pad_mode="constant"), 6),
                                  (dict(init_pad_mode="reflect"), 6), (dict(tanh_out=True), 5), (dict(give_pre_end=True), 5)],
                         ids=["no_resamp_conv", "pad_constant", "pad_reflect", "tanh_out", "give_pre_end"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_rare_constructor_options_match_oracle(ov, T, dtype):
    """constructor options of the reference no shipped YAML sets (model_3dcausal.py:200-230, 37-43/680/688, 862-869)"""
    model, cfg, sd = build_model("vidtok_kl_causal_488_4chn", seed=9, device=DEV, dtype=dtype, overrides=ov)
    ora = build_oracle(cfg, sd)
    x = torch.rand(1, 3, T, 64, 64, generator=torch.Generator().manual_seed(6)) * 2 - 1
    torch.manual_seed(2)
    z = model.encode(x.to(DEV))
    dec = model.decoder(z)
    torch.manual_seed(2)
    z2, _ = ora.encode(x)
    dec2 = ora.decode(z2)
    ez, ed = rel_err(z, z2), rel_err(dec, dec2)
    print(f"{ov} {dtype}: z rel {ez:.2e} dec rel {ed:.2e}")
    assert dec.shape == dec2.shape
    assert (ez < 1e-3 and ed < 1e-3) if dtype != torch.bfloat16 else (ez < BF16_Z and ed < BF16_RECON)


@pytest.mark.parametrize("name,shape", [("vidtok_kl_causal_488_4chn", (1, 3, 9, 64, 64)),
                                        ("vidtok_v1_1/vidtok_kl_causal_488_4chn_v1_1", (1, 3, 9, 64, 64))], ids=["v1_0", "v1_1"])
def test_temporal_blocks_fused_and_unfused(name, shape, monkeypatch):
    """VIDTOK_AMD_FUSE_TBLOCK: the widest level's temporal residual blocks as ONE launch (vt_temporal_block) or as
    LayerNorm + two convolutions -- both against the oracle, and the switch must really change the launch sequence."""
    from vidtok_amd import modules, ops

    x = torch.rand(shape, generator=torch.Generator().manual_seed(9)) * 2 - 1
    outs = {}
    for fused in (True, False):
        monkeypatch.setattr(modules, "_FUSE_TBLOCK", fused)
        model, cfg, sd = build_model(name, seed=21, device=DEV, dtype=torch.bfloat16)
        if hasattr(model.regularization, "sample"):
            model.regularization.sample = False
        ops.CONV_RECORD = []
        z, dec, log = model(x.to(DEV))
        rec, ops.CONV_RECORD = ops.CONV_RECORD, None
        n_fused = sum(1 for d, _, _ in rec if isinstance(d, ops.L.TBlockDesc))
        assert (n_fused > 0) == fused, (fused, n_fused)
        outs[fused] = (z, dec)
    ora = build_oracle(cfg, sd)
    ora.sample = False
    z2, dec2, _ = ora(x)
    for fused, (z, dec) in outs.items():
        ez, ed = rel_err(z, z2), rel_err(dec, dec2)
        print(f"{name} fused={fused}: z rel {ez:.3e} dec rel {ed:.3e}")
        assert ez < BF16_Z and ed < BF16_RECON


def test_encoder_tail_precision():
    """set_compute_dtype(bf16, encoder_tail=fp32, tail_level=k): the deep encoder levels, mid and conv_out in fp32 on a
    bf16 pass.  tail_level 0 = everything after conv_in in fp32; the latent gets closer to the fp32 kernels' as the tail
    grows, a tail in the pass's own type is the plain pass bit for bit, and the decoder is untouched."""
    name, shape = "vidtok_fsq_causal_488_32768", (1, 3, 9, 128, 128)
    model, cfg, sd = build_model(name, device=DEV, dtype=torch.float32)
    torch.manual_seed(11)
    x = torch.rand(shape, device=DEV) * 2 - 1
    ref = model._run_encoder(x)
    model.set_compute_dtype(torch.bfloat16)
    plain_bf16 = model._run_encoder(x)
    err = [rel_err(plain_bf16, ref)]
    n = model.encoder.num_resolutions
    for level in (n, n - 1, 0):
        model.set_compute_dtype(torch.bfloat16, encoder_tail=torch.float32, tail_level=level)
        assert model.decoder.compute_dtype == torch.bfloat16 and model.encoder.tail_level == level
        err.append(rel_err(model._run_encoder(x), ref))
    assert err[3] < err[2] < err[0] and err[1] < err[0], err
    model.set_compute_dtype(torch.bfloat16, encoder_tail=torch.bfloat16)
    assert torch.equal(model._run_encoder(x), plain_bf16)
    model.set_compute_dtype(torch.float32, encoder_tail=torch.bfloat16, tail_level=n)        # the other direction runs too
    assert rel_err(model._run_encoder(x), ref) < err[0]
    model.set_compute_dtype(torch.bfloat16)
    assert model.encoder.tail_dtype is None and torch.equal(model._run_encoder(x), plain_bf16)
    model.enable_graphs()
    model.set_compute_dtype(torch.bfloat16, encoder_tail=torch.float32)
    eager = None
    for _ in range(3):                                    # eager, captured, replayed
        got = model._run_encoder(x)
        eager = got if eager is None else eager
        assert torch.equal(got, eager)


# ---- the model handle of the C-ABI (vt_create / vt_load_weight / vt_encode / vt_regularize_* / vt_decode) -------------------
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32, X3, F16], ids=["bf16", "f32", "bf16x3", "f16"])
@pytest.mark.parametrize("name,shape", [("vidtok_kl_causal_488_4chn", (2, 3, 9, 64, 64)), ("vidtok_kl_causal_488_4chn", (1, 3, 17, 256, 256)),
                                        ("vidtok_fsq_causal_488_32768", (1, 3, 17, 128, 128)), ("vidtok_kl_causal_41616_4chn", (1, 3, 9, 128, 128)),
                                        ("vidtok_v1_1/vidtok_kl_causal_488_4chn_v1_1", (2, 3, 18, 64, 64)),      # front pad 2, trilinear
                                        ("vidtok_v1_1/vidtok_fsq_causal_888_32768_v1_1", (1, 3, 16, 128, 128)),  # three time up-samplers
                                        ("vidtok_v1_1/vidtok_kl_causal_288_8chn_v1_1", (1, 3, 33, 64, 64)),
                                        ("vidtok_v1_1/vidtok_kl_causal_488_4chn_v1_1:nearest", (1, 3, 12, 64, 64)),
                                        ("vidtok_kl_noncausal_488_4chn", (2, 3, 16, 64, 64)),                     # the non-causal family: centred windows
                                        ("vidtok_fsq_noncausal_41616_262144", (1, 3, 8, 128, 128))],
                         ids=["kl_488_small", "kl_488_full_size", "fsq_488", "kl_41616", "v11_kl_488", "v11_fsq_888", "v11_kl_288", "v11_nearest",
                              "noncausal_kl_488", "noncausal_fsq_41616"])
def test_model_handle_matches_engine(name, shape, dtype):
    """the handle-level C-ABI drives the stage graph from C++ (csrc/model.cpp).  Everything below goes
    through ctypes only -- create from the YAML's constructor arguments, load the reference state_dict key by key from
    host memory, encode, regularize, decode into caller buffers -- and must give the bits of the Python engine."""
    import ctypes as C

    from vidtok_amd import lib as L
    from vidtok_amd import ops

    if dtype == torch.float32 and shape[-1] >= 256:
        pytest.skip("fp32 at full size is covered by the bf16 case of the same graph and the small fp32 case")
    name, _, interp = name.partition(":")
    model, cfg, sd = build_model(name, device=DEV, dtype=dtype, overrides={"interpolation_mode": interp} if interp else None)
    _handle_vs_engine(model, cfg, sd, shape, dtype)


# the constructor arguments no shipped YAML sets: `norm_type: groupnorm` (model_3dcausal.py:30-34) in each family,
# FSQ with several codebooks and with project_in / project_out (regularizers.py:95-146) -- vt_model_config carries them
# (norm_type, fsq_num_codebooks, fsq_dim), and the handle must give the Python engine's bits for them as for the shipped YAMLs
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
@pytest.mark.parametrize("name,shape,ov,reg", [
    ("vidtok_kl_causal_488_4chn", (1, 3, 9, 64, 64), dict(norm_type="groupnorm"), None),
    ("vidtok_v1_1/vidtok_kl_causal_488_4chn_v1_1", (1, 3, 10, 64, 64), dict(norm_type="groupnorm"), None),
    ("vidtok_kl_noncausal_488_4chn", (1, 3, 8, 64, 64), dict(norm_type="groupnorm"), None),
    ("vidtok_fsq_causal_488_32768", (1, 3, 9, 64, 64), dict(z_channels=6), dict(levels=[8, 5, 5], num_codebooks=2)),
    ("vidtok_fsq_causal_488_32768", (2, 3, 5, 64, 64), dict(z_channels=8), dict(levels=[8, 5, 5], num_codebooks=2, dim=8)),
    ("vidtok_fsq_causal_488_32768", (1, 3, 9, 64, 64), dict(z_channels=7), dict(levels=[8, 5, 5], dim=7)),
], ids=["groupnorm_v10", "groupnorm_v11", "groupnorm_noncausal", "fsq_two_codebooks", "fsq_two_codebooks_projected", "fsq_projected"])
def test_model_handle_constructor_variants(name, shape, ov, reg, dtype):
    if reg is not None:
        reg = dict(reg, entropy_loss_weight=0.0, commitment_loss_weight=0.0)
    model, cfg, sd = build_model(name, seed=17, device=DEV, dtype=dtype, overrides=ov, reg_overrides=reg)
    _handle_vs_engine(model, cfg, sd, shape, dtype)


def _handle_vs_engine(model, cfg, sd, shape, dtype):
    import ctypes as C

    from vidtok_amd import lib as L
    from vidtok_amd import ops

    prm = cfg["model"]["params"]
    lib = L.load()
    h = C.c_void_p()
    mc = handle_config(L, prm["encoder_config"]["params"], prm["regularizer_config"]["target"], prm["regularizer_config"].get("params", {}),
                        prm["encoder_config"]["target"])
    v11 = mc.version == 1
    if v11:
        # one pass per clip: what AutoencodingEngineV11.encode / decode set up before running un-tiled
        assert not model.use_tiling
        for part in (model.encoder, model.decoder):
            model._empty_causal_cached(part)
        model._set_first_chunk(True)
        model._set_fused_temporal()
    L.check(lib.vt_create(C.byref(mc), {torch.bfloat16: L.VT_BF16, torch.float32: L.VT_F32, X3: L.VT_BF16X3, F16: L.VT_F16}[dtype], C.byref(h)), "vt_create")
    try:
        names = [lib.vt_weight_name(h, i).decode() for i in range(lib.vt_weight_count(h))]
        assert set(names) == {k for k in sd if not k.startswith("regularization") or ".project_" in k}, \
            "the handle reads exactly the encoder / decoder tensors of the reference state_dict (+ FSQ's projections)"
        for k in names:
            t = sd[k].detach().float().contiguous().cpu()
            shp = (C.c_int64 * t.dim())(*t.shape)
            L.check(lib.vt_load_weight(h, k.encode(), t.data_ptr(), shp, t.dim()), "vt_load_weight")
        B, _, T, H, W = shape
        torch.manual_seed(5)
        x = (torch.rand(shape, device=DEV) * 2 - 1).contiguous()
        nbytes = lib.vt_workspace_bytes(h, B, T, H, W)
        assert nbytes > 0
        ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
        ld = (C.c_int32 * 4)()
        L.check(lib.vt_latent_dims(h, T, H, W, ld), "vt_latent_dims")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        # encoder
        ref_h = model._run_encoder(x)
        assert tuple(ref_h.shape) == (B, ld[0], ld[1], ld[2], ld[3])
        got_h = torch.empty_like(ref_h)
        L.check(lib.vt_encode(h, x.data_ptr(), B, T, H, W, got_h.data_ptr(), ws.data_ptr(), nbytes, st), "vt_encode")
        torch.cuda.synchronize()
        assert torch.equal(got_h, ref_h), f"encoder: {rel_err(got_h, ref_h):.3e}"
        # regularizer (KL: the mode, no noise; FSQ: codes and indices)
        zc = mc.z_channels
        z = torch.empty((B, zc, ld[1], ld[2], ld[3]), dtype=torch.float32, device=DEV)
        if mc.regularizer == 0:
            kl = torch.zeros(1, dtype=torch.float32, device=DEV)
            L.check(lib.vt_regularize_kl(h, got_h.data_ptr(), None, z.data_ptr(), kl.data_ptr(), B, ld[1], ld[2], ld[3], st), "vt_regularize_kl")
            ref_z, ref_kl = ops.kl_sample(ref_h.contiguous(), None)
            assert torch.equal(z, ref_z) and torch.equal(kl.reshape(()), ref_kl.reshape(()))
        else:
            ncb = max(1, mc.fsq_num_codebooks)
            idx = torch.empty((B, ld[1], ld[2], ld[3]) + ((ncb,) if ncb > 1 else ()), dtype=torch.int32, device=DEV)
            L.check(lib.vt_regularize_fsq(h, got_h.data_ptr(), z.data_ptr(), idx.data_ptr(), B, ld[1], ld[2], ld[3], st), "vt_regularize_fsq")
            ref_z, ref_log = model.regularization(ref_h)
            assert ref_log["indices"].numel() == idx.numel()
            assert torch.equal(z, ref_z) and torch.equal(idx, ref_log["indices"].to(torch.int32).reshape(idx.shape))
            z2 = torch.empty_like(z)          # decode(indices, decode_from_indices=True) starts here
            L.check(lib.vt_indices_to_latent(h, idx.data_ptr(), z2.data_ptr(), B, ld[1], ld[2], ld[3], st), "vt_indices_to_latent")
            assert torch.equal(z2, model.indices_to_latent(ref_log["indices"])) and torch.equal(z2, z)
        # decoder
        ref_x = model._run_decoder(z)
        got_x = torch.empty_like(ref_x)
        L.check(lib.vt_decode(h, z.data_ptr(), B, ld[1], ld[2], ld[3], got_x.data_ptr(), ws.data_ptr(), nbytes, st), "vt_decode")
        torch.cuda.synchronize()
        assert torch.equal(got_x, ref_x), f"decoder: {rel_err(got_x, ref_x):.3e}"
        # a second pass over the same workspace gives the same bits (nothing stale between calls)
        L.check(lib.vt_encode(h, x.data_ptr(), B, T, H, W, got_h.data_ptr(), ws.data_ptr(), nbytes, st), "vt_encode")
        torch.cuda.synchronize()
        assert torch.equal(got_h, ref_h)
        assert lib.vt_reset_cache(h) == 0
        # a workspace that is too small is refused, not overrun
        assert lib.vt_encode(h, x.data_ptr(), B, T, H, W, got_h.data_ptr(), ws.data_ptr(), 4096, st) != 0 and b"workspace" in lib.vt_last_error()
    finally:
        lib.vt_destroy(h)


def _make_handle(L, lib, cfg, sd, dtype):
    """vt_create from the YAML's constructor arguments + every encoder / decoder tensor of the reference state_dict, through ctypes"""
    import ctypes as C

    prm = cfg["model"]["params"]
    mc = handle_config(L, prm["encoder_config"]["params"], prm["regularizer_config"]["target"], prm["regularizer_config"].get("params", {}),
                        prm["encoder_config"]["target"])
    h = C.c_void_p()
    L.check(lib.vt_create(C.byref(mc), {torch.bfloat16: L.VT_BF16, torch.float32: L.VT_F32, X3: L.VT_BF16X3, F16: L.VT_F16}[dtype], C.byref(h)), "vt_create")
    for i in range(lib.vt_weight_count(h)):
        k = lib.vt_weight_name(h, i).decode()
        t = sd[k].detach().float().contiguous().cpu()
        L.check(lib.vt_load_weight(h, k.encode(), t.data_ptr(), (C.c_int64 * t.dim())(*t.shape), t.dim()), "vt_load_weight")
    L.check(lib.vt_prepare(h), "vt_prepare")             # every weight packed + uploaded now: the calls below never block on one
    return h, mc


@pytest.mark.parametrize("name,shape,tc,overlap,dtype", [
    ("vidtok_v1_1/vidtok_kl_causal_488_16chn_v1_1", (1, 3, 41, 64, 64), 16, True, torch.bfloat16),      # BASELINE configs[4]'s protocol, ragged last chunk
    ("vidtok_v1_1/vidtok_kl_causal_488_16chn_v1_1", (2, 3, 41, 64, 64), 16, True, torch.float32),
    ("vidtok_v1_1/vidtok_kl_causal_488_16chn_v1_1", (1, 3, 37, 64, 64), 8, False, X3),
    ("vidtok_v1_1/vidtok_fsq_causal_888_32768_v1_1", (1, 3, 33, 64, 64), 16, True, torch.bfloat16),      # f = 8: three time up-samplers, offsets 1 / 2 / 4 / 8
    ("vidtok_v1_1/vidtok_kl_causal_288_8chn_v1_1", (1, 3, 21, 64, 64), 8, True, torch.bfloat16),         # f = 2
    ("vidtok_v1_1/vidtok_kl_causal_488_4chn_v1_1:nearest", (1, 3, 25, 64, 64), 8, True, torch.bfloat16),
    ("vidtok_v1_1/vidtok_kl_causal_488_16chn_v1_1", (1, 3, 129, 256, 256), 16, True, torch.bfloat16),    # BASELINE configs[4] itself
    ("vidtok_v1_1/vidtok_kl_causal_488_16chn_v1_1", (1, 3, 41, 64, 64), 16, True, F16),
], ids=["kl16_t41", "kl16_t41_f32_b2", "kl16_t37_x3_no_overlap", "fsq888_t33", "kl288_t21", "kl488_nearest_t25", "configs4_129x256x256", "kl16_t41_f16"])
def test_model_handle_tiled_matches_engine(name, shape, tc, overlap, dtype):
    """The v1.1 temporal tiling driven from C++ (vt_tile_encode / vt_tile_decode: chunk schedule, per-module causal caches owned
    by the handle, look-ahead decode with the doubling cache offsets) against the Python engine's tiled pass: same bits.
    ctypes only.  VERDICT r3 item 8: a C host can now run BASELINE.json configs[4]."""
    import ctypes as C

    from vidtok_amd import lib as L
    from vidtok_amd import ops

    name, _, interp = name.partition(":")
    model, cfg, sd = build_model(name, device=DEV, dtype=dtype, overrides={"interpolation_mode": interp} if interp else None)
    f = model.encoder.time_downsample_factor
    model.use_tiling, model.t_chunk_enc, model.t_chunk_dec, model.use_overlap = True, tc, tc // f, overlap
    if hasattr(model.regularization, "sample"):
        model.regularization.sample = False
    lib = L.load()
    h, mc = _make_handle(L, lib, cfg, sd, dtype)
    try:
        B, _, T, H, W = shape
        torch.manual_seed(6)
        x = (torch.rand(shape, device=DEV) * 2 - 1).contiguous()
        z_ref, log_ref = model.encode(x, return_reg_log=True)
        dec_ref = model.decode(z_ref)
        tz = lib.vt_tile_latent_frames(h, T, tc)
        assert tz == z_ref.shape[2]
        nbytes = lib.vt_tile_workspace_bytes(h, B, T, H, W, tc, int(overlap))
        assert nbytes > 0, lib.vt_last_error()
        ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        ld = (C.c_int32 * 4)()
        L.check(lib.vt_latent_dims(h, T, H, W, ld), "vt_latent_dims")
        got_h = torch.empty((B, ld[0], tz, ld[2], ld[3]), dtype=torch.float32, device=DEV)
        for _ in range(2):          # twice: the second pass starts from the caches the first one left (reset inside)
            L.check(lib.vt_tile_encode(h, x.data_ptr(), B, T, H, W, tc, got_h.data_ptr(), ws.data_ptr(), nbytes, st), "vt_tile_encode")
        z = torch.empty((B, mc.z_channels, tz, ld[2], ld[3]), dtype=torch.float32, device=DEV)
        if mc.regularizer == 0:
            kl = torch.zeros(1, dtype=torch.float32, device=DEV)
            L.check(lib.vt_regularize_kl(h, got_h.data_ptr(), None, z.data_ptr(), kl.data_ptr(), B, tz, ld[2], ld[3], st), "vt_regularize_kl")
        else:
            idx = torch.empty((B, tz, ld[2], ld[3]), dtype=torch.int32, device=DEV)
            L.check(lib.vt_regularize_fsq(h, got_h.data_ptr(), z.data_ptr(), idx.data_ptr(), B, tz, ld[2], ld[3], st), "vt_regularize_fsq")
            assert torch.equal(idx, log_ref["indices"].to(torch.int32).reshape(idx.shape))
            # the auxiliary-loss statistics of the whole latent = the operator's on the same tensor
            arr = (C.c_int32 * mc.n_levels)(*[mc.levels[i] for i in range(mc.n_levels)])
            work = torch.empty(lib.vt_fsq_aux_work_floats(arr, mc.n_levels, B, tz * ld[2] * ld[3]), dtype=torch.float32, device=DEV)
            out3 = torch.empty(3, dtype=torch.float32, device=DEV)
            L.check(lib.vt_regularize_fsq_aux(h, got_h.data_ptr(), B, tz, ld[2], ld[3], 100.0, work.data_ptr(), out3.data_ptr(), st), "vt_regularize_fsq_aux")
            ref3 = ops.fsq_aux_stats(got_h, [mc.levels[i] for i in range(mc.n_levels)], 100.0)
            assert torch.allclose(out3, ref3, rtol=1e-5, atol=1e-7), (out3, ref3)
        torch.cuda.synchronize()
        assert torch.equal(z, z_ref), f"tiled encode: {rel_err(z, z_ref):.3e}"
        got_x = torch.empty_like(dec_ref)
        assert tuple(dec_ref.shape) == (B, mc.out_ch, tz * f, H, W)
        L.check(lib.vt_tile_decode(h, z.data_ptr(), B, tz, ld[2], ld[3], tc // f, int(overlap), got_x.data_ptr(), ws.data_ptr(), nbytes, st), "vt_tile_decode")
        torch.cuda.synchronize()
        assert torch.equal(got_x, dec_ref), f"tiled decode: {rel_err(got_x, dec_ref):.3e}"
        # the one-pass entry points still work on the same handle afterwards, and the cache buffers can be given back
        assert lib.vt_reset_cache(h) == 0
        assert lib.vt_tile_decode(h, z.data_ptr(), B, tz, ld[2], ld[3], tc // f, int(overlap), got_x.data_ptr(), ws.data_ptr(), 4096, st) != 0
    finally:
        lib.vt_destroy(h)


def test_model_handle_tiling_refused_for_v10():
    import ctypes as C

    from vidtok_amd import lib as L

    model, cfg, sd = build_model("vidtok_kl_causal_488_4chn", device="cpu")
    lib = L.load()
    h, _ = _make_handle(L, lib, cfg, sd, torch.bfloat16)
    try:
        assert lib.vt_tile_workspace_bytes(h, 1, 17, 64, 64, 16, 1) < 0 and b"v1.1" in lib.vt_last_error()
    finally:
        lib.vt_destroy(h)


def test_autocast_region_selects_the_kernels():
    """The reference's README runs `model(x)` under torch.autocast (README.md:336-340,375-385).  The engine follows the caller's
    context: autocast(bfloat16) / autocast(float16) = the bf16 / fp16 kernels for that call (bit-identical to
    set_compute_dtype(that dtype)), the chosen mode returns afterwards; set_autocast_policy can map a region elsewhere; captured
    graphs of both modes survive the switching (no recapture on every transition)."""
    name, shape = "vidtok_kl_causal_488_4chn", (1, 3, 9, 64, 64)
    model, cfg, sd = build_model(name, seed=31, device=DEV, dtype=torch.float32)
    model.regularization.sample = False
    x = (torch.rand(shape, generator=torch.Generator().manual_seed(3)) * 2 - 1).to(DEV)
    z32, dec32, _ = model(x)
    outs = {}
    for adt, arith in ((torch.bfloat16, "bf16"), (torch.float16, "fp16")):
        with torch.no_grad(), torch.autocast(device_type="cuda", dtype=adt):
            za, deca, _ = model(x)
            assert model.arith == arith
            zb = model.encode(x)                                   # every entry point reads the context
            assert torch.equal(zb, za)
        z32b, dec32b, _ = model(x)                                 # the region ended: fp32 kernels again, same bits as before
        assert model.arith == "fp32" and torch.equal(dec32b, dec32) and torch.equal(z32b, z32)
        model.set_compute_dtype(adt)
        z16, dec16, _ = model(x)
        assert torch.equal(deca, dec16) and torch.equal(za, z16)
        assert not torch.equal(dec16, dec32) and rel_err(dec16, dec32) < RECON[adt]
        outs[arith] = dec16
        model.set_compute_dtype(torch.float32)
    assert not torch.equal(outs["bf16"], outs["fp16"])              # two arithmetics did run
    with torch.autocast(device_type="cuda"):                       # scripts/inference_*.py --precision autocast: torch's default dtype = float16
        model(x)
        assert model.arith == "fp16"
    model.set_autocast_policy(float16="error")
    with torch.autocast(device_type="cuda", dtype=torch.float16):
        with pytest.raises(NotImplementedError, match="float16"):
            model(x)
    model.set_autocast_policy(float16="bf16x3")
    with torch.autocast(device_type="cuda", dtype=torch.float16):
        _, decx, _ = model(x)
        assert model.arith == "bf16x3"
    assert rel_err(decx, dec32) < 1e-3 and model(x) is not None and model.arith == "fp32"
    # graphs: a caller alternating plain and autocast calls replays both captured sets
    model.set_autocast_policy(float16="fp16").enable_graphs()
    seen = []
    for rnd in range(4):
        _, d_plain, _ = model(x)
        with torch.autocast(device_type="cuda", dtype=torch.float16):
            _, d_auto, _ = model(x)
        seen.append((d_plain, d_auto))
        assert torch.equal(d_plain, dec32) and torch.equal(d_auto, outs["fp16"]), rnd
    assert sum(1 for e in model._gdec.entries.values() if isinstance(e, tuple)) == 2      # one captured graph per mode, neither dropped
    # an fp32 encoder tail chosen with the mode stays in force inside the region
    model.enable_graphs(False).set_compute_dtype(torch.bfloat16, encoder_tail=torch.float32)
    with torch.autocast(device_type="cuda", dtype=torch.float16):
        model(x)
        assert model.arith == "fp16" and model.encoder.tail_dtype == torch.float32


def test_reference_readme_snippet_runs_unmodified():
    """The "Easy Usage" snippet of the reference (README.md:324-341) with only the config's `target:` lines naming this package (the
    shipped configs/*.yaml): load_model_from_config, .to('cuda').eval(), a random (1, 3, 17, 256, 256) clip, `model(x_input)` under
    torch.autocast(device_type='cuda', dtype=torch.float16) -- and the timing loop of README.md:375-385.  The result is gated against
    the fp32 kernels' on the same clip."""
    import vidtok_amd
    from util import config_path, seeded_state_dict

    cfg_path = config_path("vidtok_kl_causal_488_4chn")
    model = vidtok_amd.load_model_from_config(vidtok_amd.load_config(cfg_path), verbose=False)
    model.load_state_dict(seeded_state_dict({k: v.shape for k, v in model.state_dict().items()}, 77))
    model.to('cuda').eval()
    num_frames = 17 if model.is_causal else 16
    torch.manual_seed(0)
    x_input = (torch.rand(1, 3, num_frames, 256, 256) * 2 - 1).to('cuda')
    with torch.no_grad(), torch.autocast(device_type='cuda', dtype=torch.float16):
        _, x_recon, _ = model(x_input)
    assert x_input.shape == x_recon.shape and x_recon.dtype == torch.float32 and torch.isfinite(x_recon).all()
    with torch.no_grad(), torch.autocast(device_type='cuda', dtype=torch.float16):
        for i in range(3):
            _, x_recon, _ = model(x_input)
    torch.cuda.synchronize()
    model.regularization.sample = False
    with torch.no_grad(), torch.autocast(device_type='cuda', dtype=torch.float16):
        _, r16, _ = model(x_input)
    _, r32, _ = model(x_input)                                  # outside the region: the construction default, fp32 kernels
    assert model.arith == "fp32" and rel_err(r16, r32) < RECON[F16]
