"""Pins oracle/vidtok_oracle.py against the UNMODIFIED reference (build container only: needs
/root/reference; on the GPU box the same comparison is replayed from tests/golden/)."""
import pytest
import torch

from oracle.refload import load_reference_model, randomize_weights
from oracle.vidtok_oracle import OracleEngine
from util import rel_err

pytestmark = pytest.mark.reference

CASES = [
    ("vidtok_kl_causal_488_4chn", (1, 3, 9, 32, 32)),
    ("vidtok_fsq_causal_488_32768", (2, 3, 5, 32, 32)),
    ("vidtok_kl_causal_488_16chn", (1, 3, 8, 32, 32)),
    ("vidtok_kl_causal_288_8chn", (1, 3, 5, 32, 32)),
    ("vidtok_kl_causal_444_4chn", (1, 3, 5, 16, 16)),
    ("vidtok_kl_causal_41616_4chn", (1, 3, 5, 32, 32)),
    # v1.1 schedules beyond 4x8x8 (un-tiled; the tiled protocol has its own tests below)
    ("vidtok_v1_1/vidtok_fsq_causal_888_32768_v1_1", (1, 3, 9, 32, 32)),
    ("vidtok_v1_1/vidtok_kl_causal_41616_16chn_v1_1", (1, 3, 5, 32, 32)),
    ("vidtok_v1_1/vidtok_kl_causal_288_8chn_v1_1", (1, 3, 5, 32, 32)),
    # non-causal family: T a multiple of the temporal factor
    ("vidtok_kl_noncausal_488_4chn", (1, 3, 8, 32, 32)),
    ("vidtok_fsq_noncausal_488_262144", (2, 3, 4, 32, 32)),
    ("vidtok_kl_noncausal_41616_16chn", (1, 3, 8, 32, 32)),
]


@pytest.mark.parametrize("cfg,shape", CASES)
def test_oracle_matches_reference_forward(cfg, shape):
    ref, c = load_reference_model(cfg)
    randomize_weights(ref)
    ora = OracleEngine(c["model"]["params"], ref.state_dict())
    x = torch.rand(*shape) * 2 - 1
    with torch.no_grad():
        torch.manual_seed(7)
        z, dec, log = ref(x)
        torch.manual_seed(7)
        z2, dec2, log2 = ora(x)
    assert dec.shape == dec2.shape and z.shape == z2.shape
    assert rel_err(z2, z) < 2e-5 and rel_err(dec2, dec) < 5e-5
    if "indices" in log:
        assert torch.equal(log["indices"], log2["indices"])
        assert abs(float(log["aux_loss"]) - float(log2["aux_loss"])) < 1e-4
        # decode_from_indices identity (SURVEY.md section 8c free KAT)
        back = ora.decode(log2["indices"], decode_from_indices=True)
        assert rel_err(back[:, :, -dec.shape[2]:], dec) < 5e-5         # (v1.1 forward trims the front padding, decode does not)
    else:
        assert abs(float(log["kl_loss"]) - float(log2["kl_loss"])) < 1e-3 * abs(float(log["kl_loss"]))


@pytest.mark.parametrize("overlap", [False, True])
def test_oracle_matches_reference_v11_tiled(overlap):
    cfg = "vidtok_v1_1/vidtok_kl_causal_488_16chn_v1_1"
    ref, c = load_reference_model(cfg)
    randomize_weights(ref)
    ora = OracleEngine(c["model"]["params"], ref.state_dict())
    x = torch.rand(1, 3, 41, 32, 32) * 2 - 1
    for m in (ref, ora):
        m.use_tiling, m.t_chunk_enc, m.use_overlap = True, 16, overlap
    ref.t_chunk_dec = 4
    with torch.no_grad():
        torch.manual_seed(3)
        z, dec, log = ref(x)
        torch.manual_seed(3)
        z2, dec2, log2 = ora(x)
    assert dec.shape == dec2.shape == x.shape
    assert rel_err(z2, z) < 2e-5 and rel_err(dec2, dec) < 5e-5


def test_oracle_matches_reference_v11_fsq_untiled():
    cfg = "vidtok_v1_1/vidtok_fsq_causal_488_32768_v1_1"
    ref, c = load_reference_model(cfg)
    randomize_weights(ref)
    ora = OracleEngine(c["model"]["params"], ref.state_dict())
    x = torch.rand(1, 3, 17, 32, 32) * 2 - 1
    with torch.no_grad():
        z, dec, log = ref(x)
        z2, dec2, log2 = ora(x)
    assert torch.equal(log["indices"], log2["indices"])
    assert rel_err(dec2, dec) < 5e-5


FSQ_PROJ = dict(overrides=dict(z_channels=8), reg_overrides=dict(dim=8, levels=[8, 8, 8, 5, 5, 5]))


def test_oracle_matches_reference_fsq_with_projections():
    """dim != len(levels): project_in / project_out around the quantiser (no shipped YAML sets it)"""
    ref, c = load_reference_model("vidtok_fsq_causal_488_32768", **FSQ_PROJ)
    randomize_weights(ref)
    assert ref.regularization.has_projections
    ora = OracleEngine(c["model"]["params"], ref.state_dict())
    x = torch.rand(1, 3, 5, 32, 32) * 2 - 1
    with torch.no_grad():
        z, dec, log = ref(x)
        z2, dec2, log2 = ora(x)
    assert z.shape == z2.shape == (1, 8, 2, 4, 4)
    assert rel_err(z2, z) < 2e-5 and rel_err(dec2, dec) < 5e-5
    assert torch.equal(log["indices"], log2["indices"])
    assert rel_err(ora.decode(log2["indices"], decode_from_indices=True), dec) < 5e-5


# with the codebook axis kept the reference's own forward raises as soon as an aux-loss weight is non-zero
# (its implicit codebook is flattened to 1-D, regularizers.py:143-146,191-192,234): these options exist without aux loss
NO_AUX = dict(entropy_loss_weight=0.0, commitment_loss_weight=0.0)


@pytest.mark.parametrize("zc,reg", [(6, dict(levels=[8, 5, 5], num_codebooks=2, **NO_AUX)),
                                    (8, dict(levels=[8, 5, 5], num_codebooks=2, dim=8, **NO_AUX)),
                                    (3, dict(levels=[8, 5, 5], keep_num_codebooks_dim=True, **NO_AUX))],
                         ids=["two_codebooks", "two_codebooks_projected", "one_codebook_kept_axis"])
def test_oracle_matches_reference_fsq_num_codebooks(zc, reg):
    """num_codebooks > 1 / keep_num_codebooks_dim (regularizers.py:100-150,227,247-262): no shipped YAML sets them
    (VERDICT r1 "unused reference branches"); indices carry a trailing codebook axis, the entropy terms are per codebook."""
    ref, c = load_reference_model("vidtok_fsq_causal_488_32768", overrides=dict(z_channels=zc), reg_overrides=reg)
    randomize_weights(ref)
    ora = OracleEngine(c["model"]["params"], ref.state_dict())
    x = torch.rand(1, 3, 5, 32, 32) * 2 - 1
    with torch.no_grad():
        z, dec, log = ref(x)
        z2, dec2, log2 = ora(x)
    assert log["indices"].shape == log2["indices"].shape == (1, 2, 4, 4, reg.get("num_codebooks", 1))
    assert torch.equal(log["indices"], log2["indices"])
    assert z.shape == z2.shape == (1, zc, 2, 4, 4) and rel_err(z2, z) < 2e-5 and rel_err(dec2, dec) < 5e-5
    assert abs(float(log["aux_loss"]) - float(log2["aux_loss"])) < 2e-5 * max(1.0, abs(float(log["aux_loss"])))
    # (the reference ENGINE's indices_to_latent only takes index maps without the codebook axis, autoencoder.py:204-213;
    #  its regularizer's indices_to_codes is the defined inverse)
    with torch.no_grad():
        back = ref.decoder(ref.regularization.indices_to_codes(log["indices"]))
    assert rel_err(ora.decode(log2["indices"], decode_from_indices=True), back) < 5e-5


def _perturb_norm_affines(model, seed=3):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "norm" in n and p.dim() == 1:
                p.copy_((1.0 if n.endswith("weight") else 0.0) + 0.1 * torch.randn(p.shape, generator=g))


@pytest.mark.parametrize("cfg,shape,tiled", [
    ("vidtok_kl_causal_488_4chn", (1, 3, 5, 32, 32), False),
    ("vidtok_fsq_causal_488_32768", (1, 3, 5, 32, 32), False),
    ("vidtok_v1_1/vidtok_kl_causal_488_4chn_v1_1", (1, 3, 25, 32, 32), True),
    ("vidtok_kl_noncausal_488_4chn", (1, 3, 8, 32, 32), False),
])
def test_oracle_matches_reference_groupnorm(cfg, shape, tiled):
    """`norm_type: groupnorm` (no shipped YAML): torch.nn.GroupNorm(32) whose statistics follow the view of each call
    site -- frames, single positions, pixels over time or whole clips (oracle norm_c)."""
    ref, c = load_reference_model(cfg, overrides=dict(norm_type="groupnorm"))
    randomize_weights(ref)
    _perturb_norm_affines(ref)
    assert "encoder.down.0.block.0.norm1.weight" in ref.state_dict()
    ora = OracleEngine(c["model"]["params"], ref.state_dict())
    if tiled:
        for m in (ref, ora):
            m.use_tiling, m.t_chunk_enc, m.use_overlap = True, 8, True
        ref.t_chunk_dec = 2
    x = torch.rand(*shape) * 2 - 1
    with torch.no_grad():
        torch.manual_seed(7)
        z, dec, log = ref(x)
        torch.manual_seed(7)
        z2, dec2, log2 = ora(x)
    assert rel_err(z2, z) < 2e-5 and rel_err(dec2, dec) < 1e-4


RARE = [  # constructor options no shipped YAML sets (VERDICT r1 "unused reference branches"); (overrides, input frames)
    (dict(resamp_with_conv=False), 5),
    (dict(init_pad_mode="constant"), 6),
    (dict(init_pad_mode="reflect"), 6),
    (dict(tanh_out=True), 5),
    (dict(give_pre_end=True), 5),
]


@pytest.mark.parametrize("ov,T", RARE, ids=[next(iter(o)) + "=" + str(next(iter(o.values()))) for o, _ in RARE])
def test_oracle_matches_reference_rare_options(ov, T):
    """Upsample / Downsample(with_conv=False) (model_3dcausal.py:200-230), init_pad_mode constant / reflect (:37-43,680,688),
    tanh_out / give_pre_end (:862-869)."""
    ref, c = load_reference_model("vidtok_kl_causal_488_4chn", overrides=ov)
    randomize_weights(ref)
    if "resamp_with_conv" in ov:
        assert not any(".downsample.conv." in k and "down_temporal" not in k for k in ref.state_dict())
    ora = OracleEngine(c["model"]["params"], ref.state_dict())
    x = torch.rand(1, 3, T, 32, 32) * 2 - 1
    with torch.no_grad():
        torch.manual_seed(7)
        z = ref.encode(x)
        dec = ref.decoder(z)
        torch.manual_seed(7)
        z2, _ = ora.encode(x)
        dec2 = ora.decode(z2)
    assert dec.shape == dec2.shape and (dec.shape[1] == (128 if "give_pre_end" in ov else 3))
    assert rel_err(z2, z) < 2e-5 and rel_err(dec2, dec) < 5e-5
