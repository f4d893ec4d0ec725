"""CPU tests of the host side (vidtok_amd.modules / engine / config / packing): the HIP operators are
replaced, for these tests only, by the torch statements of their contracts (tests/torch_ops_ref.py), so
padding bookkeeping, weight packing, causal caches and tiling are checked end-to-end against the
oracle and the reference goldens without a GPU."""
import os

import pytest
import torch
from safetensors.torch import load_file

import torch_ops_ref
from golden_cases import CASES, apply_tiling, make_input
from util import GOLDEN_DIR, build_model, build_oracle, config_path, rel_err


@pytest.fixture()
def emulated_ops(monkeypatch):
    torch_ops_ref.patch_ops(monkeypatch)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_host_graph_matches_reference_golden(case, emulated_ops):
    gold = load_file(os.path.join(GOLDEN_DIR, case["name"] + ".safetensors"))
    model, cfg, sd = build_model(case["config"], seed=case["weight_seed"])
    apply_tiling(model, case)
    torch.manual_seed(case["noise_seed"])
    z, dec, log = model(make_input(case))
    assert dec.shape == gold["dec"].shape
    assert rel_err(z, gold["z"]) < 2e-5 and rel_err(dec, gold["dec"]) < 5e-5
    if "indices" in gold:
        assert torch.equal(log["indices"], gold["indices"])
        assert abs(float(log["aux_loss"]) - float(gold["aux_loss"])) < 1e-4
        dec2 = model.decode(log["indices"], decode_from_indices=True)
        assert rel_err(dec2[:, :, -gold["dec"].shape[2]:], gold["dec"]) < 5e-5   # forward() keeps the last T frames
    else:
        assert abs(float(log["kl_loss"]) - float(gold["kl_loss"])) < 1e-4 * abs(float(gold["kl_loss"]))


def test_encoder_tail_precision_plumbing(emulated_ops):
    """set_compute_dtype(bf16, encoder_tail=fp32, tail_level=k) (host logic with the torch stand-ins for the operators): the
    stages from level k on run in fp32 -- their inputs arrive converted, un-normalised, and every one of them is handed
    fp32; with the whole encoder behind conv_in in the tail the latent is much closer to the fp32 pass's; tail = the pass's own
    type is the plain pass."""
    import vidtok_amd.ops as ops

    model, cfg, sd = build_model("vidtok_kl_causal_488_4chn", seed=4)
    x = torch.rand(1, 3, 5, 32, 32) * 2 - 1
    ref = model.encoder(x)
    model.set_compute_dtype(torch.bfloat16)
    plain = model.encoder(x)
    seen = []
    conv = ops.conv
    ops.conv = lambda xx, *a, **k: (seen.append(xx.dtype), conv(xx, *a, **k))[1]      # (monkeypatched module attribute: restored below)
    try:
        n = model.encoder.num_resolutions
        errs = []
        for level in (n, 2, 0):
            seen.clear()
            model.set_compute_dtype(torch.bfloat16, encoder_tail=torch.float32, tail_level=level)
            errs.append(rel_err(model.encoder(x), ref))
            first32 = seen.index(torch.float32)
            assert seen[0] == torch.bfloat16 and all(d == torch.float32 for d in seen[first32:]) and all(d == torch.bfloat16 for d in seen[:first32])
        assert errs[2] < 0.5 * rel_err(plain, ref), errs          # everything after conv_in in fp32 (a short tail on a 4-frame clip is within the noise)
    finally:
        ops.conv = conv
    model.set_compute_dtype(torch.bfloat16, encoder_tail=torch.bfloat16)
    assert torch.equal(model.encoder(x), plain)
    model.set_compute_dtype(torch.bfloat16)
    assert model.encoder.tail_dtype is None and model.encoder.tail_level is None


@pytest.mark.parametrize("name,shape", [
    ("vidtok_kl_causal_288_8chn", (1, 3, 5, 32, 32)),
    ("vidtok_kl_causal_444_4chn", (1, 3, 5, 16, 16)),
    ("vidtok_kl_causal_41616_4chn", (1, 3, 5, 32, 32)),
    ("vidtok_fsq_causal_488_4096", (1, 3, 4, 32, 32)),
    ("vidtok_v1_1/vidtok_fsq_causal_888_32768_v1_1", (1, 3, 9, 32, 32)),
    ("vidtok_kl_noncausal_41616_4chn", (1, 3, 8, 32, 32)),
    ("vidtok_kl_noncausal_488_16chn", (2, 3, 4, 32, 32)),
    ("vidtok_fsq_noncausal_41616_262144", (1, 3, 8, 32, 32)),
])
def test_other_causal_configs_match_oracle(name, shape, emulated_ops):
    model, cfg, sd = build_model(name, seed=3)
    ora = build_oracle(cfg, sd)
    x = torch.rand(*shape) * 2 - 1
    torch.manual_seed(1)
    z, dec, log = model(x)
    torch.manual_seed(1)
    z2, dec2, log2 = ora(x)
    assert dec.shape == dec2.shape
    assert rel_err(z, z2) < 2e-5 and rel_err(dec, dec2) < 5e-5


def test_split_bf16_weight_planes_and_mode_plumbing(emulated_ops):
    """"bf16x3": pack_split3's plane layout ([hi 16 | lo 16] bf16 per 16 k, K padded to 32, hi + lo within 2^-16 of w), and
    set_compute_dtype("bf16x3") reaching every convolution's PackedCache (not the attention's W_v row operand) with fp32
    storage -- the host graph in that mode stays within the fp32 tolerance of the oracle; switching back restores fp32 rows"""
    from vidtok_amd.packing import PackedCache, pack_split3

    w = torch.randn(5, 40)
    p = pack_split3(w)
    assert p.dtype == torch.int32 and p.shape == (5, 64) and p[1:3].contiguous().dtype == torch.int32     # the tag survives views
    planes = p.view(torch.bfloat16).reshape(5, 4, 2, 16)
    hi, lo = planes[:, :, 0].reshape(5, 64).float(), planes[:, :, 1].reshape(5, 64).float()
    assert torch.equal(hi[:, :40], w.to(torch.bfloat16).float()) and torch.equal(lo[:, :40], (w - hi[:, :40]).to(torch.bfloat16).float())
    assert hi[:, 40:].abs().sum() == 0 and lo[:, 40:].abs().sum() == 0
    assert ((hi + lo)[:, :40] - w).abs().max() <= w.abs().max() * 2.0 ** -16

    for name, shape in (("vidtok_fsq_causal_488_32768", (1, 3, 5, 32, 32)), ("vidtok_kl_noncausal_488_4chn", (1, 3, 8, 32, 32)),
                        ("vidtok_v1_1/vidtok_kl_causal_488_16chn_v1_1", (1, 3, 9, 32, 32))):
        model, cfg, sd = build_model(name, seed=3, dtype="bf16x3")
        assert model.arith == "bf16x3" and model.encoder.compute_dtype == torch.float32
        caches = [v for m in model.modules() for v in m.__dict__.values() if isinstance(v, PackedCache)]
        assert caches and all((c.arith == "bf16x3") != c.pin_native for c in caches) and any(c.pin_native for c in caches)
        ora = build_oracle(cfg, sd)
        x = torch.rand(*shape) * 2 - 1
        torch.manual_seed(1)
        z, dec, log = model(x)
        torch.manual_seed(1)
        z2, dec2, log2 = ora(x)
        ez, ed = rel_err(z, z2), rel_err(dec, dec2)
        assert ez < 1e-3 and ed < 1e-3 and (ez > 0 or "indices" in log2), (name, ez, ed)      # not bit-equal to fp32: the planes were used
        if "indices" in log2:
            assert torch.equal(log["indices"], log2["indices"])
        model.set_compute_dtype(torch.float32)
        assert all(c.arith is None for c in caches) and model.arith == "fp32"
        torch.manual_seed(1)
        z3, dec3, _ = model(x)
        assert rel_err(dec3, dec2) < 5e-5


@pytest.mark.parametrize("name,shape", [
    ("vidtok_kl_causal_488_4chn", (1, 3, 5, 32, 32)),
    ("vidtok_v1_1/vidtok_kl_causal_488_16chn_v1_1", (1, 3, 9, 32, 32)),
    ("vidtok_kl_noncausal_488_4chn", (1, 3, 8, 32, 32)),
])
def test_groupnorm_variants_match_oracle(name, shape, emulated_ops):
    """norm_type: groupnorm -- reference key names (no `.norm` level), site-dependent statistics, no epilogue fusion"""
    model, cfg, sd = build_model(name, seed=6, overrides=dict(norm_type="groupnorm"))
    assert "encoder.down.0.block.0.norm1.weight" in sd and "encoder.down.0.block.0.norm1.norm.weight" not in sd
    ora = build_oracle(cfg, sd)
    x = torch.rand(*shape) * 2 - 1
    torch.manual_seed(1)
    z, dec, log = model(x)
    torch.manual_seed(1)
    z2, dec2, log2 = ora(x)
    assert dec.shape == dec2.shape and rel_err(z, z2) < 2e-5 and rel_err(dec, dec2) < 1e-4


def test_fsq_with_projections_matches_oracle(emulated_ops):
    """dim != len(levels): nn.Linear project_in / project_out (reference parameter names) around the quantiser"""
    model, cfg, sd = build_model("vidtok_fsq_causal_488_32768", seed=5, overrides=dict(z_channels=8),
                                 reg_overrides=dict(dim=8, levels=[8, 8, 8, 5, 5, 5]))
    assert "regularization.project_in.weight" in sd and sd["regularization.project_out.weight"].shape == (8, 6)
    ora = build_oracle(cfg, sd)
    x = torch.rand(1, 3, 5, 32, 32) * 2 - 1
    z, dec, log = model(x)
    z2, dec2, log2 = ora(x)
    assert z.shape == (1, 8, 2, 4, 4) and rel_err(z, z2) < 2e-5 and rel_err(dec, dec2) < 5e-5
    assert torch.equal(log["indices"], log2["indices"])
    assert rel_err(model.decode(log["indices"], decode_from_indices=True), dec) < 1e-6


def test_api_surface_and_aliases(emulated_ops):
    import vidtok_amd
    from vidtok_amd.engine import AutoencodingEngine, AutoencodingEngineV11

    m, cfg, _ = build_model("vidtok_kl_causal_488_4chn")
    assert isinstance(m, AutoencodingEngine) and m.is_causal and m.encoder.time_downsample_factor == 4
    assert m.encode_decode.__func__ is m.forward.__func__
    x = torch.rand(1, 3, 5, 16, 16) * 2 - 1
    m.regularization.sample = False
    z = m.encode(x)
    z2, log = m.encode(x, return_reg_log=True)
    assert torch.equal(z, z2) and log["kl_loss"].dim() == 0
    assert m.decode(z).shape == x.shape
    m11, _, _ = build_model("vidtok_v1_1/vidtok_kl_causal_488_4chn_v1_1")
    assert isinstance(m11, AutoencodingEngineV11)
    assert m11.build_chunk_start_end(33) == [[0, 1], [1, 17], [17, 33]]
    assert m11.build_chunk_start_end(9, decoder_mode=True) == [[0, 1], [1, 5], [5, 9]]
    with pytest.raises(AssertionError):
        m.encoder(torch.zeros(1, 3, 8, 8))
    # reference target strings resolve to the MI355X classes (unmodified reference YAML surface)
    assert vidtok_amd.config.get_obj_from_str("vidtok.models.autoencoder.AutoencodingEngine") is AutoencodingEngine


def test_unsupported_variants_fail_loudly():
    from vidtok_amd.modules import EncoderCausal3DPadding
    from vidtok_amd.regularizers import FSQRegularizer

    with pytest.raises(NotImplementedError):
        EncoderCausal3DPadding(ch=32, out_ch=3, ch_mult=(1, 2), num_res_blocks=1, in_channels=3, z_channels=4,
                               norm_type="batchnorm")
    with pytest.raises(AssertionError):                      # as in the reference: several codebooks keep their axis
        FSQRegularizer(levels=[8, 8, 8], num_codebooks=2, keep_num_codebooks_dim=False)


# with the codebook axis kept the reference's own forward raises as soon as an aux-loss weight is non-zero
# (its implicit codebook is flattened to 1-D, regularizers.py:143-146,191-192,234): these options exist without aux loss
NO_AUX = dict(entropy_loss_weight=0.0, commitment_loss_weight=0.0)


@pytest.mark.parametrize("zc,reg", [(6, dict(levels=[8, 5, 5], num_codebooks=2, **NO_AUX)),
                                    (8, dict(levels=[8, 5, 5], num_codebooks=2, dim=8, **NO_AUX)),
                                    (3, dict(levels=[8, 5, 5], keep_num_codebooks_dim=True, **NO_AUX))],
                         ids=["two_codebooks", "two_codebooks_projected", "one_codebook_kept_axis"])
def test_fsq_num_codebooks_matches_oracle(zc, reg, emulated_ops):
    """several FSQ codebooks / the kept codebook axis: host graph on the same operators vs the oracle (which is pinned
    to the reference for these options in test_oracle_vs_reference.py)"""
    model, cfg, sd = build_model("vidtok_fsq_causal_488_32768", seed=5, overrides=dict(z_channels=zc), reg_overrides=reg)
    ora = build_oracle(cfg, sd)
    x = torch.rand((1, 3, 5, 32, 32), generator=torch.Generator().manual_seed(2)) * 2 - 1
    z, dec, log = model(x)
    z2, dec2, log2 = ora(x)
    assert log["indices"].shape == (1, 2, 4, 4, reg.get("num_codebooks", 1)) and torch.equal(log["indices"], log2["indices"])
    assert z.shape == (1, zc, 2, 4, 4) and rel_err(z, z2) < 2e-5 and rel_err(dec, dec2) < 5e-5
    assert abs(float(log["aux_loss"]) - float(log2["aux_loss"])) < 1e-4 * max(1.0, abs(float(log2["aux_loss"])))
    assert rel_err(model.decode(log["indices"], decode_from_indices=True), dec) < 1e-6
    model.regularization.entropy_loss_weight = 0.1           # ... and, like the reference, not with an aux loss
    with pytest.raises(NotImplementedError):
        model(x)


def test_product_path_has_no_cpu_fallback():
    """Without the monkeypatch the operators refuse CPU tensors instead of computing on the host."""
    import vidtok_amd.lib as L

    m, _, _ = build_model("vidtok_kl_causal_488_4chn")
    with pytest.raises((L.VtError, OSError, AttributeError)):
        m(torch.zeros(1, 3, 5, 16, 16))


def test_weight_packing_layout():
    from vidtok_amd.packing import pack_conv_weight

    w = torch.arange(2 * 3 * 2 * 3 * 3, dtype=torch.float32).reshape(2, 3, 2, 3, 3)
    p = pack_conv_weight(w, torch.float32)          # Cin 3 -> 8
    assert p.shape == (2, 2 * 3 * 3 * 8)
    p5 = p.reshape(2, 2, 3, 3, 8)
    assert torch.equal(p5[..., :3], w.permute(0, 2, 3, 4, 1)) and p5[..., 3:].abs().sum() == 0
    w1 = torch.randn(4, 16, 3)
    assert torch.equal(pack_conv_weight(w1, torch.float32).reshape(4, 3, 16), w1.permute(0, 2, 1))


def test_config_interpolation_and_checkpoint_roundtrip(tmp_path, emulated_ops):
    import vidtok_amd
    from safetensors.torch import save_file

    cfg = vidtok_amd.load_config(config_path("vidtok_fsq_causal_488_32768"))
    p = cfg["model"]["params"]
    assert p["decoder_config"]["params"] == p["encoder_config"]["params"]
    m, _, sd = build_model("vidtok_fsq_causal_488_32768", seed=5)
    path = str(tmp_path / "w.safetensors")
    extra = dict(sd)
    extra["loss.logvar"] = torch.zeros(())          # checkpoints carry loss.* keys; must be ignored
    save_file({k: v.contiguous() for k, v in extra.items()}, path)
    m2 = vidtok_amd.load_model_from_config(cfg, ckpt=path, verbose=False)
    for k, v in m2.state_dict().items():
        assert torch.equal(v, sd[k]), k
    m3 = vidtok_amd.load_model_from_config(cfg, ckpt=path, ignore_keys=[r"decoder\.conv_out\..*"], verbose=False)
    assert not torch.equal(m3.state_dict()["decoder.conv_out.conv.weight"], sd["decoder.conv_out.conv.weight"])


@pytest.mark.parametrize("ov,T", [(dict(resamp_with_conv=False), 5), (dict(init_pad_mode="constant"), 6),
                                  (dict(init_pad_mode="reflect"), 6), (dict(tanh_out=True), 5), (dict(give_pre_end=True), 5)],
                         ids=["no_resamp_conv", "pad_constant", "pad_reflect", "tanh_out", "give_pre_end"])
def test_rare_constructor_options_match_oracle(ov, T, emulated_ops):
    """options no shipped YAML sets run (they used to raise): same operators, host graph vs the oracle"""
    model, cfg, sd = build_model("vidtok_kl_causal_488_4chn", seed=9, overrides=ov)
    ora = build_oracle(cfg, sd)
    x = torch.rand(1, 3, T, 32, 32) * 2 - 1
    torch.manual_seed(2)
    z = model.encode(x)
    dec = model.decoder(z)
    torch.manual_seed(2)
    z2, _ = ora.encode(x)
    dec2 = ora.decode(z2)
    assert dec.shape == dec2.shape and rel_err(z, z2) < 2e-5 and rel_err(dec, dec2) < 5e-5


def test_chunk_cache_buffers_are_released_between_eager_passes(emulated_ops):
    """the persistent v1.1 chunk-cache buffers (modules.py::_CausalState._persistent) must not accumulate:
    an eager tiled pass leaves one buffer per (module, shape), the next encode / decode starts by dropping them, and
    invalidate_graphs() / .to() drop them when the graph cache is on."""
    model, cfg, sd = build_model("vidtok_v1_1/vidtok_kl_causal_488_16chn_v1_1", seed=3)
    model.use_tiling, model.t_chunk_enc, model.use_overlap = True, 8, True
    model.t_chunk_dec = 2

    def n_bufs():
        return sum(len(m.__dict__.get("_cache_bufs", {})) for m in model.modules())

    x = torch.rand(1, 3, 17, 16, 16) * 2 - 1
    model(x)
    assert n_bufs() > 0
    per_pass = n_bufs()
    model(torch.rand(1, 3, 17, 16, 24) * 2 - 1)             # another resolution: the first one's buffers are gone
    assert n_bufs() <= per_pass
    model._empty_causal_cached(model)
    assert n_bufs() == 0 and all(getattr(m, "causal_cache", None) is None for m in model.modules())
    model.use_graphs = True                                   # (no capture on the host; only the bookkeeping is tested)
    model(x)
    model._empty_causal_cached(model.encoder)
    assert n_bufs() > 0                                       # captured chunks would replay against these: kept ...
    model.invalidate_graphs()
    assert n_bufs() == 0                                      # ... until the graphs go


def test_graph_cache_bookkeeping_and_engine_copies():
    """GraphedCall keeps at most MAX_ENTRIES shapes (LRU; stateful chunk graphs are only dropped by the owner between
    passes), and a deep copy / pickle of an engine runs ITS OWN encoder and decoder (closures over `self`)."""
    import copy
    import pickle

    from vidtok_amd.graphs import GraphedCall

    g = GraphedCall(lambda t: t)
    for i in range(GraphedCall.MAX_ENTRIES + 5):
        g.entries[i] = "warm" if i % 2 else (None, None, None, None)
        g._touch(i)
    assert len(g.entries) == GraphedCall.MAX_ENTRIES and 0 not in g.entries and not g.overfull
    g._touch(5)
    assert list(g.entries)[-1] == 5
    for i in range(100, 100 + GraphedCall.MAX_ENTRIES + 3):    # stateful entries pile up until the owner resets
        g.entries[i] = (None, None, None, [("module", "cache")])
        g._touch(i)
    assert g.overfull and all(k >= 100 for k in g.entries)
    g.clear()
    assert not g.entries and not g.overfull

    model, cfg, sd = build_model("vidtok_kl_causal_488_4chn", seed=3)
    model._genc.entries["x"] = "warm"
    for other in (copy.deepcopy(model), pickle.loads(pickle.dumps(model))):
        assert other._genc is not model._genc and not other._genc.entries
        assert other._genc.fn.__self__ is other and other._gdec.fn.__self__ is other
        assert other._genc.state_get.__self__ is other and other._genc.fn.__func__ is type(model)._encoder_fn
        assert torch.equal(other.state_dict()["encoder.conv_in.conv.weight"], model.state_dict()["encoder.conv_in.conv.weight"])


def test_autocast_region_switches_the_arithmetic_mode():
    """AutoencodingEngine._sync_autocast (host logic only, no launch): the caller's torch.autocast region selects the kernels --
    autocast(bfloat16) / autocast(float16) = the bf16 / fp16 kernels whatever set_compute_dtype chose, the chosen mode returns when
    the region ends (an fp32 encoder tail chosen with it stays in force inside the region), a policy can map a region elsewhere or
    refuse it, "ignore" keeps the chosen mode; captured graphs are kept across the switches (their keys carry the mode)."""
    model, cfg, sd = build_model("vidtok_kl_causal_488_4chn", seed=3)
    x = torch.zeros(1, 3, 5, 16, 16)
    assert model.arith == "fp32"
    model._sync_autocast(x)
    assert model.arith == "fp32" and model.encoder.compute_dtype == torch.float32
    with torch.autocast("cpu", dtype=torch.bfloat16):
        model._sync_autocast(x)
        assert model.arith == "bf16" and model.encoder.compute_dtype == model.decoder.compute_dtype == torch.bfloat16
        model._sync_autocast(x)                                    # idempotent inside the region
        assert model.arith == "bf16"
    model._sync_autocast(x)
    assert model.arith == "fp32" and model.encoder.compute_dtype == torch.float32
    model.set_compute_dtype("bf16x3", encoder_tail=None)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        model._sync_autocast(x)
        assert model.arith == "bf16"
    model._sync_autocast(x)
    assert model.arith == "bf16x3" and model.encoder.compute_dtype == torch.float32
    model.set_compute_dtype(torch.bfloat16, encoder_tail=torch.float32, tail_level=2)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        model._sync_autocast(x)                                    # bf16 asked and chosen: nothing to switch, the fp32 tail stays
        assert model.arith == "bf16" and model.encoder.tail_dtype == torch.float32
    with torch.autocast("cpu", dtype=torch.float16):
        model._sync_autocast(x)                                    # fp16 for the region, the tail chosen with the mode stays in force
        assert model.arith == "fp16" and model.encoder.compute_dtype == torch.float16
        assert model.encoder.tail_dtype == torch.float32 and model.encoder.tail_level == 2
    model._sync_autocast(x)
    assert model.arith == "bf16" and model.encoder.tail_dtype == torch.float32 and model.encoder.tail_level == 2
    model.set_compute_dtype(torch.float32)
    model._genc.entries["kept"] = "warm"
    with torch.autocast("cpu", dtype=torch.float16):
        model._sync_autocast(x)                                    # the README's region: the fp16 kernels
        assert model.arith == "fp16" and model.decoder.compute_dtype == torch.float16
        assert "kept" in model._genc.entries                       # a mode switch does not drop captured graphs
    model._sync_autocast(x)
    assert model.arith == "fp32" and "kept" in model._genc.entries
    model.set_autocast_policy(float16="error")
    with torch.autocast("cpu", dtype=torch.float16):
        with pytest.raises(NotImplementedError, match="float16"):
            model._sync_autocast(x)
        model.set_autocast_policy(float16="ignore")
        model._sync_autocast(x)
        assert model.arith == "fp32"
        model.set_autocast_policy(float16="bf16x3")
        model._sync_autocast(x)
        assert model.arith == "bf16x3"
    model._sync_autocast(x)
    assert model.arith == "fp32"
    with pytest.raises(AssertionError):
        model.set_autocast_policy(float16="fp8")


def test_parity_mix_tables_equal_the_tap_sum_statements():
    """The tap-sum tables vt_pack_conv_weight gets for the up-samplers' parity classes (packing.{time,space}_upsample_parity_mix) against
    the torch statements they replace on the GPU ({time,space}_upsample_parity_weights + pack_conv_weight): the kernel's contract --
    out tap j = (w[m0] + w[m1]) + (w[m2] + w[m3]), absent terms left out, channels padded -- evaluated on the host gives the same bits."""
    from vidtok_amd import packing as P

    def by_table(w, cin_p, mix):
        cout, cin = w.shape[:2]
        wf = w.reshape(cout, cin, -1)
        out = torch.zeros(cout, len(mix), cin_p)
        for j, m in enumerate(mix):
            m = (list(m) + [-1] * 4)[:4]
            a = wf[:, :, m[0]].clone()
            if m[1] >= 0:
                a = a + wf[:, :, m[1]]
            if m[2] >= 0:
                b = wf[:, :, m[2]].clone()
                if m[3] >= 0:
                    b = b + wf[:, :, m[3]]
                a = a + b
            out[:, j, :cin] = a
        return out.reshape(cout, -1)

    g = torch.Generator().manual_seed(1)
    for early in (True, False):
        w = torch.randn(16, 5, 3, 3, 3, generator=g)
        mix = P.time_upsample_parity_mix((3, 3, 3), early)
        assert len(mix) == 18 and torch.equal(P.pack_conv_weight(P.time_upsample_parity_weights(w, early), torch.float32, 8), by_table(w, 8, mix))
    for py in (0, 1):
        for px in (0, 1):
            w = torch.randn(8, 24, 3, 3, generator=g)
            mix = P.space_upsample_parity_mix((3, 3), py, px)
            assert len(mix) == 4 and torch.equal(P.pack_conv_weight(P.space_upsample_parity_weights(w, py, px), torch.float32, 24), by_table(w, 24, mix))


def test_fsq_mismatch_report_names_the_boundary_cases():
    """tests/util.py::fsq_mismatch_report ("count of mismatches and their distance to a rounding boundary"): a latent
    nudged across a rounding boundary flips exactly that digit, and the report names token, channel, both bounded values and their
    distances to the boundary in fp32 ulps."""
    from oracle.vidtok_oracle import fsq_regularize
    from util import fsq_mismatch_report

    levels = [8, 8, 8, 5, 5, 5]
    h = torch.randn(2, 6, 3, 8, 8, generator=torch.Generator().manual_seed(0)) * 0.3
    # put channel 4 of one token right below the boundary 0.5 of an odd level (bounded = tanh(h) * half_l), then push it across
    half_l = (5 - 1) * (1 + 1e-3) / 2
    h[1, 4, 2, 3, 5] = float(torch.atanh(torch.tensor(0.5 / half_l))) - 2e-6
    h2 = h.clone()
    h2[1, 4, 2, 3, 5] += 4e-6
    _, l0 = fsq_regularize(h, levels, with_aux=False)
    _, l1 = fsq_regularize(h2, levels, with_aux=False)
    rep = fsq_mismatch_report(levels, h2, h, l1["indices"], l0["indices"])
    assert rep["tokens"] == 2 * 3 * 8 * 8 and rep["mismatches"] == 1 and len(rep["detail"]) == 1
    d = rep["detail"][0]
    assert d["token"] == [1, 2, 3, 5] and d["channel"] == 4 and d["digit"] == d["digit_ref"] + 1
    assert d["bounded_ref"] < 0.5 < d["bounded"] and 0 < d["ref_to_boundary_ulps"] < 200 and 0 < d["ours_to_boundary_ulps"] < 200
    assert 100 < d["h_diff_ulps"] < 170                              # 4e-6 at |h| = 0.255: ulp 3e-8
    same = fsq_mismatch_report(levels, h, h, l0["indices"], l0["indices"])
    assert same["mismatches"] == 0 and same["detail"] == []
