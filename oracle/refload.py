"""TEST INFRASTRUCTURE ONLY -- loader for the *unmodified* reference (microsoft/VidTok).

Only usable in the build container, where /root/reference exists; nothing on the GPU
box may import this at run time (the GPU box has no /root/reference).  It is used by
`scripts/make_golden.py` (fixture generation) and by the CPU tests that pin
`oracle/vidtok_oracle.py` against the real reference.

Three packages that the reference imports are not installed in this image and carry no
arithmetic: `beartype` (argument type checks,
reference vidtok/modules/model_3dcausal.py:2-3), `lightning.pytorch` (base class only,
reference vidtok/models/autoencoder.py:9,18) and `omegaconf` (one annotation,
autoencoder.py:5).  They are replaced by inert stand-ins in sys.modules *before* the first
`import vidtok`.  `loss_config` is overridden with torch.nn.Identity because the real
loss constructs LPIPS, which needs torchvision and a network download
(reference vidtok/modules/lpips.py:9,55); the loss is never called in forward/encode/decode.
"""
import os
import sys
import types
import typing

import torch
import yaml

REFERENCE_ROOT = os.environ.get("VIDTOK_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "vidtok"))


def _install_stubs():
    if "vidtok" in sys.modules:
        return

    def _mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    if "beartype" not in sys.modules:
        bt = _mod("beartype")
        bt.beartype = lambda f: f
        btt = _mod("beartype.typing")
        btt.Tuple, btt.Union = typing.Tuple, typing.Union
    if "lightning" not in sys.modules:
        L = _mod("lightning")
        LP = _mod("lightning.pytorch")
        L.pytorch = LP

        class LightningModule(torch.nn.Module):
            global_step = 0
            automatic_optimization = True

        LP.LightningModule = LightningModule
        _mod("lightning.pytorch.utilities")
        _mod("lightning.pytorch.utilities.rank_zero").rank_zero_only = lambda f: f
    if "omegaconf" not in sys.modules:
        _mod("omegaconf").ListConfig = list
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def load_reference_config(cfg_rel: str) -> dict:
    """cfg_rel e.g. 'vidtok_kl_causal_488_4chn' or 'vidtok_v1_1/vidtok_kl_causal_488_16chn_v1_1'."""
    with open(os.path.join(REFERENCE_ROOT, "configs", cfg_rel + ".yaml")) as f:
        cfg = yaml.safe_load(f)
    p = cfg["model"]["params"]
    # resolves the single OmegaConf interpolation ${model.params.encoder_config.params}
    p["decoder_config"]["params"] = dict(p["encoder_config"]["params"])
    p["loss_config"] = {"target": "torch.nn.Identity"}
    return cfg


def load_reference_model(cfg_rel: str, seed: int = 0, overrides: dict = None, reg_overrides: dict = None):
    """Instantiate the reference AutoencodingEngine (random init under `seed`), eval mode.  `overrides` update the
    encoder/decoder params, `reg_overrides` the regularizer params (variants no shipped YAML exercises)."""
    _install_stubs()
    from vidtok.modules.util import instantiate_from_config  # reference vidtok/modules/util.py:69

    cfg = load_reference_config(cfg_rel)
    if overrides:
        for k in ("encoder_config", "decoder_config"):
            cfg["model"]["params"][k]["params"].update(overrides)
    if reg_overrides:
        cfg["model"]["params"]["regularizer_config"].setdefault("params", {}).update(reg_overrides)
    torch.manual_seed(seed)
    model = instantiate_from_config(cfg["model"]).eval()
    return model, cfg


def randomize_weights(model, seed: int = 1):
    """Make every layer numerically non-trivial (SURVEY.md finding 3): the temporal blocks'
    conv2 is zero-initialised in the reference (model_3dcausal.py:460-462) so a fresh model
    never exercises it; LayerNorm affines are identity.  Deterministic under `seed`."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if (name.endswith("conv2.conv.weight") or name.endswith("conv2.weight")) and p.dim() == 3:   # zero-init temporal conv2
                p.copy_(torch.randn(p.shape, generator=g) * (1.0 / (p.shape[1] * p.shape[2]) ** 0.5))
            elif (name.endswith("conv2.conv.bias") or name.endswith("conv2.bias")) and "temporal" in name:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
            elif ".norm" in name and name.endswith("weight") and p.dim() == 1:
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif ".norm" in name and name.endswith("bias") and p.dim() == 1:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif name.endswith("mix_factor"):
                p.copy_(torch.tensor([0.3]) + 0.5 * torch.randn(p.shape, generator=g))
    return model
