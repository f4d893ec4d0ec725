"""Shared helpers of the test-suite (TEST INFRASTRUCTURE)."""
import math
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIG_DIR = os.path.join(ROOT, "configs")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def config_path(name):
    return os.path.join(CONFIG_DIR, name + ".yaml")


def seeded_state_dict(shapes: dict, seed: int = 1234) -> dict:
    """Deterministic, numerically non-trivial weights for a VidTok state_dict (SURVEY.md finding 3:
    a fresh reference model has zero temporal conv2 and identity LayerNorms).  Depends only on the key
    names/shapes and the torch CPU generator, so the build container (where the reference produces the
    golden outputs) and the GPU box regenerate bit-identical tensors."""
    out = {}
    for i, name in enumerate(sorted(shapes)):
        shape = tuple(shapes[name])
        g = torch.Generator().manual_seed(seed * 100003 + i)
        if name.endswith("mix_factor"):
            t = 0.3 + 0.5 * torch.randn(shape, generator=g)
        elif ".norm" in name and name.endswith(".weight") and len(shape) == 1:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            t = 0.05 * torch.randn(shape, generator=g)
        else:
            fan_in = max(1, int(math.prod(shape[1:])))
            t = torch.randn(shape, generator=g) / math.sqrt(fan_in)
        out[name] = t
    return out


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max |a-b| / max |b| -- the parity metric of SURVEY.md section 8(d)."""
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def build_model(name: str, seed: int = 1234, device="cpu", dtype=torch.float32, overrides=None, reg_overrides=None):
    """vidtok_amd engine from configs/<name>.yaml with seeded weights; returns (model, cfg, state_dict).
    `overrides` / `reg_overrides` update the encoder+decoder / regularizer params (variants without a YAML)."""
    import vidtok_amd

    cfg = vidtok_amd.load_config(config_path(name))
    prm = cfg["model"]["params"]
    if overrides:
        prm["encoder_config"]["params"].update(overrides)
        if isinstance(prm["decoder_config"]["params"], dict):
            prm["decoder_config"]["params"].update(overrides)
    if reg_overrides:
        prm["regularizer_config"].setdefault("params", {}).update(reg_overrides)
    model = vidtok_amd.load_model_from_config(cfg, verbose=False)
    sd = seeded_state_dict({k: v.shape for k, v in model.state_dict().items()}, seed)
    missing, unexpected = model.load_state_dict(sd, strict=True)
    model = model.to(device).eval()
    model.set_compute_dtype(dtype)
    return model, cfg, sd


def build_oracle(cfg: dict, sd: dict):
    from oracle.vidtok_oracle import OracleEngine

    return OracleEngine(cfg["model"]["params"], sd)
