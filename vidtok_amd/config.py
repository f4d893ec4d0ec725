"""YAML config surface of the reference (configs/*.yaml) and its plug-in loader.

`instantiate_from_config` / `get_obj_from_str` have the semantics of the reference loader
(vidtok/modules/util.py:69-86): a dict with a dotted `target:` and optional `params:`.  The only
additions: (1) reference target strings of the hot path resolve to the vidtok_amd classes, so an
UNMODIFIED reference YAML builds the MI355X model; (2) the training-only `loss_config` target is
not importable offline and is mapped to torch.nn.Identity (the loss is never called by
forward / encode / decode); (3) `${model.params.encoder_config.params}` -- the one OmegaConf
interpolation the shipped configs use (e.g. configs/vidtok_kl_causal_488_4chn.yaml:31) -- is
resolved by the loader itself because omegaconf is not a dependency.
"""
import copy
import importlib
import re

import yaml

TARGET_ALIASES = {
    "vidtok.models.autoencoder.AutoencodingEngine": "vidtok_amd.engine.AutoencodingEngine",
    "vidtok.models.autoencoder_v1_1.AutoencodingEngine": "vidtok_amd.engine.AutoencodingEngineV11",
    "vidtok.modules.model_3dcausal.EncoderCausal3DPadding": "vidtok_amd.modules.EncoderCausal3DPadding",
    "vidtok.modules.model_3dcausal.DecoderCausal3DPadding": "vidtok_amd.modules.DecoderCausal3DPadding",
    "vidtok.modules.model_3dcausal_v1_1.EncoderCausal3DPadding": "vidtok_amd.modules.EncoderCausal3DPaddingV11",
    "vidtok.modules.model_3dcausal_v1_1.DecoderCausal3DPadding": "vidtok_amd.modules.DecoderCausal3DPaddingV11",
    "vidtok.modules.model_3dnoncausal.Encoder3D": "vidtok_amd.modules_noncausal.Encoder3D",
    "vidtok.modules.model_3dnoncausal.Decoder3D": "vidtok_amd.modules_noncausal.Decoder3D",
    "vidtok.modules.regularizers.DiagonalGaussianRegularizer": "vidtok_amd.regularizers.DiagonalGaussianRegularizer",
    "vidtok.modules.regularizers.FSQRegularizer": "vidtok_amd.regularizers.FSQRegularizer",
    "vidtok.modules.losses.GeneralLPIPSWithDiscriminator": "torch.nn.Identity",
}

_INTERP = re.compile(r"^\$\{([A-Za-z0-9_.]+)\}$")


def get_obj_from_str(string: str, reload: bool = False):
    string = TARGET_ALIASES.get(string, string)
    module, cls = string.rsplit(".", 1)
    if reload:
        importlib.reload(importlib.import_module(module))
    return getattr(importlib.import_module(module, package=None), cls)


def instantiate_from_config(config):
    if "target" not in config:
        raise KeyError("Expected key `target` to instantiate.")
    target = TARGET_ALIASES.get(config["target"], config["target"])
    params = config.get("params", dict()) or dict()
    if target == "torch.nn.Identity":
        params = {}
    return get_obj_from_str(target)(**params)


def _lookup(root, dotted):
    node = root
    for part in dotted.split("."):
        node = node[part]
    return node


def resolve_interpolations(root, node=None):
    """Replace '${a.b.c}' string values by a deep copy of the referenced node (in place)."""
    node = root if node is None else node
    items = node.items() if isinstance(node, dict) else enumerate(node) if isinstance(node, list) else []
    for k, v in list(items):
        if isinstance(v, str):
            m = _INTERP.match(v.strip())
            if m:
                node[k] = copy.deepcopy(_lookup(root, m.group(1)))
        elif isinstance(v, (dict, list)):
            resolve_interpolations(root, v)
    return root


def load_config(path: str) -> dict:
    with open(path) as f:
        cfg = yaml.safe_load(f)
    return resolve_interpolations(cfg)


def load_model_from_config(config, ckpt: str = None, ignore_keys=(), verbose: bool = True):
    """Counterpart of scripts/inference_evaluate.py:26-32 of the reference."""
    cfg = load_config(config) if isinstance(config, str) else config
    cfg = copy.deepcopy(cfg)
    params = cfg["model"].setdefault("params", {})
    params["ckpt_path"] = ckpt
    params["ignore_keys"] = list(ignore_keys)
    params["verbose"] = verbose
    return instantiate_from_config(cfg["model"])
