"""Build-container helper: derive configs/*.yaml (model section only, vidtok_amd targets) from the
hyper-parameters of the reference's causal and non-causal configs under /root/reference/configs.  The emitted
files are plain data (channel counts, levels, flags); the `data:` and `lightning:` sections and the training loss
are out of scope and are not carried over."""
import glob
import os
import sys

import yaml

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from vidtok_amd.config import TARGET_ALIASES  # noqa: E402

REF = "/root/reference/configs"
OUT = os.path.join(os.path.dirname(__file__), "..", "configs")


def main():
    paths = sorted(glob.glob(f"{REF}/vidtok_*causal_*.yaml") + glob.glob(f"{REF}/vidtok_v1_1/*.yaml"))
    for p in paths:
        src = yaml.safe_load(open(p))["model"]
        prm = src["params"]
        enc = dict(prm["encoder_config"]["params"])
        for k in ("dropout", "use_checkpoint", "fix_encoder", "fix_decoder"):
            enc.pop(k, None)  # training-only switches
        model = {
            "target": TARGET_ALIASES[src["target"]],
            "params": {
                "encoder_config": {"target": TARGET_ALIASES[prm["encoder_config"]["target"]], "params": enc},
                "decoder_config": {"target": TARGET_ALIASES[prm["decoder_config"]["target"]],
                                   "params": "${model.params.encoder_config.params}"},
                "regularizer_config": {"target": TARGET_ALIASES[prm["regularizer_config"]["target"]]},
            },
        }
        if prm["regularizer_config"].get("params"):
            model["params"]["regularizer_config"]["params"] = prm["regularizer_config"]["params"]
        if "use_tiling" in prm:
            model["params"]["use_tiling"] = bool(prm["use_tiling"])
        rel = os.path.relpath(p, REF)
        dst = os.path.join(OUT, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        with open(dst, "w") as f:
            f.write(f"# vidtok_amd model config; hyper-parameters of the reference's configs/{rel}\n")
            yaml.safe_dump({"model": model}, f, sort_keys=False, default_flow_style=None, width=100)
        print("wrote", dst)


if __name__ == "__main__":
    main()
