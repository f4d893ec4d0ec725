"""Oracle vs the committed outputs of the unmodified reference (tests/golden/, made by
scripts/make_golden.py).  Runs anywhere -- this is how the oracle stays pinned on the GPU box."""
import os

import pytest
import torch
from safetensors.torch import load_file

from golden_cases import CASES, apply_tiling, make_input
from util import GOLDEN_DIR, build_oracle, config_path, rel_err, seeded_state_dict


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_reproduces_reference_golden(case):
    import vidtok_amd

    gold = load_file(os.path.join(GOLDEN_DIR, case["name"] + ".safetensors"))
    cfg = vidtok_amd.load_config(config_path(case["config"]))
    # shapes of the state_dict come from the (CPU-constructible) host mirror; no kernels run here
    model = vidtok_amd.load_model_from_config(cfg, verbose=False)
    sd = seeded_state_dict({k: v.shape for k, v in model.state_dict().items()}, case["weight_seed"])
    ora = build_oracle(cfg, sd)
    apply_tiling(ora, case)
    x = make_input(case)
    assert torch.equal(x, gold["x"])
    torch.manual_seed(case["noise_seed"])
    z, dec, log = ora(x)
    assert rel_err(z, gold["z"]) < 2e-5
    assert rel_err(dec, gold["dec"]) < 5e-5
    if "indices" in gold:
        assert torch.equal(log["indices"], gold["indices"])
        assert abs(float(log["aux_loss"]) - float(gold["aux_loss"])) < 1e-4
    else:
        assert abs(float(log["kl_loss"]) - float(gold["kl_loss"])) < 1e-4 * abs(float(gold["kl_loss"]))
