"""Build-container helper: run the UNMODIFIED reference (oracle/refload.py) on seeded inputs with the
seeded weights of tests/util.seeded_state_dict and store its outputs as fixtures under tests/golden/.
The fixtures let the GPU box (which has no /root/reference) check both the oracle and the HIP path
against outputs of the reference itself.  Re-run: `python scripts/make_golden.py`."""
import os
import sys

import torch
from safetensors.torch import save_file

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from golden_cases import CASES, make_input  # noqa: E402
from oracle.refload import load_reference_model  # noqa: E402
from util import GOLDEN_DIR, seeded_state_dict  # noqa: E402


def main():
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    for case in CASES:
        ref, _ = load_reference_model(case["config"])
        sd = seeded_state_dict({k: v.shape for k, v in ref.state_dict().items()}, case["weight_seed"])
        ref.load_state_dict(sd, strict=True)
        if case.get("tiling"):
            ref.use_tiling, ref.t_chunk_enc = True, case["tiling"]["t_chunk_enc"]
            ref.t_chunk_dec = case["tiling"]["t_chunk_enc"] // 4
            ref.use_overlap = case["tiling"]["use_overlap"]
        x = make_input(case)
        with torch.no_grad():
            torch.manual_seed(case["noise_seed"])
            z, dec, log = ref(x)
        out = {"x": x, "z": z.contiguous(), "dec": dec.contiguous()}
        for k, v in log.items():
            out[k] = v.detach().clone().reshape(-1) if v.dim() == 0 else v.detach().contiguous()
        path = os.path.join(GOLDEN_DIR, case["name"] + ".safetensors")
        save_file(out, path)
        print(case["name"], {k: tuple(v.shape) for k, v in out.items()}, os.path.getsize(path))


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
