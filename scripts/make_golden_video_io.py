"""Build-container helper: run the reference's OWN scripts/inference_reconstruct.py (unmodified; oracle/refscript.py
supplies the codec and torchvision stand-ins) over the synthetic videos of tests/golden_cases.VIDEO_CASES with seeded
weights and store what it hands to write_video under tests/golden/video_io.safetensors.  Re-run:
`python scripts/make_golden_video_io.py`."""
import os
import sys

import torch
from safetensors.torch import save_file

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from golden_cases import VIDEO_CASES, make_video  # noqa: E402
from oracle.refload import load_reference_model  # noqa: E402
from oracle.refscript import run_reference_reconstruct  # noqa: E402
from util import GOLDEN_DIR, seeded_state_dict  # noqa: E402


def reference_output(case):
    ref, _ = load_reference_model(case["config"])
    ref.load_state_dict(seeded_state_dict({k: v.shape for k, v in ref.state_dict().items()}, case["weight_seed"]), strict=True)
    if hasattr(ref.regularization, "sample"):
        ref.regularization.sample = False            # KL: the posterior mode, so the fixture does not depend on a noise stream
    arr, fps = run_reference_reconstruct(
        ref, make_video(case), case["fps"], config_rel=case["config"], input_height=case["input_height"], input_width=case["input_width"],
        sample_fps=case["sample_fps"], chunk_size=case["chunk_size"], read_long_video=case["read_long_video"],
        pad_gen_frames=case["pad_gen_frames"], concate_input=case["concate_input"])
    assert fps == case["sample_fps"]
    return torch.from_numpy(arr).contiguous()


def main():
    out = {}
    for case in VIDEO_CASES:
        out[case["name"]] = reference_output(case)
        print(case["name"], tuple(out[case["name"]].shape))
    path = os.path.join(GOLDEN_DIR, "video_io.safetensors")
    save_file(out, path)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
