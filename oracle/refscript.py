"""TEST INFRASTRUCTURE ONLY -- runs the reference's OWN video reconstruction script, unmodified, in the build container.

R = /root/reference.  `run_reference_reconstruct(...)` executes R/scripts/inference_reconstruct.py as `__main__`
(`SingleVideoDataset` :28-73, `tensor_to_uint8` :76-80, the whole `main()` :83-246 incl. the DataLoader, the
--pad_gen_frames chaining :211-221, clamp / rearrange / concatenate :226-235) and returns the array it hands to
`write_video`.  Nothing of the script is restated here; what this file supplies are stand-ins for the packages the script
imports that are absent from the image (no network):

  decord                  the codec.  `VideoReader(path)` serves a numpy / torch uint8 array [N, H0, W0, 3] registered under
                          `path` (len, get_avg_fps, get_batch -> torch tensor, as with `decord.bridge.set_bridge("torch")`).
  torchvision.transforms  Compose / Resize(int, antialias=True) / CenterCrop / Normalize for float tensors exactly as
                          torchvision 0.17 (pinned by R/environment.yaml) defines them: resize = the ATen operator it calls,
                          torch.nn.functional.interpolate(mode="bilinear", align_corners=False, antialias=True), to
                          _compute_resized_output_size(); center_crop offsets int(round((H - h) / 2.0)); (x - mean) / std.
  torchvision.io          write_video -> captures (path, array, fps).
  scripts.inference_evaluate.load_model_from_config -> returns the model the caller passes in (the real module drags in
                          LPIPS / torchvision models; the script only needs the constructed model).
  omegaconf.OmegaConf.load, lightning.pytorch.seed_everything -> yaml load / torch.manual_seed.

Used by tests/test_video_io.py (reference-marked) to pin oracle/video_io_oracle.py and by scripts/make_golden_video_io.py.
"""
import os
import runpy
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

from .refload import REFERENCE_ROOT, _install_stubs

_VIDEOS = {}          # path -> (uint8 tensor [N, H0, W0, 3], fps)
_CAPTURE = {}


def _install_script_stubs(model):
    _install_stubs()

    def _mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    # ---- decord -------------------------------------------------------------------------------------------------
    dec = _mod("decord")
    dec.bridge = types.SimpleNamespace(set_bridge=lambda name: None)

    class VideoReader:
        def __init__(self, path, num_threads=0):
            self.frames, self.fps = _VIDEOS[path]

        def __len__(self):
            return int(self.frames.shape[0])

        def get_avg_fps(self):
            return float(self.fps)

        def get_batch(self, ids):
            return self.frames[torch.as_tensor(list(ids), dtype=torch.long)]

    dec.VideoReader = VideoReader

    # ---- torchvision.transforms / torchvision.io (torchvision 0.17 semantics for float tensors) -------------------
    tv = _mod("torchvision")
    tr = _mod("torchvision.transforms")
    tio = _mod("torchvision.io")
    tv.transforms, tv.io = tr, tio

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class Resize:
        def __init__(self, size, antialias=True):
            assert isinstance(size, int) and antialias is True
            self.size = size

        def __call__(self, img):                      # functional.resize + _compute_resized_output_size, size: int
            h, w = img.shape[-2:]
            short, long = (w, h) if w <= h else (h, w)
            new_short, new_long = self.size, int(self.size * long / short)
            new_w, new_h = (new_short, new_long) if w <= h else (new_long, new_short)
            if (new_h, new_w) == (h, w):
                return img
            return F.interpolate(img, size=(new_h, new_w), mode="bilinear", align_corners=False, antialias=True)

    class CenterCrop:
        def __init__(self, size):
            self.size = tuple(size)

        def __call__(self, img):                      # functional.center_crop (images at least as large as the crop)
            ch, cw = self.size
            h, w = img.shape[-2:]
            assert h >= ch and w >= cw
            top, left = int(round((h - ch) / 2.0)), int(round((w - cw) / 2.0))
            return img[..., top:top + ch, left:left + cw]

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = mean, std

        def __call__(self, t):                        # functional.normalize on [..., C, H, W]
            mean = torch.as_tensor(self.mean, dtype=t.dtype).view(-1, 1, 1)
            std = torch.as_tensor(self.std, dtype=t.dtype).view(-1, 1, 1)
            return (t - mean) / std

    tr.Compose, tr.Resize, tr.CenterCrop, tr.Normalize = Compose, Resize, CenterCrop, Normalize

    def write_video(path, array, fps, **kw):
        _CAPTURE.update(path=path, array=np.array(array), fps=fps)

    tio.write_video = write_video

    # ---- the script's sibling module, omegaconf.OmegaConf, lightning seed_everything ----------------------------------
    if "scripts" not in sys.modules:
        _mod("scripts").__path__ = []
    ev = _mod("scripts.inference_evaluate")
    ev.load_model_from_config = lambda config, ckpt: model
    import yaml

    oc = sys.modules["omegaconf"]
    oc.OmegaConf = types.SimpleNamespace(load=lambda p: yaml.safe_load(open(p)))

    def seed_everything(seed):
        torch.manual_seed(seed)
        return seed

    sys.modules["lightning.pytorch"].seed_everything = seed_everything


def run_reference_reconstruct(model, frames_u8, fps, *, config_rel, input_height, input_width, sample_fps, chunk_size,
                              read_long_video=False, pad_gen_frames=False, concate_input=True, seed=42):
    """Run R/scripts/inference_reconstruct.py::main() on the uint8 frames [N, H0, W0, 3] with `model` (a reference
    AutoencodingEngine); returns the uint8 array [n, h, w | 2w, 3] the script passes to write_video."""
    _install_script_stubs(model)
    frames = torch.as_tensor(np.asarray(frames_u8)).to(torch.uint8)
    path = f"/virtual/video_{len(_VIDEOS)}.mp4"
    _VIDEOS[path] = (frames, fps)
    argv = ["inference_reconstruct.py", "--config", os.path.join(REFERENCE_ROOT, "configs", config_rel + ".yaml"), "--ckpt", "unused",
            "--output_video_dir", "/tmp/vidtok_refscript_out", "--input_video_path", path, "--input_height", str(input_height),
            "--input_width", str(input_width), "--sample_fps", str(sample_fps), "--chunk_size", str(chunk_size),
            "--concate_input", "true" if concate_input else "false", "--seed", str(seed)]
    if read_long_video:
        argv.append("--read_long_video")
    if pad_gen_frames:
        argv.append("--pad_gen_frames")
    _CAPTURE.clear()
    old_argv, old_cwd = sys.argv, os.getcwd()
    sys.argv = argv
    try:
        os.chdir(REFERENCE_ROOT)                       # the script appends os.getcwd() to sys.path for `scripts.` / `vidtok.`
        runpy.run_path(os.path.join(REFERENCE_ROOT, "scripts", "inference_reconstruct.py"), run_name="__main__")
    finally:
        sys.argv = old_argv
        os.chdir(old_cwd)
    assert "array" in _CAPTURE, "the reference script did not reach write_video"
    return _CAPTURE["array"], _CAPTURE["fps"]
