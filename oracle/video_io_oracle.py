"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's video front / back end around `model(x)`
(SURVEY.md section 8 row f4): everything of scripts/inference_reconstruct.py except the codec itself.

R = /root/reference.  The reference decodes with `decord` and transforms with `torchvision` (both absent from this
image, neither vendored under R): `decord.VideoReader.get_batch` hands over uint8 frames [T, H0, W0, 3];
torchvision 0.17 (pinned by R/environment.yaml next to torch 2.2.2) implements, for float tensors,
  * transforms.Resize(size:int, antialias=True) as torch.nn.functional.interpolate(mode="bilinear",
    align_corners=False, antialias=True) to (new_h, new_w) where the SHORTER side becomes `size` and the longer one
    int(size * long / short); an image that already has that size is returned unchanged
    (torchvision/transforms/functional.py::resize, _compute_resized_output_size),
  * transforms.CenterCrop((h, w)) as img[..., top:top+h, left:left+w] with top = int(round((H - h) / 2.0)) -- Python's
    round, i.e. half to even (functional.py::center_crop),
  * transforms.Normalize(mean, std) as (x - mean) / std.
`resize_aa` below restates ATen's anti-aliased bilinear filter (aten/src/ATen/native/cpu/UpSampleKernel.cpp,
`_compute_indices_weights_aa` + separable horizontal-then-vertical passes, fp32) and is PINNED in tests against
torch.nn.functional.interpolate of this image -- the operator the reference reaches through torchvision.

PARITY PINNED: `frame_id_batches`, `preprocess_frames` (with the ATen resize), `tensor_to_uint8` and `reconstruct` are
checked against the reference's own script run UNMODIFIED in the build container (oracle/refscript.py executes
R/scripts/inference_reconstruct.py::main() with stand-ins for the absent codec / torchvision packages): bit-exact uint8
output for causal + --pad_gen_frames, FSQ without concatenation on a portrait source, v1.1 --read_long_video tiling and a
non-causal model (tests/test_video_io.py::test_oracle_loop_equals_reference_script), and the script's outputs are
committed as tests/golden/video_io.safetensors (scripts/make_golden_video_io.py) for replay where R is absent.

Call sites restated: R/scripts/inference_reconstruct.py:28-82 (SingleVideoDataset), :76-82 (tensor_to_uint8),
:206-239 (main loop with --pad_gen_frames chaining and --concate_input), R/vidtok/data/vidtok.py:180-188, 204-265."""
import math
from typing import Callable, List

import numpy as np
import torch


# --------------------------------------------------------------------------------------------------
# frame index batching (no arithmetic on pixels)
# --------------------------------------------------------------------------------------------------
def frame_id_batches(total_frames: int, fps: float, sample_fps: int, chunk_size: int, is_causal: bool,
                     read_long_video: bool) -> List[List[int]]:
    """R/scripts/inference_reconstruct.py:49-66 (same rule in R/vidtok/data/vidtok.py:216-236 with
    last_frames_handle="drop")."""
    interval = round(fps / sample_fps)
    frame_ids = list(range(0, total_frames, interval))
    out = []
    if read_long_video:
        n = len(frame_ids)
        if is_causal and n > chunk_size:
            out.append(frame_ids[:chunk_size * ((n - 1) // chunk_size) + 1])
        elif not is_causal and n >= chunk_size:
            out.append(frame_ids[:chunk_size * (n // chunk_size)])
    else:
        per = chunk_size + 1 if is_causal else chunk_size
        for s in range(0, len(frame_ids), per):
            if len(frame_ids[s:s + per]) == per:
                out.append(frame_ids[s:s + per])
    return out


# --------------------------------------------------------------------------------------------------
# resize / crop / normalise
# --------------------------------------------------------------------------------------------------
def resized_size(h: int, w: int, size: int):
    """torchvision _compute_resized_output_size for an int `size`: shorter side -> size, longer -> int(size*long/short)"""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)      # (new_h, new_w)


def aa_weights(in_size: int, out_size: int):
    """ATen _compute_indices_weights_aa for the bilinear (triangle) filter, align_corners=False, fp32:
    per output index: (first input index, [weights])."""
    f32 = np.float32
    scale = f32(in_size) / f32(out_size)
    support = f32(scale) if scale >= 1.0 else f32(1.0)               # interp_size/2 * scale, interp_size = 2
    invscale = f32(1.0) / scale if scale >= 1.0 else f32(1.0)
    out = []
    for i in range(out_size):
        center = scale * f32(i + 0.5)
        xmin = max(int(np.int64(center - support + f32(0.5))), 0)
        xsize = min(int(np.int64(center + support + f32(0.5))), in_size) - xmin
        ws, total = [], f32(0.0)
        for j in range(xsize):
            x = (f32(j + xmin) - center + f32(0.5)) * invscale
            a = abs(x)
            w = f32(1.0) - a if a < 1.0 else f32(0.0)
            ws.append(f32(w))
            total = f32(total + w)
        if total != 0:
            ws = [f32(w / total) for w in ws]
        out.append((xmin, ws))
    return out


def resize_aa(x: torch.Tensor, new_h: int, new_w: int) -> torch.Tensor:
    """x [..., H, W] fp32 -> [..., new_h, new_w]: horizontal pass then vertical pass, each a weighted sum in fp32
    accumulated in tap order (ATen's separable anti-aliased bilinear)."""
    H, W = x.shape[-2:]
    if W != new_w:
        wt = aa_weights(W, new_w)
        cols = []
        for xmin, ws in wt:
            acc = torch.zeros(x.shape[:-1], dtype=torch.float32)
            for j, w in enumerate(ws):
                acc = acc + x[..., xmin + j] * float(w)
            cols.append(acc)
        x = torch.stack(cols, dim=-1)
    if H != new_h:
        wt = aa_weights(H, new_h)
        rows = []
        for ymin, ws in wt:
            acc = torch.zeros(x.shape[:-2] + x.shape[-1:], dtype=torch.float32)
            for j, w in enumerate(ws):
                acc = acc + x[..., ymin + j, :] * float(w)
            rows.append(acc)
        x = torch.stack(rows, dim=-2)
    return x


def center_crop_offsets(H: int, W: int, h: int, w: int):
    assert H >= h and W >= w, "the reference pads smaller images; no VidTok script feeds one"
    return int(round((H - h) / 2.0)), int(round((W - w) / 2.0))


def preprocess_frames(frames_u8: torch.Tensor, input_height: int, input_width: int, resize_fn=resize_aa) -> torch.Tensor:
    """uint8 [T, H0, W0, 3] -> fp32 [3, T, h, w] in [-1, 1]: R/scripts/inference_reconstruct.py:39-45, 70-73."""
    x = frames_u8.permute(0, 3, 1, 2).float() / 255.0
    H0, W0 = x.shape[-2:]
    nh, nw = resized_size(H0, W0, input_height)
    if (nh, nw) != (H0, W0):
        x = resize_fn(x, nh, nw)
    top, left = center_crop_offsets(nh, nw, input_height, input_width)
    x = x[..., top:top + input_height, left:left + input_width]
    x = (x - 0.5) / 0.5
    return x.permute(1, 0, 2, 3).contiguous()


def tensor_to_uint8(t: torch.Tensor) -> np.ndarray:
    """R/scripts/inference_reconstruct.py:76-80: clamp, (x+1)/2, *255 in fp32, truncation to uint8."""
    t = torch.clamp(t, -1.0, 1.0)
    t = (t + 1.0) / 2.0
    return (t.cpu().numpy() * 255).astype(np.uint8)


# --------------------------------------------------------------------------------------------------
# the reconstruction loop
# --------------------------------------------------------------------------------------------------
def reconstruct(model: Callable, clips: List[torch.Tensor], is_causal: bool, time_downsample_factor: int,
                read_long_video: bool, pad_gen_frames: bool, concate_input: bool) -> np.ndarray:
    """R/scripts/inference_reconstruct.py:206-239.  `clips` = what the DataLoader yields: fp32 [1, 3, T, h, w] each;
    model(x) -> (z, xrec, log).  Returns the uint8 frames [N, h, w or 2w, 3] handed to write_video."""
    inputs, outputs = [], []
    last_gen = None
    for i, x in enumerate(clips):
        if is_causal and not read_long_video and pad_gen_frames:
            if i == 0:
                _, xrec, _ = model(x)
            else:
                _, xrec, _ = model(torch.cat([last_gen, x], dim=2))
            xrec = xrec[:, :, -x.shape[2]:].clamp(-1, 1)
            last_gen = xrec[:, :, (1 - time_downsample_factor):, :, :]
        else:
            _, xrec, _ = model(x)
        inputs.append(x.permute(0, 2, 1, 3, 4).reshape((-1,) + tuple(x.shape[1:2]) + tuple(x.shape[3:])))
        xr = xrec.clamp(-1, 1)
        outputs.append(xr.permute(0, 2, 1, 3, 4).reshape((-1,) + tuple(xr.shape[1:2]) + tuple(xr.shape[3:])))
    inp = np.transpose(tensor_to_uint8(torch.cat(inputs, dim=0)), (0, 2, 3, 1))
    out = np.transpose(tensor_to_uint8(torch.cat(outputs, dim=0)), (0, 2, 3, 1))
    n = min(inp.shape[0], out.shape[0])
    return np.concatenate([inp[:n], out[:n]], axis=2) if concate_input else out[:n]
