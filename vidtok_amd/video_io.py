"""Video front / back end around `model(x)`: what scripts/inference_reconstruct.py of the reference does between the
codec and the model.  The codec itself (decord / torchvision.io.write_video; neither is
in this image) stays with the caller: `VideoReconstructor.reconstruct` takes the decoded uint8 frames and returns the
uint8 frames the reference hands to `write_video`.  Everything between -- /255, Resize(antialias) + CenterCrop +
Normalize, the per-clip model calls with `--pad_gen_frames` chaining, clamp + uint8 conversion, side-by-side
concatenation -- runs on the GPU through the C-ABI (vt_frames_u8_to_ncthw, vt_ncthw_copy_frames,
vt_ncthw_to_frames_u8); no frame goes back to the host in between.

Reference: SingleVideoDataset (inference_reconstruct.py:28-73), tensor_to_uint8 (:76-80), main loop (:206-239);
the same transform in vidtok/data/vidtok.py:180-188."""
from typing import List

import torch

from . import ops


def frame_id_batches(total_frames: int, fps: float, sample_fps: int, chunk_size: int, is_causal: bool,
                     read_long_video: bool) -> List[List[int]]:
    """Frame indices of each clip (inference_reconstruct.py:49-66): every round(fps / sample_fps)-th frame, in clips of
    chunk_size (+1 when causal), incomplete tails dropped; one long clip with `read_long_video`."""
    interval = round(fps / sample_fps)
    ids = list(range(0, total_frames, interval))
    out = []
    if read_long_video:
        n = len(ids)
        if is_causal and n > chunk_size:
            out.append(ids[:chunk_size * ((n - 1) // chunk_size) + 1])
        elif not is_causal and n >= chunk_size:
            out.append(ids[:chunk_size * (n // chunk_size)])
        return out
    per = chunk_size + 1 if is_causal else chunk_size
    for s in range(0, len(ids), per):
        if len(ids[s:s + per]) == per:
            out.append(ids[s:s + per])
    return out


def resized_size(h: int, w: int, size: int):
    """torchvision Resize(int): the shorter side becomes `size`, the longer one int(size * long / short) -> (new_h, new_w)"""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def center_crop_offsets(H: int, W: int, h: int, w: int):
    """torchvision CenterCrop: top = int(round((H - h) / 2.0)) with Python's round (half to even)"""
    if H < h or W < w:
        raise ValueError(f"frame {H}x{W} is smaller than the crop {h}x{w} after resizing (the reference would pad)")
    return int(round((H - h) / 2.0)), int(round((W - w) / 2.0))


def preprocess_frames(frames_u8: torch.Tensor, input_height: int, input_width: int, out=None, t_off: int = 0):
    """uint8 [T, H0, W0, 3] (device) -> fp32 [1, 3, T, h, w] in [-1, 1] (SingleVideoDataset.transform + permutes)"""
    H0, W0 = frames_u8.shape[1:3]
    nh, nw = resized_size(H0, W0, input_height)
    top, left = center_crop_offsets(nh, nw, input_height, input_width)
    return ops.frames_u8_to_ncthw(frames_u8.contiguous(), (nh, nw), (top, left), (input_height, input_width), out=out, t_off=t_off)


class VideoReconstructor:
    """The loop of scripts/inference_reconstruct.py:197-235 on device tensors."""

    def __init__(self, model, input_height=256, input_width=256, sample_fps=30, chunk_size=16, read_long_video=False,
                 pad_gen_frames=False, concate_input=True):
        self.model = model
        self.h, self.w = input_height, input_width
        self.sample_fps, self.chunk_size = sample_fps, chunk_size
        self.read_long_video, self.pad_gen_frames, self.concate_input = read_long_video, pad_gen_frames, concate_input
        f = model.encoder.time_downsample_factor
        assert chunk_size % f == 0
        if read_long_video:                                   # :187-193
            assert hasattr(model, "use_tiling"), "Tiling inference is needed to conduct long video reconstruction."
            model.use_tiling = True
            model.t_chunk_enc = chunk_size
            model.t_chunk_dec = chunk_size // f
            model.use_overlap = True

    @torch.no_grad()
    def reconstruct(self, frames_u8: torch.Tensor, fps: float) -> torch.Tensor:
        """frames_u8 uint8 [N, H0, W0, 3] on the device (the decoded video), fps of the file -> uint8
        [n, h, w | 2w, 3] on the device: the frames the reference writes with write_video(..., sample_fps)."""
        model = self.model
        f = model.encoder.time_downsample_factor
        batches = frame_id_batches(frames_u8.shape[0], fps, self.sample_fps, self.chunk_size, model.is_causal,
                                   self.read_long_video)
        chain = model.is_causal and not self.read_long_video and self.pad_gen_frames
        total = sum(len(b) for b in batches)
        wtot = 2 * self.w if self.concate_input else self.w
        out = torch.empty((total, self.h, wtot, 3), dtype=torch.uint8, device=frames_u8.device)
        last_rec, done = None, 0
        for i, ids in enumerate(batches):
            clip_u8 = frames_u8[ids[0]:ids[-1] + 1:ids[1] - ids[0]] if len(ids) > 1 else frames_u8[ids[0]:ids[0] + 1]
            T = len(ids)
            if chain and i > 0:
                # model(cat([last f-1 generated frames, input])): the clip buffer is filled in place by the two kernels
                x = torch.empty((1, 3, f - 1 + T, self.h, self.w), dtype=torch.float32, device=frames_u8.device)
                ops.ncthw_copy_frames(last_rec, x, last_rec.shape[2] - (f - 1), 0, f - 1, clamp=True)
                preprocess_frames(clip_u8, self.h, self.w, out=x, t_off=f - 1)
            else:
                x = preprocess_frames(clip_u8, self.h, self.w)
            _, xrec, _ = model(x)
            if chain:
                last_rec = xrec                                  # its last f-1 frames, clamped, seed the next clip
            t_in0 = x.shape[2] - T
            if self.concate_input:
                ops.ncthw_to_frames_u8(x, t0=t_in0, n=T, out=out[done:done + T], w_off=0)
            ops.ncthw_to_frames_u8(xrec, t0=xrec.shape[2] - T, n=T, out=out[done:done + T], w_off=self.w if self.concate_input else 0)
            done += T
        return out
