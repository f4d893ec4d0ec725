"""Host mirror of the reference's evaluation metrics (vidtok/modules/util.py:146-178) on the HIP kernels of
csrc/metrics.hip -- same names, same argument convention ([0,1] images, 4-D NCHW or 5-D NCTHW), same scalar result
(mean over all frames).  `evaluate_clip` is the per-batch body of the eval loop (scripts/inference_evaluate.py:167-192)
with the post-processing fused into the kernel."""
import torch

from . import ops


def _as5d(x):
    if x.dim() == 4:    # (N, C, H, W): frames along the batch axis
        return x.unsqueeze(2)
    assert x.dim() == 5
    return x


def _frames(x, y, raw):
    x, y = _as5d(x), _as5d(y)
    assert x.shape == y.shape
    return ops.eval_psnr_ssim(x.float().contiguous(), y.float().contiguous(), raw=raw)


def compute_psnr(x, y):
    return _frames(x, y, False)[0].mean()


def compute_ssim(x, y):
    return _frames(x, y, False)[1].mean()


def evaluate_clip(model, x):
    """x: NCTHW in [-1,1] on the GPU.  Returns (reconstruction, psnr [B,T], ssim [B,T]); the reference's running
    averages over 16-frame splits are means of these per-frame values."""
    with torch.no_grad():
        _, xrec, _ = model(x)
        xrec = xrec[:, :, -x.shape[2]:] if xrec.shape[2] != x.shape[2] else xrec
        psnr, ssim = _frames(x, xrec, True)
    return xrec, psnr, ssim
