"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's evaluation metrics.
Pinned against the unmodified reference by tests/test_oracle_vs_reference.py (build container) and against the
committed outputs of the reference in tests/golden/metrics.safetensors (anywhere).  Only tests/ may import this.

Follows, per frame:
  post-processing  scripts/inference_evaluate.py:175-176   out.clamp(-1,1); x,out -> (.+1)/2
  PSNR             vidtok/modules/util.py:146-154          -10 log10(mean_{c,h,w}(x-y)^2 + 1e-8)
  SSIM             vidtok/modules/util.py:157-222,306-324  11x11 Gaussian sigma 1.5, valid depthwise conv,
                                                           k1 .01, k2 .03, f x f avg-pool, f = max(1, round(min(H,W)/256))
"""
import torch
import torch.nn.functional as F


def postprocess(x, y):
    return (x + 1) / 2, (y.clamp(-1, 1) + 1) / 2


def gaussian_kernel2d(size=11, sigma=1.5, dtype=torch.float32):
    coords = torch.arange(size, dtype=dtype) - (size - 1) / 2.0
    g = coords ** 2
    g = (-(g.unsqueeze(0) + g.unsqueeze(1)) / (2 * sigma ** 2)).exp()
    return g / g.sum()


def psnr_frames(x, y):
    """x, y NCTHW already post-processed -> [B, T]"""
    mse = ((x - y) ** 2).mean(dim=(1, 3, 4))
    return -10 * torch.log10(mse + 1e-8)


def ssim_frames(x, y):
    B, C, T, H, W = x.shape
    xf = x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W)
    yf = y.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W)
    f = max(1, round(min(H, W) / 256))
    if f > 1:
        xf, yf = F.avg_pool2d(xf, kernel_size=f), F.avg_pool2d(yf, kernel_size=f)
    if xf.shape[-1] < 11 or xf.shape[-2] < 11:
        raise ValueError("Kernel size can't be greater than actual input size")
    k = gaussian_kernel2d(dtype=x.dtype).expand(C, 1, 11, 11).contiguous()
    filt = lambda v: F.conv2d(v, k, groups=C)
    mu_x, mu_y = filt(xf), filt(yf)
    s_xx = filt(xf * xf) - mu_x * mu_x
    s_yy = filt(yf * yf) - mu_y * mu_y
    s_xy = filt(xf * yf) - mu_x * mu_y
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    cs = (2.0 * s_xy + c2) / (s_xx + s_yy + c2)
    ss = (2.0 * mu_x * mu_y + c1) / (mu_x * mu_x + mu_y * mu_y + c1) * cs
    return ss.mean(dim=(-1, -2)).mean(1).reshape(B, T)


def eval_psnr_ssim(x, y):
    """raw model input / output in [-1,1] (NCTHW) -> per-frame (psnr, ssim), each [B, T]"""
    xp, yp = postprocess(x, y)
    return psnr_frames(xp, yp), ssim_frames(xp, yp)
