"""Shared helpers of the test-suite (TEST INFRASTRUCTURE)."""
import math
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIG_DIR = os.path.join(ROOT, "configs")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def config_path(name):
    return os.path.join(CONFIG_DIR, name + ".yaml")


def seeded_state_dict(shapes: dict, seed: int = 1234) -> dict:
    """Deterministic, numerically non-trivial weights for a VidTok state_dict (SURVEY.md finding 3:
    a fresh reference model has zero temporal conv2 and identity LayerNorms).  Depends only on the key
    names/shapes and the torch CPU generator, so the build container (where the reference produces the
    golden outputs) and the GPU box regenerate bit-identical tensors."""
    out = {}
    for i, name in enumerate(sorted(shapes)):
        shape = tuple(shapes[name])
        g = torch.Generator().manual_seed(seed * 100003 + i)
        if name.endswith("mix_factor"):
            t = 0.3 + 0.5 * torch.randn(shape, generator=g)
        elif ".norm" in name and name.endswith(".weight") and len(shape) == 1:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            t = 0.05 * torch.randn(shape, generator=g)
        else:
            fan_in = max(1, int(math.prod(shape[1:])))
            t = torch.randn(shape, generator=g) / math.sqrt(fan_in)
        out[name] = t
    return out


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max |a-b| / max |b| -- the parity metric of SURVEY.md section 8(d)."""
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def build_model(name: str, seed: int = 1234, device="cpu", dtype=torch.float32, overrides=None, reg_overrides=None):
    """vidtok_amd engine from configs/<name>.yaml with seeded weights; returns (model, cfg, state_dict).
    `overrides` / `reg_overrides` update the encoder+decoder / regularizer params (variants without a YAML)."""
    import vidtok_amd

    cfg = vidtok_amd.load_config(config_path(name))
    prm = cfg["model"]["params"]
    if overrides:
        prm["encoder_config"]["params"].update(overrides)
        if isinstance(prm["decoder_config"]["params"], dict):
            prm["decoder_config"]["params"].update(overrides)
    if reg_overrides:
        prm["regularizer_config"].setdefault("params", {}).update(reg_overrides)
    model = vidtok_amd.load_model_from_config(cfg, verbose=False)
    sd = seeded_state_dict({k: v.shape for k, v in model.state_dict().items()}, seed)
    missing, unexpected = model.load_state_dict(sd, strict=True)
    model = model.to(device).eval()
    model.set_compute_dtype(dtype)
    return model, cfg, sd


def build_oracle(cfg: dict, sd: dict):
    from oracle.vidtok_oracle import OracleEngine

    return OracleEngine(cfg["model"]["params"], sd)


def handle_config(L, enc, reg_target, reg_params, enc_target=""):
    """vt_model_config from the constructor arguments of the reference's YAML (the defaults of EncoderCausal3D /
    DecoderCausal3D for the lists the YAML leaves out, model_3dcausal.py:560-566,738-741)"""
    c = L.ModelConfig()
    n = len(enc["ch_mult"])
    c.version = 1 if enc_target.endswith("V11") else (2 if "noncausal" in enc_target else 0)
    c.interpolation_mode = {"nearest": 0, "trilinear": 1}[enc.get("interpolation_mode", "nearest")] if c.version else 0
    c.ch, c.num_res_blocks, c.in_channels, c.out_ch, c.z_channels = enc["ch"], enc["num_res_blocks"], enc["in_channels"], enc["out_ch"], enc["z_channels"]
    c.double_z, c.num_resolutions = int(enc.get("double_z", True)), n
    lists = dict(ch_mult=enc["ch_mult"], spatial_ds=enc.get("spatial_ds") or list(range(0, n - 1)), tempo_ds=enc.get("tempo_ds") or [n - 2, n - 3],
                 spatial_us=enc.get("spatial_us") or list(range(1, n)), tempo_us=enc.get("tempo_us") or [1, 2])
    for k, v in lists.items():
        for i, e in enumerate(v):
            getattr(c, k)[i] = int(e)
        if k != "ch_mult":
            setattr(c, "n_" + k, len(v))
    c.time_downsample_factor = enc.get("time_downsample_factor", 4)
    c.norm_type = {"layernorm": 0, "groupnorm": 1}[enc.get("norm_type", "layernorm")]
    if reg_target.endswith("FSQRegularizer"):
        c.regularizer, c.n_levels = 1, len(reg_params["levels"])
        for i, e in enumerate(reg_params["levels"]):
            c.levels[i] = int(e)
        c.fsq_num_codebooks = int(reg_params.get("num_codebooks", 1))
        c.fsq_dim = int(reg_params.get("dim") or 0)
    return c


def fsq_mismatch_report(levels, h, h_ref, idx, idx_ref, limit=16):
    """Forensics of FSQ code mismatches ("report count of mismatches and their distance to a rounding
    boundary").  h / h_ref: pre-quantisation encoder outputs [B, D, T, H, W] (ours / the checker's), idx / idx_ref: integer
    codes [B, T, H, W].  For every token whose code differs, and every channel of it whose digit differs: the bounded value
    b = tanh(h + shift) * half_l - offset the digit is rounded from (reference regularizers.py:153-163), its distance to the
    nearest rounding boundary k + 1/2 in fp32 ulps of b, and |h - h_ref| in fp32 ulps of h_ref.  A digit can only flip where
    the checker's own value sits within the two paths' distance of a boundary; the report shows exactly that."""
    import numpy as np

    lv = torch.tensor([int(v) for v in levels], dtype=torch.float64)
    half_l = (lv - 1) * (1 + 1e-3) / 2
    offset = torch.where(lv % 2 == 0, 0.5, 0.0).double()
    shift = (offset / half_l).atanh()
    idx, idx_ref = idx.cpu().long(), idx_ref.cpu().long()
    bad = (idx != idx_ref).nonzero()
    out = {"mismatches": int(bad.shape[0]), "tokens": int(idx_ref.numel()), "detail": []}
    basis = torch.cumprod(torch.tensor([1] + [int(v) for v in levels[:-1]]), 0)
    for b, t, y, x in bad.tolist()[:limit]:
        hv, hr = h[b, :, t, y, x].double().cpu(), h_ref[b, :, t, y, x].double().cpu()
        d_ours = (idx[b, t, y, x] // basis) % lv.long()
        d_ref = (idx_ref[b, t, y, x] // basis) % lv.long()
        for c in (d_ours != d_ref).nonzero().flatten().tolist():
            bo, br = (float(torch.tanh(v[c] + shift[c]) * half_l[c] - offset[c]) for v in (hv, hr))
            edge = np.floor(br) + 0.5
            ulp_b = float(np.spacing(np.float32(abs(br))))
            ulp_h = float(np.spacing(np.float32(abs(float(hr[c])))))
            out["detail"].append({"token": [b, t, y, x], "channel": c, "digit": int(d_ours[c]), "digit_ref": int(d_ref[c]),
                                  "bounded_ref": round(br, 9), "bounded": round(bo, 9),
                                  "ref_to_boundary_ulps": round(float(abs(br - edge) / ulp_b), 2),
                                  "ours_to_boundary_ulps": round(float(abs(bo - edge) / ulp_b), 2),
                                  "h_diff_ulps": round(float(abs(float(hv[c]) - float(hr[c])) / ulp_h), 2)})
    return out


_FP16_PROBE = {}


def cpu_autocast_usable(dtype, factor=6.0):
    """Can the oracle be run under torch.autocast("cpu", dtype) in reasonable time on THIS host?  torch's CPU convolutions in float16 fall
    back to a scalar path on hosts without fp16 vector support (the first GPU-box run of round 6 sat in one such call for the rest of its
    45 minutes): a 3x3x3 convolution 64 -> 64 on a 3 x 32 x 32 block is timed in fp32 and under autocast(dtype), once per session; callers skip
    the autocast comparison (not the kernel checks against the fp32 oracle) when the autocast form is more than `factor` times slower."""
    import time

    if dtype not in _FP16_PROBE:
        x, w = torch.randn(1, 64, 3, 32, 32), torch.randn(64, 64, 3, 3, 3)

        def run(ctx):
            with ctx:
                torch.nn.functional.conv3d(x[:, :, :1, :8, :8], w, padding=1)        # warm-up (dispatch, thread pool)
                t0 = time.perf_counter()
                torch.nn.functional.conv3d(x, w, padding=1)
                return time.perf_counter() - t0

        import contextlib

        t32 = run(contextlib.nullcontext())
        t16 = run(torch.autocast("cpu", dtype=dtype))
        _FP16_PROBE[dtype] = t16 < max(0.25, factor * t32)
        print(f"[cpu_autocast_usable] conv3d 3x3x3 64->64 on 3x32x32: fp32 {t32:.3f} s, autocast({dtype}) {t16:.3f} s -> usable {_FP16_PROBE[dtype]}")
    return _FP16_PROBE[dtype]
