"""GPU check: the weight-stationary kernels must be deterministic and independent of how tiles / columns are split
over workgroups (run-to-run bit equality, batch slice == batch-of-one)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vidtok_amd import lib as L  # noqa: E402
from vidtok_amd import ops  # noqa: E402
from vidtok_amd.ops import ConvGeom  # noqa: E402

dev, dt, C_ = "cuda", torch.bfloat16, 128
torch.manual_seed(0)
G3 = ConvGeom(kh=3, kw=3, ph=1, pw=1, ph_hi=1, pw_hi=1)


def eq(a, b):
    a = a if isinstance(a, tuple) else (a,)
    b = b if isinstance(b, tuple) else (b,)
    return all(torch.equal(u, v) for u, v in zip(a, b))


for shape in [(2, 5, 128, 128), (2, 17, 256, 256)]:
    B, T, H, W = shape
    x = torch.randn((B, T, H, W, C_), device=dev).to(dt)
    res = torch.randn((B, T, H, W, C_), device=dev).to(dt)
    w = (torch.randn((C_, 9 * C_), device=dev) / math.sqrt(9 * C_)).to(dt)
    bias = torch.randn((C_,), device=dev)
    ln = (torch.ones(C_, device=dev), torch.zeros(C_, device=dev), 1e-6, True)
    for name, kw in [("plain", {}), ("res", dict(res=res, res_mode=L.VT_RES_ADD)), ("res+ln", dict(res=res, res_mode=L.VT_RES_ADD, ln=ln)),
                     ("ln only", dict(ln=ln, ln_keep_y=False))]:
        outs = [ops.conv(x, w, bias, G3, cout=C_, **kw) for _ in range(4)]
        torch.cuda.synchronize()
        rep = all(eq(outs[0], o) for o in outs[1:])
        kw1 = dict(kw)
        if "res" in kw1:
            kw1["res"] = res[1:2].contiguous()
        one = ops.conv(x[1:2].contiguous(), w, bias, G3, cout=C_, **kw1)
        o0 = outs[0] if isinstance(outs[0], tuple) else (outs[0],)
        sl = tuple(t[1:2] for t in o0)
        bi = eq(sl, one)
        with L.options(conv_ws=0):
            ig = ops.conv(x, w, bias, G3, cout=C_, **kw)
        ig = ig if isinstance(ig, tuple) else (ig,)
        dmax = max(float((a.float() - b.float()).abs().max()) for a, b in zip(o0, ig))
        print(f"ws128 {shape} {name:8s}: repeatable {rep}, batch-independent {bi}, max |ws - igemm| {dmax:.3e}")
    ws = [(torch.randn((C_, 3 * C_), device=dev) / math.sqrt(3 * C_)).to(dt) for _ in range(2)]
    bs = [torch.randn((C_,), device=dev) for _ in range(2)]
    nm = (torch.ones(C_, device=dev) * 1.1, torch.zeros(C_, device=dev) + 0.1)
    for nxt in (None, (nm[0], nm[1], True)):
        outs = [ops.temporal_block(x, ws[0], bs[0], ws[1], bs[1], nm, nm, next_ln=nxt) for _ in range(4)]
        torch.cuda.synchronize()
        rep = all(eq(outs[0], o) for o in outs[1:])
        one = ops.temporal_block(x[1:2].contiguous(), ws[0], bs[0], ws[1], bs[1], nm, nm, next_ln=nxt)
        o0 = outs[0] if isinstance(outs[0], tuple) else (outs[0],)
        bi = eq(tuple(t[1:2] for t in o0), one)
        print(f"tblock {shape} next={nxt is not None}: repeatable {rep}, batch-independent {bi}")
