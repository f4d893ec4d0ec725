"""Weight re-packing: reference parameter layouts -> the [Cout][taps*Cin_p] rows vt_conv reads.

The reference stores nn.Conv3d / Conv2d / Conv1d weights as [Cout, Cin, kT, kH, kW] /
[Cout, Cin, kH, kW] / [Cout, Cin, k] (state_dict shapes listed in SURVEY.md section 8b).  vt_conv wants
k = ((kt*KH + kh)*KW + kw)*Cin_p + c contiguous per output channel, in the arithmetic dtype,
with Cin zero-padded to the activation's stored channel count.  Packing is a one-time permute /
pad / cast done with torch tensor ops (data movement only) and cached per (dtype, version).
"""
import torch

from .ops import pad_channels


def pack_conv_weight(weight: torch.Tensor, dtype: torch.dtype, cin_stored: int = None) -> torch.Tensor:
    """weight [Cout, Cin, *k] (1, 2 or 3 kernel dims) -> [Cout, taps * cin_stored] contiguous."""
    cout, cin = weight.shape[:2]
    cin_p = cin_stored or pad_channels(cin)
    nd = weight.dim() - 2
    w = weight.detach().to(torch.float32)
    # channels-last: [Cout, *k, Cin]
    perm = (0,) + tuple(range(2, 2 + nd)) + (1,)
    w = w.permute(*perm)
    if cin_p != cin:
        w = torch.nn.functional.pad(w, (0, cin_p - cin))
    return w.reshape(cout, -1).to(dtype).contiguous()


ARITH_SPLIT3 = "bf16x3"
SPLIT3_DTYPE = torch.int32        # element type of a split-bf16 weight container (4 bytes per k, like the fp32 row it replaces)


def pack_split3(w2d: torch.Tensor) -> torch.Tensor:
    """Packed fp32 rows [Cout, K] -> the operand of vt_conv's VT_BF16X3 arithmetic (include/vidtok_amd.h): every value as
    two bf16 planes, hi = bf16(w) (round to nearest even) and lo = bf16(w - hi), stored per group of 16 k as
    [hi 16 x bf16 | lo 16 x bf16] (64 bytes, the size of the 16 fp32 values they replace), K zero-padded to 32 (a whole
    128-byte K step of the kernel).  Returned as an INT32-typed container [Cout, K32] (ldw = K32): the element type is the
    tag ops.conv reads (`SPLIT3_DTYPE`) -- unlike a Python attribute it survives .view / .contiguous / slicing, and a
    split container can never be taken for fp32 rows.  Data movement + two roundings."""
    cout, K = w2d.shape
    Kp = (K + 31) // 32 * 32
    w = w2d.detach().to(torch.float32)
    if Kp != K:
        w = torch.nn.functional.pad(w, (0, Kp - K))
    hi = w.to(torch.bfloat16)
    lo = (w - hi.to(torch.float32)).to(torch.bfloat16)
    planes = torch.stack([hi.view(cout, Kp // 16, 16), lo.view(cout, Kp // 16, 16)], dim=2).contiguous()   # [Cout, K/16, 2, 16]
    return planes.view(SPLIT3_DTYPE).reshape(cout, Kp)


def set_arith(root: torch.nn.Module, arith):
    """Select the weight arithmetic of every convolution under `root` for fp32 storage: None = fp32 rows (fp32 MFMA),
    ARITH_SPLIT3 = split-bf16 planes (three bf16 MFMAs per product).  Visits the PackedCache objects the modules hold
    (directly or inside tuples / lists); caches created with pin_native=True (operands of an activation x activation GEMM)
    keep fp32 rows."""
    assert arith in (None, ARITH_SPLIT3)

    def visit(v):
        if isinstance(v, PackedCache):
            if not v.pin_native:
                v.arith = arith
        elif isinstance(v, (tuple, list)):
            for e in v:
                visit(e)

    for m in root.modules():
        for v in m.__dict__.values():
            visit(v)


def time_upsample_parity_weights(weight: torch.Tensor, early: bool):
    """A k=3 temporal conv over a nearest-x2 frame-repeated input u[t] = x[t >> 1] touches only two input frames per
    output frame, so it equals a k=2 conv over x with pre-summed taps (fp32 sums, rounded once when packed):
      window (j-1, j)  when the three taps cover u[2j-2 .. 2j]   : [W0 + W1, W2]   `early=True`
                       or                 u[2j-1 .. 2j+1] causal : [W0, W1 + W2]   `early=False`
    (the caller pairs each output parity with its window, see TimeUpsampleRes*2x).  weight [Co, Ci, 3, kh, kw]."""
    w = weight.detach().to(torch.float32)
    assert w.dim() == 5 and w.shape[2] == 3
    if early:
        return torch.stack([w[:, :, 0] + w[:, :, 1], w[:, :, 2]], dim=2)
    return torch.stack([w[:, :, 0], w[:, :, 1] + w[:, :, 2]], dim=2)


def space_upsample_parity_weights(weight: torch.Tensor, py: int, px: int):
    """A 3x3 conv (pad 1) over a nearest-x2 up-sampled frame v[Y][X] = x[Y>>1][X>>1] reads, for an output pixel of
    parity (py, px), only a 2x2 window of x: rows (a-1, a) with taps [W0, W1+W2] for py = 0, rows (a, a+1) with
    [W0+W1, W2] for py = 1, likewise for columns.  weight [Co, Ci, 3, 3] -> [Co, Ci, 2, 2] (fp32 sums)."""
    w = weight.detach().to(torch.float32)
    assert w.dim() == 4 and tuple(w.shape[2:]) == (3, 3)
    rows = torch.stack([w[:, :, 0], w[:, :, 1] + w[:, :, 2]], dim=2) if py == 0 else \
        torch.stack([w[:, :, 0] + w[:, :, 1], w[:, :, 2]], dim=2)                     # [Co, Ci, 2, 3]
    return torch.stack([rows[..., 0], rows[..., 1] + rows[..., 2]], dim=3) if px == 0 else \
        torch.stack([rows[..., 0] + rows[..., 1], rows[..., 2]], dim=3)               # [Co, Ci, 2, 2]


def time_upsample_parity_mix(kdims, early: bool):
    """the tap sums of time_upsample_parity_weights as a mix table for vt_pack_conv_weight: kdims = (3, kh, kw) of the reference
    weight -> 2 * kh * kw output taps, each [m0, m1] (indices into the 3 * kh * kw reference taps)"""
    kt, kh, kw = kdims
    assert kt == 3
    s = kh * kw
    first = [[0 * s + i, 1 * s + i] if early else [0 * s + i] for i in range(s)]
    second = [[2 * s + i] if early else [1 * s + i, 2 * s + i] for i in range(s)]
    return first + second


def space_upsample_parity_mix(kdims, py: int, px: int):
    """the tap sums of space_upsample_parity_weights as a mix table: (3, 3) -> 4 output taps (r, c), each (w[r0][c0] + w[r1][c0]) +
    (w[r0][c1] + w[r1][c1]) over its row set and column set -- rows first, then columns, as the host statement sums them"""
    assert tuple(kdims) == (3, 3)
    rows = [[0], [1, 2]] if py == 0 else [[0, 1], [2]]
    cols = [[0], [1, 2]] if px == 0 else [[0, 1], [2]]
    out = []
    for R in rows:
        for Cc in cols:
            m = [-1, -1, -1, -1]
            for ci, c in enumerate(Cc):
                for ri, r in enumerate(R):
                    m[2 * ci + ri] = r * 3 + c
            out.append(m)
    return out


class PackedCache:
    """Caches the packed weight / fp32 bias of one conv-like parameter holder.  `transform` (optional) maps the
    parameter tensor to the tensor that is packed (e.g. the parity weights of a time up-sampler); `mix` is the same
    transform as a tap-sum table (kernel dims -> [[m0..m3], ...]) for the device packer.  Parameters that live on the GPU
    are packed there by vt_pack_conv_weight (same bits as the host statements; no foreign kernel in the process);
    host parameters (CPU tests, tools) by the torch statements above."""

    def __init__(self, transform=None, pin_native=False, mix=None):
        self._entries = {}           # (dtype, arith, cin_stored) -> (validity key, (w, b)): the packed rows of every mode the module has run
        self._transform = transform  # in stay alive, so a graph captured in one mode still replays after a detour through another
        self._mix = mix
        assert (transform is None) == (mix is None)
        self.arith = None            # set_arith(): ARITH_SPLIT3 = fp32 requests are answered with split-bf16 planes
        self.pin_native = pin_native

    def get(self, weight: torch.nn.Parameter, bias, dtype, cin_stored=None):
        arith = self.arith if dtype == torch.float32 else None
        key = (dtype, weight.device, weight._version, None if bias is None else bias._version, cin_stored,
               weight.data_ptr(), arith)
        slot = (dtype, arith, cin_stored)
        ent = self._entries.get(slot)
        if ent is None or ent[0] != key:
            if weight.is_cuda:
                from . import ops

                wd = weight.detach()
                wd = wd if wd.dtype == torch.float32 and wd.is_contiguous() else wd.float().contiguous()
                mix = None if self._mix is None else self._mix(tuple(wd.shape[2:]))
                w = ops.pack_conv_weight(wd, dtype, cin_stored, mix=mix, split3=arith == ARITH_SPLIT3)
            else:
                w = pack_conv_weight(weight if self._transform is None else self._transform(weight), dtype, cin_stored)
                if arith == ARITH_SPLIT3:
                    w = pack_split3(w)
            b = None if bias is None else bias.detach().to(torch.float32).contiguous()
            ent = self._entries[slot] = (key, (w, b))
        return ent[1]
