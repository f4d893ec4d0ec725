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


class PackedCache:
    """Caches the packed weight / fp32 bias of one conv-like parameter holder."""

    def __init__(self):
        self._key = None
        self._val = None

    def get(self, weight: torch.nn.Parameter, bias, dtype, cin_stored=None):
        key = (dtype, weight.device, weight._version, None if bias is None else bias._version, cin_stored,
               weight.data_ptr())
        if key != self._key:
            w = pack_conv_weight(weight, dtype, cin_stored)
            b = None if bias is None else bias.detach().to(torch.float32).contiguous()
            self._key, self._val = key, (w, b)
        return self._val
