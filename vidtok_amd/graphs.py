"""Per-shape hipGraph cache for the encoder / decoder launch sequences.

An encode or decode of one clip batch is 150-250 kernel launches issued from Python; eager, the host paces the GPU
(measured round 1: FSQ 559 frames/s eager vs 680 captured).  `GraphedCall` wraps a pure device function
`f(x) -> tensor` (EncoderCausal3DPadding.forward, DecoderCausal3DPadding.forward): the first call of a shape runs
eagerly (it also packs the weights), the second is captured into a hipGraph on a side stream, later calls copy the
input into the graph's static buffer and replay.  Outputs are returned as fresh tensors (a device-to-device copy of the
graph's static output), so the reference's "caller owns the outputs" contract holds.

Opt-in (`AutoencodingEngine.enable_graphs()`): a replay does not see in-place edits of parameters made after the
capture -- `load_state_dict`, `.to()`, `set_compute_dtype` and `enable_graphs` reset the cache, anything else needs
`invalidate_graphs()`.

Stateful chunked passes (v1.1 temporal tiling) replay too: the modules keep their chunk-to-chunk caches in persistent
buffers that are updated in place (vidtok_amd/modules.py::_CausalState), so a captured chunk reads and writes the same
addresses on every replay.  Such a call (`stateful=True`) must run exactly once per chunk: the usual side-stream warm-up
run before the capture is skipped (the first, eager call of the shape already packed the weights), and capturing does
not execute.  `frames=(start, end)` passes the chunk as a frame range of a larger NCTHW fp32 tensor (copied straight
into the graph's input by vt_ncthw_copy_frames); `borrow=True` returns the graph's own output tensor -- valid until the
next call of the same shape -- instead of a copy.
"""
import torch


class GraphedCall:
    def __init__(self, fn, state_get=None, state_set=None):
        self.fn = fn
        self.entries = {}
        # stateful calls: host-side state the function leaves behind (module attributes bound to cache buffers), read
        # after the capture and re-applied after every replay -- a replay runs no Python
        self.state_get, self.state_set = state_get, state_set

    def clear(self):
        self.entries.clear()

    def __call__(self, x, key_extra=(), stateful=False, frames=None, borrow=False):
        if not x.is_cuda:
            return self.fn(x if frames is None else x[:, :, frames[0]:frames[1]].contiguous())
        x = x.contiguous()
        if frames is not None:
            from . import ops

            x = x.float()
            shape = tuple(x.shape[:2]) + (frames[1] - frames[0],) + tuple(x.shape[3:])

            def fill(dst):
                return ops.ncthw_copy_frames(x, dst, frames[0], 0, frames[1] - frames[0])
        else:
            shape = tuple(x.shape)

            def fill(dst):
                return dst.copy_(x)
        key = (shape, x.dtype, x.device, key_extra)
        e = self.entries.get(key)
        if e is None:                       # first sight of this shape: eager (packs weights, sizes the allocator)
            y = self.fn(x if frames is None else fill(torch.empty(shape, dtype=x.dtype, device=x.device)))
            self.entries[key] = "warm"
            return y
        if e == "warm":
            sx = fill(torch.empty(shape, dtype=x.dtype, device=x.device))
            if not stateful:
                cur = torch.cuda.current_stream()
                side = torch.cuda.Stream()
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    self.fn(sx)
                cur.wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                sy = self.fn(sx)
            self.entries[key] = e = (g, sx, sy, self.state_get() if stateful and self.state_get else None)
        g, sx, sy, state = e
        fill(sx)
        g.replay()
        if state is not None:
            self.state_set(state)
        return sy if borrow else sy.clone()
