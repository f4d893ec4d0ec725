"""KL and FSQ regularizers with the reference's module API (vidtok/modules/regularizers.py:74-268),
computed by the HIP kernels vt_kl_sample / vt_fsq_* on the NCTHW fp32 latent."""
from typing import Any, List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops


class DiagonalGaussianRegularizer(nn.Module):
    """regularizers.py:74-92 + distributions.py:5-28.

    `noise_source="host"` (default) reproduces the reference bit-for-bit in its use of the RNG:
    one `torch.randn(mean.shape)` on the CPU default generator per call, uploaded to the device
    (distributions.py:16-18).  On a GPU the draw goes into one of two PINNED host buffers and is
    uploaded with an asynchronous copy on the launch stream (same numbers, same generator
    consumption as `torch.randn(shape)`): the host never waits for the device, so the draw of a
    step hides behind the encoder launches that are still running -- the noise tensor is what the
    captured decoder graph's input z is computed from (bench.py times this path).  A buffer is
    reused only after the copy that read it has completed (one event per buffer).
    `noise_source="device"` draws the noise with the device generator instead (ATen's philox
    kernel: no host work at all) -- same distribution, different stream; not a parity mode.
    """

    def __init__(self, sample: bool = True, noise_source: str = "host"):
        super().__init__()
        self.sample = sample
        assert noise_source in ("host", "device")
        self.noise_source = noise_source
        self._pinned = {}        # (shape, device) -> [[pinned buffer, device buffer, event] x 2, next index]

    def get_trainable_parameters(self) -> Any:
        yield from ()

    def _host_noise(self, shape, device):
        """torch.randn(shape) of the CPU default generator, on `device`"""
        if device.type != "cuda":
            return torch.randn(shape).to(device=device)
        key = (tuple(shape), device)
        ring = self._pinned.get(key)
        if ring is None:
            if len(self._pinned) >= 8:           # a process that keeps meeting new latent shapes does not pile up pinned memory
                self._pinned.clear()
            ring = self._pinned[key] = [[[torch.empty(shape, dtype=torch.float32).pin_memory(),
                                          torch.empty(shape, dtype=torch.float32, device=device), None] for _ in range(2)], 0]
        slot = ring[0][ring[1]]
        ring[1] ^= 1
        host, dev, ev = slot
        if ev is not None:
            ev.synchronize()                     # the upload that last read this buffer (two steps ago) has completed
        torch.randn(shape, out=host)             # the reference's draw: same generator, same count, same values
        dev.copy_(host, non_blocking=True)
        if ev is None:
            ev = slot[2] = torch.cuda.Event()
        ev.record()
        return dev

    def __getstate__(self):                      # pinned buffers / events do not pickle or deep-copy
        st = dict(self.__dict__)
        st["_pinned"] = {}
        return st

    def __deepcopy__(self, memo):
        new = type(self)(self.sample, self.noise_source)
        memo[id(self)] = new
        return new

    @torch.no_grad()
    def forward(self, z: torch.Tensor, n_steps=None) -> Tuple[torch.Tensor, dict]:
        assert z.dim() >= 3 and z.shape[1] % 2 == 0
        shape = (z.shape[0], z.shape[1] // 2) + tuple(z.shape[2:])
        noise = None
        if self.sample:
            if self.noise_source == "host":
                noise = self._host_noise(shape, z.device)
            else:
                # device random numbers must never end up inside a captured launch sequence: the engine replays its graphs through
                # vt_graph_launch, which does not refresh a device generator's philox offset -- every replay would reuse the same noise
                if z.is_cuda and torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("DiagonalGaussianRegularizer(noise_source='device') inside a hipGraph capture: the replay path "
                                       "(vt_graph_launch) does not advance the device generator; regularize outside the captured region")
                noise = torch.randn(shape, device=z.device, dtype=torch.float32)
        zs, kl = ops.kl_sample(z.contiguous(), noise)
        return zs, {"kl_loss": kl}


class _Identity(nn.Identity):
    pass


class FSQRegularizer(nn.Module):
    """Finite scalar quantisation (regularizers.py:95-268): tanh bound, round half to even, integer
    codes packed in mixed radix; aux loss = entropy terms over the implicit codebook + commitment."""

    def __init__(self, levels: List[int], dim: Optional[int] = None, num_codebooks=1,
                 keep_num_codebooks_dim: Optional[bool] = None, scale: Optional[float] = None,
                 entropy_loss_weight: float = 0.0, entropy_loss_annealing_steps: int = 0,
                 entropy_loss_annealing_factor: float = 1.0, commitment_loss_weight: float = 0.0,
                 diversity_gamma: float = 1.0, compute_aux_loss: bool = True):
        super().__init__()
        self.levels = [int(v) for v in levels]
        self.num_codebooks = int(num_codebooks)
        assert self.num_codebooks >= 1
        self.codebook_dim = len(self.levels)
        self.effective_codebook_dim = self.codebook_dim * self.num_codebooks
        # regularizers.py:131-133: several codebooks always keep their axis on the indices
        self.keep_num_codebooks_dim = (self.num_codebooks > 1) if keep_num_codebooks_dim is None else bool(keep_num_codebooks_dim)
        assert not (self.num_codebooks > 1 and not self.keep_num_codebooks_dim)
        self.dim = self.effective_codebook_dim if dim is None else dim
        self.has_projections = self.dim != self.effective_codebook_dim
        # same parameter names as the reference (regularizers.py:137-139): checkpoints load unchanged
        self.project_in = nn.Linear(self.dim, self.effective_codebook_dim) if self.has_projections else _Identity()
        self.project_out = nn.Linear(self.effective_codebook_dim, self.dim) if self.has_projections else _Identity()
        self.scale = scale
        self.entropy_loss_weight = entropy_loss_weight
        self.entropy_loss_annealing_steps = entropy_loss_annealing_steps
        self.entropy_loss_annealing_factor = entropy_loss_annealing_factor
        self.commitment_loss_weight = commitment_loss_weight
        self.diversity_gamma = diversity_gamma
        self.compute_aux_loss = compute_aux_loss
        cs = 1
        for v in self.levels:
            cs *= v
        self.codebook_size = cs
        self.register_buffer("_levels", torch.tensor(self.levels, dtype=torch.int32), persistent=False)
        basis, b = [], 1
        for v in self.levels:
            basis.append(b)
            b *= v
        self.register_buffer("_basis", torch.tensor(basis, dtype=torch.int32), persistent=False)
        self.register_buffer("zero", torch.tensor(0.0), persistent=False)

    def get_trainable_parameters(self) -> Any:
        return self.parameters()

    def calculate_entropy_loss_weight(self, n_steps):
        if n_steps >= self.entropy_loss_annealing_steps:
            return self.entropy_loss_weight
        start = self.entropy_loss_annealing_factor * self.entropy_loss_weight
        return start - (n_steps / self.entropy_loss_annealing_steps) * (start - self.entropy_loss_weight)

    @torch.no_grad()
    def indices_to_codes(self, indices: torch.Tensor, project_out=True) -> torch.Tensor:
        """indices int32 [B, ...] (with keep_num_codebooks_dim: [B, ..., c]) -> codes [B, D, ...]
        (regularizers.py:180-198, image/video form)."""
        assert indices.dim() >= 3 + int(self.keep_num_codebooks_dim), "expects [B, T, H, W] (or [B, H, W]) index maps"
        idx = indices if indices.dtype == torch.int32 else indices.to(torch.int32)
        c = self.num_codebooks
        if self.keep_num_codebooks_dim and c == 1:
            idx = idx.reshape(idx.shape[:-1])          # a kept axis of one codebook: [B, ..., 1] is [B, ...] in memory
        assert c == 1 or idx.shape[-1] == c
        codes = ops.fsq_indices_to_codes(idx.contiguous(), self.levels, c)
        if project_out and self.has_projections:
            codes = self._linear(self.project_out, codes)
        return codes

    @staticmethod
    def _linear(lin: nn.Linear, x):
        return ops.channel_linear(x, lin.weight.detach().float().contiguous(),
                                  None if lin.bias is None else lin.bias.detach().float().contiguous())

    def _aux_stats(self, h, inv_temperature):
        """(stats [3], codebook entropy or None = stats[1]).  Like the reference, the batch-mean code distribution is averaged over the
        ranks whenever torch.distributed runs with world > 1 -- in eval too (maybe_distributed_mean,
        regularizers.py:49-59,240): one all_reduce (RCCL on GPUs) of prod(levels) floats, then its entropy."""
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            st = ops.fsq_aux_stats(h, self.levels, inv_temperature)
            return st, None              # the codebook entropy is st[1]
        st, avg = ops.fsq_aux_stats(h, self.levels, inv_temperature, return_avg=True)
        dist.all_reduce(avg)
        avg = avg / dist.get_world_size()
        return st, ops.entropy(avg)

    @torch.no_grad()
    def forward(self, z: torch.Tensor, inv_temperature: float = 100.0, n_steps: int = 0):
        assert z.dim() >= 4, "expects [B, D, T, H, W]"
        assert z.shape[1] == self.dim, f"expected dimension of {self.dim} but found dimension of {z.shape[1]}"
        h = z.float().contiguous()
        if self.has_projections:
            h = self._linear(self.project_in, h)
        c = self.num_codebooks
        # "b n (c d) -> b n c d" (regularizers.py:227): in NCTHW the d channels of codebook k of a clip are contiguous; the kernel
        # quantises each codebook on its own and writes the indices with the codebook axis last ([B, ..., c]) for c > 1
        codes, indices = ops.fsq_quantize(h, self.levels, c)
        if self.keep_num_codebooks_dim and c == 1:
            indices = indices.unsqueeze(-1)
        want_aux = self.compute_aux_loss and (self.entropy_loss_weight > 0 or self.commitment_loss_weight > 0)
        if want_aux and not self.keep_num_codebooks_dim:
            st, codebook_entropy = self._aux_stats(h, inv_temperature)
            # (st[0] - gamma * codebook_entropy) * w + st[2] * commitment_weight (regularizers.py:241,264-266), one launch
            aux = ops.fsq_aux_loss(st, codebook_entropy, self.diversity_gamma, self.calculate_entropy_loss_weight(n_steps),
                                   self.commitment_loss_weight)
        elif want_aux:
            # the reference cannot run this combination either: with the codebook axis kept, its implicit codebook is
            # flattened to one dimension and the distance einsum raises (regularizers.py:143-146,191-192,234)
            raise NotImplementedError("FSQ entropy / commitment loss with keep_num_codebooks_dim (num_codebooks > 1): "
                                      "the reference's forward raises for it (regularizers.py:234); set both weights to 0")
        elif z.is_cuda:
            # zero * w + zero * commitment_weight of the reference (regularizers.py:246,264-266): a fresh zero, from the same launch
            z3 = self.__dict__.get("_zero3")
            if z3 is None or z3.device != z.device:
                z3 = self.__dict__["_zero3"] = torch.zeros(3, device=z.device)
            aux = ops.fsq_aux_loss(z3, None, self.diversity_gamma, 0.0, 0.0)
        else:
            aux = self.zero.to(z.device) * 1.0
        if self.has_projections:
            codes = self._linear(self.project_out, codes)
        return codes, dict(indices=indices, aux_loss=aux)
