"""AutoencodingEngine -- the model API of the reference (vidtok/models/autoencoder.py:98-229 and
autoencoder_v1_1.py:98-342) over the MI355X modules: forward / encode / decode /
indices_to_latent / init_from_ckpt, plus the v1.1 temporal tiling with causal caches.

`forward(x) -> (z, dec, reg_log)` is the reference's round-trip entry point; `encode_decode` is an
alias (BASELINE.json names it that way).  Training (`training_step`, optimizers, EMA, losses) is out
of scope: `loss_config` is accepted and ignored.
"""
import re
from typing import Any, Dict, Optional, Tuple, Union

import torch
import torch.nn as nn

from . import ops, packing
from .config import instantiate_from_config
from .graphs import GraphedCall


def _print0(msg):
    try:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized() and dist.get_rank() != 0:
            return
    except Exception:
        pass
    print(msg)


class AutoencodingEngine(nn.Module):
    version = "v1_0"
    arith = "fp32"          # set_compute_dtype(): "fp32" | "bf16" | "fp16" | "bf16x3"

    def __init__(self, *args, encoder_config: Dict, decoder_config: Dict, loss_config: Dict = None,
                 regularizer_config: Dict, optimizer_config: Union[Dict, None] = None, lr_g_factor: float = 1.0,
                 compile_model: bool = False, **kwargs):
        ckpt_path = kwargs.pop("ckpt_path", None)
        ignore_keys = kwargs.pop("ignore_keys", ())
        verbose = kwargs.pop("verbose", True)
        self.use_tiling = kwargs.pop("use_tiling", False)
        self.t_chunk_enc = kwargs.pop("t_chunk_enc", 16)
        # AbstractAutoencoder keywords of the reference (autoencoder.py:26-33); training only
        self.input_key = kwargs.pop("input_key", "jpg")
        for k in ("ema_decay", "monitor", "mode", "base_learning_rate"):
            kwargs.pop(k, None)
        super().__init__()
        self.global_step = 0
        self.encoder = instantiate_from_config(encoder_config)
        self.decoder = instantiate_from_config(decoder_config)
        self.loss = nn.Identity()  # loss_config is training-only
        self.regularization = instantiate_from_config(regularizer_config)
        self.is_causal = self.encoder.is_causal
        self.t_chunk_dec = self.t_chunk_enc // self.encoder.time_downsample_factor
        self.use_overlap = False
        if self.version == "v1_0" and self.use_tiling:
            raise NotImplementedError("temporal tiling exists only in the v1.1 models of the reference")
        self.use_graphs = False
        # bound methods, not closures: copy.deepcopy / pickle of the engine then rebind them to the copy
        self._genc = GraphedCall(self._encoder_fn, self._encoder_state, self._set_chunk_state)
        self._gdec = GraphedCall(self._decoder_fn, self._decoder_state, self._set_chunk_state)
        if verbose:
            _print0(f"[vidtok_amd.engine][AutoencodingEngine] Use ckpt_path: {ckpt_path}")
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys, verbose=verbose)

    # ---- numeric mode ---------------------------------------------------------------------------
    def set_compute_dtype(self, dtype, encoder_tail: Optional[torch.dtype] = None, tail_level: Optional[int] = None):
        """torch.float32: fp32 storage + fp32-input MFMA (parity mode); torch.bfloat16 / torch.float16: 16-bit storage +
        bf16 / fp16 MFMA with fp32 accumulation (throughput modes: what the reference computes in under torch.autocast with
        that dtype; fp16 values beyond 65 504 become inf, as in the reference's fp16 convolutions); "bf16x3": fp32 storage,
        every convolution on the bf16 matrix cores from bf16 hi / lo planes of both operands (vt_conv VT_BF16X3: three
        MFMAs per product, ~2^-17 relative per product) -- the fast mode that stays inside the reference's fp32 tolerance;
        `self.arith` says which ("fp32" | "bf16" | "fp16" | "bf16x3").
        `encoder_tail` (causal encoders): the encoder levels from `tail_level` on (default: the last level), its mid section
        and conv_out run in that type instead -- the small deep layers, whose rounding decides most of the FSQ code flips
        of a bf16 pass, in fp32 while the wide levels stay on the bf16 kernels."""
        self._chosen = (dtype, encoder_tail, tail_level)         # what to return to when an autocast region ends
        self._autocast_active = None
        self._apply_compute_dtype(dtype, encoder_tail, tail_level)
        self.invalidate_graphs()
        return self

    _ARITH_NAME = {torch.float32: "fp32", torch.bfloat16: "bf16", torch.float16: "fp16"}

    def _apply_compute_dtype(self, dtype, encoder_tail=None, tail_level=None):
        """switch the arithmetic mode.  Captured graphs stay: their keys carry (dtype, arith, tail), the packed weights of every
        mode a module has run in stay cached (packing.PackedCache), the v1.1 chunk-cache buffers are kept per dtype -- a caller
        that alternates autocast and plain calls replays both sets"""
        split3 = dtype == packing.ARITH_SPLIT3
        if split3:
            dtype = torch.float32
        assert dtype in self._ARITH_NAME and encoder_tail in (None,) + tuple(self._ARITH_NAME)
        packing.set_arith(self, packing.ARITH_SPLIT3 if split3 else None)
        self.arith = packing.ARITH_SPLIT3 if split3 else self._ARITH_NAME[dtype]
        self.encoder.compute_dtype = dtype
        self.decoder.compute_dtype = dtype
        if hasattr(self.encoder, "tail_dtype"):
            n = self.encoder.num_resolutions
            self.encoder.tail_dtype = encoder_tail
            self.encoder.tail_level = (n - 1 if tail_level is None else int(tail_level)) if encoder_tail is not None else None
            assert encoder_tail is None or 0 <= self.encoder.tail_level <= n
        else:
            assert encoder_tail is None, "encoder_tail: causal encoders only"

    # ---- the caller's torch.autocast region (reference README.md:336-340,375-385: `with torch.autocast(device_type="cuda",
    # dtype=...): model(x)`) ---------------------------------------------------------------------------------------------
    # The reference has no precision switch of its own: the ambient autocast context decides what its convolutions and
    # matmuls compute in.  Here the kernels are chosen by `set_compute_dtype`, so the engine reads the context at every
    # entry point: autocast(bfloat16) -> the bf16 kernels for this call, autocast(float16) -- the dtype of the README's
    # snippets and of `--precision autocast` in scripts/inference_*.py -- -> the fp16 kernels (whatever was set; the chosen
    # mode, and an fp32 `encoder_tail` chosen with it, return / stay).  `autocast_policy` can map a region elsewhere.
    autocast_policy = {"bfloat16": "bf16", "float16": "fp16"}

    def set_autocast_policy(self, **kw):
        """e.g. set_autocast_policy(float16="bf16x3"): run autocast(float16) regions in the split-bf16 mode (inside fp32
        tolerance); "bf16" / "fp16" = those kernels; "ignore" = keep the mode chosen by set_compute_dtype; "error" = raise"""
        pol = dict(self.autocast_policy)
        for k, v in kw.items():
            assert k in ("bfloat16", "float16") and v in ("bf16", "fp16", "bf16x3", "error", "ignore"), (k, v)
            pol[k] = v
        self.autocast_policy = pol
        return self

    _MODE_DTYPE = {"bf16": torch.bfloat16, "fp16": torch.float16, "bf16x3": packing.ARITH_SPLIT3}

    def _sync_autocast(self, x):
        """make the arithmetic mode follow the caller's autocast context (device of `x`)"""
        dev = x.device.type if isinstance(x, torch.Tensor) else "cuda"
        try:
            on = torch.is_autocast_enabled(dev)
            adt = torch.get_autocast_dtype(dev) if on else None
        except (TypeError, AttributeError, RuntimeError):      # older torch: the CUDA-only spellings
            on = torch.is_autocast_enabled()
            adt = torch.get_autocast_gpu_dtype() if on else None
        want = None
        if on:
            pol = self.autocast_policy.get({torch.bfloat16: "bfloat16", torch.float16: "float16"}.get(adt, ""), "error")
            if pol == "error":
                raise NotImplementedError(
                    f"vidtok_amd: called under torch.autocast(dtype={adt}): no kernels for that dtype (bfloat16 and float16 regions "
                    f"select the bf16 / fp16 kernels), or the policy for it is 'error' (model.set_autocast_policy)")
            if pol != "ignore":
                want = self._MODE_DTYPE[pol]
        active = getattr(self, "_autocast_active", None)
        if want is not None:
            name = self._ARITH_NAME.get(want, want)
            if active is None and self.arith == name:
                return                                          # already the mode the region asks for: nothing to switch, nothing to restore
            if active != want:
                if getattr(self, "_chosen", None) is None:      # never set: the construction default
                    self._chosen = (self.encoder.compute_dtype, None, None)
                # an encoder tail chosen with the mode (fp32 deep levels for FSQ code stability) stays in force inside the region
                _, tail, level = self._chosen
                self._apply_compute_dtype(want, tail, level)
                self._autocast_active = want
        elif active is not None:
            self._autocast_active = None
            self._apply_compute_dtype(*self._chosen)

    # ---- launch mode: per-shape hipGraphs of the encoder / decoder launch sequences (vidtok_amd/graphs.py) ----
    def enable_graphs(self, on: bool = True):
        self.use_graphs = bool(on)
        self.invalidate_graphs()
        return self

    def invalidate_graphs(self):
        """forget captured graphs (call after editing parameters in place) and the v1.1 chunk-cache buffers they replay
        against (vidtok_amd/modules.py::_CausalState._persistent)"""
        self._genc.clear()
        self._gdec.clear()
        self._drop_cache_buffers(self)

    @staticmethod
    def _drop_cache_buffers(root):
        for m in root.modules():
            if "_cache_bufs" in m.__dict__:
                m.__dict__["_cache_bufs"].clear()
                if hasattr(m, "causal_cache"):
                    m.causal_cache = None

    def _encoder_fn(self, t):
        return self.encoder(t)

    def _decoder_fn(self, t):
        return self.decoder(t)

    def _encoder_state(self):
        return self._chunk_state(self.encoder)

    def _decoder_state(self):
        return self._chunk_state(self.decoder)

    def _apply(self, fn, *a, **kw):           # .to() / .cuda() / .float(): parameters move, captured graphs are stale
        self.invalidate_graphs()
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, *a, **kw):
        self.invalidate_graphs()
        return super().load_state_dict(*a, **kw)

    @staticmethod
    def _chunk_state(root):
        """[(module, its chunk-to-chunk cache)] of a sub-tree (v1.1 tiling; empty for v1.0 models)"""
        return [(m, m.causal_cache) for m in root.modules() if hasattr(m, "causal_cache")]

    @staticmethod
    def _set_chunk_state(state):
        for m, c in state:
            m.causal_cache = c

    def _run_encoder(self, x):
        key = (self.encoder.compute_dtype, getattr(self, "arith", None), getattr(self.encoder, "tail_dtype", None), getattr(self.encoder, "tail_level", None))
        return self._genc(x, key) if self.use_graphs else self.encoder(x)

    def _run_decoder(self, z):
        return self._gdec(z, (self.decoder.compute_dtype, getattr(self, "arith", None))) if self.use_graphs else self.decoder(z)

    # ---- checkpoints (autoencoder.py:146-176) ---------------------------------------------------
    def init_from_ckpt(self, path: str, ignore_keys=tuple(), verbose: bool = True) -> None:
        if path.endswith("ckpt"):
            ckpt = torch.load(path, map_location="cpu")
            weights = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
        elif path.endswith("safetensors"):
            from safetensors.torch import load_file

            weights = load_file(path)
        else:
            raise NotImplementedError(f"Unknown checkpoint: {path}")
        for k in list(weights.keys()):
            for ik in ignore_keys:
                if re.match(ik, k):
                    _print0(f"[vidtok_amd.engine] Deleting key {k} from state_dict.")
                    del weights[k]
                    break
        missing, unexpected = self.load_state_dict(weights, strict=False)
        _print0(f"[vidtok_amd.engine] Restored from {path} with {len(missing)} missing and "
                f"{len(unexpected)} unexpected keys")
        if verbose:
            if missing:
                _print0(f"[vidtok_amd.engine] Missing Keys: {missing}")
            if unexpected:
                _print0(f"[vidtok_amd.engine] Unexpected Keys: {unexpected}")

    def get_last_layer(self):
        return self.decoder.get_last_layer()

    # ---- encode / decode ------------------------------------------------------------------------
    @torch.no_grad()
    def encode(self, x: Any, return_reg_log: bool = False) -> Any:
        self._sync_autocast(x)
        z = self._run_encoder(x)
        z, reg_log = self.regularization(z, n_steps=self.global_step // 2)
        if return_reg_log:
            return z, reg_log
        return z

    @torch.no_grad()
    def indices_to_latent(self, token_indices: torch.Tensor) -> torch.Tensor:
        """int32 [B, T', H', W'] -> latent [B, D, T', H', W'] (autoencoder.py:205-213)."""
        return self.regularization.indices_to_codes(token_indices)

    @torch.no_grad()
    def decode(self, z: Any, decode_from_indices: bool = False) -> torch.Tensor:
        self._sync_autocast(z)
        if decode_from_indices:
            z = self.indices_to_latent(z)
        return self._run_decoder(z)

    @torch.no_grad()
    def forward(self, x: Any) -> Tuple[torch.Tensor, torch.Tensor, dict]:
        z, reg_log = self.encode(x, return_reg_log=True)
        dec = self.decode(z)
        return z, dec, reg_log

    encode_decode = forward


class AutoencodingEngineV11(AutoencodingEngine):
    """v1.1: first-frame-replicate causal padding and sequential temporal tiling with per-module causal
    caches and a one-latent-frame decoder look-ahead (autoencoder_v1_1.py:202-342)."""

    version = "v1_1"

    # -- cache / chunk protocol: walks the module tree exactly like the reference (:202-216) ------
    def _empty_causal_cached(self, parent):
        for _, module in parent.named_modules():
            if hasattr(module, "causal_cache"):
                module.causal_cache = None
        if not self.use_graphs:
            # eager passes own no captured addresses: the chunk-cache buffers go back to the allocator between clips, as
            # the reference's caches do
            self._drop_cache_buffers(parent)
        elif self._genc.overfull or self._gdec.overfull:
            # captured chunks replay against those buffers, so they live as long as the graphs: a process that keeps
            # meeting new chunk kinds (clip lengths, resolutions) starts over here, between two passes
            self.invalidate_graphs()

    def _set_first_chunk(self, is_first_chunk=True):
        for module in self.modules():
            if hasattr(module, "is_first_chunk"):
                module.is_first_chunk = is_first_chunk

    def _set_fused_temporal(self):
        """Un-tiled passes never read the chunk caches, so blocks that can run as one fused launch (which keeps their
        convolutions' inputs on chip and therefore cannot leave caches behind) may do so; tiled passes may not."""
        for module in self.modules():
            if hasattr(module, "allow_fused"):
                module.allow_fused = not self.use_tiling

    def _set_cache_offset(self, modules, cache_offset=0):
        for module in modules:
            for submodule in module.modules():
                if hasattr(submodule, "cache_offset"):
                    submodule.cache_offset = cache_offset

    def build_chunk_start_end(self, t, decoder_mode=False):
        """[[0,1],[1,1+c],[1+c,1+2c],...]: the first chunk is the single leading frame (:218-228)."""
        step = self.t_chunk_dec if decoder_mode else self.t_chunk_enc
        assert step > 0
        start_end, start = [[0, 1]], 1
        while start < t:
            end = min(t, start + step)
            start_end.append([start, end])
            start = end
        return start_end

    @torch.no_grad()
    def encode(self, x: Any, return_reg_log: bool = False) -> Any:
        self._sync_autocast(x)
        self._empty_causal_cached(self.encoder)
        self._set_first_chunk(True)
        self._set_fused_temporal()
        if self.use_tiling:
            z, reg_log = self.tile_encode(x)
        else:
            z = self._run_encoder(x)
            z, reg_log = self.regularization(z, n_steps=self.global_step // 2)
        if return_reg_log:
            return z, reg_log
        return z

    def _chunk_of(self, x, start, end):
        """frames [start, end) of an NCTHW fp32 tensor as a contiguous tensor (one vt_ncthw_copy_frames launch)"""
        if not x.is_cuda:
            return x[:, :, start:end].contiguous()
        x = x.contiguous().float()
        c = torch.empty((x.shape[0], x.shape[1], end - start) + tuple(x.shape[3:]), dtype=torch.float32, device=x.device)
        return ops.ncthw_copy_frames(x, c, start, 0, end - start)

    def _chunk_call(self, graphed, module, x, start, end, first):
        """module(frames [start, end) of x).  With the graph cache on, a chunk replays once its kind has been seen
        twice; the kind is the chunk shape plus everything that shapes the launch sequence: first / later chunk, the
        cache offsets of an overlapped decode, and how many frames each module's cache holds right now (the chunk
        after the single-frame first one meets shorter caches than the ones after it)."""
        if not (self.use_graphs and x.is_cuda):
            return module(self._chunk_of(x, start, end))
        sig = tuple(0 if c is None else c.shape[1] for _, c in self._chunk_state(module))
        key = (module.compute_dtype, getattr(self, "arith", None), getattr(module, "tail_dtype", None), getattr(module, "tail_level", None),
               "chunk", bool(first), bool(self.use_overlap), sig)
        return graphed(x, key, stateful=True, frames=(start, end), borrow=True)

    def tile_encode(self, x: Any) -> Any:
        """chunks are encoded in order (module caches carry the causal state); their latents land in one preallocated
        tensor -- no list + torch.cat (autoencoder_v1_1.py:244-264)"""
        chunks = self.build_chunk_start_end(x.shape[2])
        z, idx, logs, done = None, None, [], 0
        for i, (start, end) in enumerate(chunks):
            self._set_first_chunk(i == 0)
            chunk_z = self._chunk_call(self._genc, self.encoder, x, start, end, i == 0)
            chunk_z, chunk_log = self.regularization(chunk_z, n_steps=self.global_step // 2)
            if z is None:      # a chunk of n frames is front-padded to a multiple of f: ceil(n / f) latent frames
                f = self.encoder.time_downsample_factor
                tz = sum(-(-(e - s) // f) for s, e in chunks)
                z = torch.empty((chunk_z.shape[0], chunk_z.shape[1], tz) + tuple(chunk_z.shape[3:]), dtype=chunk_z.dtype,
                                device=chunk_z.device)
                if "indices" in chunk_log:
                    # [B, T', H', W'] (+ a trailing codebook axis with keep_num_codebooks_dim)
                    idx = torch.empty((chunk_z.shape[0], tz) + tuple(chunk_log["indices"].shape[2:]), dtype=torch.int32, device=chunk_z.device)
            n = chunk_z.shape[2]
            if chunk_z.is_cuda:
                ops.ncthw_copy_frames(chunk_z.contiguous(), z, 0, done, n)
                if idx is not None:
                    ops.gather_frames(chunk_log["indices"].contiguous(), list(range(n)), out=idx, out_t0=done)
            else:
                z[:, :, done:done + n] = chunk_z
                if idx is not None:
                    idx[:, done:done + n] = chunk_log["indices"]
            done += n
            logs.append(chunk_log)
        assert done == z.shape[2], (done, z.shape)
        if "kl_loss" in logs[0]:
            return z, {"kl_loss": torch.mean(torch.stack([d["kl_loss"] for d in logs]))}
        return z, {"aux_loss": torch.mean(torch.stack([d["aux_loss"] for d in logs])), "indices": idx}

    def tile_indices_to_latent(self, token_indices: torch.Tensor) -> torch.Tensor:
        # per-position look-up: chunking changes nothing in the result (the reference chunks to bound memory)
        return self.indices_to_latent(token_indices.contiguous())

    @torch.no_grad()
    def decode(self, z: Any, decode_from_indices: bool = False) -> torch.Tensor:
        self._sync_autocast(z)
        if decode_from_indices:
            z = self.tile_indices_to_latent(z) if self.use_tiling else self.indices_to_latent(z)
        self._empty_causal_cached(self.decoder)
        self._set_first_chunk(True)
        self._set_fused_temporal()
        if self.use_tiling:
            return self.tile_decode(z)
        return self._run_decoder(z)

    def _overlap_offsets(self):
        """cache_offset per decoder sub-tree when chunks carry one look-ahead latent frame: 1 at latent
        rate, doubling after each temporal up-sampler (autoencoder_v1_1.py:307-320)."""
        f, d = self.encoder.time_downsample_factor, self.decoder
        assert f in [2, 4, 8], "Only support 2x, 4x or 8x temporal downsampling now."
        self._set_cache_offset([d], 1)
        if f == 4:
            self._set_cache_offset([d.up_temporal[2].upsample, d.up_temporal[1]], 2)
            self._set_cache_offset([d.up_temporal[1].upsample, d.up_temporal[0], d.conv_out], 4)
        elif f == 2:
            self._set_cache_offset([d.up_temporal[2].upsample, d.up_temporal[1], d.up_temporal[0], d.conv_out], 2)
        else:
            self._set_cache_offset([d.up_temporal[3].upsample, d.up_temporal[2]], 2)
            self._set_cache_offset([d.up_temporal[2].upsample, d.up_temporal[1]], 4)
            self._set_cache_offset([d.up_temporal[1].upsample, d.up_temporal[0], d.conv_out], 8)

    def tile_decode(self, z: Any) -> torch.Tensor:
        """chunks decoded in order, each with one look-ahead latent frame when `use_overlap` (its f trailing output
        frames are dropped); outputs land in one preallocated tensor (autoencoder_v1_1.py:302-331)"""
        num_frames = z.shape[2]
        f = self.encoder.time_downsample_factor
        if self.use_overlap:
            self._overlap_offsets()
        chunks = self.build_chunk_start_end(num_frames, decoder_mode=True)
        out, done = None, 0
        for idx, (start, end) in enumerate(chunks):
            self._set_first_chunk(idx == 0)
            look = self.use_overlap and end + 1 <= num_frames
            chunk = self._chunk_call(self._gdec, self.decoder, z, start, end + 1 if look else end, idx == 0)
            n = chunk.shape[2] - (f if look else 0)
            if out is None:   # first chunk (1 latent frame) yields 1 frame, every other latent frame f frames
                total = n + sum(f * (e - s) for s, e in chunks[1:])
                out = torch.empty((chunk.shape[0], chunk.shape[1], total) + tuple(chunk.shape[3:]), dtype=chunk.dtype, device=chunk.device)
            if chunk.is_cuda:
                ops.ncthw_copy_frames(chunk.contiguous(), out, 0, done, n)
            else:
                out[:, :, done:done + n] = chunk[:, :, :n]
            done += n
        assert done == out.shape[2], (done, out.shape)
        return out

    @torch.no_grad()
    def forward(self, x: Any) -> Tuple[torch.Tensor, torch.Tensor, dict]:
        z, reg_log = self.encode(x, return_reg_log=True)
        dec = self.decode(z)
        if dec.shape[2] != x.shape[2]:               # drop the frames that decode the encoder's front padding (:339-341)
            n = x.shape[2]
            if dec.is_cuda:                           # as a tensor of its own (one vt_ncthw_copy_frames), not a strided view
                out = torch.empty(tuple(dec.shape[:2]) + (n,) + tuple(dec.shape[3:]), dtype=dec.dtype, device=dec.device)
                dec = ops.ncthw_copy_frames(dec.contiguous(), out, dec.shape[2] - n, 0, n)
            else:
                dec = dec[:, :, -n:, ...]
        return z, dec, reg_log

    encode_decode = forward
