"""vidtok_amd -- MI355X (gfx950) native encode/decode path for microsoft/VidTok's causal tokenizers.

Layout: `csrc/` HIP kernels + C-ABI (include/vidtok_amd.h), `lib.py` ctypes binding, `ops.py`
tensor-facing operator wrappers, `modules.py` / `regularizers.py` / `engine.py` the host-side mirror
of the reference's module API, `config.py` the YAML `target:` plug-in loader.
"""
from .config import instantiate_from_config, load_config, load_model_from_config  # noqa: F401

__all__ = ["instantiate_from_config", "load_config", "load_model_from_config"]
