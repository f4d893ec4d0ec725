"""Timing of the BASELINE.json configurations that are parity cases rather than the bench line (configs[2], [4]):
  * vidtok_fsq_causal_488_32768, bf16, B=4, 17x256x256  -- frames/s and the FSQ code agreement bf16 vs fp32 kernels
  * vidtok_kl_causal_488_16chn_v1_1, bf16, 1 clip of 129x256x256, temporal tiling t_chunk_enc=16 with decoder overlap
Synthetic inputs, seeded weights (tests/util.seeded_state_dict).  Prints one line per configuration; the output of a
run on the MI355X is kept under profiles/."""
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from util import build_model  # noqa: E402


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, out


def main():
    dev = "cuda:0"
    g = torch.Generator().manual_seed(0)
    # configs[2]
    x = (torch.rand((4, 3, 17, 256, 256), generator=g) * 2 - 1).to(dev)
    m16, _, _ = build_model("vidtok_fsq_causal_488_32768", seed=21, device=dev, dtype=torch.bfloat16)
    dt_eager, _ = timed(lambda: m16(x), 3)
    m16.enable_graphs()
    m16(x); m16(x)                                   # eager pass, capture pass
    dt, (z, dec, log) = timed(lambda: m16(x), 5)
    m32, _, _ = build_model("vidtok_fsq_causal_488_32768", seed=21, device=dev, dtype=torch.float32)
    _, _, log32 = m32(x)
    rate = (log["indices"] == log32["indices"]).float().mean().item()
    print(f"vidtok_fsq_causal_488_32768 bf16 B=4 17x256x256: engine graph cache {4 * 17 / dt:.1f} frames/s, {dt * 1e3:.1f} ms/step "
          f"(eager {4 * 17 / dt_eager:.1f} frames/s); "
          f"FSQ codes bf16 vs fp32 kernels equal: {100 * rate:.2f} % of {log['indices'].numel()} tokens "
          f"(fp32 kernels vs the reference: 100 %, tests/test_gpu_e2e.py)")
    del m16, m32
    # configs[4]
    xl = (torch.rand((1, 3, 129, 256, 256), generator=g) * 2 - 1).to(dev)
    m, _, _ = build_model("vidtok_v1_1/vidtok_kl_causal_488_16chn_v1_1", seed=22, device=dev, dtype=torch.bfloat16)
    m.use_tiling, m.t_chunk_enc, m.use_overlap = True, 16, True
    dt, (z, dec, log) = timed(lambda: m(xl), 3)
    assert dec.shape[2] >= 129 and torch.isfinite(dec).all()
    print(f"vidtok_kl_causal_488_16chn_v1_1 bf16 1 clip 129x256x256, tiled t_chunk_enc=16 + overlap (eager, stateful chunks): "
          f"{129 / dt:.1f} frames/s, {dt * 1e3:.1f} ms per clip, z {tuple(z.shape)}")
    m.enable_graphs()
    m(xl); m(xl)                                     # every chunk kind seen twice: later calls replay
    dt, _ = timed(lambda: m(xl), 3)
    print(f"vidtok_kl_causal_488_16chn_v1_1 bf16 1 clip 129x256x256, tiled t_chunk_enc=16 + overlap (engine graph cache: chunks replay): "
          f"{129 / dt:.1f} frames/s, {dt * 1e3:.1f} ms per clip")
    m.use_tiling = False
    m.enable_graphs()
    m(xl); m(xl)
    dt, (z, dec, log) = timed(lambda: m(xl), 3)
    print(f"vidtok_kl_causal_488_16chn_v1_1 bf16 1 clip 129x256x256, un-tiled (engine graph cache): "
          f"{129 / dt:.1f} frames/s, {dt * 1e3:.1f} ms per clip")


if __name__ == "__main__":
    main()
