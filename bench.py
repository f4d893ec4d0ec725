#!/usr/bin/env python
"""bench.py -- encode+decode frames/s of the MI355X VidTok path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--dtype bf16|fp32|bf16x3] [--batch B] [--no-graph]

One "step" = one pass of the hot path, `model(x)` = encode -> KL regularizer -> decode
(reference AutoencodingEngine.forward, vidtok/models/autoencoder.py:221-229), over one batch of
synthetic clips x = rand(B,3,17,256,256)*2-1 that is resident in HBM before the timed region.
Workload: N=1 -> BASELINE.json configs[1], vidtok_kl_causal_488_4chn, bf16, B=4 clips; N>1 -> configs[3],
vidtok_kl_causal_488_16chn, global batch 4*N clips batch-sharded over the N GPUs (B=32 at N=8), 4 clips per
GPU (weak scaling: the path has no data-path collective; only the timing / metrics
reduction crosses ranks, one tiny RCCL all_reduce).  The two configs differ in two layers (encoder.conv_out /
decoder.conv_in width), < 0.01 % of the FLOPs.  Prints ONE JSON line on rank 0.

Launch: `python bench.py --gpus N` spawns its N ranks itself (re-executes under torch.distributed.run on
127.0.0.1) when it is not already running under a launcher; under `python -m torch.distributed.run ... bench.py
--gpus N` it reads RANK / LOCAL_RANK / WORLD_SIZE from the environment.

  value       real frames/s over the whole job = N*B*17*K / max-over-ranks(time of K steps)
  noise       the timed step draws its KL noise from the reference's host stream (one torch.randn of the CPU generator per step, pinned double
              buffer, async upload: the parity graph); the same K steps with ATen's device philox are timed next to it (host_over_device)
  roofline    achieved / frac: algorithmic FLOPs of one step (1.0345 TFLOP per padded 256x256 frame) / ms_per_step -- the
              WHOLE step the line reports -- against the dense MFMA peak of the dtype; *_kernel_only: the same FLOPs / the time of the MFMA kernels alone
              (conv_igemm_glds_kernel, conv3x3_ws2_kernel, conv_in8_kernel, conv3d_narrow_kernel, tblock_pair_kernel, flash_attn_kernel: all convolutions +
              the attention = every MFMA FLOP of the path); *_executed: the MACs the launches really execute.  The kernel time is measured live: the conv
              launches of one step (same descriptors, same tensors) are replayed back to back from a
              hipGraph that contains nothing else, bracketed by HIP events on the launch stream -- no
              per-launch event overhead, so it is <= ms_per_step by construction; the rocprofv3
              --kernel-trace average of the same command is committed under profiles/.  peak = dense MFMA
              peak of the dtype; traffic = HBM-side bytes per launch, measured by two rocprofv3 PMC passes
              (FETCH_SIZE x2 + WRITE_SIZE) over a child run of this command (counters cannot be read inside
              an un-profiled process); falls back to the committed pass under profiles/ (committed_traffic).
              roofline.hbm: the HBM-shaped kernel classes (LayerNorm passes, conv_in, 1x1 convolutions, conv_out, the
              temporal k3 convolutions and the fused temporal block: FLOP per byte below the 312 FLOP/B ridge), each
              replayed alone the same way: algorithmic bytes (every operand / result once) / time, against 8 TB/s
  parity_mode / modes / other_configs   (N = 1) the same workload in the other arithmetic modes -- fp32 (fp32 MFMA) and bf16x3
              (split-bf16: fp32 storage, three bf16 MFMAs per product; the fast mode inside the reference's fp32 tolerance) --
              with each mode's distance from the CPU oracle on clip 0 of the batch, and the frames/s of BASELINE.json
              configs[2] (FSQ, with the integer-code agreement) and configs[4] (129-frame clip, temporal tiling)
  cpu_baseline  the CPU oracle (port of the reference, oracle/vidtok_oracle.py) timed on this host's
              cores on a bounded sample of the same workload; a baseline, not a target.  At N > 1 rank 0 runs it (and
              the traffic passes) after the process group is gone, so the line of a scaling run carries both too
"""
import argparse
import json
import math
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL across processes on this driver); before the HIP runtime starts

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_PADDED_FRAME_256 = 1.0345e12     # SURVEY.md section 8(d), conv + attention MACs x 2
# MI355X_MICROARCH.md: dense MFMA peaks.  bf16x3 (split-bf16: fp32 storage, three bf16 MFMAs per product) is priced per
# ALGORITHMIC FLOP like the others: a third of the bf16 peak
PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3, "bf16x3": 2500.0 / 3}
PEAK_HBM_GBS = 8000.0                           # MI355X_MICROARCH.md: HBM3E 8 TB/s (6.3 TB/s achievable by a copy)
T_REAL, T_PADDED, RES = 17, 20, 256
CONFIG_1GPU = "vidtok_kl_causal_488_4chn"       # BASELINE.json configs[1]
CONFIG_NGPU = "vidtok_kl_causal_488_16chn"      # BASELINE.json configs[3]
CONFIG = CONFIG_1GPU


def randomize_weights(model, seed=0):
    """Random-init weights of the named architecture, with the zero-initialised temporal conv2 and the
    identity LayerNorm affines re-drawn so no layer is numerically trivial (SURVEY.md finding 3)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("mix_factor"):
                p.copy_(0.3 + 0.5 * torch.randn(p.shape, generator=g))
            elif ".norm" in name and p.dim() == 1:
                p.copy_((1.0 if name.endswith("weight") else 0.0) + 0.1 * torch.randn(p.shape, generator=g))
            elif p.dim() == 1:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) / math.sqrt(p[0].numel()))


# seconds of the UNMODIFIED reference's forward / seconds of the oracle port's, same 17x256x256 clip, same threads: measured where the reference can
# be imported (scripts/cpu_port_vs_reference.py in the build container, 8 vCPUs) -- the reference is that much slower than the port timed here
REFERENCE_OVER_PORT = {"ratio": 1.14, "measured_on": "build container, 8 vCPU, 8 torch threads, fp32, vidtok_kl_causal_488_4chn 17x256x256",
                       "source": "profiles/r05_cpu_port_vs_reference.txt (re-measured per round by scripts/cpu_port_vs_reference.py)"}


def cpu_baseline(config=None, clip=None, seed=77):
    """Time the CPU oracle (a port: the reference is Python under /root/reference, which does not exist on the GPU box and
    may not be copied into the repo) on ONE unscaled 17x256x256 clip of the bench workload -- about 30 s of CPU work.
    The thread count is the best of a sweep over 8 / 16 / 32 / 64 / 128 threads (capped by the host) on a 17x64x64 probe,
    reported in `thread_sweep`: torch's CPU convolutions collapse when given all 256 hardware threads of the GPU box
    (measured: 17x64x64 took 122 s on 256 threads).  In the build container the unmodified reference runs this same clip
    1.0-1.1x slower than the oracle at the same thread count (profiles/r05_cpu_port_vs_reference.txt), so the port is a
    fair stand-in."""
    import vidtok_amd
    from oracle.vidtok_oracle import OracleEngine

    config = config or CONFIG_1GPU
    cores = os.cpu_count() or 1
    cfg = vidtok_amd.load_config(os.path.join(ROOT, "configs", config + ".yaml"))
    model = vidtok_amd.load_model_from_config(cfg, verbose=False)
    randomize_weights(model, 0)
    ora = OracleEngine(cfg["model"]["params"], model.state_dict())
    del model

    def run(res):
        x = torch.rand(1, 3, T_REAL, res, res) * 2 - 1
        t0 = time.perf_counter()
        ora(x)
        return time.perf_counter() - t0

    best_t, threads, sweep = None, 1, {}
    for nt in sorted({min(cores, n) for n in (8, 16, 32, 64, 128)}):
        torch.set_num_threads(nt)
        run(32)                   # warm-up (thread pool, allocator)
        t = min(run(64), run(64))
        sweep[str(nt)] = round(t, 3)
        if best_t is None or t < best_t:
            best_t, threads = t, nt
        if t > 3.0 * best_t or t > 10.0:
            break                 # past the knee: more threads only get slower
    if clip is None:
        clip = torch.rand(1, 3, T_REAL, RES, RES) * 2 - 1
    # the probe's winner AND its faster neighbour in the sweep run the real 17x256x256 clip (the knee of a 64x64 probe need
    # not be the knee of the 256x256 clip); `value` is the faster of the two
    order = sorted(int(k) for k in sweep)
    i = order.index(threads)
    cands = [threads] + [n for n in (order[i + 1] if i + 1 < len(order) else None, order[i - 1] if i > 0 else None) if n is not None][:1]
    full, best = {}, None
    for nt in cands:
        torch.set_num_threads(nt)
        torch.manual_seed(seed)      # the KL noise stream: the engine's host-noise pass of the same clip draws the same numbers
        t0 = time.perf_counter()
        z, dec, _ = ora(clip)
        t = time.perf_counter() - t0
        full[str(nt)] = round(t, 2)
        if best is None or t < best[0]:
            best = (t, nt, z, dec)
    t, threads, z, dec = best
    torch.set_num_threads(threads)
    return {"value": round(T_REAL / t, 4), "unit": "frames/s", "cores": threads, "kind": "port",
            "thread_sweep": {"probe": "17x64x64 clip, seconds per forward by torch thread count", **sweep,
                             "full_clip": {"what": f"17x{RES}x{RES} clip, seconds per forward: the probe's winner and its neighbour", **full}},
            # the unmodified reference (importable only in the build container: /root/reference does not exist on the GPU box) against this
            # port on the same clip at the same thread count: scripts/cpu_port_vs_reference.py, profiles/r06_cpu_port_vs_reference.txt
            "reference_over_port": REFERENCE_OVER_PORT,
            "sample": f"oracle/vidtok_oracle.py forward, fp32, 1 unscaled clip 17x{RES}x{RES} in {t:.2f}s on {threads} of "
                      f"{cores} host threads, the best of the sweep ({config})"}, (z, dec)


def rel_err(a, b):
    """max |a - b| / max |b|: the parity metric of SURVEY.md section 8(d)"""
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def time_steps(fn, steps, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, out


MODES = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32, "bf16x3": "bf16x3"}


def mode_measurements(model, x, main_mode, seed=77):
    """rank 0, N = 1: the bench workload in the OTHER arithmetic modes (frames/s, graph-replayed like the main line) and, in
    every mode, the engine's output for clip 0 with the reference's host-side noise stream -- compared by the caller with the
    CPU oracle's output for that clip (the run cpu_baseline times anyway)."""
    B = x.shape[0]
    res = {}
    for name, dt in MODES.items():
        model.set_compute_dtype(dt)
        model.enable_graphs(False)
        model.regularization.noise_source = "host"
        torch.manual_seed(seed)
        z, dec, _ = model(x[:1].contiguous())
        res[name] = {"z": z.cpu(), "dec": dec.cpu()}
        if name != main_mode:
            model.enable_graphs(True)
            dt_s, _ = time_steps(lambda: model(x), 3 if name == "fp32" else 5)
            res[name].update(value=round(B * T_REAL / dt_s, 2), ms_per_step=round(dt_s * 1e3, 3))
    return res


def other_config_measurements(dev, x):
    """rank 0, N = 1: the BASELINE.json configurations that are not the bench line -- configs[2] (vidtok_fsq_causal_488_32768,
    B=4 17x256x256: frames/s in bf16, fp16 and bf16x3, FSQ integer codes of those modes against the fp32 kernels', which
    reproduce the CPU oracle's codes exactly in the GPU tests and in smoke()) and configs[4] (vidtok_kl_causal_488_16chn_v1_1,
    one clip of 129x256x256, t_chunk_enc = 16 tiling with decoder look-ahead, bf16; chunks replayed from the graph cache)."""
    import vidtok_amd

    out = {}
    B = x.shape[0]
    # configs[3]'s single-GPU shard: what every rank of the 8-GPU job runs (4 of its 32 clips), on one GPU -- the N = 1 anchor
    # of the scaling curve the driver measures with `bench.py --gpus N`
    name = CONFIG_NGPU
    m = vidtok_amd.load_model_from_config(os.path.join(ROOT, "configs", name + ".yaml"), verbose=False)
    randomize_weights(m, 0)
    m = m.to(dev).eval().set_compute_dtype(torch.bfloat16)
    m.enable_graphs(True)
    dt_s, o = time_steps(lambda: m(x), 10)
    out["configs[3] shard"] = {"workload": f"{name} forward (encode+KL+decode), bf16, B={B} clips = one rank's shard of the B={8 * B} / 8-GPU job, 17x256x256",
                               "unit": "frames/s", "value": round(B * T_REAL / dt_s, 2), "ms_per_step": round(dt_s * 1e3, 3),
                               "noise": "host", "output_finite": bool(torch.isfinite(o[1].cpu()).all())}
    del m, o
    torch.cuda.empty_cache()
    name = "vidtok_fsq_causal_488_32768"
    m = vidtok_amd.load_model_from_config(os.path.join(ROOT, "configs", name + ".yaml"), verbose=False)
    randomize_weights(m, 0)
    m = m.to(dev).eval()
    codes, e, hs = {}, {}, {}
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import fsq_mismatch_report       # checker (test infrastructure): which codes differ and how close to a rounding boundary

    levels = m.regularization.levels
    for mode in ("fp32", "bf16x3", "bf16", "fp16"):
        m.set_compute_dtype(MODES[mode])
        m.enable_graphs(False)
        hs[mode] = m._run_encoder(x)            # the pre-quantisation latent
        codes[mode] = m.regularization(hs[mode])[1]["indices"]
        if mode != "fp32":
            m.enable_graphs(True)
            dt_s, _ = time_steps(lambda: m(x), 5)
            e[mode] = {"value": round(B * T_REAL / dt_s, 2), "ms_per_step": round(dt_s * 1e3, 3),
                       "code_rate_vs_fp32_kernels": round(float((codes[mode] == codes["fp32"]).float().mean()), 6)}
            if mode == "bf16x3":                # the tolerance mode: every differing code, its digit and its distance from the boundary
                e[mode]["code_mismatches_vs_fp32_kernels"] = fsq_mismatch_report(levels, hs[mode], hs["fp32"], codes[mode], codes["fp32"])
    out["configs[2]"] = {"workload": f"{name} forward (encode+FSQ+decode), B={B} clips, 17x256x256", "unit": "frames/s",
                         "codes_compared": int(codes["fp32"].numel()), **e}
    del m, codes, hs
    torch.cuda.empty_cache()
    name = "vidtok_v1_1/vidtok_kl_causal_488_16chn_v1_1"
    m = vidtok_amd.load_model_from_config(os.path.join(ROOT, "configs", name + ".yaml"), verbose=False)
    randomize_weights(m, 0)
    m = m.to(dev).eval().set_compute_dtype(torch.bfloat16)
    xl = (torch.rand((1, 3, 129, RES, RES), generator=torch.Generator().manual_seed(2)) * 2 - 1).to(dev)
    e = {}
    for tiled in (True, False):
        m.use_tiling, m.t_chunk_enc, m.use_overlap = tiled, 16, True
        m.enable_graphs(True)
        dt_s, o = time_steps(lambda: m(xl), 3, warmup=2)
        e["tiled" if tiled else "untiled"] = {"value": round(129 / dt_s, 2), "ms_per_clip": round(dt_s * 1e3, 2),
                                               "output_finite": bool(torch.isfinite(o[1].cpu()).all())}
    out["configs[4]"] = {"workload": f"{name} forward, bf16, 1 clip 129x256x256; tiled = t_chunk_enc 16 + decoder look-ahead",
                         "unit": "frames/s", **e}
    return out



MFMA_KERNELS = ("conv_igemm", "conv3x3_ws2", "conv3x3_ws128", "conv3d_narrow", "conv_in8", "tblock_pair", "flash_attn")


def measure_traffic(dtype, batch, config, timeout_s=200, table=None):
    """HBM-side bytes per MFMA-kernel launch, measured NOW: two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot
    share one; counters only with --kernel-trace) over a child run of this file that does three eager steps and
    nothing else.  FETCH_SIZE is doubled (gfx950 reports half the bytes of wide streaming reads, MI355X_MICROARCH.md
    section HBM); both are KiB.  Returns (bytes_per_launch, note) or (None, why)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    out = tempfile.mkdtemp(prefix="vt_pmc_", dir="/tmp")
    vals, per = {}, {}
    labels_path = os.path.join(out, "labels.json")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", os.path.join(out, counter), "-o", "p",
                   "--", sys.executable, os.path.abspath(__file__), "--pmc-child", "--dtype", dtype, "--batch", str(batch), "--config", config,
                   "--pmc-labels", labels_path]
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE",
                      "TORCHELASTIC_RUN_ID", "ROLE_RANK", "ROLE_WORLD_SIZE"):
                env.pop(k, None)
            r = subprocess.run(cmd, cwd="/tmp", env=env, timeout=timeout_s, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            v = []
            for f in glob.glob(os.path.join(out, counter, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] == counter and any(t in row["Kernel_Name"] for t in MFMA_KERNELS):
                        v.append((int(row.get("Dispatch_Id", len(v))), float(row["Counter_Value"])))
            if not v:
                return None, f"no {counter} samples (rocprofv3 rc={r.returncode})"
            vals[counter] = sum(x for _, x in v) / len(v)
            per[counter] = [x for _, x in sorted(v)]
        if table is not None and os.path.exists(labels_path):
            # per layer group: the child's MFMA launches come in the order it recorded them, step after step
            labs = json.load(open(labels_path))
            n = len(labs)
            if n and all(len(per[c]) % n == 0 for c in per):
                groups = {}
                for c, mul in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
                    steps = len(per[c]) // n
                    for i, val in enumerate(per[c]):
                        g = groups.setdefault(tuple(labs[i % n]["label"]), {"launches": 0, "algorithmic": 0.0, "measured": 0.0})
                        g["measured"] += mul * val * 1024.0 / steps
                for lab in labs:
                    g = groups[tuple(lab["label"])]
                    g["launches"] += 1
                    g["algorithmic"] += lab["bytes"]
                table.extend({"M": k[0], "N": k[1], "K": k[2], **g} for k, g in groups.items())
        return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run"
    except Exception as e:  # a profiler problem must not cost the bench line
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(out, ignore_errors=True)


def committed_traffic(dtype, batch):
    """HBM-side bytes per MFMA-kernel launch from the committed PMC passes under profiles/ (newest round first) -- the
    fallback when the live rocprofv3 passes are not possible (no rocprofv3, time-out, empty CSV).  The files of
    different rounds name the figure differently; both spellings are read.  Returns (bytes | None, source)."""
    if batch != 4:
        return None, "committed passes are for B=4"
    import re

    sfx = "" if dtype == "bf16" else "_" + dtype
    pdir = os.path.join(ROOT, "profiles")
    names = sorted((f for f in (os.listdir(pdir) if os.path.isdir(pdir) else []) if re.fullmatch(rf"r\d+_conv_traffic_pmc{sfx}\.json", f)), reverse=True)
    for name in names:
        tpath = os.path.join(pdir, name)
        try:
            rec = json.load(open(tpath))
        except (OSError, ValueError):
            continue
        for key in ("bytes_per_launch", "traffic_bytes_per_launch"):
            if key in rec:
                return round(float(rec[key])), f"profiles/{os.path.basename(tpath)} (committed pass)"
    return None, "no committed pass under profiles/"


def selftest_host_loop(steps, batch, world):
    """Host seconds per step of the bench's real step loop with the GPU stubbed (CPU only; `--selftest-spawn --selftest-steps K`,
    all ranks at once): `model(x)` of the N-GPU workload's engine in graph-replay mode -- AutoencodingEngine.forward, the
    REAL GraphedCall.__call__ (key, cache lookup, LRU, state) for encoder and decoder, the REAL KL regularizer with its
    torch.randn draw of the reference's noise stream on this rank's share of the host threads -- where every device operation
    (graph input fill, replay, output clone, vt_kl_sample) is a no-op on preallocated tensors.  What is left is exactly the
    per-step host work a rank adds next to its GPU; with N ranks on one box it must stay far below the ~75 ms a step takes."""
    import vidtok_amd
    from vidtok_amd import ops
    from vidtok_amd.graphs import GraphedCall

    cores = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(16, cores // max(1, world))))
    cfg = vidtok_amd.load_config(os.path.join(ROOT, "configs", CONFIG_NGPU + ".yaml"))
    zc = int(cfg["model"]["params"]["encoder_config"]["params"]["z_channels"])
    with torch.device("meta"):                       # no parameters are touched: the launch sequences are stubbed
        model = vidtok_amd.load_model_from_config(cfg, verbose=False)
    model.eval()
    tl, hl = T_PADDED // 4, RES // 8
    h_enc = torch.zeros((batch, 2 * zc, tl, hl, hl))
    dec = torch.empty((batch, 3, T_REAL, RES, RES))
    z_reg, kl = torch.zeros((batch, zc, tl, hl, hl)), torch.zeros(())

    class StubGraphed(GraphedCall):
        def _on_device(self, x):
            return True

        def _fill(self, dst, x, frames):
            return dst

        def _replay(self, g):
            pass

        def _result(self, sy, borrow):
            return sy

    x = torch.empty((batch, 3, T_REAL, RES, RES))
    for name, out in (("_genc", h_enc), ("_gdec", dec)):
        g = StubGraphed(getattr(model, name).fn)
        setattr(model, name, g)
    model.use_graphs = True
    # pre-populate the two entries the way a captured graph would sit there: (graph, static input, static output, state)
    key_e = (model.encoder.compute_dtype, getattr(model, "arith", None), getattr(model.encoder, "tail_dtype", None), getattr(model.encoder, "tail_level", None))
    model._genc.entries[(tuple(x.shape), x.dtype, x.device, key_e)] = (None, x, h_enc, None)
    model._gdec.entries[(tuple(z_reg.shape), z_reg.dtype, z_reg.device, (model.decoder.compute_dtype, getattr(model, "arith", None)))] = (None, z_reg, dec, None)
    real_kl = ops.kl_sample
    ops.kl_sample = lambda h, noise: (z_reg, kl)
    try:
        for _ in range(3):
            model(x)
        dt = None
        for _ in range(3):                    # best of three blocks: a neighbour's burst on a shared host is not this loop's cost
            t0 = time.perf_counter()
            for _ in range(steps):
                out = model(x)
            d = (time.perf_counter() - t0) / steps
            dt = d if dt is None else min(dt, d)
    finally:
        ops.kl_sample = real_kl
    assert out[1] is dec and out[0] is z_reg
    return dt


def spawn_ranks(n):
    """`python bench.py --gpus N` outside a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] spawning {n} ranks: {' '.join(cmd)}", file=sys.stderr)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", choices=["bf16", "fp16", "fp32", "bf16x3"], default="bf16",
                    help="bf16 / fp16: 16-bit storage + MFMA (throughput modes; fp16 = the reference README's autocast dtype); fp32: fp32 storage + fp32 MFMA; bf16x3: fp32 storage, every "
                         "convolution as three bf16 MFMAs per product (the fast mode inside the reference's fp32 tolerance)")
    ap.add_argument("--batch", type=int, default=4, help="clips per GPU")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="print the per-layer-shape conv timeline to stderr")
    ap.add_argument("--traffic", choices=["pmc", "profile", "none"], default="pmc",
                    help="roofline.traffic: measure now with two rocprofv3 PMC passes over a child run on rank 0's GPU (default), quote profiles/, or null")
    ap.add_argument("--pmc-child", action="store_true", help="internal: three eager steps for the PMC passes, no output")
    ap.add_argument("--pmc-labels", default=None, help="internal: where the PMC child writes its per-launch labels")
    ap.add_argument("--selftest-spawn", action="store_true",
                    help="CPU/gloo check of the N-rank launch path only (no GPU work): prints the world size reached")
    ap.add_argument("--config", default=None, help="override the workload's YAML (default: BASELINE configs[1] / [3])")
    ap.add_argument("--noise", choices=["host", "device"], default="host",
                    help="KL noise of the timed step: host = the reference's CPU-generator stream through a pinned double buffer (default, the "
                         "parity graph); device = ATen philox on the GPU")
    ap.add_argument("--selftest-steps", type=int, default=0,
                    help="with --selftest-spawn: also run this many steps of the real per-step host loop of the bench (graph-replay mode: input "
                         "fill, encoder replay, KL regularizer with the host noise draw, decoder replay, output clone) with the GPU work stubbed, "
                         "and report the max-over-ranks host time per step")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the N = 1 extras of the line: the other arithmetic modes (parity_mode, modes) and BASELINE configs[2] / [4]")
    args = ap.parse_args()

    if not torch.cuda.is_available() and not args.selftest_spawn:
        raise SystemExit("bench.py needs a GPU: the vidtok_amd path has no CPU fallback")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        if not args.selftest_spawn and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible")
        raise SystemExit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    import torch.distributed as dist

    if args.selftest_spawn:
        # launch plumbing only (CPU, gloo): every rank joins, the metrics reduction runs, rank 0 reports the world
        from vidtok_amd.sharding import reduce_metrics, shard_range

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        a, b = shard_range(4 * world, world, rank)
        red = reduce_metrics(1.0 + rank, {"clips": float(b - a), "ranks": 1.0})
        rec = {"selftest": "spawn", "n_gpus": red["world"], "ranks_seen": int(round(red["ranks"])), "clips": red["clips"], "elapsed_s": red["elapsed_s"]}
        if args.selftest_steps > 0:
            host_s = selftest_host_loop(args.selftest_steps, args.batch, world)
            if world > 1:
                dist.barrier()
            h = reduce_metrics(host_s, {})
            rec.update(host_ms_per_step=round(h["elapsed_s"] * 1e3, 4), steps=args.selftest_steps,
                       host_threads_per_rank=torch.get_num_threads(),
                       host_loop="graph-replay step of bench.py: input fill, encoder replay, KL regularizer with the host noise draw, "
                                 "decoder replay, output clone; device work stubbed; max over ranks")
        if rank == 0:
            print(json.dumps(rec), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm

    import vidtok_amd
    from vidtok_amd import ops

    dtype = MODES[args.dtype]
    config = args.config or (CONFIG_1GPU if world == 1 else CONFIG_NGPU)
    model = vidtok_amd.load_model_from_config(os.path.join(ROOT, "configs", config + ".yaml"), verbose=False)
    randomize_weights(model, 0)
    model = model.to(dev).eval().set_compute_dtype(dtype)
    # KL noise: the reference's stream -- one torch.randn(shape) of the CPU generator per call (distributions.py:16-18), drawn
    # into a pinned double buffer and uploaded asynchronously while the encoder launches are still running
    # (vidtok_amd/regularizers.py): the timed graph IS the parity graph.  --noise device = ATen's philox kernel instead.
    model.regularization.noise_source = args.noise
    # host threads of this rank: the only host arithmetic of a step is that draw (82 K normals); N ranks share the box
    cores = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(16, cores // max(1, world))))
    B = args.batch
    g = torch.Generator().manual_seed(1234 + rank)
    x_host = torch.rand((B, 3, T_REAL, RES, RES), generator=g) * 2 - 1
    x = x_host.to(dev)
    clip0 = x_host[:1].clone()
    del x_host

    def step():
        return model(x)

    if args.pmc_child:           # the profiler's subject: a few eager steps (every kernel of the path), nothing else
        ops.CONV_RECORD = []
        for i in range(3):
            step()
            if i == 0 and args.pmc_labels:   # what each MFMA launch of a step is, in launch order (joined with the counters by the parent)
                json.dump([{"label": list(lab), "bytes": ops.launch_bytes(d)} for d, _, lab in ops.CONV_RECORD], open(args.pmc_labels, "w"))
            ops.CONV_RECORD = None
        torch.cuda.synchronize()
        return
    # launch mode: the engine's own per-shape hipGraph cache (vidtok_amd/graphs.py: encoder and decoder launch
    # sequences captured on their second call, replayed afterwards); --no-graph launches every kernel eagerly
    model.enable_graphs(not args.no_graph)
    for _ in range(max(3, args.warmup)):
        out = step()
    torch.cuda.synchronize()
    graph = None if args.no_graph else True
    run = step

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    from vidtok_amd.sharding import reduce_metrics

    # one tiny RCCL all_reduce; "ranks": every rank contributes 1 -- the sum is the number of ranks the collective really reached
    red = reduce_metrics(elapsed, {"frames": float(B * T_REAL * args.steps), "ranks": 1.0}, device=dev)
    elapsed, total_frames = red["elapsed_s"], red["frames"]
    # the same K steps with the OTHER noise source (host stream vs device philox), so the line shows what the reference's
    # host-side noise protocol costs per step (per rank: it is the host cost SURVEY.md section 8e names as the 8-GPU risk)
    other_noise = "device" if args.noise == "host" else "host"
    noise_rec = {"timed": args.noise, f"ms_per_step_{args.noise}": round(elapsed / args.steps * 1e3, 3), "host_threads_per_rank": torch.get_num_threads(),
                 "note": "host = one torch.randn(shape) of the CPU generator per step (the reference's stream, distributions.py:16-18), pinned "
                         "double buffer + async upload; device = ATen philox kernel"}
    if not args.no_extras:       # (--no-extras, the profiled form of the command, launches nothing but the timed step's kernels)
        model.regularization.noise_source = other_noise
        for _ in range(2):
            run()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed_other = reduce_metrics(time.perf_counter() - t0, {}, device=dev)["elapsed_s"]
        model.regularization.noise_source = args.noise
        noise_rec[f"ms_per_step_{other_noise}"] = round(elapsed_other / args.steps * 1e3, 3)
        noise_rec["host_over_device"] = round((elapsed if args.noise == "host" else elapsed_other) / (elapsed_other if args.noise == "host" else elapsed), 4)

    z, dec, log = out
    ok = bool(torch.isfinite(dec.cpu()).all()) and dec.shape == x.shape      # checked on the host: no foreign kernel in the profiled process

    # ---- roofline leg: the conv launches of one step, replayed alone from a hipGraph, timed with HIP events -----
    roof = None
    if rank == 0:
        model.enable_graphs(False)
        ops.CONV_RECORD, ops.LN_RECORD = [], []
        step()                                          # eager: records descriptors + keeps their tensors alive
        torch.cuda.synchronize()
        rec, ops.CONV_RECORD = ops.CONV_RECORD, None
        lnrec, ops.LN_RECORD = ops.LN_RECORD, None
        tl = [lab for _, _, lab in rec]

        def time_replay(records, reps=3, fn=ops.replay_convs):
            """ms per replay of `records` (that kernel only) from a hipGraph, HIP events on the launch stream"""
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2):
                fn(records)
            try:                 # hipGraphLaunch and nothing else (vt_graph_launch), as the engine replays its graphs: torch's replay() also
                ex = int(g2.raw_cuda_graph_exec()) or None    # launches two generator-state fill kernels
            except (AttributeError, RuntimeError):
                ex = None

            def replay():
                if ex is None:
                    g2.replay()
                else:
                    import ctypes as C

                    from vidtok_amd import lib as L

                    L.check(L.load().vt_graph_launch(C.c_void_p(ex), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "vt_graph_launch")

            replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                replay()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps

        conv_ms = time_replay(rec, reps=max(3, min(args.steps, 10)))
        if args.breakdown:
            groups = {}
            for r in rec:
                groups.setdefault(r[2], []).append(r)
            rows = [(lab, len(rs), time_replay(rs)) for lab, rs in groups.items()]
            print("[bench] conv launches of one step by (pixels, Cout, K=taps*Cin), each group replayed alone:", file=sys.stderr)
            tot = sum(ms for _, _, ms in rows)
            def served_by(rs):          # which kernel / tile / epilogue the dispatcher gives the group's launches (vt_conv_plan)
                kinds = []
                for d, _k, _l in rs:
                    if isinstance(d, tuple):
                        k = "flash_attn"
                    elif not hasattr(d, "ln_mode"):
                        k = "tblock_pair (fused temporal block)"
                    else:
                        pl = ops.conv_plan(d)
                        k = f"{pl['kernel']} {pl['tile'][0]}x{pl['tile'][1]}" + (" +LN" if pl["ln_fused"] else "") + \
                            (" lds-epilogue" if pl["lds_epilogue"] else "") + (" deep-ring" if pl["deep_ring"] else "") + \
                            (f" x{pl['launches']} launches" if pl["launches"] > 1 else "")
                    if k not in kinds:
                        kinds.append(k)
                return " | ".join(kinds)
            for (M, N, K), n, ms in sorted(rows, key=lambda r: -r[2]):
                print(f"[bench]   M={M:8d} N={N:4d} K={K:6d}  x{n:3d}  {ms:8.3f} ms  {2.0 * M * N * K * n / ms / 1e9:8.1f} TFLOP/s"
                      f"  {100 * ms / tot:5.1f}%   {served_by(groups[(M, N, K)])}", file=sys.stderr)
        # HBM-shaped kernel classes (SURVEY.md section 8d asks for both roofs): LayerNorm passes and the MFMA-kernel
        # launches whose FLOP per byte is below the ridge, each class replayed alone; algorithmic bytes = every operand
        # and result moved once (vidtok_amd/ops.py::launch_class)
        hbm = []
        classes = {}
        for r in rec:
            c = ops.launch_class(r[0])
            if c is not None:
                e = classes.setdefault(c[0], [[], 0])
                e[0].append(r)
                e[1] += c[1]
        for name, (rs, nbytes) in classes.items():
            ms = time_replay(rs)
            hbm.append({"class": name, "launches": len(rs), "ms_per_step": round(ms, 3), "GB_per_step": round(nbytes / 1e9, 3),
                        "achieved": round(nbytes / ms / 1e6, 1), "frac": round(nbytes / ms / 1e6 / PEAK_HBM_GBS, 4)})
        if lnrec:
            nbytes = sum(M_ * x_.shape[-1] * (x_.element_size() + y_.element_size()) for x_, y_, _, _, M_, _, _, _ in lnrec)
            ms = time_replay(lnrec, fn=ops.replay_layernorms)
            hbm.append({"class": "layernorm_act_kernel (LayerNorm + SiLU passes not fused into a producer)", "launches": len(lnrec),
                        "ms_per_step": round(ms, 3), "GB_per_step": round(nbytes / 1e9, 3), "achieved": round(nbytes / ms / 1e6, 1),
                        "frac": round(nbytes / ms / 1e6 / PEAK_HBM_GBS, 4)})
        hbm.sort(key=lambda h: -h["ms_per_step"])
        flops = FLOP_PER_PADDED_FRAME_256 * B * T_PADDED
        achieved = flops / (conv_ms * 1e-3) / 1e12
        # MACs the launches really execute: the up-sampler convs run as parity classes with pre-summed taps (2/3 resp.
        # 4/9 of the reference's MACs for the same result), so `achieved` -- algorithmic FLOPs of the reference per
        # unit over kernel time -- is an effective rate; the executed rate is reported next to it
        executed = sum(2.0 * M * N * K for (M, N, K) in tl)
        peak = PEAK_TFLOPS[args.dtype]
        # `achieved` / `frac`: algorithmic FLOPs of a step over the STEP time the line reports (ms_per_step: everything a step
        # launches -- LayerNorm passes, layout kernels, the regularizer, graph input / output copies -- not only the MFMA kernels).
        # *_kernel_only: the same FLOPs over the HIP-event time of the MFMA-kernel launches alone; *_executed: the MACs the
        # launches really execute (parity-class up-samplers do 4/9 resp. 2/3 of the reference's) over the step time.
        step_ms = elapsed / args.steps * 1e3
        roof = {"bound": "mfma", "kernel": "conv_igemm_glds_kernel + conv3x3_ws2_kernel + conv_in8_kernel + conv3d_narrow_kernel + tblock_pair_kernel + flash_attn_kernel",
                "achieved": round(flops / (step_ms * 1e-3) / 1e12, 2), "peak": peak,
                "unit": "TFLOP/s", "frac": round(flops / (step_ms * 1e-3) / 1e12 / peak, 4), "traffic": None, "traffic_source": "not measured",
                "basis": "algorithmic TFLOP per step / ms_per_step (whole step); *_kernel_only = / HIP-event time of the MFMA launches alone",
                "launches_per_step": len(tl), "kernel_ms_per_step": round(conv_ms, 3),
                "avg_launch_ms": round(conv_ms / max(1, len(tl)), 4), "algorithmic_tflop_per_step": round(flops / 1e12, 3),
                "achieved_kernel_only": round(achieved, 2), "frac_kernel_only": round(achieved / peak, 4),
                "executed_tflop_per_step": round(executed / 1e12, 3),
                "achieved_executed": round(executed / (step_ms * 1e-3) / 1e12, 2),
                "frac_executed": round(executed / (step_ms * 1e-3) / 1e12 / peak, 4),
                "achieved_executed_kernel_only": round(executed / (conv_ms * 1e-3) / 1e12, 2),
                "hbm": {"peak": PEAK_HBM_GBS, "unit": "GB/s", "classes": hbm}}
        del rec, lnrec

    # the remaining legs belong to rank 0 alone (a profiled child run of this command on its GPU, the CPU baseline on the
    # host cores): the other ranks are done -- the job's time was taken above, between the barriers
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return

    # HBM-side bytes per launch: PMC counters cannot be read inside an un-profiled process, so two rocprofv3 passes over
    # a child run of this command measure them now; the committed pass of profiles/ is the fallback (and what
    # --traffic profile quotes)
    extras_on = world == 1 and not args.no_extras
    modes = others = None
    extras_error = None
    if extras_on:
        try:        # an extra must never cost the line
            modes = mode_measurements(model, x, args.dtype)
            others = other_config_measurements(dev, x)
        except Exception as e:  # noqa: BLE001
            extras_error = f"{type(e).__name__}: {e}"
            print(f"[bench] extras failed: {extras_error}", file=sys.stderr)
    if args.traffic == "pmc":
        del model, out, z, dec
        torch.cuda.empty_cache()
        ttab = [] if args.breakdown else None
        tb, src = measure_traffic(args.dtype, B, config, table=ttab)
        if tb is not None:
            roof["traffic"], roof["traffic_source"] = round(tb), src
            if ttab:
                print("[bench] HBM-side traffic per step by (pixels, Cout, K) group: measured (PMC FETCH_SIZE x2 + WRITE_SIZE) vs algorithmic "
                      "(operands and results once):", file=sys.stderr)
                for g in sorted(ttab, key=lambda g: -g["measured"]):
                    print(f"[bench]   M={g['M']:8d} N={g['N']:4d} K={g['K']:6d}  x{g['launches']:3d}  measured {g['measured'] / 1e9:8.3f} GB  "
                          f"algorithmic {g['algorithmic'] / 1e9:8.3f} GB  ratio {g['measured'] / max(g['algorithmic'], 1):5.2f}", file=sys.stderr)
                print(f"[bench]   total measured {sum(g['measured'] for g in ttab) / 1e9:.1f} GB, algorithmic "
                      f"{sum(g['algorithmic'] for g in ttab) / 1e9:.1f} GB per step", file=sys.stderr)
        else:
            roof["traffic_source"] = f"live measurement failed: {src}"
    if roof["traffic"] is None and args.traffic != "none":
        tb, src = committed_traffic(args.dtype, B)
        if tb is not None:
            roof["traffic"] = tb
            roof["traffic_source"] = src + ("" if args.traffic == "profile" else f"; {roof['traffic_source']}")

    cpu, ref = (None, None) if args.no_cpu_baseline else cpu_baseline(config, clip0)
    parity_mode = mode_table = None
    if modes is not None:
        # every arithmetic mode on the bench workload: frames/s, and the distance of its output for clip 0 from the CPU
        # oracle's (z: the latent, recon: the reconstruction; max-norm relative).  parity_mode = the fastest mode inside the
        # reference's fp32 tolerance (north_star: 1e-3 relative, FSQ codes bit-exact): bf16x3
        mode_table = {}
        for name, r in modes.items():
            e = {"value": r.get("value", round(total_frames / elapsed, 2)), "ms_per_step": r.get("ms_per_step", round(elapsed / args.steps * 1e3, 3))}
            if ref is not None:
                e["z_rel"], e["recon_rel"] = float(f"{rel_err(r['z'], ref[0]):.3e}"), float(f"{rel_err(r['dec'], ref[1]):.3e}")
            mode_table[name] = e
        parity_mode = {"dtype": "bf16x3", **mode_table["bf16x3"],
                       "code_rate": (others or {}).get("configs[2]", {}).get("bf16x3", {}).get("code_rate_vs_fp32_kernels"),
                       "code_mismatches": (others or {}).get("configs[2]", {}).get("bf16x3", {}).get("code_mismatches_vs_fp32_kernels"),
                       "code_rate_basis": "FSQ integer codes of configs[2] (B=4, 20 480 tokens) against the fp32 kernels' codes; fp32 kernels vs "
                                          "the CPU oracle: all codes equal (tests/test_gpu_e2e.py, smoke())",
                       "tolerance": "recon / z <= 1e-3 relative to the fp32 oracle (SURVEY.md section 8d)"}

    ms = elapsed / args.steps * 1e3
    value = total_frames / elapsed
    line = {
        "metric": f"encode+decode frames/sec, {config} 17x256x256", "value": round(value, 2),
        "unit": "frames/s", "n_gpus": world, "ranks_seen": int(round(red["ranks"])), "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic uniform[-1,1] clips, random-init weights (temporal convs un-zeroed)",
        "config": {"workload": f"{config} forward (encode+KL+decode), {args.dtype}, B={B} clips/GPU, 17x256x256",
                   "global_batch": world * B, "parallelism": f"dp{world} (batch-sharded, no data-path collective; ranks_seen = the world size the RCCL "
                                                              f"metrics all_reduce counted)",
                   "launch": "hipGraph replay (engine graph cache)" if graph is not None else "eager"},
        "output_finite": ok, "noise": noise_rec, "roofline": roof, "cpu_baseline": cpu,
    }
    if parity_mode is not None:
        line["parity_mode"], line["modes"], line["other_configs"] = parity_mode, mode_table, others
    if extras_error is not None:
        line["extras_error"] = extras_error
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
