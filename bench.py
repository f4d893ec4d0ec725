#!/usr/bin/env python
"""bench.py -- encode+decode frames/s of the MI355X VidTok path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--dtype bf16|fp32] [--batch B] [--no-graph]

One "step" = one pass of the hot path, `model(x)` = encode -> KL regularizer -> decode
(reference AutoencodingEngine.forward, vidtok/models/autoencoder.py:221-229), over one batch of
synthetic clips x = rand(B,3,17,256,256)*2-1 that is resident in HBM before the timed region.
Workload at every N: BASELINE.json configs[1] -- vidtok_kl_causal_488_4chn, bf16, B=4 clips per GPU
(weak scaling: every rank runs its own B clips; the path has no data-path collective, SURVEY.md
section 8e; only the timing/metrics reduction crosses ranks).  Prints ONE JSON line on rank 0.

  value       real frames/s over the whole job = N*B*17*K / max-over-ranks(time of K steps)
  roofline    conv_igemm_glds_kernel (all convolutions + the attention GEMMs = every MFMA FLOP of the
              path): algorithmic FLOPs of one step (1.0345 TFLOP per padded 256x256 frame, SURVEY.md
              section 8d) / sum of that kernel's launch durations in one step, measured live with HIP
              events on the launch stream; peak = dense MFMA peak of the dtype; traffic = HBM-side
              bytes per launch from the committed rocprofv3 PMC pass (profiles/)
  cpu_baseline  the CPU oracle (port of the reference, oracle/vidtok_oracle.py) timed on this host's
              cores on a bounded sample of the same workload; a baseline, not a target
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_PADDED_FRAME_256 = 1.0345e12     # SURVEY.md section 8(d), conv + attention MACs x 2
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}   # MI355X_MICROARCH.md: dense MFMA peaks
T_REAL, T_PADDED, RES = 17, 20, 256
CONFIG = "vidtok_kl_causal_488_4chn"


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


def cpu_baseline(budget_s=25.0):
    """Time the CPU oracle on a bounded sample: one 17-frame clip at the largest square resolution in
    {64,128,256} whose forward is expected to fit the budget; throughput is scaled to 256x256 frames
    by the pixel ratio (every op of the path is linear in H*W).  The thread count is the better of
    16 / 64 (capped by the host) on a small probe: torch's CPU convolutions collapse when given all
    256 hardware threads of the GPU box (measured: 17x64x64 took 122 s on 256 threads)."""
    import vidtok_amd
    from oracle.vidtok_oracle import OracleEngine

    cores = os.cpu_count() or 1
    cfg = vidtok_amd.load_config(os.path.join(ROOT, "configs", CONFIG + ".yaml"))
    model = vidtok_amd.load_model_from_config(cfg, verbose=False)
    randomize_weights(model, 0)
    ora = OracleEngine(cfg["model"]["params"], model.state_dict())
    del model

    def run(res):
        x = torch.rand(1, 3, T_REAL, res, res) * 2 - 1
        t0 = time.perf_counter()
        ora(x)
        return time.perf_counter() - t0

    best_t, threads = None, 1
    for nt in sorted({min(cores, 16), min(cores, 64)}):
        torch.set_num_threads(nt)
        run(32)                   # warm-up (thread pool, allocator)
        t = run(64)
        if best_t is None or t < best_t:
            best_t, threads = t, nt
        if t > 10.0:
            break
    torch.set_num_threads(threads)
    t64, res = best_t, 64
    for cand in (128, 256):
        if t64 * (cand / 64) ** 2 * 1.15 <= budget_s:
            res = cand
    t = run(res) if res != 64 else t64
    fps256 = T_REAL / (t * (RES / res) ** 2)
    return {"value": round(fps256, 4), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"oracle/vidtok_oracle.py forward, fp32, 1 clip 17x{res}x{res} in {t:.2f}s on {threads} of "
                      f"{cores} host threads, scaled x{(res / RES) ** 2:.4g} to 256x256 frames"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--batch", type=int, default=4, help="clips per GPU")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="print the per-layer-shape conv timeline to stderr")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"[bench] note: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the vidtok_amd path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm

    import vidtok_amd
    from vidtok_amd import ops

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model = vidtok_amd.load_model_from_config(os.path.join(ROOT, "configs", CONFIG + ".yaml"), verbose=False)
    randomize_weights(model, 0)
    model = model.to(dev).eval().set_compute_dtype(dtype)
    model.regularization.noise_source = "device"   # reparameterisation noise drawn on the GPU (capturable)
    B = args.batch
    g = torch.Generator().manual_seed(1234 + rank)
    x = (torch.rand((B, 3, T_REAL, RES, RES), generator=g) * 2 - 1).to(dev)

    def step():
        return model(x)

    for _ in range(max(1, args.warmup)):
        out = step()
    torch.cuda.synchronize()

    graph = None
    if not args.no_graph:
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                step()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = step()
            graph.replay()
            torch.cuda.synchronize()
        except Exception as e:  # stay on the HIP path, just launch eagerly
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); launching eagerly", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()

    run = graph.replay if graph is not None else step

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    from vidtok_amd.sharding import reduce_metrics

    red = reduce_metrics(elapsed, {"frames": float(B * T_REAL * args.steps)}, device=dev)  # one tiny RCCL all_reduce
    elapsed, total_frames = red["elapsed_s"], red["frames"]

    z, dec, log = out
    ok = bool(torch.isfinite(dec).all()) and dec.shape == x.shape

    # ---- roofline leg: per-launch HIP-event timeline of the conv kernel over one eager step --------
    roof = None
    if rank == 0:
        # The eager step is enqueued behind a ~0.3 s device-side spin, so the host runs ahead and the conv kernels
        # execute back to back: an event pair then brackets kernel time only, not the Python launch gaps of an
        # un-graphed step (they inflated the short launches by 5-10 %).
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        torch.cuda._sleep(10_000_000)
        c1.record()
        torch.cuda.synchronize()
        cycles_per_ms = 10_000_000 / max(c0.elapsed_time(c1), 1e-3)
        ops.CONV_TIMELINE = []
        torch.cuda._sleep(int(300 * cycles_per_ms))
        step()
        torch.cuda.synchronize()
        tl, ops.CONV_TIMELINE = ops.CONV_TIMELINE, None
        conv_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in tl)
        if args.breakdown:
            agg = {}
            for e0, e1, (M, N, K) in tl:
                a = agg.setdefault((M, N, K), [0, 0.0])
                a[0] += 1
                a[1] += e0.elapsed_time(e1)
            print("[bench] conv launches of one step by (pixels, Cout, K=taps*Cin):", file=sys.stderr)
            for (M, N, K), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                print(f"[bench]   M={M:8d} N={N:4d} K={K:6d}  x{n:3d}  {ms:8.3f} ms  {2.0 * M * N * K * n / ms / 1e9:8.1f} TFLOP/s"
                      f"  {100 * ms / conv_ms:5.1f}%", file=sys.stderr)
        flops = FLOP_PER_PADDED_FRAME_256 * B * T_PADDED
        achieved = flops / (conv_ms * 1e-3) / 1e12
        # MACs the launches really execute: the up-sampler convs run as parity classes with pre-summed taps (2/3 resp.
        # 4/9 of the reference's MACs for the same result), so `achieved` -- algorithmic FLOPs of the reference per
        # unit over kernel time -- is an effective rate; the executed rate is reported next to it
        executed = sum(2.0 * M * N * K for _, _, (M, N, K) in tl)
        peak = PEAK_TFLOPS[args.dtype]
        # HBM-side bytes per launch come from rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE,
        # scripts/pmc_bench.sh + scripts/pmc_traffic.py); counters cannot be read inside a normal run, so the
        # figure of the committed pass for this dtype is quoted (null when there is none)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", f"r01_conv_traffic_pmc{'' if args.dtype == 'bf16' else '_' + args.dtype}.json")
        if os.path.exists(tpath) and B == 4:
            traffic = round(json.load(open(tpath))["traffic_bytes_per_launch"])
        roof = {"bound": "mfma", "kernel": "conv_igemm_glds_kernel", "achieved": round(achieved, 2), "peak": peak,
                "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                "launches_per_step": len(tl), "kernel_ms_per_step": round(conv_ms, 3),
                "avg_launch_ms": round(conv_ms / max(1, len(tl)), 4), "algorithmic_tflop_per_step": round(flops / 1e12, 3),
                "executed_tflop_per_step": round(executed / 1e12, 3),
                "achieved_executed": round(executed / (conv_ms * 1e-3) / 1e12, 2),
                "frac_executed": round(executed / (conv_ms * 1e-3) / 1e12 / peak, 4)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = total_frames / elapsed
        line = {
            "metric": "encode+decode frames/sec, vidtok_kl_causal_488_4chn 17x256x256", "value": round(value, 2),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic uniform[-1,1] clips, random-init weights (temporal convs un-zeroed)",
            "config": {"workload": f"{CONFIG} forward (encode+KL+decode), {args.dtype}, B={B} clips/GPU, 17x256x256",
                       "global_batch": world * B, "parallelism": f"dp{world} (batch-sharded, no data-path collective)",
                       "launch": "hipGraph replay" if graph is not None else "eager"},
            "output_finite": ok, "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
