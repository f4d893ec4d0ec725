"""Where does a K step of the 8-wave 256x256 implicit-GEMM tile spend its cycles?  Records the descriptor of a
benchmark-sized layer (3x3x3 256->256 at 128x128, B=4, 20 frames: M = 1.3 M pixels, 108 K steps of 64), replays it
through vt_conv_profile and prints, per wave of workgroup 0, the shader-clock ticks between phase boundaries of K steps
8..11: [wait for my DMA pieces] [barrier] [address set-up of the step after next] [32 MFMAs with 8 DMA pieces and 24
ds_read_b128 in between].  MFMA-bound time of a step: 32 MFMAs x 32 cycles = 1 024 per wave, two waves per SIMD."""
import ctypes as C
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vidtok_amd import lib as L  # noqa: E402
from vidtok_amd import ops  # noqa: E402
from vidtok_amd.ops import ConvGeom  # noqa: E402

# plain schedule (VT_CONV_SCHED=0): stamps at stage start / after the vmcnt wait / after the barrier / after the address
# set-up / at stage end; schedule 1: stage start / in front of the waits of the last sub-step / after vmcnt / after the
# barrier / stage end; schedule 2 (two-group ping-pong): the eight phase boundaries of a step.  NB: a stamp is an s_memtime plus
# its lgkmcnt(0) -- it drains the fragment reads in flight, so the stamped kernel is slower than the shipped one (K = 13 824
# layers: ~2 400 cycles per step by the launch time against ~3 100 here); the PROPORTIONS are what to read
SCHED = 2            # the two-group ping-pong (bf16) / schedule 5 (split-bf16): the shipped schedules; the others are in the history
X3 = os.environ.get("MODE", "bf16") == "bf16x3"      # split-bf16 arithmetic: fp32 tensors, schedule 3 (K steps of 16)
if X3:
    NAMES = ["LOAD: 12 ds_read + set-up + 4 pieces + x split", "waits + barrier", "COMPUTE: 24 MFMAs", "barrier"]
elif SCHED == 0:
    NAMES = ["wait my DMA (vmcnt)", "barrier", "prep_step (addresses)", "MFMAs + DMA issue + ds_read"]
elif SCHED == 1:
    NAMES = ["sub-steps 0-2: 24 MFMAs + 8 DMA pieces + set-up", "wait lgkm + my DMA (vmcnt)", "barrier", "sub-step 3: 8 MFMAs + next fragments"]
else:
    NAMES = ["LOAD(2s): 16 ds_read + 2 pieces", "waits + barrier", "COMPUTE(2s): 16 MFMAs", "barrier", "LOAD(2s+1): 8 ds_read + set-up + 6 pieces",
             "waits + barrier", "COMPUTE(2s+1): 16 MFMAs"]
NS = len(NAMES)


def main():
    dev = "cuda:0"
    cases = [("3x3x3 256->256 @128^2 (K = 6912)", (4, 20, 128, 128), 256, 256, ConvGeom(kt=3, kh=3, kw=3, pt=2, ph=1, pw=1, ph_hi=1, pw_hi=1), 27),
             ("3x3 512->512 @64^2 (K = 4608)", (4, 10, 64, 64), 512, 512, ConvGeom(kh=3, kw=3, ph=1, pw=1, ph_hi=1, pw_hi=1), 9)]
    for label, (B, T, H, W), cin, cout, geom, taps in cases:
        torch.manual_seed(0)
        x = torch.randn((B, T, H, W, cin), device=dev, dtype=torch.float32 if X3 else torch.bfloat16)
        w = (torch.randn((cout, taps * cin), device=dev) / math.sqrt(taps * cin)).to(torch.bfloat16)
        if X3:
            from vidtok_amd.packing import pack_split3
            w = pack_split3(w.float())
        bias = torch.randn((cout,), device=dev)
        ops.CONV_RECORD = []
        y = ops.conv(x, w, bias, geom, cout=cout)
        rec, ops.CONV_RECORD = ops.CONV_RECORD, None
        d = rec[0][0]
        plan = ops.conv_plan(d)
        modes = [(0, "as shipped"), (1, "activation pieces = zero fills"), (2, "weight pieces = zero fills"), (3, "no memory traffic in the K loop"), (4, "no DMA requests"), (8, "no address arithmetic"), (12, "no DMA requests, no address arithmetic")]
        if X3:    # schedule 3 only: bit 4 = no x split, bit 5 = no MFMAs
            modes += [(16, "no x split"), (28, "no DMA requests, no address arithmetic, no x split"), (32, "no MFMAs"), (60, "fragment reads and barriers only")]
        nsteps_k = taps * cin // (16 if X3 else 64)
        for pm, pml in modes:
            L.set_option("ws_prof_mode", pm)
            stamps = torch.zeros((8, 4, 8), dtype=torch.int64, device=dev)
            lib = L.load()
            for _ in range(2):
                L.check(lib.vt_conv_profile(C.byref(d), stamps.data_ptr(), None), "vt_conv_profile")
            torch.cuda.synchronize()
            # launch time of the instrumented kernel (only workgroup 0 takes stamps): what the mode does to the real K step
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                L.check(lib.vt_conv_profile(C.byref(d), stamps.data_ptr(), None), "vt_conv_profile")
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            rounds = -(-plan["workgroups"] // 256)
            print(f"{label}: {pml}: launch {ms:.3f} ms = {ms * 1e3 / rounds / nsteps_k:.3f} us per K step and tile round ({rounds} rounds x {nsteps_k} steps, epilogue included)")
            s = stamps.cpu()
            print(f"{label}: tile {plan['tile']}, {plan['workgroups']} workgroups, {pml}")
            for wv in range(8 if pm == 0 else 0):
                for st in range(4):
                    dl = [int(s[wv, st, k + 1] - s[wv, st, k]) for k in range(NS)]
                    nxt = int(s[wv, st + 1, 0] - s[wv, st, 0]) if st < 3 else sum(dl)
                    print(f"  wave {wv} step {8 + st}: step period {nxt:6d} | " + " | ".join(f"{n} {v}" for n, v in zip(NAMES, dl)))
            L.set_option("ws_prof_mode", 0)
            avg = [sum(int(s[wv, st, k + 1] - s[wv, st, k]) for wv in range(8) for st in range(4)) / 32 for k in range(NS)]
            per = sum(int(s[wv, st + 1, 0] - s[wv, st, 0]) for wv in range(8) for st in range(3)) / 24
            print(f"  average step period {per:.0f} cycles (MFMA-bound: {'1536 per SIMD = 2 waves x 24' if X3 else '2048 per SIMD = 2 waves x 32'} MFMAs x 32); phases of one wave:")
            for n, v in zip(NAMES, avg):
                print(f"    {n:50s} {v:8.1f}")


if __name__ == "__main__":
    main()
