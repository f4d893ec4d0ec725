"""the layer class -> kernel -> measured table, from a `bench.py --breakdown` stderr file:
    python scripts/state_table.py profiles/r06_bench_bf16_conv_breakdown.txt"""
import re
import sys

KIND = {1152: "3×3, Cin 128", 768: "temporal k = 3, Cin 256", 2304: "3×3, Cin 256", 4608: "3×3, Cin 512 / time up-sampler parity 2×3×3, Cin 256",
        9216: "time up-sampler parity 2×3×3, Cin 512", 13824: "3×3×3, Cin 512", 6912: "3×3×3, Cin 256", 1536: "temporal k = 3, Cin 512",
        1024: "time up-sampler parity k = 2, Cin 512 / space parity 2×2, Cin 256", 2048: "space up-sampler parity 2×2, Cin 512", 512: "1×1, Cin 512",
        256: "1×1 shortcut, Cin 256", 128: "1×1, Cin 128", 216: "conv_in 3×3×3, 3 (8 stored) channels", 3456: "conv_out 3×3×3 → 3 channels"}


def kind(M, N, K):
    if N == 128 and K == 768:
        return "ResnetCausalBlock1D at C = 128: both k = 3 convolutions, LN1, LN2, residual (+ the next norm)"
    if N == 1024:
        return "attention: QKᵀ and PV"
    return KIND.get(K, "")


def main():
    rows = []
    for ln in open(sys.argv[1]):
        m = re.match(r"\[bench\]\s+M=\s*(\d+) N=\s*(\d+) K=\s*(\d+)\s+x\s*(\d+)\s+([\d.]+) ms\s+([\d.]+) TFLOP/s\s+([\d.]+)%\s*(.*)", ln)
        if m:
            rows.append((int(m[1]), int(m[2]), int(m[3]), int(m[4]), float(m[5]), float(m[6]), m[8].strip().replace(" | ", "; ")))
    tot = sum(r[4] for r in rows)
    print("| pixels M | Cout | K | launches | ms per step | TFLOP/s (executed) | share | layer class | kernel, tile, epilogue (vt_conv_plan) |")
    print("|---|---|---|---|---|---|---|---|---|")
    for M, N, K, n, ms, tf, kern in rows:
        if ms < 0.3:
            continue
        sp = lambda v: f"{v:,.0f}".replace(",", " ")  # noqa: E731
        print(f"| {sp(M)} | {N} | {sp(K)} | {n} | {ms:.2f} | {sp(tf)} | {100 * ms / tot:.1f} % | {kind(M, N, K)} | {kern} |")
    rest = [r for r in rows if r[4] < 0.3]
    print(f"| … | | | {sum(r[3] for r in rest)} | {sum(r[4] for r in rest):.2f} | | {100 * sum(r[4] for r in rest) / tot:.1f} % | {len(rest)} smaller groups "
          f"(1×1 shortcuts and q / k / v at the deep levels, the encoder's conv_out, …) | igemm 256×256 / 256×32, flash_attn |")
    print(f"| **all MFMA-kernel launches** | | | {sum(r[3] for r in rows)} | **{tot:.2f}** | | | | |")


if __name__ == "__main__":
    main()
