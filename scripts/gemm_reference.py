"""What the vendor GEMM (hipBLASLt through torch.matmul) sustains on this GPU, bf16 -> fp32 accumulate, on plain dense
matrices shaped like the convolution layers of the benchmark seen as GEMMs (pixels x K times K x Cout) and on a large
square.  A calibration for DESIGN.md section 5: the convolution kernels do the same MFMA work plus the gather, so the
vendor's rate on the bare GEMM is the practical ceiling for them on this power envelope (the 2.5 PFLOP/s dense peak is
not reachable by any kernel we have measured).  Not part of the product path."""
import time

import torch

CASES = [  # label, M, N, K
    ("square 8192^3", 8192, 8192, 8192),
    ("L1 3x3x3 256->256 (M=2.6M/4, K=4608... as GEMM)", 655360, 256, 4608),
    ("L0 3x3 128->128 (M=5.2M/4, K=1152)", 1310720, 128, 1152),
    ("L2 512->512 K=9216 (M=327680)", 327680, 512, 9216),
    ("L0 temporal k3 128->128 (M=5.2M/4, K=384)", 1310720, 128, 384),
]


def main():
    dev = "cuda:0"
    for label, M, N, K in CASES:
        for fill in ("randn", "zeros"):
            a = torch.randn((M, K), device=dev, dtype=torch.bfloat16) if fill == "randn" else torch.zeros((M, K), device=dev, dtype=torch.bfloat16)
            b = torch.randn((N, K), device=dev, dtype=torch.bfloat16) if fill == "randn" else torch.zeros((N, K), device=dev, dtype=torch.bfloat16)
            for _ in range(3):
                c = a @ b.t()
            torch.cuda.synchronize()
            n = 20
            t0 = time.perf_counter()
            for _ in range(n):
                c = a @ b.t()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            print(f"{label:55s} {fill:6s} {dt * 1e3:8.3f} ms  {2.0 * M * N * K / dt / 1e12:8.1f} TFLOP/s", flush=True)
            del a, b, c


if __name__ == "__main__":
    main()
