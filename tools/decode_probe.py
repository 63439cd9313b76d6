"""The spatial decoder's second layer (fp32 output [snapshots x patches, n_fields x n_inp], 2.5 GB per 2024-snapshot rollout) as a stand-alone grouped GEMM:
does the output row stride matter?  (n_inp = 1628: rows of 3256 / 1628 floats start at odd multiples of 16 B.)  Development aid."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sea_amd import ops
from tools.bench_ops import timeit

dev = torch.device("cuda:0")
bf = torch.bfloat16


def main():
    M, K = 2024 * 64, 480
    for ld_total, name in ((4884, "dense [M, 4884]"), (4896, "rows padded to 4896 (a multiple of 32 floats)"), (5120, "rows padded to 5120")):
        out = torch.empty(M, ld_total, device=dev)
        gs = []
        col = 0
        for n in (3256, 1628):
            A = torch.randn(M, K, device=dev).to(bf)
            W = torch.randn(n, K, device=dev).to(bf)
            gs.append(dict(A=A, W=W, bias=torch.zeros(n, device=dev), C32=out[:, col:col + n]))
            col += n
        us = timeit(lambda: ops.gemm_grouped(gs, bf), iters=5, reps=3)
        nbytes = M * 4884 * 4
        print(f"{name:48s} {us:8.1f} us  output {nbytes / us / 1e6:5.2f} TB/s  {2 * M * 4884 * K / us / 1e6:6.0f} TFLOP/s", flush=True)
    # bf16 output for comparison
    gs = []
    for n in (3256, 1628):
        A = torch.randn(M, K, device=dev).to(bf)
        W = torch.randn(n, K, device=dev).to(bf)
        gs.append(dict(A=A, W=W, bias=torch.zeros(n, device=dev), Cact=torch.empty(M, n, device=dev, dtype=bf)))
    us = timeit(lambda: ops.gemm_grouped(gs, bf), iters=5, reps=3)
    print(f"{'bf16 outputs (two dense tensors)':48s} {us:8.1f} us  {2 * M * 4884 * K / us / 1e6:6.0f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
