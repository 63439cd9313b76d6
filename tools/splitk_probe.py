"""Is a split-K fc2 worth building?  The cfg2 fc2 (M = 2024, N = 256, K = 2048, 3 fields) as 3 groups of the grouped GEMM against the same contraction cut
into 4 K-quarters (12 groups writing fp32 partial tiles) — development aid: python tools/splitk_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sea_amd import ops
from tools.bench_ops import timeit

dev = torch.device("cuda:0")
bf = torch.bfloat16


def main():
    for M in (2024, 16192):
        F, N, K = 3, 256, 2048
        A = [torch.randn(M, K, device=dev).to(bf) for _ in range(F)]
        W = [torch.randn(N, K, device=dev).to(bf) for _ in range(F)]
        C = [torch.empty(M, N, device=dev) for _ in range(F)]
        us = timeit(lambda: ops.gemm_grouped([dict(A=A[i], W=W[i], C32=C[i]) for i in range(F)], bf))
        print(f"M={M}: fc2 as 3 groups, K=2048: {us:7.1f} us  ({F * 2 * M * N * K / us / 1e6:6.1f} TFLOP/s)", flush=True)
        for KQ in (2, 4):
            P = [[torch.empty(M, N, device=dev) for _ in range(KQ)] for _ in range(F)]
            ks = K // KQ
            gs = [dict(A=A[i][:, q * ks:(q + 1) * ks], W=W[i][:, q * ks:(q + 1) * ks], C32=P[i][q]) for i in range(F) for q in range(KQ)]
            us = timeit(lambda: ops.gemm_grouped(gs, bf))
            print(f"M={M}: fc2 as {F * KQ} groups, K={ks} (fp32 partials): {us:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
