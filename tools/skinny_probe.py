"""Development aid (round 4): the shipped multiphase MLP GEMM shapes (M = 796 rows, 2 fields) under the grouped GEMM's variants — whole contraction against split-K
groups (fp32 partial outputs), ring depth via SEA_TUNE=gemm_dma_ns, the LDS-DMA loop forced via gemm_dma_force: python tools/skinny_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sea_amd import ops
from tools.bench_ops import timeit

dev = torch.device("cuda:0")
bf = torch.bfloat16
M, F = 796, 2


def run(N, K, splits):
    A = [torch.randn(M, K, device=dev).to(bf) for _ in range(F)]
    W = [(torch.randn(N, K, device=dev) * 0.02).to(bf) for _ in range(F)]
    out = []
    for S in splits:
        ks = K // S
        P = [[torch.empty(M, N, device=dev) for _ in range(S)] for _ in range(F)]
        gs = [dict(A=A[i][:, q * ks:(q + 1) * ks], W=W[i][:, q * ks:(q + 1) * ks], C32=P[i][q]) for i in range(F) for q in range(S)]
        if S == 1:
            C = [torch.empty(M, N, device=dev, dtype=bf) for _ in range(F)]
            gs = [dict(A=A[i], W=W[i], Cact=C[i]) for i in range(F)]
        us = timeit(lambda: ops.gemm_grouped(gs, bf))
        out.append(f"S={S}: {us:6.1f} us ({F * 2 * M * N * K / us / 1e6:5.0f} TF/s)")
    print(f"N={N} K={K} SEA_TUNE={os.environ.get('SEA_TUNE', '')}: " + "  ".join(out), flush=True)


run(2048, 16384, (1, 2, 4, 8))
run(16384, 2048, (1,))
