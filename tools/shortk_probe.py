"""Development aid: cfg3's short-contraction GEMM launches (3 fields x 16192 rows) under the grouped GEMM's variants (SEA_TUNE): python tools/shortk_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sea_amd import ops
from tools.bench_ops import timeit

dev = torch.device("cuda:0")
bf = torch.bfloat16


def run(name, G, M, N, K):
    A = [torch.randn(M, K, device=dev).to(bf) for _ in range(G)]
    W = [(torch.randn(N, K, device=dev) * 0.02).to(bf) for _ in range(G)]
    C = [torch.empty(M, N, device=dev, dtype=bf) for _ in range(G)]
    b = [torch.zeros(N, device=dev) for _ in range(G)]
    gs = [dict(A=A[i], W=W[i], bias=b[i], Cact=C[i]) for i in range(G)]
    us = timeit(lambda: ops.gemm_grouped(gs, bf))
    mb = G * (M * K + N * K + M * N) * 2 / 1e6
    print(f"{name:12s} {G} x ({M}, N={N}, K={K}) SEA_TUNE={os.environ.get('SEA_TUNE', ''):36s}: {us:6.1f} us  {G * 2 * M * N * K / us / 1e6:5.0f} TF/s  {mb / us / 1e3:5.2f} TB/s", flush=True)


run("mlp.fc1", 3, 16192, 2048, 256)
run("mlp.fc2", 3, 16192, 256, 2048)
run("cond(9)", 9, 16192, 512, 512)
run("qkv", 3, 16192, 768, 256)
