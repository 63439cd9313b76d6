"""Does a padded row stride of the weights help the 32-row MLP kernels?  (W2's rows are 4 KiB apart: a K-tile of a stage is 256 pieces of 128 B at a 4 KiB stride.)
Development aid: python tools/stride_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sea_amd import ops
from tools.bench_ops import timeit

dev = torch.device("cuda:0")
bf = torch.bfloat16


def main():
    M, E, S, F = 2024, 256, 2048, 3
    for pad2, pad1, padp, padh in ((0, 0, 0, 0), (64, 0, 0, 0), (64, 64, 64, 0), (8, 8, 8, 0), (256, 0, 0, 0), (0, 0, 0, 64), (0, 0, 0, 256), (64, 0, 0, 64)):
        g2, g1 = [], []
        for i in range(F):
            Hg = torch.randn(M, S + padh, device=dev).to(bf)[:, :S]
            W2 = torch.randn(E, S + pad2, device=dev).to(bf)[:, :S]
            Wp = torch.randn(E, E + padp, device=dev).to(bf)[:, :E]
            g2.append(dict(Hg=Hg, W2=W2, b2=torch.zeros(E, device=dev), R=torch.randn(M, E, device=dev), Wproj=Wp, bproj=torch.zeros(E, device=dev),
                           Y32=torch.empty(M, E, device=dev), gamma=torch.ones(E, device=dev)))
            A = torch.randn(M, E, device=dev).to(bf)
            W1 = torch.randn(S, E + pad1, device=dev).to(bf)[:, :E]
            g1.append(dict(A=A, W1=W1, b1=torch.zeros(S, device=dev), lnw=torch.ones(S, device=dev), lnb=torch.zeros(S, device=dev),
                           Hg=torch.empty(M, S + padh, device=dev, dtype=bf)[:, :S]))
        t2 = timeit(lambda: ops.mlp_fc2_proj_norm(g2))
        t1 = timeit(lambda: ops.mlp_fc1_ln_gelu(g1))
        print(f"row padding W2 +{pad2}, W1 +{pad1}, Wproj +{padp}, Hg +{padh} elements: fc2_proj_norm {t2:6.1f} us   fc1_ln_gelu {t1:6.1f} us", flush=True)


if __name__ == "__main__" and len(sys.argv) == 1:
    main()


def cond_probe():
    """The AdaLN condition GEMM (12 modules, A = silu rows [M, 512] / [M, 256]) with padded operand / output rows."""
    M = 2024
    for pa, pc in ((0, 0), (64, 0), (0, 64), (64, 64)):
        gs = []
        for n, k in [(512, 512)] * 9 + [(256, 256)] * 3:
            A = torch.randn(M, k + pa, device=dev).to(bf)[:, :k]
            W = torch.randn(n, k, device=dev).to(bf)
            gs.append(dict(A=A, W=W, Cact=torch.empty(M, n + pc, device=dev, dtype=bf)[:, :n]))
        us = timeit(lambda: ops.gemm_grouped(gs, bf))
        print(f"cond GEMM: A rows +{pa}, output rows +{pc} elements: {us:6.1f} us", flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "cond":
    cond_probe()
