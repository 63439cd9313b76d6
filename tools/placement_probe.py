"""Stand-alone durations of the cfg2 plan's GEMM-shaped launches on freshly allocated operands, to compare with their durations inside the plan
(profiles/r02_forward_cfg2_launch_breakdown.txt): a launch that is slower in the plan than alone is paying for WHERE its operands sit.  Development aid."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sea_amd import ops
from tools.bench_ops import timeit

dev = torch.device("cuda:0")
bf = torch.bfloat16
M = 2024


def gemm(groups_nk, name, **kw):
    gs = []
    for n, k in groups_nk:
        A = torch.randn(M, k, device=dev).to(bf)
        W = torch.randn(n, k, device=dev).to(bf)
        gs.append(dict(A=A, W=W, Cact=torch.empty(M, n, device=dev, dtype=bf), **kw))
    us = timeit(lambda: ops.gemm_grouped(gs, bf))
    print(f"{name:34s} {us:7.1f} us", flush=True)


def main():
    gemm([(512, 512)] * 9 + [(256, 256)] * 3, "adaln.cond_gemm (12 modules)")
    gemm([(256, 256)] * 3, "self.out_proj-like (3 x 256x256)")
    H, T = 8, 2024
    for hd, nprob, nm in ((32, 3, "self.attention"), (16, 2, "cross.attention")):
        probs = []
        for _ in range(nprob):
            Q = torch.randn(1, H, T, hd, device=dev).to(bf)
            K = torch.randn(1, H, T, hd, device=dev).to(bf)
            Vt = torch.randn(1, H, hd, T, device=dev).to(bf)
            probs.append(dict(Q=Q, K=K, Vt=Vt, O=torch.empty(1, T, H * hd, device=dev, dtype=bf)))
        us = timeit(lambda: ops.attention_fwd(probs, 1, H, hd, T, T, T, 0, 0, bf))
        print(f"{nm:34s} {us:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
