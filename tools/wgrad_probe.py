"""The cfg3 weight-gradient launches stand-alone (fc1: dW1 [2048, 256] x 3 fields over 16192 rows; fc2: [256, 2048]; the condition matrices), timed, and —
under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` — their HBM-side fetch bytes per launch.  Development aid.

    python tools/wgrad_probe.py [repeats]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sea_amd import ops
from tools.bench_ops import timeit

dev = torch.device("cuda:0")
bf = torch.bfloat16
M = 8 * 2024


def case(name, shapes, ld_pad=64):
    gs = []
    for n, k in shapes:
        dY = torch.randn(M, n + (ld_pad if n % 1024 == 0 else 0), device=dev).to(bf)[:, :n]
        X = torch.randn(M, k + (ld_pad if k % 1024 == 0 else 0), device=dev).to(bf)[:, :k]
        gs.append(dict(dY=dY, X=X, dW=torch.zeros(n, k, device=dev), db=torch.zeros(n, device=dev)))
    rep = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    if rep:
        for _ in range(rep):
            ops.wgrad_grouped(gs, bf)
        torch.cuda.synchronize()
        return
    us = timeit(lambda: ops.wgrad_grouped(gs, bf))
    gf = sum(2.0 * M * n * k for n, k in shapes) / 1e9
    mb = sum(2.0 * M * (n + k) for n, k in shapes) / 1e6
    print(f"{name:28s} {us:7.1f} us  {gf / us * 1e3:6.0f} TF/s  algorithmic {mb:6.1f} MB -> {mb / us:5.2f} TB/s", flush=True)


case("fc1.wgrad  3 x [2048,256]", [(2048, 256)] * 3)
case("fc2.wgrad  3 x [256,2048]", [(256, 2048)] * 3)
case("cond.wgrad 9x512^2+3x256^2", [(512, 512)] * 9 + [(256, 256)] * 3)
case("qkv.wgrad  3 x [768,256]", [(768, 256)] * 3)
