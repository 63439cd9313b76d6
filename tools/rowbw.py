"""Streaming rates of the row kernels at the cfg3 shapes against plain copies of the same bytes (development aid): python tools/rowbw.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sea_amd import ops
from tools.bench_ops import timeit

dev = torch.device("cuda:0")
M, F, E, S = 16192, 3, 256, 2048
bf, f32 = torch.bfloat16, torch.float32


def report(name, us, nbytes):
    print(f"{name:44s} {us:8.1f} us  {nbytes / 1e6:8.1f} MB  {nbytes / us / 1e6:6.2f} TB/s", flush=True)


def main():
    h = [torch.randn(M, S, device=dev).to(bf) for _ in range(F)]
    hg = [torch.empty(M, S, device=dev, dtype=bf) for _ in range(F)]
    big = torch.cat(h)
    big2 = torch.empty_like(big)
    report("torch copy bf16 [3M, S]", timeit(lambda: big2.copy_(big)), 2 * big.numel() * 2)
    b32 = torch.randn(M * F, E, device=dev)
    b32o = torch.empty_like(b32)
    report("torch copy f32 [3M, E]", timeit(lambda: b32o.copy_(b32)), 2 * b32.numel() * 4)
    report("torch gelu bf16 [3M, S]", timeit(lambda: torch.nn.functional.gelu(big, approximate="none")), 2 * big.numel() * 2)
    lnw, lnb = torch.ones(S, device=dev), torch.zeros(S, device=dev)
    mean, rstd = [torch.empty(M, device=dev) for _ in range(F)], [torch.empty(M, device=dev) for _ in range(F)]
    gs = [dict(X=h[i], gamma=lnw, beta=lnb, Yact=hg[i], mean=mean[i], rstd=rstd[i]) for i in range(F)]
    report("sea_rownorm LN+GELU bf16 S=2048 (mlp.ln_gelu)", timeit(lambda: ops.rownorm(gs, M, S, True, True, 1e-5, bf)), 2 * big.numel() * 2)
    gs2 = [dict(X=h[i], gamma=lnw, beta=lnb, Yact=hg[i], mean=mean[i], rstd=rstd[i]) for i in range(F)]
    report("sea_rownorm LN only bf16 S=2048", timeit(lambda: ops.rownorm(gs2, M, S, True, False, 1e-5, bf)), 2 * big.numel() * 2)
    x = [torch.randn(M, E, device=dev) for _ in range(F)]
    mod = [torch.randn(M, 2 * E, device=dev).to(bf) for _ in range(F)]
    n = [torch.empty(M, E, device=dev, dtype=bf) for _ in range(F)]
    gw, gb = torch.ones(E, device=dev), torch.zeros(E, device=dev)
    ga = [dict(X=x[i], mod=mod[i], gamma=gw, beta=gb, Yact=n[i], mean=mean[i], rstd=rstd[i]) for i in range(F)]
    report("sea_rownorm AdaLN f32->bf16 E=256 (adaln0)", timeit(lambda: ops.rownorm(ga, M, E, False, False, 1e-5, bf)), F * M * (E * 4 + 2 * E * 2 + E * 2))
    dS = [torch.randn(M, S, device=dev).to(bf) for _ in range(F)]
    dg, db = torch.zeros(S, device=dev), torch.zeros(S, device=dev)
    gbk = [dict(dY=dS[i], X=h[i], gamma=lnw, beta=lnb, mean=mean[i], rstd=rstd[i], dXact=dS[i], dgamma=dg, dbeta=db) for i in range(F)]
    ws = torch.empty(64 * 1024 * 1024 // 4, device=dev)
    report("sea_rownorm_bwd LN+GELU bf16 S=2048 (bwd.mlp.ln_gelu)", timeit(lambda: ops.rownorm_bwd(gbk, M, S, True, True, True, False, bf, ws)), 3 * big.numel() * 2)


if __name__ == "__main__":
    main()
