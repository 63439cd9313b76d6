"""Development aid (round 4): sea_mlp_block at the cfg2 shape (3 fields x 2024 rows, E = 256, S = 2048, norm prologue with the ib add, final AdaLN) timed
stand-alone, beside the two launches it replaces.  SEA_TUNE=blk_probe=n ends the kernel after stage n (see mlp_block.hip): python tools/mlp_probe.py [M]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sea_amd import ops
from tools.bench_ops import timeit

dev = torch.device("cuda:0")
dt = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2024
E, S, F = 256, 2048, 3


def rnd(*shape, dtype=torch.float32, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(dtype)


blk, one, two = [], [], []
out = torch.empty(M, F * E, device=dev)
for i in range(F):
    W1, b1, lnw, lnb = rnd(S, E, dtype=dt, scale=0.08), rnd(S), 1 + 0.1 * rnd(S), 0.1 * rnd(S)
    W2, b2, Wp, bp = rnd(E, S, dtype=dt, scale=0.03), rnd(E), rnd(E, E, dtype=dt, scale=0.08), rnd(E)
    x, add = rnd(M, E), rnd(M, E)
    nrm = dict(X32=x, gamma=1 + 0.1 * rnd(E), beta=0.1 * rnd(E), mod=rnd(M, 2 * E, dtype=dt, scale=0.3), addend=add)
    fin = dict(gamma=1 + 0.1 * rnd(E), beta=0.1 * rnd(E), mod=rnd(M, 2 * E, dtype=dt, scale=0.3))
    y = out[:, i * E:(i + 1) * E]
    blk.append(dict(W1=W1, b1=b1, lnw=lnw, lnb=lnb, norm=nrm, W2=W2, b2=b2, R=None, Wproj=Wp, bproj=bp, Y32=y, **fin))
    hg, xq = torch.empty(M, S + 64, device=dev, dtype=dt)[:, :S], torch.empty(M, E, device=dev)
    one.append(dict(W1=W1, b1=b1, lnw=lnw, lnb=lnb, Hg=hg, norm=dict(nrm, Xout=xq)))
    two.append(dict(Hg=hg, W2=W2, b2=b2, R=xq, Wproj=Wp, bproj=bp, Y32=y, **fin))
t_blk = timeit(lambda: ops.mlp_block(blk))
t1 = timeit(lambda: ops.mlp_fc1_ln_gelu(one))
t2 = timeit(lambda: ops.mlp_fc2_proj_norm(two))
print(f"M={M} SEA_TUNE={os.environ.get('SEA_TUNE', '')}: block {t_blk:.1f} us; fc1_ln_gelu {t1:.1f} + fc2_proj_norm {t2:.1f} = {t1 + t2:.1f} us", flush=True)
