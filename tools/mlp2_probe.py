"""Development aid: time sea_mlp_fc2_ln_gelu (and the fc1 launch in front of it) alone at the cfg2 shape, 20 back-to-back launches in a captured graph;
SEA_MLP2_PROBE switches parts of the kernel off (see mlp_fused.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sea_amd import ops, _native as N

dev = torch.device("cuda:0")
dt = torch.bfloat16
M, E, S, F = int(os.environ.get("M", "2024")), 256, 2048, 3
ks = int(os.environ.get("KSPLIT", "2"))
g1, g2 = [], []
for i in range(F):
    A = torch.randn(M, E, device=dev).to(dt)
    W1, b1 = (0.08 * torch.randn(S, E, device=dev)).to(dt), 0.3 * torch.randn(S, device=dev)
    lnw, lnb = 1 + 0.1 * torch.randn(S, device=dev), 0.1 * torch.randn(S, device=dev)
    W2, b2 = (0.03 * torch.randn(E, S, device=dev)).to(dt), 0.2 * torch.randn(E, device=dev)
    R = torch.randn(M, E, device=dev)
    h = torch.empty(M, S, device=dev, dtype=dt)
    st = torch.empty(M, S // 32, 2, device=dev)
    out = torch.empty(ks, M, E, device=dev, dtype=dt)
    g1.append(dict(A=A, W=W1, bias=b1, Cact=h, stats=st))
    g2.append(dict(H=h, stats=st, lnw=lnw, lnb=lnb, W2=W2, b2=b2, R=R, Out=out))
ops.gemm_grouped(g1, dt)
ops.mlp_fc2_ln_gelu(g2, ksplit=ks)
torch.cuda.synchronize()

def timed(fn, reps=20):
    st = torch.cuda.Stream()
    gk = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(gk, stream=st):
            for _ in range(reps):
                fn()
    gk.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gk.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best * 1e3

print(f"M={M} ksplit={ks} probe={os.environ.get('SEA_MLP2_PROBE', '0')}: fc1+stats {timed(lambda: ops.gemm_grouped(g1, dt)):.1f} us, ln_gelu_fc2 {timed(lambda: ops.mlp_fc2_ln_gelu(g2, ksplit=ks)):.1f} us")

# the one-launch form of the first half (sea_mlp_fc1_ln_gelu) and the plain second Linear, for comparison
hg = [torch.empty(M, S, device=dev, dtype=dt) for _ in range(F)]
go = [dict(A=g["A"], W1=g["W"], b1=g["bias"], lnw=d["lnw"], lnb=d["lnb"], Hg=h_) for g, d, h_ in zip(g1, g2, hg)]
xm = [torch.empty(M, E, device=dev, dtype=dt) for _ in range(F)]
gf2 = [dict(A=h_, W=d["W2"], bias=d["b2"], R=d["R"], Cact=x_) for h_, d, x_ in zip(hg, g2, xm)]
ops.mlp_fc1_ln_gelu(go); ops.gemm_grouped(gf2, dt); torch.cuda.synchronize()
print(f"   one-launch fc1+LN+GELU {timed(lambda: ops.mlp_fc1_ln_gelu(go)):.1f} us, plain fc2 {timed(lambda: ops.gemm_grouped(gf2, dt)):.1f} us")
