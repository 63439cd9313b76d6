import os, sys, torch
sys.path.insert(0, "/root/repo")
from sea_amd import ops
dev = torch.device("cuda:0"); dt = torch.bfloat16
def gelu(x): return torch.nn.functional.gelu(x)
M, E, S, ks = int(os.environ.get("M", "77")), 256, 2048, int(os.environ.get("KSPLIT", "2"))
torch.manual_seed(0)
A = torch.randn(M, E, device=dev).to(dt); W1 = (0.08 * torch.randn(S, E, device=dev)).to(dt); b1 = 0.3 * torch.randn(S, device=dev)
lnw = 1 + 0.1 * torch.randn(S, device=dev); lnb = 0.1 * torch.randn(S, device=dev)
W2 = (0.03 * torch.randn(E, S, device=dev)).to(dt); b2 = 0.2 * torch.randn(E, device=dev); R = torch.randn(M, E, device=dev)
h = torch.empty(M, S, device=dev, dtype=dt); st = torch.empty(M, S // 32, 2, device=dev); out = torch.zeros(ks, M, E, device=dev, dtype=dt)
ops.gemm_grouped([dict(A=A, W=W1, bias=b1, Cact=h, stats=st)], dt)
ops.mlp_fc2_ln_gelu([dict(H=h, stats=st, lnw=lnw, lnb=lnb, W2=W2, b2=b2, R=R, Out=out)], ksplit=ks)
pre = A.float() @ W1.float().t() + b1
act = gelu(torch.nn.functional.layer_norm(pre, (S,), lnw, lnb, 1e-5))
ref = R + act @ W2.float().t() + b2
got = out.float().sum(0)
print("probe", os.environ.get("SEA_MLP2_PROBE", "0"), "rel", float((got - ref).norm() / ref.norm()))
# per k-half partials
for p in range(ks):
    span = S // ks
    part = act[:, p * span:(p + 1) * span] @ W2.float()[:, p * span:(p + 1) * span].t() + ((R + b2) if p == 0 else 0)
    print("  segment", p, "rel", float((out[p].float() - part).norm() / part.norm()))
