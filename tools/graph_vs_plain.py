"""Diagnostic: cfg2 bf16 forward through the plain replay, a second plain plan, and the captured graph; where do they differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sea_amd.models.temporal import TemporalModel
from sea_amd.engine import Plan

dev = torch.device("cuda:0")
torch.manual_seed(42)
m = TemporalModel(1, 256, 8, 2024, 8, 0, 3, 2, 0.0, "sea", "learnable", "mlp", "add", 1, 1, True, "adaln")
m.set_compute_dtype("bf16"); m = m.to(dev).eval()
x = torch.randn(1, 2024, 3, 256, generator=torch.Generator().manual_seed(1234)).to(dev)
ib = torch.rand(1, 2024, 1, generator=torch.Generator().manual_seed(1235)).to(dev)
eng = m.engine()

def rel(a, b):
    return ((a - b).norm() / b.norm()).item()

with torch.no_grad():
    a = eng.forward(x, ib).clone()
    a2 = eng.forward(x, ib).clone()
    p2 = Plan(eng, 1, 2024, "full")
    out2 = torch.empty_like(x)
    p2.bind(x, ib, out2)
    # poison the second plan's workspace before its first run
    for t in p2._bufs if hasattr(p2, "_bufs") else []:
        t.fill_(float("nan")) if t.is_floating_point() else None
    p2.run(); torch.cuda.synchronize()
    b = out2.clone()
    p2.run(); torch.cuda.synchronize()
    b2 = out2.clone()
    g1 = eng.forward_graphed(x, ib).clone()
    g2 = eng.forward_graphed(x, ib).clone()
print("plain vs plain again", torch.equal(a, a2))
print("plain vs second plan (first run)", torch.equal(a, b), rel(b, a), " second run", torch.equal(a, b2), rel(b2, a))
print("graph vs graph again", torch.equal(g1, g2), " graph vs plain", torch.equal(g1, a), rel(g1, a))
d = (g1 - a).abs()
print("nan in outputs", torch.isnan(a).any().item(), torch.isnan(b).any().item(), torch.isnan(g1).any().item())
rows = (d.amax(dim=(0, 2, 3)) > 0).nonzero().flatten()
print("rows that differ:", rows.numel(), rows[:10].tolist(), rows[-5:].tolist())
for f in range(3):
    print("field", f, "max abs diff", d[:, :, f].max().item(), "first differing row", (d[:, :, f].amax(dim=(0, 2)) > 0).nonzero().flatten()[:3].tolist())
