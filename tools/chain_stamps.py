"""In-kernel time stamps of every sea_rowchain launch of the cfg2 forward plan (workgroup (0,0)): python tools/chain_stamps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sea_amd.models.temporal import TemporalModel
from sea_amd import _native as N

dev = torch.device("cuda:0")
torch.manual_seed(42)
m = TemporalModel(1, 256, 8, 2024, 8, 0, 3, 2, 0.0, "sea", "learnable", "mlp", "add", 1, 1, True, "adaln")
m.set_compute_dtype("bf16"); m = m.to(dev).eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
x = torch.randn(B, 2024, 3, 256).to(dev); ib = torch.rand(B, 2024, 1).to(dev)
eng = m.engine(); out = torch.empty_like(x)
plan = eng.plan(B, 2024, "full"); plan.bind(x, ib, out)
bufs = {}
for r in plan.records:
    if r.fn is N.lib().sea_rowchain:
        prog, L = r.keep
        t = torch.zeros(64, dtype=torch.int64, device=dev); bufs[r.name] = (t, prog); L.dbg = t.data_ptr()
for _ in range(5):
    plan.run()
torch.cuda.synchronize()
for name, (t, prog) in bufs.items():
    v = t.cpu().tolist(); n = prog.first[1] - prog.first[0]
    st = [(v[i + 1] - v[i]) / 100.0 for i in range(0, n + 1)]  # 100 MHz -> us
    kinds = [prog.host[i].kind for i in range(n)]
    print(f"{name:28s} total {sum(st):6.2f} us | warm {st[0]:5.2f} | " + " ".join(f"k{k}:{s:4.2f}" for k, s in zip(kinds, st[1:])))
    fine = []
    for i in range(n):
        if kinds[i] != 2 and v[32 + 2 * i]:
            a, b = v[32 + 2 * i], v[33 + 2 * i]
            fine.append(f"[{i}] gemm/load {(a - v[1 + i]) / 100:4.2f} epi {(b - a) / 100 if b else -1:4.2f} norm {(v[2 + i] - b) / 100 if b else -1:4.2f}")
    print("      " + " ".join(fine))
