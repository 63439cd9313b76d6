"""Diagnostic: replay ONE launch of the cfg2 bf16 plan (REC, default self.qkv_rope) many times on fixed inputs and describe the elements that change."""
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
eng = m.engine(); eng.params.sync()
p = Plan(eng, 1, 2024, "full")
out = torch.empty_like(x)
p.bind(x, ib, out)
p.run(); torch.cuda.synchronize()
bufs = [t for t in p._keep if isinstance(t, torch.Tensor)]
name = os.environ.get("REC", "self.qkv_rope")
recs = [r for r in p.records if r.fn is not None]
k = [r.name for r in recs].index(name)
# bring the workspace to the state just before launch k
stream = torch.cuda.current_stream().cuda_stream
for r in recs[:k]:
    assert r.fn(*r.args, stream) == 0
torch.cuda.synchronize()
pre = [t.clone() for t in bufs]
def one():
    for t, s in zip(bufs, pre):
        t.copy_(s)
    assert recs[k].fn(*recs[k].args, stream) == 0
    torch.cuda.synchronize()
    return [t.clone() for t in bufs]
ref = one()
if "qkv" in name:   # which of the launch's outputs is which buffer
    arr, n = recs[k].args[0], recs[k].args[1]
    for gi in range(n):
        g = arr[gi]
        for field in ("Qout", "Kout", "Vtout"):
            ptr = getattr(g, field)
            for i, t in enumerate(bufs):
                if ptr and t.data_ptr() == ptr:
                    print("group", gi, field, "= buffer", i, "N", g.N, "col0", g.col0)
n_bad = 0
for it in range(int(os.environ.get("RUNS", "200"))):
    cur = one()
    for i, (a, b) in enumerate(zip(ref, cur)):
        if not torch.equal(a.view(torch.int16 if a.element_size() == 2 else torch.int32), b.view(torch.int16 if b.element_size() == 2 else torch.int32)):
            idx = (a.float() != b.float()).nonzero()
            n_bad += 1
            if n_bad <= 6:
                print("run", it, "buffer", i, tuple(a.shape), "elements", idx.shape[0], "min idx", idx.min(dim=0).values.tolist(), "max idx", idx.max(dim=0).values.tolist(),
                      "max |diff|", (a.float() - b.float()).abs().max().item())
                if a.dim() == 4:
                    _, h_, t_, d_ = idx[0].tolist()
                    d0 = d_ & ~3
                    for tt in (t_, t_ + 1, t_ + 7):
                        print("   t", tt, "ref d%d..%d" % (d0, d0 + 3), [round(v, 4) for v in a[0, h_, tt, d0:d0 + 4].float().tolist()], "bad", [round(v, 4) for v in b[0, h_, tt, d0:d0 + 4].float().tolist()])
print(name, ": runs that differ from the first:", n_bad)
