"""Diagnostic: replay the cfg2 bf16 plan several times on the same inputs and report which workspace buffers (in allocation order) are not
bit-identical from run to run — the first one names the launch that is not deterministic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sea_amd.models.temporal import TemporalModel
from sea_amd.engine import Plan

dev = torch.device("cuda:0")
torch.manual_seed(42)
B = int(os.environ.get("B", "1"))
m = TemporalModel(1, 256, 8, 2024, 8, 0, 3, 2, 0.0, "sea", "learnable", "mlp", "add", 1, 1, True, "adaln")
m.set_compute_dtype("bf16"); m = m.to(dev).eval()
x = torch.randn(B, 2024, 3, 256, generator=torch.Generator().manual_seed(1234)).to(dev)
ib = torch.rand(B, 2024, 1, generator=torch.Generator().manual_seed(1235)).to(dev)
eng = m.engine()
eng.params.sync()
p = Plan(eng, B, 2024, "full")
out = torch.empty_like(x)
p.bind(x, ib, out)
bufs = [t for t in p._keep if isinstance(t, torch.Tensor)] + [out]
print("launches:", [r.name for r in p.records])
ref = None
for it in range(int(os.environ.get("RUNS", "12"))):
    p.run(); torch.cuda.synchronize()
    snap = [t.clone() for t in bufs]
    if ref is None:
        ref = snap
        continue
    bad = []
    for k, (a, b) in enumerate(zip(ref, snap)):
        if a.dtype.is_floating_point:
            same = torch.equal(a.view(torch.int16 if a.element_size() == 2 else torch.int32), b.view(torch.int16 if b.element_size() == 2 else torch.int32))
        else:
            same = torch.equal(a, b)
        if not same:
            d = (a.float() - b.float()).abs()
            rows = (d.reshape(d.shape[0], -1).amax(dim=1) > 0).nonzero().flatten() if d.dim() >= 2 else torch.tensor([])
            bad.append((k, tuple(a.shape), str(a.dtype)[6:], int(rows.numel()), rows[:6].tolist()))
    print("run", it, "differing buffers:", bad[:8] if bad else "none")

# ---- per-launch localisation: checksums of every buffer after every launch, run after run, until a run disagrees with the first
def checksums():
    return [int(t.view(torch.int16 if t.element_size() == 2 else torch.int32).to(torch.int64).sum().item()) if t.dtype.is_floating_point else int(t.to(torch.int64).sum().item()) for t in bufs]

recs = [r for r in p.records if r.fn is not None]
stream = torch.cuda.current_stream().cuda_stream
def traced_run():
    sums = []
    for r in recs:
        rc = r.fn(*r.args, stream)
        assert rc == 0, r.name
        torch.cuda.synchronize()
        sums.append(checksums())
    return sums
if os.environ.get("TRACED", "30") == "0":
    sys.exit(0)
base = traced_run()
found = False
for it in range(int(os.environ.get("TRACED", "30"))):
    cur = traced_run()
    for k, (s0, s1) in enumerate(zip(base, cur)):
        if s0 != s1:
            which = [i for i, (u, v) in enumerate(zip(s0, s1)) if u != v]
            print("traced run", it, ": first launch whose outputs differ:", k, recs[k].name, "buffers", [(i, tuple(bufs[i].shape), str(bufs[i].dtype)[6:]) for i in which])
            found = True
            break
print("traced: difference found" if found else "traced: no difference in the launch-by-launch runs (synchronised after every launch)")
