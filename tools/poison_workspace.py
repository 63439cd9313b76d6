"""Diagnostic: does any launch of the plan read workspace it has not written?  Fill every workspace buffer of a fresh plan with NaN (0xFFFF... bit
patterns for the integer ones) before its first run and compare the output with a normal run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sea_amd.models.temporal import TemporalModel
from sea_amd.engine import Plan

dev = torch.device("cuda:0")
for dtype, B, T, E, L in (("bf16", 1, 2024, 256, 1), ("bf16", 2, 333, 256, 2), ("fp32", 2, 333, 128, 2), ("bf16", 8, 2024, 256, 1)):
    torch.manual_seed(42)
    m = TemporalModel(L, E, 8, 2024, 8, 0, 3, 2, 0.0, "sea", "learnable", "mlp", "add", 1, 1, True, "adaln")
    m.set_compute_dtype(dtype); m = m.to(dev).eval()
    x = torch.randn(B, T, 3, E, generator=torch.Generator().manual_seed(1234)).to(dev)
    ib = torch.rand(B, T, 1, generator=torch.Generator().manual_seed(1235)).to(dev)
    eng = m.engine(); eng.params.sync()
    with torch.no_grad():
        ref = eng.forward(x, ib).clone()
    p = Plan(eng, B, T, "full")
    out = torch.full_like(x, float("nan"))
    p.bind(x, ib, out)
    n = 0
    for t in p._keep:
        if isinstance(t, torch.Tensor) and t.data_ptr() not in (x.data_ptr(), ib.data_ptr()):
            if t.dtype.is_floating_point:
                t.fill_(float("nan"))
            else:
                t.fill_(-1)
            n += 1
    p.run(); torch.cuda.synchronize()
    print(dtype, "B", B, "T", T, "E", E, "L", L, ": poisoned", n, "buffers; output finite:", bool(torch.isfinite(out).all()), "equal to the normal run:", torch.equal(out, ref))
