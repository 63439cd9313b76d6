"""Runs WARM + N plain (launch-by-launch) cfg2 forwards and nothing else, and writes the launch names in order, so that a rocprofv3
--pmc pass over this script can be folded back onto the plan's launches:  rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python tools/pmc_forward.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sea_amd.models.temporal import TemporalModel

N, WARM = 10, 3
dev = torch.device("cuda:0")
torch.manual_seed(42)
m = TemporalModel(1, 256, 8, 2024, 8, 0, 3, 2, 0.0, "sea", "learnable", "mlp", "add", 1, 1, True, "adaln")
m.set_compute_dtype("bf16"); m = m.to(dev).eval()
x = torch.randn(1, 2024, 3, 256, generator=torch.Generator().manual_seed(1234)).to(dev)
ib = torch.rand(1, 2024, 1, generator=torch.Generator().manual_seed(1235)).to(dev)
eng = m.engine()
out = torch.empty_like(x)
plan = eng.plan(1, 2024, "full"); plan.bind(x, ib, out)
torch.cuda.synchronize()
for _ in range(WARM + N):
    plan.run()
torch.cuda.synchronize()
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"n": N, "names": [r.name for r in plan.records]}, open("gpurun_out/pmc_forward_names.json", "w"))
