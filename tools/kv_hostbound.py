import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sea_amd.models.temporal import TemporalModel
from sea_amd.utils.train_utils import rollout
dev = torch.device("cuda:0")
for E, ln in ((1024, "adaln"), (2048, "ln")):
    torch.manual_seed(42)
    m = TemporalModel(1, E, 8, 2024, 8, 0, 2, 2, 0.0, "sea", "learnable", "mlp", "add", 1, 1, True, ln)
    m.set_compute_dtype("bf16"); m = m.to(dev).eval()
    x0 = torch.randn(1, 1, 2, E, device=dev); ib = torch.rand(1, 100, 1, device=dev)
    for _ in range(2): rollout(m, x0, ib, 100, mode="kv")
    torch.cuda.synchronize()
    t0 = time.perf_counter(); r = rollout(m, x0, ib, 100, mode="kv"); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(E, "host enqueue %.2f ms, total %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
