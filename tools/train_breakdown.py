"""Per-launch device times of the training plan (forward + backward) at a bench shape: python tools/train_breakdown.py B [T]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sea_amd.models.temporal import TemporalModel
from sea_amd import _native as N

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 2024
dev = torch.device("cuda:0")
torch.manual_seed(42)
m = TemporalModel(1, 256, 8, 2024, 8, 0, 3, 2, 0.0, "sea", "learnable", "mlp", "add", 1, 1, True, "adaln")
m.set_compute_dtype("bf16"); m = m.to(dev).train()
x = torch.randn(B, T, 3, 256, device=dev); tgt = torch.randn_like(x); ib = torch.rand(B, T, 1, device=dev)
eng = m.engine()
out, plan = eng.forward_train(x, ib)
loss, dout = eng.mse_loss_and_grad(out, tgt)
eng.backward(plan, dout)
torch.cuda.synchronize()

def time_list(recs, iters=5):
    stream = N.stream_ptr(); n = len(recs); tot = [0.0] * n
    for _ in range(iters):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        evs[0].record()
        for k, r in enumerate(recs):
            rc = r.fn(*r.args, stream); assert rc == 0, r.name
            evs[k + 1].record()
        torch.cuda.synchronize()
        for k in range(n): tot[k] += evs[k].elapsed_time(evs[k + 1])
    return [(r.name, t / iters * 1e3) for r, t in zip(recs, tot)]

for title, recs in (("FORWARD", plan.records), ("BACKWARD", plan.bwd)):
    ts = time_list(recs)
    print(f"== {title}: {sum(t for _, t in ts):9.1f} us in {len(ts)} launches")
    agg = {}
    for n_, t in ts:
        key = n_.replace("cross0", "crossI").replace("cross1", "crossI").replace("cross2", "crossI")
        agg[key] = agg.get(key, 0.0) + t
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
        print(f"   {v:9.1f} us  {k}")
