"""Times sea_exchange_tail at the cfg2 shape (M = 2024, D = 128, E = 256, 2 segments), 50 back-to-back launches in one captured graph:
python tools/tail_probe.py [has_down].  (The probe build behind DESIGN.md §5 — stages switched off one at a time — was this script over a kernel
with a debug switch; the switch is gone, the timing harness stays.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sea_amd import ops

has_down = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev, dt = torch.device("cuda:0"), torch.bfloat16
M, D, E, S = 2024, 128, 256, 2
g = torch.Generator().manual_seed(0)
rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev)
att = [rnd(M, D).to(dt) for _ in range(S)]
Wp = [rnd(D, D, scale=0.1).to(dt) for _ in range(S)]
Wup, bup, Wd, bd = rnd(E, D, scale=0.1).to(dt), rnd(E), rnd(D, E, scale=0.1).to(dt), rnd(D)
x, mod = rnd(M, E), rnd(M, 2 * D).to(dt)
gamma, beta = rnd(D), rnd(D)
nd = torch.empty(M, D, device=dev, dtype=dt)
down = dict(W=Wd, bias=bd, gamma=gamma, beta=beta, mod=mod, Yact=nd) if has_down else None
graph = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    ops.exchange_tail(att, Wp, Wup, bup, float(S), x, None, down)
torch.cuda.synchronize()
with torch.cuda.graph(graph):
    for _ in range(50):
        ops.exchange_tail(att, Wp, Wup, bup, float(S), x, None, down)
for _ in range(3):
    graph.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    graph.replay()
e1.record()
torch.cuda.synchronize()
print(f"has_down={has_down}: {e0.elapsed_time(e1) / 500 * 1e3:.2f} us per launch")
