"""Kernel sequence of ONE KV-cache step at a shipped width from a rocprofv3 kernel trace (names, durations, gaps) — development aid.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kvt -o run -- python3 tools/kv_shipped_trace.py run 2048 ln
    python tools/kv_shipped_trace.py show gpurun_out/kvt/run_kernel_trace.csv <launches per step>"""
import csv
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(E, ln):
    import torch

    from sea_amd.models.temporal import TemporalModel
    from sea_amd.utils.train_utils import rollout

    dev = torch.device("cuda:0")
    torch.manual_seed(42)
    m = TemporalModel(1, E, 8, 2024, 8, 0, 2, 2, 0.0, "sea", "learnable", "mlp", "add", 1, 1, True, ln)
    m.set_compute_dtype("bf16")
    m = m.to(dev).eval()
    x0 = torch.randn(1, 1, 2, E, device=dev)
    ib = torch.rand(1, 100, 1, device=dev)
    for _ in range(3):
        rollout(m, x0, ib, 100, mode="kv")
    torch.cuda.synchronize()


def show(path, per_step):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-per_step * 40:-per_step * 39] if per_step else rows[-60:]   # one step in the middle of the last rollout
    prev = None
    tot = 0.0
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = re.sub(r"\(.*", "", r["Kernel_Name"])[:70]
        gap = (s - prev) / 1e3 if prev else 0.0
        print(f"{name:70s} {(e - s) / 1e3:7.1f} us   gap {gap:6.1f}")
        tot += (e - s) / 1e3
        prev = e
    print(f"{len(rows)} launches, kernels {tot:.1f} us")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]), sys.argv[3])
    else:
        show(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 0)
