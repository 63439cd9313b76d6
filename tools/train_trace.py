"""Kernel sequence of ONE cfg3 train step from a rocprofv3 kernel trace.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tt -o run -- python3 tools/train_trace.py run        (8 fused train steps, nothing else)
    python tools/train_trace.py show gpurun_out/tt/.../run_kernel_trace.csv                                           (one step: names, durations, gaps)
The step is delimited by its AdamW launch; everything between two of them is listed, so launches that are not plan records (fills, copies) show up."""
import csv
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch

    import bench
    from sea_amd.utils.train_utils import initialize_optimizer

    dev = torch.device("cuda:0")
    c = bench.CFG
    model = bench.build_model(dev, "bf16").train()
    x, tgt, ib = bench.inputs(8, c["max_len"], c["F"], c["E"], 0, dev)
    eng = model.engine(dev)
    opt = initialize_optimizer(model, {"learning_rate": 1e-4})
    for _ in range(8):
        eng.train_step(x, tgt, ib, opt, allreduce=False)
    torch.cuda.synchronize()


def show(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "adamw_flat" in r["Kernel_Name"]]
    a, b = idx[-2], idx[-1]
    t_prev = int(rows[a]["End_Timestamp"])
    tot = gaps = 0.0
    for r in rows[a + 1:b + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        n = re.sub(r"^void ", "", r["Kernel_Name"])
        m = re.match(r"_Z\d+([a-zA-Z_0-9]+?)I", n)
        n = (m.group(1) if m else n.split("(")[0])[:60]
        print(f"{n:60s} {(e - s) / 1e3:8.1f} us   gap {(s - t_prev) / 1e3:7.1f}")
        tot += (e - s) / 1e3
        gaps += (s - t_prev) / 1e3
        t_prev = e
    print(f"{b - a} launches, kernels {tot:.1f} us, gaps {gaps:.1f} us, step {tot + gaps:.1f} us")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        show(sys.argv[2])
