"""Every launch of the cfg3 training plan (forward + backward) with its algorithmic work and measured duration:
    python tools/train_launch_table.py [B]      -> name, us, GFLOP, TFLOP/s, MB, GB/s   (HIP events around single launches, bench.py's _time_list)
Measurement aid for the round notes; run on the MI355X box from the repository root."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda:0")
    c = bench.CFG
    model = bench.build_model(dev, "bf16").train()
    x, tgt, ib = bench.inputs(B, c["max_len"], c["F"], c["E"], 0, dev)
    eng = model.engine(dev)
    from sea_amd.utils.train_utils import initialize_optimizer

    opt = initialize_optimizer(model, {"learning_rate": 1e-4})
    for _ in range(3):
        eng.train_step(x, tgt, ib, opt, allreduce=False)
    torch.cuda.synchronize()
    plan = eng.train_plan(B, c["max_len"])
    times = bench._time_list(list(plan.records) + list(plan.bwd), iters=5)
    tot = 0.0
    print(f"{'launch':34s} {'us':>8s} {'GFLOP':>8s} {'TF/s':>7s} {'MB':>8s} {'GB/s':>7s}")
    for rec, ms in times:
        fl, by = bench.record_work(rec, 2)
        tot += ms
        print(f"{rec.name:34s} {ms * 1e3:8.1f} {fl / 1e9:8.2f} {fl / (ms * 1e-3) / 1e12 if fl else 0:7.0f} {by / 1e6:8.1f} {by / (ms * 1e-3) / 1e9 if by else 0:7.0f}")
    print(f"sum of single-launch durations {tot:.3f} ms over {len(times)} launches")


if __name__ == "__main__":
    main()
