"""cfg2 forward with the condition lane (SEA_PLAN=lanes=cond) replayed on two plain streams (events, no graph), against the one-stream replay.
The graph form of the lanes lost; this measures the stream form, with and without stream priorities.  Development aid.

    python tools/lane_probe.py            # prints ms per step for: one stream | lanes, equal priority | lanes, main stream high priority
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench


def timed(fn, steps=300, warmup=30):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    dev = torch.device("cuda:0")
    c = bench.CFG
    F, E, T = c["F"], c["E"], 2024
    x, _, ib = bench.inputs(1, T, F, E, 0, dev)
    res = {}
    ref = None
    for mode in ("none", "cond"):
        os.environ["SEA_PLAN"] = "lanes=" + mode
        model = bench.build_model(dev, "bf16").eval()
        eng = model.engine(dev)
        from sea_amd.engine import Plan

        with torch.no_grad():
            out = torch.empty_like(x)
            eng.params.sync()
            p = Plan(eng, 1, T, "full")
            p.bind(x, ib, out)
            p.run()
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            names = [(r.name, r.lane) for r in p.records]
            print(mode, len([r for r in p.records if r.fn is not None]), "launches:", names, flush=True)
            if mode == "none":
                res["one stream (sea_run_list)"] = timed(lambda: p.run())
                res["one stream (python loop)"] = timed(lambda: _loop(p))
            else:
                res["lanes: one stream, record order"] = timed(lambda: _loop(p))
                res["lanes: two streams, equal priority"] = timed(lambda: p.run(concurrent=True))
                torch.cuda.synchronize()
                assert torch.equal(out, ref), "lane replay differs"
                main = torch.cuda.Stream(device=dev, priority=-1)
                p._lane_streams.clear()
                p._lane_streams[1] = torch.cuda.Stream(device=dev, priority=0)
                with torch.cuda.stream(main):
                    res["lanes: two streams, main high priority"] = timed(lambda: p.run(concurrent=True))
                torch.cuda.synchronize()
                assert torch.equal(out, ref), "lane replay differs"
    for k, v in res.items():
        print(f"{k:44s} {v:.4f} ms/step", flush=True)


def _loop(p):
    s = torch.cuda.current_stream().cuda_stream
    for r in p.records:
        if r.fn is not None:
            r.fn(*r.args, s)


if __name__ == "__main__":
    main()
