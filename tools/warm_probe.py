"""Where a rollout step's HOST time goes, and how the device comes out of idle (development aid):  python tools/warm_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench


def host_ms(fn, n=200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e3, (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    m = bench.build_model(dev, "bf16").eval()
    c = bench.CFG
    x, _, ib = bench.inputs(1, 2024, c["F"], c["E"], 0, dev)
    eng = m.engine(dev)
    with torch.no_grad():
        for _ in range(150):
            eng.forward(x, ib)
        p = eng.plan(1, 2024, "full")
        outs = [torch.empty_like(x) for _ in range(2)]
        p.bind(x, ib, outs[0])
        print("eng.forward            host %.4f ms, with drain %.4f ms" % host_ms(lambda: eng.forward(x, ib)))
        print("plan.run (sea_run_list) host %.4f ms, with drain %.4f ms" % host_ms(p.run))
        k = [0]

        def rebind():
            k[0] ^= 1
            p.bind(x, ib, outs[k[0]])

        print("plan.bind (alternating outputs) host %.4f ms" % host_ms(rebind)[0])
        print("torch.empty_like       host %.4f ms" % host_ms(lambda: torch.empty_like(x))[0])
        print("params.sync            host %.4f ms" % host_ms(eng.params.sync)[0])
        time.sleep(0.5)
        for rep in range(8):
            t0 = time.perf_counter()
            for _ in range(20):
                eng.forward(x, ib)
            torch.cuda.synchronize()
            print(f"after 0.5 s idle, burst {rep} of 20 steps: {(time.perf_counter() - t0) / 20 * 1e3:.4f} ms/step", flush=True)


if __name__ == "__main__":
    main()
