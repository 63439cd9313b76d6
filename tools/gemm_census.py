"""Development aid: every sea_gemm_grouped launch of the cfg3 training plan with its groups' shapes and epilogue features (which would a new GEMM form have to support?):
    python tools/gemm_census.py [B]"""
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
    from sea_amd import _native as N
    from sea_amd.utils.train_utils import initialize_optimizer

    opt = initialize_optimizer(model, {"learning_rate": 1e-4})
    for _ in range(3):
        eng.train_step(x, tgt, ib, opt, allreduce=False)
    torch.cuda.synchronize()
    plan = eng.train_plan(B, c["max_len"])
    L = N.lib()
    times = bench._time_list(list(plan.records) + list(plan.bwd), iters=5)
    for rec, ms in times:
        if rec.fn is not L.sea_gemm_grouped:
            continue
        arr, n = rec.args[0], rec.args[1]
        feats = []
        for i in range(n):
            g = arr[i]
            f = f"M{g.M} N{g.N} K{g.K}"
            if g.n_seg != 1: f += f" seg{g.n_seg}"
            if g.act: f += f" act{g.act}"
            if g.R: f += " R" + ("=C32" if g.R == g.C32 else "")
            if g.C32: f += " C32"
            if g.Cact: f += " Cact"
            if g.Z: f += " Z"
            if g.silu_c: f += " silu"
            if g.drop.thr: f += " drop"
            feats.append(f)
        uniq = sorted(set(feats))
        print(f"{rec.name:30s} {ms * 1e3:7.1f} us  {n:2d} groups: " + " | ".join(uniq))


if __name__ == "__main__":
    main()
