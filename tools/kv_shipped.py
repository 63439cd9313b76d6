"""KV-cache and recompute rollouts at the SHIPPED widths (configs/cylinder_flow.py: embed_dim 1024, adaln; configs/multiphase_flow.py: embed_dim 2048, ln; both
2 field groups, 8 heads, 1 layer, block 2024): steps/s of a 100-step rollout and the per-launch durations of one KV step — development aid.
    python tools/kv_shipped.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sea_amd.models.temporal import TemporalModel
from sea_amd.utils.train_utils import rollout

dev = torch.device("cuda:0")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    for name, E, ln in (("cylinder_flow dims", 1024, "adaln"), ("multiphase_flow dims", 2048, "ln")):
        torch.manual_seed(42)
        m = TemporalModel(1, E, 8, 2024, 8, 0, 2, 2, 0.0, "sea", "learnable", "mlp", "add", 1, 1, True, ln)
        m.set_compute_dtype("bf16")
        m = m.to(dev).eval()
        x0 = torch.randn(1, 1, 2, E, device=dev)
        ib = torch.rand(1, n, 1, device=dev)
        for mode in ("kv", "recompute"):
            for _ in range(2):
                r = rollout(m, x0, ib, n, mode=mode)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                r = rollout(m, x0, ib, n, mode=mode)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            assert torch.isfinite(r).all()
            print(f"{name} (E={E}, {ln}): {mode:9s} rollout of {n} steps: {dt * 1e3:8.2f} ms = {dt / n * 1e3:7.4f} ms/step = {n / dt:8.0f} steps/s", flush=True)
        # per-launch durations of one KV step (generic step plan with the condition work hoisted, as rollout_kv builds it)
        import bench
        from sea_amd import kv_engine
        eng = m.engine()
        cond = ib[:, :n, 0].t().contiguous()
        cp = kv_engine.cond_plan_for(eng, n * 1)
        for t in cp.ibufs:
            t.zero_()
        cp.bind_ptrs(0, cond.data_ptr(), 0)
        cp.run()
        p = eng.plan(1, 1, "step", cond=cp)
        xs, ys = torch.randn(1, 2, E, device=dev), torch.empty(1, 2, E, device=dev)
        p.set_position(50)
        p.bind_ptrs(xs.data_ptr(), cond.data_ptr() + 50 * 4, ys.data_ptr())
        p.set_hoisted_step(50)
        times = bench._time_list(list(p.records), iters=5)
        tot = sum(t for _, t in times)
        nbytes = sum(prm.numel() for prm in m.parameters()) * 2
        print(f"   step plan: {len(times)} launches, sum {tot * 1e3:.1f} us; bf16 parameters {nbytes / 1e6:.0f} MB -> weight-read floor {nbytes / 5.0e12 * 1e6:.1f} us at 5 TB/s")
        from sea_amd import _native as N
        L = N.lib()
        for rec, ms in times:
            shapes = ""
            if rec.fn in (L.sea_gemm_grouped, L.sea_qkv_rope_grouped, L.sea_gemm_rownorm, L.sea_gemm_fewrows, L.sea_qkv_rope_fewrows):
                ng = rec.args[2] if rec.fn in (L.sea_gemm_fewrows, L.sea_qkv_rope_fewrows) else rec.args[1]
                shapes = " ".join(f"[M{g.M} N{g.N} K{g.K}]" for g in rec.args[0][:ng])
            fl, by = bench.record_work(rec, 2)
            print(f"      {rec.name:28s} {ms * 1e3:8.1f} us  {by / 1e6:7.2f} MB  {shapes}")
        del m


if __name__ == "__main__":
    main()
