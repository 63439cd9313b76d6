"""Benchmark of the SEA temporal-rollout hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, one rank per GPU)

Workload (BASELINE.json configs[1], "cfg2"): TemporalModel(1 layer, embed_dim 256, 8 heads, max_len 2024, scale_ratio 8,
3 field groups, AdaLN, SEA exchange), synthetic cylinder_flow-shaped encoded fields x ~ N(0,1) [B, 2024, 3, 256] with one
scalar condition per step, random-init weights (reference init N(0,0.02), seed 42), bf16 compute.
One STEP = one full-context forward of the model over the 2024-step window = one iteration of the reference's rollout loop
(utils/train_utils.py:203-207) at prefix length 2024 in its own recompute mode; it advances every trajectory of the batch by
one time step.  value = trajectory-steps per second over all GPUs (weak scaling: B trajectories per GPU, no collective —
rollout shards by trajectory).  The K timed steps replay the step's captured HIP graph.

Extra objects in the JSON line:
  roofline     — the dominant kernel of the step (largest share of device time): algorithmic FLOPs of one launch divided
                 by its average launch duration, measured here with HIP events on the launch stream around every launch of
                 the plan (a separate instrumented pass of the same K steps), against the dense bf16 MFMA peak (2.5 PFLOP/s,
                 MI355X_MICROARCH.md).  traffic: HBM bytes per launch from rocprofv3 PMC passes when profiles/ holds them, else null.
  cpu_baseline — the CPU oracle (oracle/sea_oracle.py, a restatement of the reference pinned by its golden vectors) timed on
                 this host's cores on the same workload (bounded sample: 1 warm-up + 3 forwards), rank 0, N = 1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA, MI355X (MI355X_MICROARCH.md "Chip-level parameters")


def algorithmic_gflop(B, T, F, E, H, D, S, L, adaln=True):
    """GEMM = 2mnk, causal attention over T(T+1)/2 pairs, redundant recomputation removed."""
    M = B * T
    A = (lambda d: 2 * M * (2 * d) + 2 * M * (2 * d) ** 2) if adaln else (lambda d: 0)
    Q = lambda d: 4 * B * (T * (T + 1) // 2) * d  # noqa: E731
    per_layer = (F * 8 * M * E * E + F * Q(E) + 2 * F * A(E) + (2 * F - 1) * 2 * M * E * D + F * A(D)
                 + F * (F - 1) * (8 * M * D * D + Q(D) + 2 * M * D * E) + (16 * M + 16 * M * E) + F * 4 * M * E * S + F * 2 * M * E * E)
    return (L * per_layer + F * A(E)) / 1e9


def kernel_flops(name, B, T, F, E, H, D, S):
    """Algorithmic FLOPs of ONE launch of the named plan record (per launch, all groups of the launch)."""
    M = B * T
    tri = T * (T + 1) // 2
    per_field = 1
    if len(name) > 3 and name[-3:-1] == ".f" and name[-1].isdigit():  # per-field launch of the lane plan ("mlp.fc1.f0")
        name, per_field = name[:-3], F
    if name == "adaln.cond_gemm.first":
        return 2 * M * F * (2 * E) ** 2
    if name == "adaln.cond_gemm.rest":
        return 2 * M * (2 * F * (2 * E) ** 2 + F * (2 * D) ** 2)
    if per_field > 1:
        return kernel_flops(name, B, T, F, E, H, D, S) // per_field
    table = {
        "adaln.cond_gemm": 2 * M * ((2 * F + F) * (2 * E) ** 2 + F * (2 * D) ** 2),
        "adaln.cond_mlp": 2 * M * ((2 * F + F) * (2 * E) ** 2 + F * (2 * D) ** 2),   # cond_mlp.2 of the 12 modules (the silu rows are not counted)
        "self.qkv_rope": F * 2 * M * E * 3 * E,
        "self.attention": F * 4 * B * tri * E,
        "self.out_proj": F * 2 * M * E * E,
        "cross.down_old": F * 2 * M * E * D,
        "cross.down_norm_old": F * 2 * M * E * D,
        "add.down_norm": F * 2 * M * E * D,
        "add.up": F * 2 * M * E * D,
        "mlp.fc1": F * 2 * M * E * S,
        "mlp.fc1_ln_gelu": F * 2 * M * E * S,
        "mlp.fc2": F * 2 * M * E * S,
        "proj": F * 2 * M * E * E,
    }
    if name in table:
        return table[name]
    # fused plan (sea_rowchain launches): the Linear layers each chain contains
    kv_old = F * (F - 1) // 2  # pairs (iq, i), iq < i: k/v projections of the old x_i; the other half reads the new x_i
    fused = {
        "self.adaln0_qkv": F * 2 * M * E * 3 * E,
        "self.proj_down_qkv": F * (2 * M * E * E + 2 * M * E * D) + F * (F - 1) * 2 * M * D * D + kv_old * 2 * M * D * 2 * D,
        "proj_final_norm": F * 2 * M * E * E,
        "proj_adaln0_qkv": F * (2 * M * E * E + 2 * M * E * 3 * E),
        "mlp.fused": F * 4 * M * E * S,
    }
    if name in fused:
        return fused[name]
    if name.startswith("cross") and name.endswith("proj_up_down_kv"):
        i = int(name[5:name.index(".")])
        return (F - 1) * (2 * M * D * D) + 2 * M * D * E + (2 * M * E * D + (F - 1 - i) * 2 * M * D * 2 * D if i < F - 1 else 0)
    if name.startswith("cross") and name.endswith("qkv_rope"):
        return (F - 1) * 2 * M * D * 3 * D
    if name.startswith("cross") and name.endswith("attention"):
        return (F - 1) * 4 * B * tri * D
    if name.startswith("cross") and name.endswith("proj_gelu"):
        return (F - 1) * 2 * M * D * D
    if name.startswith("cross") and name.endswith("up_sum"):
        return (F - 1) * 2 * M * D * E
    if name.startswith("cross") and (name.endswith("down_new") or name.endswith("down_norm_new")):
        return 2 * M * E * D
    if name.startswith("cross") and name.endswith(".tail"):
        i = int(name[5:name.index(".")])
        return (F - 1) * (2 * M * D * D + 2 * M * D * E) + (2 * M * E * D if i < F - 1 else 0)
    if name.startswith("cross") and name.endswith("up_sum_ib_adaln2"):
        return (F - 1) * 2 * M * D * E
    return 0


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def host_cores() -> int:
    """CPU threads this process may really use: scheduler affinity, capped by the cgroup CPU quota when one is set."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return min(n, 64)


def cpu_baseline(cfg_args, B, T):
    """Oracle forward on the host cores (bounded: 1 warm-up + 3 timed forwards at the bench shape)."""
    from oracle import sea_oracle as O

    L, E, H, max_len, sr, src, F = cfg_args
    cfg = O.OracleConfig(L, E, H, max_len, sr, src, F, 2, True, "adaln")
    from oracle.recipe import param_schema

    g = torch.Generator().manual_seed(42)
    p = {}
    for k, (shp, kind) in param_schema(cfg).items():
        if kind == "lin_w":
            p[k] = torch.randn(shp, generator=g) * 0.02
        elif kind == "norm_w":
            p[k] = torch.ones(shp)
        else:
            p[k] = torch.zeros(shp)
    x = torch.randn(B, T, F, E, generator=torch.Generator().manual_seed(1234))
    ib = torch.rand(B, T, 1, generator=torch.Generator().manual_seed(1235))
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu_baseline: oracle forward on {cores} host threads ...")
    with torch.no_grad():
        O.model_forward(x, ib, p, cfg)
        n = 3
        t0 = time.perf_counter()
        for _ in range(n):
            O.model_forward(x, ib, p, cfg)
        dt = (time.perf_counter() - t0) / n
    return {"value": B / dt, "unit": "trajectory-steps/s", "cores": cores, "kind": "port",
            "sample": f"oracle fp32 forward, B={B}, T={T}, 1 warm-up + {n} timed passes, {dt:.3f} s/pass, torch CPU threads={cores}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1, help="trajectories per GPU")
    ap.add_argument("--seq", type=int, default=2024)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-graph", action="store_true", help="replay launch by launch instead of the captured HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default="rollout", choices=["rollout", "train", "kv", "decode", "encode"],
                    help="rollout: full-context forward step (cfg2, the default metric); train: fwd+bwd+AdamW step (cfg3/cfg4, --batch 8); "
                         "kv: KV-cache rollout of --seq steps; decode: spatial decoder over a --seq-step rollout (SURVEY.md §8f rank 1); "
                         "encode: spatial encoder over --seq snapshots (SURVEY.md §8f rank 2)")
    args = ap.parse_args()
    if args.mode == "decode":
        return main_decode(args)
    if args.mode == "encode":
        return main_encode(args)
    if args.mode != "rollout":
        return main_other(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group("nccl", device_id=dev)
        dist = dist_mod

    from sea_amd.models.temporal import TemporalModel

    L, E, H, max_len, sr, src, F = 1, 256, 8, 2024, 8, 0, 3
    B, T = args.batch, args.seq
    D, S = E // 2, E * sr
    torch.manual_seed(42)
    model = TemporalModel(L, E, H, max_len, sr, src, F, 2, 0.0, "sea", "learnable", "mlp", "add", 1, 1, True, "adaln")
    model.set_compute_dtype(args.dtype)
    model = model.to(dev).eval()
    x = torch.randn(B, T, F, E, generator=torch.Generator().manual_seed(1234 + rank)).to(dev)
    ib = torch.rand(B, T, 1, generator=torch.Generator().manual_seed(1235 + rank)).to(dev)
    eng = model.engine(dev)

    def step():
        return eng.forward(x, ib) if args.no_graph else eng.forward_graphed(x, ib)

    log(f"model built on {dev}; warm-up ({args.warmup} steps, {'plain' if args.no_graph else 'hip-graph'} replay)")
    with torch.no_grad():
        for _ in range(max(args.warmup, 1)):
            out = step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert torch.isfinite(out).all()

    log(f"timed {args.steps} steps in {elapsed:.4f} s; per-launch event timing ...")
    # ---- per-launch device times (HIP events on the launch stream), same number of steps
    with torch.no_grad():
        plan = eng.plan(B, T, "full")
        plan.bind(x, ib, torch.empty_like(x))
        plan.run()
        torch.cuda.synchronize()
        times = plan.time_records(iters=max(args.steps, 5))
    total_ms = sum(t for _, t in times)
    name, dom_ms_events = max(times, key=lambda nt: nt[1])
    # the dominant launch alone, 20 back-to-back launches captured in one graph between two HIP events on the launch stream: the per-launch
    # figure then carries no event / launch-gap overhead and is the one to compare with rocprofv3's average duration of that kernel
    rec = next(r for r in plan.records if r.fn is not None and r.name == name)
    st = torch.cuda.Stream()
    gk = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(gk, stream=st):
            for _ in range(20):
                rc = rec.fn(*rec.args, st.cuda_stream)
                assert rc == 0, name
    gk.replay()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gk.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    dom_ms = best
    dom_flops = kernel_flops(name, B, T, F, E, H, D, S)
    achieved = dom_flops / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    esz = 2 if args.dtype == "bf16" else 4
    dom_bytes = None
    if name == "adaln.cond_gemm":   # hidden matrices in, modulations out, weights once: 12 modules at F = 3 (9 of width 2E, 3 of width 2D)
        dom_bytes = sum(n_mod * (2 * B * T * (2 * d) * esz + (2 * d) * (2 * d) * esz) for n_mod, d in ((3 * F, E), (F, D)))
    elif name == "adaln.cond_mlp":   # the condition once, weights once, modulations out: no hidden matrix
        dom_bytes = sum(n_mod * (B * T * (2 * d) * esz + (2 * d) * (2 * d) * esz + 3 * (2 * d) * 4) for n_mod, d in ((3 * F, E), (F, D))) + B * T * 4
    elif name == "mlp.fc1_ln_gelu":  # per field: normalised rows in, W1 and the three vectors once, activated hidden rows out (the pre-activation never leaves the CU)
        dom_bytes = F * (B * T * E * esz + S * E * esz + 3 * S * 4 + B * T * S * esz)
    elif name == "mlp.fc2":          # activated hidden rows and the residual in, W2 once, rows out
        dom_bytes = F * (B * T * S * esz + S * E * esz + B * T * E * 4 + B * T * E * esz)

    # HBM bytes per launch of the dominant kernel, from the committed PMC profile of this same workload (separate --pmc passes, see
    # profiles/README.md); null when the workload or the plan differs from the profiled one
    traffic, traffic_src = None, None
    prof = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_forward_cfg2_pmc_traffic.json")
    if os.path.exists(prof) and (B, T, F, E, args.dtype) == (1, 2024, 3, 256, "bf16"):
        rec = json.load(open(prof))["launches"].get(name)
        if rec is not None:
            traffic, traffic_src = rec["hbm_bytes"], "profiles/r01_forward_cfg2_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, bytes per launch)"

    # Which roof bounds the dominant launch: its algorithmic intensity (FLOP per algorithmic byte: operands + weights + outputs once) against the
    # machine balance 2500 TFLOP/s / 8 TB/s = 312 FLOP/B.  Below it the HBM roof is the bound and `achieved` is algorithmic bytes over the launch
    # duration; the MFMA-side figures stay in the object either way.
    hbm_GBps = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_bytes and dom_ms > 0 else None
    intensity = dom_flops / dom_bytes if dom_bytes else None
    hbm_bound = intensity is not None and intensity < PEAK_BF16_TFLOPS * 1e12 / 8000e9
    roofline = {"kernel": name, "bound": "hbm" if hbm_bound else "mfma",
                "achieved": hbm_GBps if hbm_bound else achieved, "peak": 8000.0 if hbm_bound else PEAK_BF16_TFLOPS, "unit": "GB/s" if hbm_bound else "TFLOP/s",
                "frac": (hbm_GBps / 8000.0) if hbm_bound else achieved / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                "flop_per_algorithmic_byte": intensity, "algorithmic_bytes": dom_bytes,
                "hbm_GBps": hbm_GBps, "hbm_frac": hbm_GBps / 8000.0 if hbm_GBps else None,
                "mfma_TFLOPs": achieved, "mfma_frac": achieved / PEAK_BF16_TFLOPS,
                "launch_ms": dom_ms, "launch_ms_single_event_pair": dom_ms_events, "launch_gflop": dom_flops / 1e9,
                "device_ms_all_launches": total_ms,
                "timing": "launch_ms: HIP events around 20 back-to-back launches of the dominant record (one captured graph) / 20; "
                          "launch_breakdown_ms: one event pair per launch of the plan (includes ~3-6 us of event / launch gap each)"}
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        gflop = algorithmic_gflop(B, T, F, E, H, D, S, L)
        line = {
            "metric": "rollout steps/sec (full-context forward, recompute mode) on cylinder_flow-shaped fields",
            "value": value, "unit": "trajectory-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "cfg2: cylinder_flow temporal model E=256 H=8 F=3 L=1 adaln, forward-only rollout step at T=2024",
                       "trajectories_per_gpu": B, "seq_len": T, "fields": F, "embed_dim": E, "replay": "plain" if args.no_graph else "hip-graph",
                       "parallelism": f"replicas x{world} (rollout shards by trajectory, no collective)"},
            "model_algorithmic_gflop_per_step": gflop,
            "model_mfma_frac": gflop / (ms_per_step * 1e-3) / 1e3 / PEAK_BF16_TFLOPS,
            "roofline": roofline,
            "launch_breakdown_ms": {n: round(t, 4) for n, t in times},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline((L, E, H, max_len, sr, src, F), B, T)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main_other(args):
    """--mode train: BASELINE.json configs[2]/[3] (fwd+bwd+fused AdamW, B trajectories per GPU, one RCCL all-reduce of the flat
    gradient per step when N > 1).  --mode kv: exact KV-cache rollout (configs[4]-style long rollout).  Same JSON contract."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group("nccl", device_id=dev)
        dist = dist_mod
    from sea_amd.models.temporal import TemporalModel
    from sea_amd.utils.train_utils import initialize_optimizer, rollout

    L, E, H, max_len, sr, src, F = 1, 256, 8, 2024, 8, 0, 3
    B, T = args.batch, args.seq
    D, S = E // 2, E * sr
    torch.manual_seed(42)
    model = TemporalModel(L, E, H, max_len, sr, src, F, 2, 0.0, "sea", "learnable", "mlp", "add", 1, 1, True, "adaln")
    model.set_compute_dtype(args.dtype)
    model = model.to(dev)
    x = torch.randn(B, T, F, E, generator=torch.Generator().manual_seed(1234 + rank)).to(dev)
    tgt = torch.randn(B, T, F, E, generator=torch.Generator().manual_seed(4321 + rank)).to(dev)
    ib = torch.rand(B, T, 1, generator=torch.Generator().manual_seed(1235 + rank)).to(dev)
    eng = model.engine(dev)
    if args.mode == "train":
        model.train()
        opt = initialize_optimizer(model, {"learning_rate": 1e-4})
        if dist is not None:
            from sea_amd.parallel import broadcast_parameters

            broadcast_parameters(eng.params.flat32)
            eng.params.sync(force=True)

        def step():
            return eng.train_step(x, tgt, ib, opt)
    else:
        model.eval()

        def step():
            return rollout(model, x[:, :1].contiguous(), ib, T, mode="kv")
    for _ in range(max(args.warmup, 1)):
        r = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert torch.isfinite(r).all()
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        if args.mode == "train":
            gflop = 3 * algorithmic_gflop(B, T, F, E, H, D, S, L)
            line = {"metric": "train steps/sec (fwd+bwd+AdamW) on cylinder_flow-shaped fields", "value": world * args.steps / elapsed * 1.0,
                    "unit": "rank-steps/s (each step = B trajectories x T positions, teacher-forced)", "trajectory_steps_per_s": world * B * args.steps / elapsed,
                    "config": {"workload": f"cfg3: cylinder_flow temporal model E=256 H=8 F=3 L=1 adaln, fwd+bwd+AdamW, B={B} per GPU, T={T}",
                               "global_batch": world * B, "seq_len": T, "parallelism": f"dp{world}: flat-gradient all-reduce (RCCL), 1 per step"},
                    "model_algorithmic_gflop_per_step": gflop, "model_mfma_frac": gflop / (ms * 1e-3) / 1e3 / PEAK_BF16_TFLOPS}
        else:
            line = {"metric": "KV-cache rollout steps/sec on cylinder_flow-shaped fields", "value": world * B * T * args.steps / elapsed,
                    "unit": "trajectory-steps/s",
                    "config": {"workload": f"KV-cache rollout of {T} steps, B={B} per GPU, E=256 H=8 F=3 L=1 adaln", "parallelism": f"replicas x{world}"}}
            ms = ms / T
        line.update({"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
                     "vs_baseline": None, "dtype": args.dtype, "data": "synthetic"})
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main_decode(args):
    """--mode decode: the decode leg of full_autoregressive_evaluation (reference utils/train_utils.py:214-222) at the shipped cylinder
    dims: 64 patches x spatial embed 16 (= temporal embed 1024), groups [[0,1],[2]], hidden 480; n_inp (mesh points per padded patch) is
    data-dependent in the reference, 512 here.  One step = decoding a whole --seq-step rollout of --batch trajectories."""
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (the product path has no CPU fallback)")
    from sea_amd.models.encoder_decoder import Decode
    from sea_amd.utils.data_processors import DataPartitioner2D, MeshUnpatcher, MinMaxScaler
    from sea_amd.utils.train_utils import decode_rollout

    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    groups, hidden, D, P = [[0, 1], [2]], 480, 16, 64
    tr, T = args.batch, args.seq
    torch.manual_seed(42)
    # synthetic 2-D mesh: 30000 points, denser towards x = 0.3 (a wake-like refinement), 8 x 8 cells as in the shipped config (m = n = 9)
    gen = torch.Generator().manual_seed(7)
    n_points = 30000
    px = torch.rand(n_points, generator=gen); px[: n_points // 3] = 0.25 + 0.1 * torch.rand(n_points // 3, generator=gen)
    py = torch.rand(n_points, generator=gen)
    part = DataPartitioner2D(px, py, m=9, n=9, device=dev)
    C_pad = part.padded_index_map.shape[1]
    n_inp = (C_pad + 3) // 4 * 4
    scalers = []
    for lo, hi in ((-1.0, 3.0), (0.5, 2.0)):
        sc = MinMaxScaler()
        sc.min_val, sc.max_val = torch.tensor(lo), torch.tensor(hi)
        scalers.append(sc)
    unpatcher = MeshUnpatcher(part, groups, scalers)
    dec = Decode(groups, n_inp, hidden, D).set_compute_dtype(args.dtype).to(dev).eval()
    roll = torch.randn(tr, T, len(groups), P * D, generator=torch.Generator().manual_seed(1234)).to(dev)
    with torch.no_grad():
        def step():
            dec_out = decode_rollout(dec, roll, P)                                              # [tr*T, P, 3, n_inp]
            return dec_out, unpatcher.inverse_scale_and_unpatch(dec_out[..., :C_pad], layout="BPFC")   # [tr*T, n_points, 3]

        for _ in range(max(args.warmup, 1)):
            dec_out, out = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            dec_out, out = step()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            unpatcher.inverse_scale_and_unpatch(dec_out[..., :C_pad], layout="BPFC")
        e1.record()
        torch.cuda.synchronize()
        u_ms = e0.elapsed_time(e1) / 5
        u_bytes = tr * T * (P * C_pad * 3 * 4 + n_points * 3 * 4) + P * C_pad * 4   # read the decoded cells + write every mesh point once
        # the dominant launch (layer2: [M, 480] x [480, 1536] + bias -> fp32) alone, HIP events on the launch stream
        M = tr * T * P
        dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
        W1, W2 = dec._weights(dt)
        hid = [torch.randn(M, hidden, device=dev).to(dt) for _ in groups]
        o2 = torch.empty(M, 3 * n_inp, device=dev)
        from sea_amd import ops
        gl = [dict(A=hid[0], W=W2[0], bias=dec.decoders[0].layer2.bias.detach(), C32=o2[:, :2 * n_inp]),
              dict(A=hid[1], W=W2[1], bias=dec.decoders[1].layer2.bias.detach(), C32=o2[:, 2 * n_inp:])]
        ops.gemm_grouped(gl, dt)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.gemm_grouped(gl, dt)
        e1.record()
        torch.cuda.synchronize()
        k_ms = e0.elapsed_time(e1) / 5
    assert torch.isfinite(out).all()
    esz = 2 if args.dtype == "bf16" else 4
    alg_bytes = M * hidden * esz + 3 * n_inp * hidden * esz + M * 3 * n_inp * 4   # read hidden + weights once, write the fp32 fields once
    flops = 2 * M * hidden * 3 * n_inp
    ms = elapsed / args.steps * 1e3
    line = {"metric": "decoded snapshots/sec (spatial decoder over a rollout)", "value": tr * T * args.steps / elapsed, "unit": "snapshots/s",
            "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"decode leg of the rollout evaluation: {tr} x {T} snapshots, 64 patches, spatial embed 16, hidden 480, groups [[0,1],[2]], "
                                   f"mesh of {n_points} points in 8x8 cells (padded cell size {C_pad}, n_inp {n_inp}); decoder + un-patchify + inverse scaling"},
            "unpatchify": {"ms": u_ms, "algorithmic_GB": u_bytes / 1e9, "achieved_GBps": u_bytes / (u_ms * 1e-3) / 1e9, "frac_of_hbm_peak": u_bytes / (u_ms * 1e-3) / 1e9 / 8000.0},
            "roofline": {"kernel": "decode.layer2 (gemm_grouped)", "bound": "hbm", "achieved": alg_bytes / (k_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                         "frac": alg_bytes / (k_ms * 1e-3) / 1e9 / 8000.0, "traffic": None, "launch_ms": k_ms, "launch_gflop": flops / 1e9,
                         "mfma_tflops": flops / (k_ms * 1e-3) / 1e12}}
    print(json.dumps(line))


def main_encode(args):
    """--mode encode: ProcessData.process_data (reference utils/data_processors.py:335-352) at the shipped cylinder spatial dims: 64 patches,
    groups [[0,1],[2]], spatial embed 16 (width 32, 8 heads), hidden 480, 12 EncoderBlocks; the padded cell size is data-dependent in the
    reference, 512 mesh points here.  One step = encoding --batch x --seq snapshots that are resident in HBM as [snapshots, P, F, C] fp32."""
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (the product path has no CPU fallback)")
    from sea_amd import ops
    from sea_amd.models.encoder_decoder import PointwiseEncode

    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    groups, hidden, D, P, H, layers, n_inp = [[0, 1], [2]], 480, 16, 64, 8, 12, 512
    n_snap = args.batch * args.seq
    torch.manual_seed(42)
    enc = PointwiseEncode(groups, n_inp, hidden, layers, D, H, 2024, 0, dropout=0.0).set_compute_dtype(args.dtype).to(dev).eval()
    x = torch.randn(n_snap, P, 3, n_inp, generator=torch.Generator().manual_seed(1234)).to(dev)
    with torch.no_grad():
        for _ in range(max(args.warmup, 1)):
            z = enc(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            z = enc(x)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        # the dominant launch: the down-scale MLPs' first Linear + GELU ([M, |g| C] x [480, |g| C], both groups), HIP events on the launch stream
        dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
        pk = enc._packed(dt, dev)
        Bc = min(enc.CHUNK, n_snap)
        ws = enc._workspace(pk, Bc, P, dt, dev)
        gl = [dict(A=A, W=g["W1"], Cact=hid, act=1) for g, A, hid in zip(pk["enc"], ws["A"], ws["hid"])]
        ops.gemm_grouped(gl, dt)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.gemm_grouped(gl, dt)
        e1.record()
        torch.cuda.synchronize()
        k_ms = e0.elapsed_time(e1) / 5
    assert torch.isfinite(z).all()
    esz = 2 if args.dtype == "bf16" else 4
    M = Bc * P
    Ksum = sum(g["Kp"] for g in pk["enc"])
    alg_bytes = M * Ksum * esz + hidden * Ksum * esz + len(groups) * M * hidden * esz    # read the cell values once + weights, write the hidden rows once
    flops = 2 * M * hidden * Ksum
    W = len(groups) * D
    # algorithmic FLOPs of one snapshot: down-scale MLPs + 12 blocks (q/k/v, scores + values over P x P, projection, MLP x4)
    per_snap = 2 * P * (hidden * 3 * n_inp + len(groups) * hidden * D) + layers * (2 * P * W * 3 * W + 4 * P * P * W + 2 * P * W * W + 2 * 2 * P * W * 4 * W)
    ms = elapsed / args.steps * 1e3
    line = {"metric": "encoded snapshots/sec (spatial encoder inference)", "value": n_snap * args.steps / elapsed, "unit": "snapshots/s",
            "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"encode leg of the data preparation: {n_snap} snapshots, 64 patches x {n_inp} padded mesh points x 3 fields, spatial embed 16 "
                                   f"(width {W}, {H} heads), hidden {hidden}, {layers} encoder blocks, chunks of {enc.CHUNK} snapshots"},
            "model_algorithmic_gflop_per_step": per_snap * n_snap / 1e9, "input_GB": x.numel() * 4 / 1e9, "input_GBps": x.numel() * 4 / (ms * 1e-3) / 1e9,
            "roofline": {"kernel": "encode.layer1 (gemm_grouped, GELU epilogue; one chunk)", "bound": "hbm", "achieved": alg_bytes / (k_ms * 1e-3) / 1e9, "peak": 8000.0,
                         "unit": "GB/s", "frac": alg_bytes / (k_ms * 1e-3) / 1e9 / 8000.0, "traffic": None, "launch_ms": k_ms, "launch_gflop": flops / 1e9,
                         "mfma_tflops": flops / (k_ms * 1e-3) / 1e12}}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
