"""Benchmark of the SEA temporal-rollout hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: the parent starts N rank processes itself, one per GPU, before any GPU call;
                                                            under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it is a rank)

Model (BASELINE.json configs[1..3], "cfg2" / "cfg3" / "cfg4"): TemporalModel(1 layer, embed_dim 256, 8 heads, max_len 2024, scale_ratio 8, 3 field
groups, AdaLN, SEA exchange), synthetic cylinder_flow-shaped encoded fields x ~ N(0,1) [B, 2024, 3, 256] with one scalar condition per step,
random-init weights (reference init N(0, 0.02), seed 42), bf16 compute.

ONE JSON line.  Its legs (all measured inside the driver-timed run):
  rollout  cfg2: one STEP = one full-context forward over the 2024-step window = one iteration of the reference's rollout loop
           (utils/train_utils.py:203-207) at prefix length 2024 in its own recompute mode; B = 1 trajectory per GPU; the K timed steps replay
           the step's captured HIP graph.  N = 1: this is the headline (`value` = trajectory-steps/s).
  train    cfg3 / cfg4: fwd + MSE + bwd + fused AdamW (train/train_temporal.py:254-258), B = 8 trajectories per GPU, T = 2024; N > 1: replicated
           model, the batch split by rank, ONE all-reduce of the flat fp32 gradient per step (RCCL), the 1/world mean folded into AdamW.
           N > 1: this is the headline (`value` = trajectory-steps/s over all GPUs), with the all-reduce timed on its own and a single-GPU
           step of the same processes beside it.
  kv       exact KV-cache rollouts: 2024 steps of the cfg2 model, a cfg5-shaped one (multiphase structure: F = 2, LayerNorm, 100 steps), and 100 steps at
           the two SHIPPED widths (configs/cylinder_flow.py: embed_dim 1024; configs/multiphase_flow.py: embed_dim 2048 — BASELINE.json configs[0] / [4]).
  cpu_baseline (N = 1, rank 0): the CPU oracle (oracle/sea_oracle.py, pinned by the reference's golden vectors) on this host's cores: forward,
           forward + backward + AdamW (1 warm-up + 1 pass), 8- and 100-step recompute rollouts.
  roofline the dominant launch of the headline leg: algorithmic FLOPs (and bytes) of ONE launch, derived from the launch's own argument
           structs, over its duration measured with HIP events on the launch stream (20 back-to-back launches in one captured graph); the
           MFMA roof is the headline (`frac`), the HBM-side figures are beside it (`hbm_frac`).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA, MI355X (MI355X_MICROARCH.md "Chip-level parameters")
PEAK_HBM_GBPS = 8000.0
CFG = dict(L=1, E=256, H=8, max_len=2024, sr=8, src=0, F=3)


def algorithmic_gflop(B, T, F, E, H, D, S, L, adaln=True):
    """GEMM = 2mnk, causal attention over T(T+1)/2 pairs, redundant recomputation removed."""
    M = B * T
    A = (lambda d: 2 * M * (2 * d) + 2 * M * (2 * d) ** 2) if adaln else (lambda d: 0)
    Q = lambda d: 4 * B * (T * (T + 1) // 2) * d  # noqa: E731
    per_layer = (F * 8 * M * E * E + F * Q(E) + 2 * F * A(E) + (2 * F - 1) * 2 * M * E * D + F * A(D)
                 + F * (F - 1) * (8 * M * D * D + Q(D) + 2 * M * D * E) + (16 * M + 16 * M * E) + F * 4 * M * E * S + F * 2 * M * E * E)
    return (L * per_layer + F * A(E)) / 1e9


HBM_PEAK_BPS = 8.0e12    # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def record_work(rec, esz):
    """(algorithmic FLOPs, algorithmic HBM bytes) of ONE launch of a plan record, from the launch's own argument structs: matrix products count
    2mnk (causal attention over the visible pairs only), operands / weights / outputs are counted once."""
    from sea_amd import _native as N

    L = N.lib()
    fl = by = 0
    a = rec.args
    if rec.fn is L.sea_gemm_grouped or rec.fn is L.sea_gemm_fewrows:
        for g in a[0][:(a[1] if rec.fn is L.sea_gemm_grouped else a[2])]:
            fl += 2 * g.M * g.N * g.K * g.n_seg
            by += (0 if g.silu_c else g.M * g.K * g.n_seg * esz) + g.N * g.K * esz + (g.M * g.N * 4 if g.C32 else 0) + (g.M * g.N * esz if g.Cact else 0) \
                + (g.M * g.N * 4 if g.R else 0) + (g.M * g.N * esz if g.Z else 0)
    elif rec.fn is L.sea_qkv_rope_grouped or rec.fn is L.sea_qkv_rope_fewrows:
        for g in a[0][:(a[1] if rec.fn is L.sea_qkv_rope_grouped else a[2])]:
            fl += 2 * g.M * g.N * g.K
            by += g.M * g.K * esz + g.N * g.K * esz + g.M * g.N * esz
    elif rec.fn is L.sea_gemm_rownorm:
        for g in a[0][:a[1]]:
            fl += 2 * g.M * g.N * g.K * g.n_seg
            by += g.M * g.K * g.n_seg * esz + g.N * g.K * esz + (g.M * g.N * esz if g.Yact else 0) + (g.M * g.N * 4 if g.Y32 else 0) + (g.M * 2 * g.N * esz if g.mod else 0)
    elif rec.fn is L.sea_mlp_fc1_ln_gelu:
        for g in a[0][:a[1]]:
            fl += 2 * g.M * g.E * g.S
            rows_in = g.M * g.E * esz if not g.X32 else g.M * g.E * 4 * (1 + (1 if g.addend else 0) + (1 if g.Xout else 0)) + (g.M * 2 * g.E * esz if g.mod else 0)
            by += rows_in + g.S * g.E * esz + 3 * g.S * 4 + g.M * g.S * esz
    elif rec.fn is L.sea_mlp_fc2_proj_norm:
        for g in a[0][:a[1]]:
            fl += 2 * g.M * g.E * g.S + 2 * g.M * g.E * g.E
            by += g.M * g.S * esz + g.E * g.S * esz + g.E * g.E * esz + g.M * g.E * 4 + (g.M * g.E * 4 if g.Y32 else 0) + (g.M * g.E * esz if g.Yact else 0) \
                + (g.M * 2 * g.E * esz if g.mod else 0)
    elif rec.fn is L.sea_splitk_finish:
        for g in a[0][:a[1]]:
            by += g.M * g.N * 4 * (g.S + (1 if g.R else 0) + (1 if g.C32 else 0)) + (g.M * g.N * esz if g.Cact else 0)
    elif rec.fn is L.sea_adaln_qkv:
        for g in a[0][:a[1]]:
            fl += 2 * g.M * (2 * g.E) * (2 * g.E) + 2 * g.M * (3 * g.E) * g.E
            by += g.M * g.E * 4 + g.M * 4 + (2 * g.E) * (2 * g.E) * esz + 3 * g.E * g.E * esz + 3 * g.M * g.E * esz
        for g in (a[3][:a[4]] if a[3] is not None else []):
            fl += 2 * g.M * g.N * g.K
            by += g.M * g.K * esz + g.N * g.K * esz + g.M * g.N * esz
    elif rec.fn is L.sea_mlp_block:
        for g, h in zip(a[0][:a[2]], a[1][:a[2]]):
            fl += 2 * g.M * g.E * g.S + 2 * g.M * g.E * g.S + 2 * g.M * g.E * g.E
            rows_in = g.M * g.E * esz if not g.X32 else g.M * g.E * 4 * (1 + (1 if g.addend else 0)) + (g.M * 2 * g.E * esz if g.mod else 0)
            by += rows_in + 2 * g.S * g.E * esz + 3 * g.S * 4 + g.E * g.E * esz + (g.M * g.E * 4 if h.R else 0) + (g.M * g.E * 4 if h.Y32 else 0) + (g.M * g.E * esz if h.Yact else 0) \
                + (g.M * 2 * g.E * esz if h.mod else 0)
    elif rec.fn is L.sea_exchange_tail:
        for g in rec.keep[:a[1]]:
            fl += g.n_seg * 2 * g.M * g.D * g.D + 2 * g.M * g.D * g.E + (2 * g.M * g.E * g.D if g.has_down else 0)
            by += g.n_seg * (g.M * g.D * esz + g.D * g.D * esz) + g.D * g.E * esz + 2 * g.M * g.E * 4 + (g.E * g.D * esz + g.M * g.D * esz if g.has_down else 0)
    elif rec.fn is L.sea_gemm_adaln:
        for g in a[0][:a[1]]:
            fl += 2 * g.M * 2 * g.d * g.K
            by += g.M * g.K * esz + 2 * g.d * g.K * esz + (g.M * g.d * (4 + esz) if g.X else g.M * 2 * g.d * esz)
    elif rec.fn is L.sea_row_chain or rec.fn is L.sea_row_chain_riders:
        if rec.fn is L.sea_row_chain_riders and a[4] > 0:   # the launch's share of the rider GEMM (tiles of 128 x 128, counted by tile)
            tot = sum(((g.M + 127) // 128) * ((g.N + 127) // 128) for g in a[3][:a[4]])
            share = a[6] / max(tot, 1)
            for g in a[3][:a[4]]:
                fl += share * 2 * g.M * g.N * g.K
                by += share * (g.N * g.K * esz + g.M * g.N * esz)
        for g in a[0][:a[1]]:
            K2 = g.D if g.n_seg > 0 else g.E
            fl += g.n_seg * 2 * g.M * g.D * g.D + 2 * g.M * K2 * g.E + (2 * g.M * g.E * g.D if g.has_down else 0)
            by += g.n_seg * (g.M * g.D * esz + g.D * g.D * esz) + (g.M * g.E * esz if g.n_seg == 0 else 0) + K2 * g.E * esz + 2 * g.M * g.E * 4 \
                + (g.E * g.D * esz + (g.M * 2 * g.D * esz if g.down.mod else 0) if g.has_down else 0)
            for q in g.proj[:g.n_proj]:
                fl += 2 * g.M * q.N * q.K
                by += q.N * q.K * esz + g.M * q.N * esz
    elif rec.fn is L.sea_attention_fwd:
        P = rec.keep
        tri = P.Tq * (P.Tq + 1) // 2 + P.Tq * (P.Tk - P.Tq)
        fl = P.n_problems * 4 * P.B * P.H * tri * P.hd
        by = P.n_problems * P.B * P.H * (2 * P.Tq + 2 * P.Tk) * P.hd * esz
    elif rec.fn is L.sea_attention_bwd:
        P = rec.keep
        tri = P.Tq * (P.Tq + 1) // 2
        fl = P.n_problems * 10 * P.B * P.H * tri * P.hd          # S, dP, dV, dK, dQ: five products where the forward has two
        by = P.n_problems * P.B * P.H * 8 * P.Tq * P.hd * esz
    elif rec.fn is L.sea_wgrad_grouped:
        for g in a[0][:a[1]]:
            fl += 2 * g.M * g.N * g.K
            by += g.M * (g.N + g.K) * esz + g.N * g.K * 4
    return fl, by


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


# ------------------------------------------------------------------------------------------------------------------ CPU baseline
def _oracle_params(cfg):
    from oracle.recipe import param_schema

    g = torch.Generator().manual_seed(42)
    p = {}
    for k, (shp, kind) in param_schema(cfg).items():
        p[k] = torch.randn(shp, generator=g) * 0.02 if kind == "lin_w" else (torch.ones(shp) if kind == "norm_w" else torch.zeros(shp))
    return p


def cpu_baseline(B, T):
    """The CPU oracle on the host cores, bounded (about 30 s): forward (1 warm-up + 3 passes), forward + backward + AdamW (1 warm-up + 1 pass,
    train/train_temporal.py:254-258), and the reference's recompute rollouts of 8 and 100 steps (utils/train_utils.py:202-209), B = 1, fp32."""
    from oracle import sea_oracle as O

    c = CFG
    cfg = O.OracleConfig(c["L"], c["E"], c["H"], c["max_len"], c["sr"], c["src"], c["F"], 2, True, "adaln")
    p = _oracle_params(cfg)
    x = torch.randn(B, T, c["F"], c["E"], generator=torch.Generator().manual_seed(1234))
    tgt = torch.randn(B, T, c["F"], c["E"], generator=torch.Generator().manual_seed(4321))
    ib = torch.rand(B, T, 1, generator=torch.Generator().manual_seed(1235))
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu_baseline: oracle on {cores} host threads: forward ...")
    with torch.no_grad():
        O.model_forward(x, ib, p, cfg)
        n = 3
        t0 = time.perf_counter()
        for _ in range(n):
            O.model_forward(x, ib, p, cfg)
        fwd = (time.perf_counter() - t0) / n
    log("cpu_baseline: forward + backward + AdamW ...")
    O.train_steps(x, ib, tgt, p, cfg, 1, lr=1e-4)
    t0 = time.perf_counter()
    O.train_steps(x, ib, tgt, p, cfg, 1, lr=1e-4)
    trn = time.perf_counter() - t0
    log("cpu_baseline: recompute rollouts (8 and 100 steps) ...")
    roll = {}
    with torch.no_grad():
        O.rollout(x[:, :1], ib, 8, p, cfg)
        for n_steps in (8, 100):
            t0 = time.perf_counter()
            O.rollout(x[:, :1], ib, n_steps, p, cfg)
            roll[n_steps] = n_steps * B / (time.perf_counter() - t0)
    return {"value": B / fwd, "unit": "trajectory-steps/s", "cores": cores, "kind": "port",
            "sample": f"oracle fp32 at the bench shape (B={B}, T={T}): forward 1 warm-up + {n} passes, {fwd:.3f} s/pass; forward+backward+AdamW 1 warm-up + 1 pass; "
                      f"recompute rollouts of 8 and 100 steps; torch CPU threads={cores}",
            "forward_s_per_pass": fwd, "train_step_s": trn, "train_trajectory_steps_per_s": B / trn,
            "rollout8_steps_per_s": roll[8], "rollout100_steps_per_s": roll[100]}


# ------------------------------------------------------------------------------------------------------------------ helpers
def _sync_barrier(dist):
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()


def _timed(step, n_steps, warmup, dist, dev):
    """W untimed steps, then exactly n_steps between barrier + synchronize on both sides; the MAX over ranks in seconds."""
    for _ in range(max(warmup, 1)):
        r = step()
    _sync_barrier(dist)
    t0 = time.perf_counter()
    for _ in range(n_steps):
        r = step()
    _sync_barrier(dist)
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    return elapsed, r


def _launch_time_ms(rec, reps=20):
    """Average duration of one launch of a record: `reps` back-to-back launches between two HIP events recorded on the launch stream — captured in one
    graph (no dispatch gap between them: the figure rocprofv3's per-kernel average agrees with) when no process group is alive; with one alive the
    launches are enqueued directly (its watchdog thread's event queries would break a global-mode capture), which adds ~1 us of launch boundary each."""
    from sea_amd import _native as N

    use_graph = not (torch.distributed.is_available() and torch.distributed.is_initialized())
    if use_graph:
        st = torch.cuda.Stream()
        gk = torch.cuda.CUDAGraph()
        with torch.cuda.stream(st):
            with torch.cuda.graph(gk, stream=st):
                for _ in range(reps):
                    rc = rec.fn(*rec.args, st.cuda_stream)
                    assert rc == 0, rec.name
        burst = gk.replay
    else:
        stream = N.stream_ptr()

        def burst():
            for _ in range(reps):
                rc = rec.fn(*rec.args, stream)
                assert rc == 0, rec.name

    burst()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        burst()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


def _time_list(recs, iters):
    """[(record, ms)] per record: one HIP event pair per launch on the launch stream (includes a few us of event / dispatch gap each)."""
    from sea_amd import _native as N

    stream = N.stream_ptr()
    recs = [r for r in recs if r.fn is not None]
    tot = [0.0] * len(recs)
    for _ in range(iters):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(recs) + 1)]
        evs[0].record()
        for k, r in enumerate(recs):
            rc = r.fn(*r.args, stream)
            assert rc == 0, r.name
            evs[k + 1].record()
        torch.cuda.synchronize()
        for k in range(len(recs)):
            tot[k] += evs[k].elapsed_time(evs[k + 1])
    return [(r, t / iters) for r, t in zip(recs, tot)]


def _roofline(recs_times, esz, traffic_file=None):
    """Roofline object of the dominant launch (largest share of device time) of a timed record list."""
    rec, dom_ms_events = max(recs_times, key=lambda rt: rt[1])
    dom_ms = _launch_time_ms(rec)
    fl, by = record_work(rec, esz)
    tflops = fl / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    gbps = by / (dom_ms * 1e-3) / 1e9 if by and dom_ms > 0 else None
    traffic = traffic_src = None
    if traffic_file and os.path.exists(traffic_file):
        e = json.load(open(traffic_file))["launches"].get(rec.name)
        if e is not None:
            traffic, traffic_src = e["hbm_bytes"], os.path.relpath(traffic_file, ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, bytes per launch)"
    return {"kernel": rec.name, "bound": "mfma", "achieved": tflops, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": tflops / PEAK_BF16_TFLOPS,
            "traffic": traffic, "traffic_source": traffic_src,
            "hbm_GBps": gbps, "hbm_frac": gbps / PEAK_HBM_GBPS if gbps else None, "algorithmic_bytes": by, "flop_per_algorithmic_byte": fl / by if by else None,
            "machine_balance_flop_per_byte": PEAK_BF16_TFLOPS * 1e12 / (PEAK_HBM_GBPS * 1e9),
            "launch_ms": dom_ms, "launch_ms_single_event_pair": dom_ms_events, "launch_gflop": fl / 1e9,
            "device_ms_all_launches": sum(t for _, t in recs_times),
            "timing": "launch_ms: HIP events on the launch stream around 20 back-to-back launches of the dominant record (one captured graph at N = 1) / 20, best of 5"}


def _pmc_file(name):
    """profiles/<this round's file>, or the previous round's while this round's has not been produced yet."""
    for rnd in ("r04", "r03", "r02"):
        f = os.path.join(ROOT, "profiles", f"{rnd}_{name}")
        if os.path.exists(f):
            return f
    return None


N_CU, CLOCK_GHZ = 256, 2.4   # MI355X_MICROARCH.md "Chip-level parameters"


def _attention_object(recs_times):
    """The attention launches of a forward plan against the VALU issue bound of their softmax ('attention is exp-bound, not MFMA-bound at head
    dims 16 / 32'): per visible (query, key) pair one v_exp_f32 (transcendental: a quarter of the fp32 rate, 16 lanes per clock and CU) and at least three
    full-rate VALU operations (running maximum, pack to bf16, the share of masking / rescaling; the subtraction of the reference rides in the MFMA's C
    operand and the row sums are an MFMA) at 64 lanes per clock and CU."""
    from sea_amd import _native as N

    L = N.lib()
    att = [r for r, _ in recs_times if r.fn is L.sea_attention_fwd]
    if not att:
        return None
    scores, flops, per = 0, 0, {}
    for r in att:
        P = r.keep
        tri = P.Tq * (P.Tq + 1) // 2 + P.Tq * (P.Tk - P.Tq)
        n = P.n_problems * P.B * P.H * tri
        ms = _launch_time_ms(r)
        bound_us = n * (1.0 / 16 + 3.0 / 64) / (N_CU * CLOCK_GHZ * 1e3)
        scores += n
        flops += 4 * n * P.hd
        per[r.name] = {"head_dim": P.hd, "scores": n, "launch_us": round(ms * 1e3, 2), "valu_bound_us": round(bound_us, 2), "frac_of_valu_bound": round(bound_us / (ms * 1e3), 3),
                       "mfma_tflops": round(4 * n * P.hd / (ms * 1e-3) / 1e12, 1)}
    tot_us = sum(v["launch_us"] for v in per.values())
    bound = scores * (1.0 / 16 + 3.0 / 64) / (N_CU * CLOCK_GHZ * 1e3)
    return {"bound": "valu (1 v_exp_f32 at quarter rate + 3 fp32 VALU operations per visible query-key pair, 256 CUs at 2.4 GHz)", "scores": scores, "achieved_us": round(tot_us, 2),
            "bound_us": round(bound, 2), "frac": round(bound / tot_us, 3), "mfma_tflops": round(flops / (tot_us * 1e-6) / 1e12, 1), "launches": per}


def build_model(dev, dtype, F=3, ln="adaln", max_len=2024):
    from sea_amd.models.temporal import TemporalModel

    c = CFG
    torch.manual_seed(42)
    m = TemporalModel(c["L"], c["E"], c["H"], max_len, c["sr"], c["src"], F, 2, 0.0, "sea", "learnable", "mlp", "add", 1, 1, True, ln)
    m.set_compute_dtype(dtype)
    return m.to(dev)


def inputs(B, T, F, E, rank, dev):
    x = torch.randn(B, T, F, E, generator=torch.Generator().manual_seed(1234 + rank)).to(dev)
    tgt = torch.randn(B, T, F, E, generator=torch.Generator().manual_seed(4321 + rank)).to(dev)
    ib = torch.rand(B, T, 1, generator=torch.Generator().manual_seed(1235 + rank)).to(dev)
    return x, tgt, ib


# ------------------------------------------------------------------------------------------------------------------ legs
def leg_rollout(args, dist, dev, rank, world, steps, warmup, B):
    c = CFG
    F, E, H, L, T = c["F"], c["E"], c["H"], c["L"], args.seq
    D, S = E // 2, E * c["sr"]
    model = build_model(dev, args.dtype).eval()
    x, _, ib = inputs(B, T, F, E, rank, dev)
    eng = model.engine(dev)
    # Default: the plan's launch list replayed by ONE native call per step (sea_run_list: the host stays ~3 steps ahead of the device).  Measured on one
    # box, same build: 0.2360 / 0.2371 ms per step against 0.2420 / 0.2424 as a captured HIP graph — a graph launch costs ~10 us of device idle time per replay
    # (rocprofv3: the gap in front of the first kernel of every replay), more than the launch boundaries it saves.  --graph times the graph replay.
    use_graph = args.graph and not args.no_graph
    step = (lambda: eng.forward_graphed(x, ib)) if use_graph else (lambda: eng.forward(x, ib))
    log(f"rollout leg: cfg2 forward, B={B} per GPU, {'hip-graph' if use_graph else 'plain'} replay, {steps} timed steps")
    with torch.no_grad():
        # The per-launch timing pass (every launch of the plan between HIP events, the dominant launch and the attention launches in bursts) runs BEFORE the
        # timed steps since round 3: a step is 0.23 ms, so `--warmup 5` is 1 ms of device work and the timed region of a short call (`--steps 20`: 5 ms)
        # used to measure the device coming out of idle — 0.2372 ms/step against 0.2248 for the same build once ~100 steps had run.
        # The timed region itself is unchanged: W untimed steps, then exactly K steps between barrier + synchronize.
        plan = eng.plan(B, T, "full")
        out_keep = torch.empty_like(x)   # stays alive until the records have been timed (a freed output block can be unmapped — e.g. by the cache flush a graph
        plan.bind(x, ib, out_keep)       # capture starts with — under launches that still write it: seen as a memory fault when the final-norm launch was timed)
        plan.run()
        torch.cuda.synchronize()
        times = _time_list(plan.records, iters=10)
        esz = 2 if args.dtype == "bf16" else 4
        pmc = _pmc_file("forward_cfg2_pmc_traffic.json") if (B, T, args.dtype) == (1, 2024, "bf16") else None
        roof = _roofline(times, esz, pmc)
        attn = _attention_object(times)
        del out_keep
        elapsed, out = _timed(step, steps, warmup, dist, dev)
        assert torch.isfinite(out).all()
    ms = elapsed / steps * 1e3
    gflop = algorithmic_gflop(B, T, F, E, H, D, S, L)

    def again():
        """The same W warm-up + K timed steps once more (run_rank calls it behind the other legs of a default run, when the device has been busy for seconds)."""
        with torch.no_grad():
            el, o = _timed(step, steps, warmup, dist, dev)
            assert torch.isfinite(o).all()
        return el

    return {"value": world * B * steps / elapsed, "unit": "trajectory-steps/s", "ms_per_step": ms, "steps": steps, "_again": again,
            "workload": f"cfg2: cylinder_flow temporal model E={E} H={H} F={F} L={L} adaln, forward-only rollout step at T={T} (recompute mode), B={B} per GPU",
            "replay": "hip-graph" if use_graph else "plain (one native call per step: sea_run_list)", "order": "per-launch timing pass, then W warm-up + K timed steps",
            "model_algorithmic_gflop_per_step": gflop,
            "model_mfma_frac": gflop / (ms * 1e-3) / 1e3 / PEAK_BF16_TFLOPS, "n_launches": len(times),
            "roofline": roof, "attention": attn, "launch_breakdown_ms": {r.name: round(t, 4) for r, t in times}}


def leg_train(args, dist, dev, rank, world, steps, warmup, B):
    from sea_amd.utils.train_utils import initialize_optimizer

    c = CFG
    F, E, H, L, T = c["F"], c["E"], c["H"], c["L"], args.seq
    D, S = E // 2, E * c["sr"]
    model = build_model(dev, args.dtype).train()
    x, tgt, ib = inputs(B, T, F, E, rank, dev)
    eng = model.engine(dev)
    opt = initialize_optimizer(model, {"learning_rate": 1e-4})
    single_ms = None
    if dist is not None:
        # the same step without the collective, on every rank at once: the single-GPU figure the data-parallel one is compared with
        for _ in range(3):
            eng.train_step(x, tgt, ib, opt, allreduce=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            eng.train_step(x, tgt, ib, opt, allreduce=False)
        torch.cuda.synchronize()
        single_ms = (time.perf_counter() - t0) / 10 * 1e3
        from sea_amd.parallel import broadcast_parameters

        broadcast_parameters(eng.params.flat32)      # the ranks drifted apart in those local steps: start the timed run in sync
        eng.params.sync(force=True)
        eng.params.sync_transposed(force=True)
    log(f"train leg: cfg3/cfg4 fwd+bwd+AdamW, B={B} per GPU, world {world}, {steps} timed steps")
    elapsed, loss = _timed(lambda: eng.train_step(x, tgt, ib, opt), steps, warmup, dist, dev)
    assert torch.isfinite(loss).all()
    ms = elapsed / steps * 1e3
    gflop = 3 * algorithmic_gflop(B, T, F, E, H, D, S, L)
    out = {"allreduce_calls_per_step": int(getattr(eng, "last_allreduce_calls", 0)), "value": world * B * steps / elapsed, "unit": "trajectory-steps/s", "train_steps_per_s": steps / elapsed, "ms_per_step": ms, "steps": steps,
           "workload": f"cfg{'3' if world == 1 else '4'}: cylinder_flow temporal model E={E} H={H} F={F} L={L} adaln, fwd+bwd+AdamW (teacher-forced, T={T}), B={B} per GPU, "
                       f"global batch {world * B}",
           "parallelism": f"dp{world}: replicated model, batch split by rank, the flat gradient all-reduced once per step (RCCL; every element once, in up to three slices "
                          "issued under the backward — allreduce_calls_per_step; SEA_DP_OVERLAP=0: one collective), 1/world folded into AdamW",
           "world_size": world, "model_algorithmic_gflop_per_step": gflop, "model_mfma_frac": gflop / (ms * 1e-3) / 1e3 / PEAK_BF16_TFLOPS}
    if dist is not None:
        from sea_amd.parallel import parameters_in_sync

        g = eng.grads[:eng.params.n_live]
        _sync_barrier(dist)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
        e1.record()
        torch.cuda.synchronize()
        ar_ms = e0.elapsed_time(e1) / 10
        out.update({"allreduce_ms": ar_ms, "allreduce_bytes": g.numel() * g.element_size(), "allreduce_dtype": str(g.dtype).replace("torch.", ""),
                    "allreduce_busbw_GBps": g.numel() * g.element_size() * 2 * (world - 1) / world / (ar_ms * 1e-3) / 1e9,
                    "backend": "nccl (RCCL)", "single_gpu_ms_per_step": single_ms, "dp_efficiency_vs_single_gpu_step": single_ms / ms,
                    "parameters_in_sync_after_run": parameters_in_sync(eng.params.flat32)})
    # per-launch breakdown of forward + backward, rank-local
    plan = eng.train_plan(B, T)
    esz = 2 if args.dtype == "bf16" else 4
    times = _time_list(list(plan.records) + list(plan.bwd), iters=3)
    out["roofline"] = _roofline(times, esz, _pmc_file("train_cfg3_pmc_traffic.json") if (B, T, args.dtype, world) == (8, 2024, "bf16", 1) else None)
    out["n_launches"] = len(times)
    out["top_launches_ms"] = {r.name: round(t, 4) for r, t in sorted(times, key=lambda rt: -rt[1])[:12]}
    return out


def leg_kv(args, dev, rank):
    from sea_amd.utils.train_utils import rollout

    c = CFG
    out = {}
    for key, F, ln, n_steps, reps, Bk in (("cfg2_2024_steps", 3, "adaln", args.seq, 3, 1), ("cfg5_shape_100_steps_F2_ln", 2, "ln", 100, 20, 1),
                                          ("cfg2_B8_2024_steps", 3, "adaln", args.seq, 2, 8)):   # B = 8: what full_autoregressive_evaluation does over a loader batch
        model = build_model(dev, args.dtype, F=F, ln=ln).eval()
        x, _, ib = inputs(Bk, max(n_steps, 1), F, c["E"], rank, dev)
        x0 = x[:, :1].contiguous()
        for _ in range(2):
            r = rollout(model, x0, ib, n_steps, mode="kv")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = rollout(model, x0, ib, n_steps, mode="kv")
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        assert torch.isfinite(r).all()
        out[key] = {"steps_per_s": n_steps / dt, "trajectory_steps_per_s": Bk * n_steps / dt, "ms_per_step": dt / n_steps * 1e3, "rollout_ms": dt * 1e3,
                    "workload": f"exact KV-cache rollout of {n_steps} steps, B={Bk}, E={c['E']} H={c['H']} F={F} L=1 {ln}"}
        if n_steps <= 100:   # the reference-equivalent recompute rollout of the same length beside it
            for _ in range(2):
                rollout(model, x0, ib, n_steps, mode="recompute")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                rollout(model, x0, ib, n_steps, mode="recompute")
            torch.cuda.synchronize()
            out[key]["recompute_steps_per_s"] = n_steps / ((time.perf_counter() - t0) / 3)
        del model
    # the SHIPPED widths (BASELINE.json configs[0] / configs[4]: configs/cylinder_flow.py embed_dim 1024 adaln, configs/multiphase_flow.py embed_dim 2048 ln; both 2
    # field groups, 8 heads, 1 layer, block 2024): 100-step exact KV-cache rollouts (generic step plan: these widths are outside sea_kv_rollout's register-
    # resident kernels) with the reference-equivalent recompute rollout beside them
    from sea_amd.models.temporal import TemporalModel

    for key, E, ln in (("cfg1_cylinder_dims_100_steps", 1024, "adaln"), ("cfg5_multiphase_dims_100_steps", 2048, "ln")):
        torch.manual_seed(42)
        model = TemporalModel(1, E, 8, 2024, 8, 0, 2, 2, 0.0, "sea", "learnable", "mlp", "add", 1, 1, True, ln)
        model.set_compute_dtype(args.dtype)
        model = model.to(dev).eval()
        x0 = torch.randn(1, 1, 2, E, generator=torch.Generator().manual_seed(77)).to(dev)
        ib = torch.rand(1, 100, 1, generator=torch.Generator().manual_seed(78)).to(dev)
        res = {}
        for mode, reps in (("kv", 10), ("recompute", 3)):
            for _ in range(2):
                r = rollout(model, x0, ib, 100, mode=mode)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                r = rollout(model, x0, ib, 100, mode=mode)
            torch.cuda.synchronize()
            res[mode] = (time.perf_counter() - t0) / reps
            assert torch.isfinite(r).all()
        # bytes ONE step really reads: the sum over the step plan's own launches (their argument structs: weights, the few operand / output rows), NOT every
        # parameter of the model — the dead parameters and the condition MLPs hoisted out of the loop (kv_engine.CondPlan) are not streamed
        esz = 2 if args.dtype == "bf16" else 4
        nbytes = sum(p.numel() for p in model.parameters()) * esz
        step_bytes, n_launch = _kv_step_plan_bytes(model, esz)
        out[key] = {"steps_per_s": 100 / res["kv"], "ms_per_step": res["kv"] / 100 * 1e3, "recompute_steps_per_s": 100 / res["recompute"],
                    "parameter_MB": nbytes / 1e6, "step_plan_MB": step_bytes / 1e6, "step_plan_launches": n_launch,
                    "weight_read_GBps": step_bytes * 100 / res["kv"] / 1e9, "hbm_frac": step_bytes * 100 / res["kv"] / HBM_PEAK_BPS,
                    "workload": f"exact KV-cache rollout of 100 steps, B=1, E={E} H=8 F=2 L=1 {ln} (shipped width; a step's {n_launch} launches read {step_bytes / 1e6:.0f} MB, "
                                f"of {nbytes / 1e6:.0f} MB of parameters: dead parameters and the hoisted condition MLPs are not streamed)"}
        del model
    return out


def _kv_step_plan_bytes(model, esz):
    """(algorithmic bytes, launches) of ONE KV-cache step at B = 1: the generic step plan with the condition work hoisted, exactly as engine.rollout_kv builds
    it (tools/kv_shipped.py prints the same sum per launch)."""
    from sea_amd import kv_engine

    eng = model.engine()
    p = eng.plan(1, 1, "step", cond=kv_engine.cond_plan_for(eng, 100))
    recs = [r for r in p.records if r.fn is not None]
    return sum(record_work(r, esz)[1] for r in recs), len(recs)


def leg_shipped(args, dev, rank):
    """The reference's OWN two temporal configurations at their own size (reference configs/cylinder_flow.py:112-141: embed_dim 1024, 2 field groups,
    AdaLN, dropout 0.1, batch 2, windows of 399 steps; configs/multiphase_flow.py:112-141: embed_dim 2048, LayerNorm, dropout 0, batch 4, windows of 199):
    the teacher-forced training step (train/train_temporal.py:252-258) and the evaluation forward, with the dominant launch of the training step on the
    roofline.  Synthetic windows of those shapes, random-init weights (seed 42), bf16."""
    from sea_amd.configs import get_config
    from sea_amd.models.temporal import TemporalModel
    from sea_amd.utils.train_utils import initialize_optimizer

    out = {}
    esz = 2 if args.dtype == "bf16" else 4
    for key, case in (("cylinder_flow", "cylinder_flow"), ("multiphase_flow", "multiphase_flow")):
        c = get_config(case, "temporal")
        E, H, F, L, B, T = c["embed_dim"], c["n_heads"], c["num_fields"], c["num_layers"], c["batch_size"], c["dataset_src_len"]
        D, S = E // c["down_proj"], E * c["scale_ratio"]
        adaln = c["LN_type"].lower() == "adaln"
        torch.manual_seed(42)
        model = TemporalModel(L, E, H, c["block_size"], c["scale_ratio"], c["src_len"], F, c["down_proj"], c["dropout"], c["exchange_mode"], c["pos_encoding_mode"],
                              c["ib_scale_mode"], c["ib_addition_mode"], c["ib_mlp_layers"], c["ib_num"], c["add_info_after_cross"], c["LN_type"])
        model.set_compute_dtype(args.dtype)
        model = model.to(dev)
        x, tgt, ib = inputs(B, T, F, E, rank, dev)
        eng = model.engine(dev)
        gflop = algorithmic_gflop(B, T, F, E, H, D, S, L, adaln=adaln)
        model.eval()
        with torch.no_grad():
            for _ in range(40):   # (40 un-timed forwards: the device ramps out of idle over ~20 ms of work, profiles/r03_idle_ramp_probe.txt; with 3 this leg once read 1.28 ms for a 0.44 ms forward)
                o = eng.forward(x, ib)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(40):
                o = eng.forward(x, ib)
            torch.cuda.synchronize()
            fwd_ms = (time.perf_counter() - t0) / 40 * 1e3
            assert torch.isfinite(o).all()
            n_fwd = len([r for r in eng.plan(B, T, "full").records if r.fn is not None])
        model.train()
        opt = initialize_optimizer(model, {"learning_rate": c["learning_rate"]})
        for _ in range(3):
            loss = eng.train_step(x, tgt, ib, opt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            loss = eng.train_step(x, tgt, ib, opt)
        torch.cuda.synchronize()
        trn_ms = (time.perf_counter() - t0) / 10 * 1e3
        assert torch.isfinite(loss).all()
        plan = eng.train_plan(B, T)
        times = _time_list(list(plan.records) + list(plan.bwd), iters=3)
        out[key] = {"workload": f"configs/{case}.py temporal model at its own size: E={E} H={H} F={F} L={L} {c['LN_type']} dropout {c['dropout']}, batch {B} x {T} steps (teacher-forced)",
                    "forward_ms": fwd_ms, "forward_trajectory_steps_per_s": B / (fwd_ms * 1e-3), "forward_launches": n_fwd,
                    "forward_model_mfma_frac": gflop / (fwd_ms * 1e-3) / 1e3 / PEAK_BF16_TFLOPS,
                    "train_ms_per_step": trn_ms, "train_trajectory_steps_per_s": B / (trn_ms * 1e-3), "train_launches": len(times),
                    "train_model_mfma_frac": 3 * gflop / (trn_ms * 1e-3) / 1e3 / PEAK_BF16_TFLOPS,
                    "model_algorithmic_gflop_forward": gflop, "parameter_MB": sum(p.numel() for p in model.parameters()) * esz / 1e6,
                    "roofline": _roofline(times, esz), "top_launches_ms": {r.name: round(t, 4) for r, t in sorted(times, key=lambda rt: -rt[1])[:8]}}
        del model, eng, opt, plan, times
        torch.cuda.empty_cache()
    return out


# ------------------------------------------------------------------------------------------------------------------ rank body
def run_rank(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # SEA_DP_REHEARSE=1 (sea_amd/parallel.py) at N = 1: a process group of one RCCL rank, every collective of the N > 1 path issued — the data-parallel
    # branch of the train leg (broadcast, the three gradient slices under the backward, the timed all-reduce, the sync check) run on the one GPU of a test box
    rehearse = world == 1 and os.environ.get("SEA_DP_REHEARSE", "0") == "1"
    if world > 1 or rehearse:
        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist_mod.init_process_group("nccl", device_id=dev)
        dist = dist_mod
    legs = {"all": ("rollout", "train", "kv", "shipped"), "rollout": ("rollout",), "train": ("train",), "kv": ("kv",), "shipped": ("shipped",)}[args.mode]
    headline = "train" if (world > 1 and "train" in legs) else legs[0]
    res = {}
    if "rollout" in legs:
        res["rollout"] = leg_rollout(args, dist, dev, rank, world, args.steps if headline == "rollout" else max(args.steps // 4, 20), args.warmup, args.batch or 1)
        torch.cuda.empty_cache()
    if "train" in legs:
        res["train"] = leg_train(args, dist, dev, rank, world, args.steps if headline == "train" else max(min(args.steps // 8, 50), 10), min(args.warmup, 5),
                                 args.batch or 8)
        torch.cuda.empty_cache()
    if "kv" in legs and (world == 1 or headline == "kv"):
        res["kv"] = leg_kv(args, dev, rank)
        torch.cuda.empty_cache()
    if "shipped" in legs and (world == 1 or headline == "shipped"):
        res["shipped"] = leg_shipped(args, dev, rank)
    if "rollout" in res:
        again = res["rollout"].pop("_again")
        if len(legs) > 1 and world == 1:
            # Round 4: the headline's W + K steps are measured TWICE in a default run — right behind the per-launch timing pass as in rounds 1-3 (the device then comes
            # out of idle: model and plan were built on the host just before; `ms_per_step_from_idle`), and again behind the train / kv / shipped legs, when the device
            # has been busy for seconds and its clocks have settled (profiles/r03_idle_ramp_probe.txt: ~80 steps of ramp after 0.5 s of idleness).  The line's value is
            # the second measurement: the rate a rollout loop or a server sees; the first stays in the line.  Same timed region both times: barrier + synchronize,
            # W un-timed steps, exactly K steps, barrier + synchronize.
            r = res["rollout"]
            r["ms_per_step_from_idle"], r["value_from_idle"] = r["ms_per_step"], r["value"]
            el = again()
            r["ms_per_step"], r["value"] = el / r["steps"] * 1e3, world * (args.batch or 1) * r["steps"] / el
            r["model_mfma_frac"] = r["model_algorithmic_gflop_per_step"] / (r["ms_per_step"] * 1e-3) / 1e3 / PEAK_BF16_TFLOPS
            r["order"] = "per-launch timing pass, W + K steps (from idle), the other legs, then W warm-up + K timed steps again (device busy): the line's value"
    if rank == 0:
        h = res[headline]
        hk = h["cfg2_2024_steps"] if headline == "kv" else h
        if headline == "shipped":   # (--mode shipped: the shipped cylinder configuration's training step as the line's value)
            cyl = h["cylinder_flow"]
            hk = h = dict(cyl, value=cyl["train_trajectory_steps_per_s"], ms_per_step=cyl["train_ms_per_step"], steps=10, **{k: v for k, v in h.items() if k != "cylinder_flow"})
        line = {
            "metric": {"rollout": "rollout steps/sec (full-context forward, recompute mode) on cylinder_flow-shaped fields",
                       "train": "train steps/sec (fwd+bwd+AdamW, data-parallel) on cylinder_flow-shaped fields, in trajectory-steps/s",
                       "kv": "KV-cache rollout steps/sec on cylinder_flow-shaped fields",
                       "shipped": "train steps/sec (fwd+bwd+AdamW) at the shipped cylinder_flow configuration, in trajectory-steps/s"}[headline],
            "value": hk["steps_per_s"] * world if headline == "kv" else h["value"], "unit": "trajectory-steps/s", "n_gpus": world,
            "steps": h.get("steps", args.steps), "warmup": args.warmup, "ms_per_step": hk["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": hk["workload"], "headline_leg": headline, "seq_len": args.seq, "fields": CFG["F"], "embed_dim": CFG["E"],
                       "parallelism": h.get("parallelism", f"replicas x{world} (rollout shards by trajectory, no collective)")},
            "roofline": h.get("roofline"),
        }
        if res.get("rollout", {}).get("attention") is not None:
            line["attention"] = res["rollout"]["attention"]
        for k in ("rollout", "train", "kv", "shipped"):
            if k in res:
                line[k] = res[k]
        if "train" in res:
            line["train"]["allreduce_calls_per_step"] = res["train"].get("allreduce_calls_per_step", 0)
        if world == 1 and not args.no_cpu_baseline and args.mode in ("all", "rollout"):
            line["cpu_baseline"] = cpu_baseline(1, args.seq)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------------------ launcher
def launch_ranks(args) -> int:
    """`python bench.py --gpus N`, N > 1, outside torch.distributed.run: start N fresh rank processes (this process has made no GPU call), pass
    rank 0's JSON line through, fail if any rank fails."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is drained by a thread while EVERY child is polled: a rank that dies first (build failure, bad device) must end the job at once —
    # rank 0 would otherwise sit in its first collective until the RCCL / gloo timeout
    import threading

    buf = []
    reader = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("SEA_BENCH_RANK_TIMEOUT_S", "3000"))
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs) or time.time() > deadline:
            failed = True
            break
        time.sleep(0.2)
    if failed:
        time.sleep(1.0)   # let the others notice on their own (a clean error message beats a kill)
        for p in procs:
            if p.poll() is None:
                p.kill()
    rcs = [p.wait() for p in procs]
    reader.join(timeout=10)
    out0 = buf[0] if buf else b""
    if any(rcs):
        print(f"[bench] rank exit codes {rcs}", file=sys.stderr)
        return 1
    lines = [ln for ln in out0.decode().splitlines() if ln.startswith("{")]
    if not lines:
        print("[bench] rank 0 printed no JSON line", file=sys.stderr)
        return 1
    print(lines[-1], flush=True)
    return 0


def _dump_maps_at_exit():
    """SEA_DUMP_MAPS=<file>: write /proc/self/maps when the interpreter shuts down (before the C library's exit handlers run) — the map that resolves
    the frame addresses of a crash inside exit() (the rocprofv3 teardown fault of the KV leg, profiles/failures/)."""
    path = os.environ.get("SEA_DUMP_MAPS")
    if path:
        import atexit

        def dump():
            with open("/proc/self/maps") as f, open(path, "w") as o:
                o.write(f.read())
        atexit.register(dump)


def main():
    _dump_maps_at_exit()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps of the headline leg (the other legs run a fixed fraction)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=0, help="trajectories per GPU (default: 1 for the rollout leg, 8 for the train leg)")
    ap.add_argument("--seq", type=int, default=2024)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--graph", action="store_true", help="rollout leg: replay the captured HIP graph instead of the plan's launch list (default: the launch list)")
    ap.add_argument("--no-graph", action="store_true", help="(the default now; kept for the measurement scripts)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default="all", choices=["all", "rollout", "train", "kv", "shipped", "decode", "encode"],
                    help="all: every leg in one line (default); rollout / train / kv: one leg; decode / encode: the spatial decoder / encoder legs "
                         "(SURVEY.md §8f ranks 1 and 2)")
    args = ap.parse_args()
    if args.mode == "decode":
        return main_decode(args)
    if args.mode == "encode":
        return main_encode(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    run_rank(args)


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
    tr, T = args.batch or 1, args.seq
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
    from sea_amd import ops
    from sea_amd.utils.train_utils import inverse_transform_processed_data

    with torch.no_grad():
        z = inverse_transform_processed_data(roll, tr, T, P, len(groups))                              # [tr*T, P, G, D]

        def step():        # the decoder over the columns the un-patchify reads (a cell's mesh points), then the scatter + inverse scaling
            return unpatcher.decode_and_unpatch(dec, z)                                                 # [tr*T, n_points, 3]

        def step_full():   # the reference's two calls: every padded cell decoded in full, then un-patchified
            dec_out = decode_rollout(dec, roll, P)                                                      # [tr*T, P, 3, n_inp]
            return dec_out, unpatcher.inverse_scale_and_unpatch(dec_out[..., :C_pad], layout="BPFC")

        def timed(fn, n):
            for _ in range(max(args.warmup, 1)):
                r = fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                r = fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n, r

        per, out = timed(step, args.steps)
        per_full, (dec_out, out_full) = timed(step_full, max(args.steps // 2, 2))
        assert torch.equal(out, out_full)                       # the same fields, bit for bit
        elapsed = per * args.steps
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            unpatcher.inverse_scale_and_unpatch(dec_out[..., :C_pad], layout="BPFC")
        e1.record()
        torch.cuda.synchronize()
        u_ms = e0.elapsed_time(e1) / 5
        u_bytes = tr * T * (n_points * 3 * 4 + n_points * 3 * 4) + n_points * 4   # read every mesh point's decoded value + write it once (gather form)
        # the dominant launch: the second layer over the needed columns (one GEMM group per (bucket, field)), HIP events on the launch stream
        M = tr * T * P
        dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
        order, buckets, _, _ = unpatcher._prefix_plan(max(1, 16 // 3))
        Cp = dec._n_inp_p
        W1, W2 = dec._weights(dt)
        bias = dec._shadow[3]
        hid = [torch.randn(M, hidden, device=dev).to(dt) for _ in groups]
        o2 = torch.empty(M, 3 * Cp, device=dev)
        gl, cols = [], 0
        for p_lo, p_hi, n_cols in buckets:
            nn = min((max(n_cols, 1) + 31) // 32 * 32, Cp)
            r0, r1, f = p_lo * tr * T, p_hi * tr * T, 0
            cols += (r1 - r0) * nn * 3
            for g, grp in enumerate(groups):
                for j in range(len(grp)):
                    gl.append(dict(A=hid[g][r0:r1], W=W2[g][j * Cp:j * Cp + nn], bias=bias[g][j * Cp:j * Cp + nn], C32=o2[r0:r1, f * Cp:f * Cp + nn]))
                    f += 1
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
    alg_bytes = M * hidden * esz + 3 * Cp * hidden * esz + cols * 4      # read hidden + weights once, write the needed fp32 cells once
    flops = 2 * cols * hidden
    ms = elapsed / args.steps * 1e3
    line = {"metric": "decoded snapshots/sec (spatial decoder over a rollout)", "value": tr * T * args.steps / elapsed, "unit": "snapshots/s",
            "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"decode leg of the rollout evaluation: {tr} x {T} snapshots, 64 patches, spatial embed 16, hidden 480, groups [[0,1],[2]], "
                                   f"mesh of {n_points} points in 8x8 cells (padded cell size {C_pad}, n_inp {n_inp}); decoder over the cells' mesh points "
                                   f"({cols / (M * 3 * Cp):.0%} of the padded columns, {len(buckets)} buckets of patches) + un-patchify + inverse scaling"},
            "two_call_chain_ms": per_full * 1e3,
            "unpatchify": {"ms": u_ms, "algorithmic_GB": u_bytes / 1e9, "achieved_GBps": u_bytes / (u_ms * 1e-3) / 1e9, "frac_of_hbm_peak": u_bytes / (u_ms * 1e-3) / 1e9 / 8000.0},
            "roofline": {"kernel": "decode.layer2 over the needed columns (gemm_grouped)", "bound": "hbm", "achieved": alg_bytes / (k_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
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
    n_snap = (args.batch or 1) * args.seq
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
