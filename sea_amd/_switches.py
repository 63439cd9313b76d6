"""The package's environment switches, all of them (defaults are the measured best):

  SEA_AMD_DTYPE   fp32 | bf16            compute dtype of new models (TemporalModel.set_compute_dtype overrides)
  SEA_CHECK_PTRS  1                      audit every launch plan's device pointers at EVERY bind (default: at a plan's first bind) — sea_amd/ptrcheck.py
  SEA_DP_OVERLAP  0                      data-parallel step: ONE gradient all-reduce after the backward instead of slices under it
  SEA_DP_REHEARSE 1                      a process group of ONE rank issues the data-parallel step's collectives instead of skipping them (sea_amd/parallel.py:
                                           how the RCCL path runs on a one-GPU box: tests/test_parallel_gpu.py, `SEA_DP_REHEARSE=1 python bench.py --mode train`)
  SEA_PLAN        key=value,...          forms of the launch plans (what the tests force to compare every form with the default one):
                                           lanes=none|cond|all   parallel graph branches          graph_lanes=0      a captured graph replays its lanes in record order
                                           norm=0                Linear + row norm as two launches  xtail=0            a field's exchange tail as three launches
                                           xtail_max_rows=N      ... from N rows up                  fold_ib=0          the info-bottleneck add as a launch of its own
                                           silu=0|1              generated GEMM operand off / on    mlp1 / mlpnorm / mlp2=0|1   the fused MLP halves off / forced
  SEA_KV          key=value,...          KV-cache rollout: fast=0 (generic step plan), hoist=0 (condition work per step), gemv=0 (step plan without the few-row launches of gemv.hip),
                                           loop=python (step loop in Python),
                                           force_err=1 (test hook: the persistent launch "reports" a hand-off that gave up)
  SEA_TUNE        key=value,...          native tuning aids read by libsea_hip.so (sea_tune() in core.hip): kv_persist, kv_pre, gemm_norm_rows, gemm_tile, gemm256, gemm_ws (0 off, 2 also short launches), attn_split4, attn_paired,
                                           attnb_mode (attention backward: 1 XCD-local order, 2 paired causal tiles, 3 both, 0 neither), ...
  SEA_EXTRA_FLAGS "..."                  extra hipcc flags for `python -m sea_amd.build` (A/B builds)
(tests only: SEA_TEST_DP_BACKEND=nccl runs the data-parallel tests one rank per GPU over RCCL.)
"""
from __future__ import annotations

import os
from typing import Dict, Optional


def _parse(var: str) -> Dict[str, str]:
    out: Dict[str, str] = {}
    for tok in os.environ.get(var, "").split(","):
        if "=" in tok:
            k, v = tok.split("=", 1)
            out[k.strip()] = v.strip()
    return out


def plan(key: str, default: Optional[str] = None) -> Optional[str]:
    """SEA_PLAN token (read at every plan build: tests change it between builds)."""
    return _parse("SEA_PLAN").get(key, default)


def kv(key: str, default: Optional[str] = None) -> Optional[str]:
    return _parse("SEA_KV").get(key, default)
