"""sea_kv_rollout (sea_amd/csrc/kvstep.hip, sea_amd/kv_engine.py): the seven-launch KV-cache step against (a) the CPU oracle's restatement of the
reference's recompute loop (utils/train_utils.py:202-209), (b) the generic step plan, which the golden-vector tests of tests/test_model_gpu.py pin,
and (c) itself (replays, batch rows independent of each other).  The golden rollouts themselves run through this path too
(tests/test_model_gpu.py::test_rollout_kv_cache_matches_reference_golden: all three fixtures are inside its limits).

Tolerances: fp32 1e-4 against the oracle (north_star's bar), 1e-5 between the two device paths; bf16 3e-2 per step-count stated in each test.
"""
import os

import numpy as np
import pytest
import torch

from oracle import sea_oracle as O
from oracle.recipe import recipe_inputs, recipe_params
from tests.conftest import rel_l2
from tests.test_model_gpu import build

pytestmark = pytest.mark.gpu


def _kv(m, x0, ib, n, monkeypatch, fast):
    from sea_amd.utils.train_utils import rollout

    monkeypatch.setenv("SEA_KV", ("fast=1" if fast else "fast=0") + (os.environ.get("SEA_KV_EXTRA", "")))
    before = len(m.engine()._kv_fast)
    out = rollout(m, x0, ib, n, mode="kv")
    if not fast:
        assert len(m.engine()._kv_fast) == before
    return out


CASES = [
    # layers, E, H, max_len, scale_ratio, src_len, F, down_proj, ib after cross, LN, exchange, ib add, ib scale | B, steps
    ((1, 64, 4, 96, 8, 0, 3, 2, True, "adaln"), 2, 40),
    ((2, 128, 8, 64, 4, 0, 2, 2, False, "ln"), 3, 30),                                   # two layers, ib before the block, head dims 16 / 8
    ((1, 256, 8, 160, 8, 0, 3, 2, True, "adaln"), 1, 130),                               # cfg2's widths
    ((1, 128, 4, 64, 8, 0, 4, 2, True, "adaln", "sea", "none", "mlp"), 2, 20),           # four fields, no info-bottleneck
    ((1, 128, 4, 64, 8, 0, 3, 2, True, "ln", "simple", "add", "fourier"), 2, 20),        # no exchange
    ((2, 64, 8, 48, 4, 0, 1, 1, True, "adaln", "sea", "add", "linear"), 5, 24),          # one field ('sea' without partners), head dim 8
    ((1, 512, 8, 40, 4, 0, 2, 4, True, "ln"), 1, 16),                                    # E = 512 (head dim 64), D = 128
]


@pytest.mark.parametrize("cfg_args,B,n", CASES)
def test_fast_kv_rollout_matches_oracle_and_generic_plan_fp32(cfg_args, B, n, monkeypatch):
    from sea_amd import kv_engine

    cfg = O.OracleConfig(*cfg_args)
    m = build(cfg, "fp32")
    assert kv_engine.supported(m.engine(), B)
    x, _, ib = recipe_inputs(B, n, cfg, seed=5)
    x0, ibg = x[:, :1].cuda().contiguous(), ib.cuda().contiguous()
    fast = _kv(m, x0, ibg, n, monkeypatch, True)
    assert len(m.engine()._kv_fast) == 1
    ref = O.rollout(x[:, :1], ib, n, recipe_params(cfg), cfg)
    assert fast.shape == ref.shape
    assert rel_l2(fast.cpu().numpy(), ref.numpy()) < 1e-4
    slow = _kv(m, x0, ibg, n, monkeypatch, False)
    assert rel_l2(fast.cpu().numpy(), slow.cpu().numpy()) < 1e-5
    again = _kv(m, x0, ibg, n, monkeypatch, True)     # caches and hand-off words are reused
    assert torch.equal(fast, again)
    short = _kv(m, x0, ibg, n // 2, monkeypatch, True)  # another step count: another batched condition pass
    assert torch.equal(short, fast[:, :n // 2])


@pytest.mark.parametrize("cfg_args,B,n", CASES[:4])
def test_fast_kv_rollout_bf16(cfg_args, B, n, monkeypatch):
    """bf16 weights and caches, fp32 activation vectors: within the single-forward bf16 tolerance of the fp32 oracle over the first 8 steps,
    and no further from it than the generic bf16 step plan over the whole rollout (+ 1e-2)."""
    cfg = O.OracleConfig(*cfg_args)
    m = build(cfg, "bf16")
    x, _, ib = recipe_inputs(B, n, cfg, seed=6)
    x0, ibg = x[:, :1].cuda().contiguous(), ib.cuda().contiguous()
    fast = _kv(m, x0, ibg, n, monkeypatch, True).cpu().numpy()
    slow = _kv(m, x0, ibg, n, monkeypatch, False).cpu().numpy()
    ref = O.rollout(x[:, :1], ib, n, recipe_params(cfg), cfg).numpy()
    e8, en, sn = rel_l2(fast[:, :8], ref[:, :8]), rel_l2(fast, ref), rel_l2(slow, ref)
    print(f"bf16 fast KV rollout rel-L2 vs fp32 oracle: 8 steps {e8:.3e}, {n} steps {en:.3e} (generic plan {sn:.3e})")
    assert e8 < 3e-2 and en < sn + 1e-2


def test_fast_kv_rows_of_a_batch_are_independent(monkeypatch):
    """Trajectory b of a batch equals the same trajectory rolled out alone (workgroup (field, b) indexing of every launch)."""
    cfg = O.OracleConfig(1, 128, 8, 64, 8, 0, 3, 2, True, "adaln")
    m = build(cfg, "fp32")
    x, _, ib = recipe_inputs(4, 24, cfg, seed=9)
    x0, ibg = x[:, :1].cuda().contiguous(), ib.cuda().contiguous()
    full = _kv(m, x0, ibg, 24, monkeypatch, True)
    monkeypatch.setenv("SEA_TUNE", "kv_persist=0")   # the same launches at both batch sizes (one trajectory alone would otherwise take the persistent form)
    for b in (0, 3):
        one = _kv(m, x0[b:b + 1].contiguous(), ibg[b:b + 1].contiguous(), 24, monkeypatch, True)
        assert torch.equal(one[0], full[b])
    monkeypatch.setenv("SEA_TUNE", "kv_persist=1")
    one = _kv(m, x0[1:2].contiguous(), ibg[1:2].contiguous(), 24, monkeypatch, True)
    assert rel_l2(one[0].cpu().numpy(), full[1].cpu().numpy()) < 1e-5


def test_fast_kv_full_length_cfg2(monkeypatch):
    """cfg2's model over its whole block (2024 steps, B = 1): fp32 fast path against the generic fp32 step plan; bf16 replays bit-identical and
    finite.  (The CPU oracle cannot run 2024 recompute steps in seconds: the generic plan is the checker here, itself pinned by the goldens.)"""
    cfg = O.OracleConfig(1, 256, 8, 2024, 8, 0, 3, 2, True, "adaln")
    x, _, ib = recipe_inputs(1, 2024, cfg, seed=3)
    x0, ibg = x[:, :1].cuda().contiguous(), ib.cuda().contiguous()
    m = build(cfg, "fp32")
    fast = _kv(m, x0, ibg, 2024, monkeypatch, True)
    slow = _kv(m, x0, ibg, 2024, monkeypatch, False)
    e = rel_l2(fast.cpu().numpy(), slow.cpu().numpy())
    print(f"fp32 fast vs generic KV rollout over 2024 steps: rel-L2 {e:.3e}")
    assert e < 1e-4
    mb = build(cfg, "bf16")
    a = _kv(mb, x0, ibg, 2024, monkeypatch, True)
    b = _kv(mb, x0, ibg, 2024, monkeypatch, True)
    assert torch.equal(a, b) and bool(torch.isfinite(a).all())
    print(f"bf16 fast vs fp32 fast over 2024 steps: rel-L2 {rel_l2(a.cpu().numpy(), fast.cpu().numpy()):.3e}")


PERSIST_CASES = [
    # one trajectory, one layer, fixed widths: the whole rollout is ONE launch (every workgroup one role, weights resident in registers)
    ((1, 256, 8, 160, 8, 0, 3, 2, True, "adaln"), 130),
    ((1, 128, 8, 96, 8, 0, 2, 2, False, "ln"), 60),                                      # two fields, ib before the block, head dims 16 / 8
    ((1, 128, 4, 64, 8, 0, 4, 2, True, "adaln", "sea", "none", "mlp"), 40),              # four fields (a third updated source streams its k / v weights)
    ((1, 64, 4, 64, 4, 0, 3, 2, True, "ln", "simple", "add", "fourier"), 30),            # no exchange: four roles only
    ((1, 256, 8, 64, 4, 0, 1, 2, True, "adaln", "sea", "add", "linear"), 40),            # one field (n / 2 > 16 rows: the batched condition pass of the
                                                                                         # shorter rollout runs the same GEMM kernel, so the prefix is bit-identical)
]


@pytest.mark.parametrize("dtype,tol_oracle,tol_paths", [("fp32", 1e-4, 1e-5), ("bf16", 3e-2, 2e-2)])
@pytest.mark.parametrize("cfg_args,n", PERSIST_CASES)
def test_persistent_kv_rollout(cfg_args, n, dtype, tol_oracle, tol_paths, monkeypatch):
    """The persistent form against the oracle, against the seven-launch form (SEA_TUNE=kv_persist=0) and against itself (a second rollout reuses the caches
    and the granule arena with fresh tags; a shorter one is its prefix)."""
    cfg = O.OracleConfig(*cfg_args)
    m = build(cfg, dtype)
    x, _, ib = recipe_inputs(1, n, cfg, seed=8)
    x0, ibg = x[:, :1].cuda().contiguous(), ib.cuda().contiguous()
    monkeypatch.setenv("SEA_TUNE", "kv_persist=1")
    a = _kv(m, x0, ibg, n, monkeypatch, True)
    ref = O.rollout(x[:, :1], ib, n, recipe_params(cfg), cfg)
    k = min(n, 8)
    assert rel_l2(a.cpu().numpy()[:, :k], ref.numpy()[:, :k]) < tol_oracle
    if dtype == "fp32":
        assert rel_l2(a.cpu().numpy(), ref.numpy()) < tol_oracle
    assert torch.equal(_kv(m, x0, ibg, n, monkeypatch, True), a)
    assert torch.equal(_kv(m, x0, ibg, n // 2, monkeypatch, True), a[:, :n // 2])
    monkeypatch.setenv("SEA_TUNE", "kv_persist=0")
    b = _kv(m, x0, ibg, n, monkeypatch, True)
    assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < tol_paths


def test_persistent_rollout_recovers_when_a_handoff_wait_gives_up(monkeypatch):
    """The recovery path of KvFast.rollout (sea_amd/kv_engine.py): when the persistent launch reports that a hand-off wait gave up — forced here by
    SEA_KV=force_err=1, in production another process holding CUs — the rollout is recomputed with the seven launches per step, the KvFast object
    stays on that form for good, and a LATER rollout on the same object (caches and granule arena reused) is still right."""
    cfg = O.OracleConfig(1, 128, 8, 96, 8, 0, 3, 2, True, "adaln")
    m = build(cfg, "fp32")
    n = 40
    x, _, ib = recipe_inputs(1, n, cfg, seed=12)
    x0, ibg = x[:, :1].cuda().contiguous(), ib.cuda().contiguous()
    ref = O.rollout(x[:, :1], ib, n, recipe_params(cfg), cfg)
    monkeypatch.setenv("SEA_TUNE", "kv_persist=1")
    good = _kv(m, x0, ibg, n, monkeypatch, True)                       # the persistent form, undisturbed
    kf = m.engine()._kv_fast[1]
    words = kf.G.handoff_words
    assert words > kf.B * kf.F * kf.D                                  # (it did take the persistent form)
    monkeypatch.setenv("SEA_KV_EXTRA", ",force_err=1")
    rec = _kv(m, x0, ibg, n, monkeypatch, True)                        # attempt 0 "fails", attempt 1 recomputes
    monkeypatch.delenv("SEA_KV_EXTRA")
    assert kf.G.handoff_words == kf.B * kf.F * kf.D < words            # the object gave the persistent form up for good
    assert int(kf.err.item()) == 0
    assert rel_l2(rec.cpu().numpy(), ref.numpy()) < 1e-4 and rel_l2(rec.cpu().numpy(), good.cpu().numpy()) < 1e-5
    later = _kv(m, x0, ibg, n // 2, monkeypatch, True)                 # a later, shorter rollout on the same KvFast
    assert rel_l2(later.cpu().numpy(), ref.numpy()[:, :n // 2]) < 1e-4
    assert torch.equal(_kv(m, x0, ibg, n, monkeypatch, True), rec)     # and the seven-launch form replays bit-identically


def test_models_outside_the_limits_keep_the_generic_plan(monkeypatch):
    from sea_amd import kv_engine

    big = build(O.OracleConfig(1, 1024, 8, 32, 4, 0, 2, 2, True, "ln"), "bf16")        # E = 1024
    add = build(O.OracleConfig(1, 64, 4, 32, 4, 0, 2, 2, True, "adaln", "addition"), "fp32")
    assert not kv_engine.supported(big.engine(), 1) and not kv_engine.supported(add.engine(), 1)
    x, _, ib = recipe_inputs(1, 8, O.OracleConfig(1, 64, 4, 32, 4, 0, 2, 2, True, "adaln", "addition"), seed=1)
    out = _kv(add, x[:, :1].cuda().contiguous(), ib.cuda().contiguous(), 8, monkeypatch, True)
    assert len(add.engine()._kv_fast) == 0 and out.shape == (1, 8, 2, 64)


@pytest.mark.parametrize("cfg_args", [(1, 64, 4, 48, 8, 0, 3, 2, True, "adaln"), (2, 128, 8, 40, 4, 0, 2, 2, True, "ln"), (1, 64, 4, 48, 8, 0, 2, 2, False, "adaln")])
def test_generic_step_plan_hoists_the_condition_work(cfg_args, monkeypatch):
    """The generic KV-cache step plan (what the shipped widths run): with the condition-only work — AdaLN modulations, the info-bottleneck term —
    evaluated for all steps by one batched pass (engine.rollout_kv, SEA_KV=hoist) the rollout equals the per-step form (fp32 <= 1e-5: the batched pass
    runs the same arithmetic through the many-row GEMM kernels), carries no condition launch in its step plan, and the native step loop and the
    per-step Python loop patch the same pointers (bitwise equal)."""
    from sea_amd.utils.train_utils import rollout

    cfg = O.OracleConfig(*cfg_args)
    B, n = 2, 20
    x, _, ib = recipe_inputs(B, n, cfg, seed=21)
    x0, ibg = x[:, :1].cuda().contiguous(), ib.cuda().contiguous()
    ref = O.rollout(x[:, :1], ib, n, recipe_params(cfg), cfg)
    for dtype, tol, tol_paths in (("fp32", 1e-4, 1e-5), ("bf16", 3e-2, 2e-2)):
        m = build(cfg, dtype)
        monkeypatch.setenv("SEA_KV", "fast=0,hoist=1")
        a = rollout(m, x0, ibg, n, mode="kv")
        eng = m.engine()
        plans = [p for k, p in eng._plans.items() if len(k) == 4 and k[:3] == (B, 1, "step")]
        assert len(plans) == 1 and plans[0]._hoisted
        names = [r.name for r in plans[0].records]
        assert not any(nm.startswith("adaln.") for nm in names)
        if cfg.add_info_after_cross:
            assert "ib_add" not in names        # the info-bottleneck rows ride in the norm pass in front of the MLP
        assert rel_l2(a.cpu().numpy(), ref.numpy()) < tol
        monkeypatch.setenv("SEA_KV", "fast=0,hoist=1,loop=python")
        assert torch.equal(rollout(m, x0, ibg, n, mode="kv"), a)
        monkeypatch.setenv("SEA_KV", "fast=0,hoist=1")
        assert torch.equal(rollout(m, x0, ibg, n // 2, mode="kv")[:, :4], a[:, :4]) or dtype == "bf16"   # (a shorter rollout batches fewer rows: same kernels from 17 rows up)
        monkeypatch.setenv("SEA_KV", "fast=0,hoist=0")
        b = rollout(m, x0, ibg, n, mode="kv")
        assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < tol_paths
