"""The pointer audit (sea_amd/ptrcheck.py) on real plans: every forward / training / KV-step / condition plan passes it with the buffers it really uses,
a corrupted or stale address is refused BEFORE any launch, and a plan keeps the tensors it is bound to alive (a dropped output tensor used to be
known to the launch list by raw address only)."""
import gc

import pytest
import torch

from oracle import sea_oracle as O
from oracle.recipe import recipe_inputs
from tests.test_model_gpu import build

pytestmark = pytest.mark.gpu

CFG = O.OracleConfig(1, 128, 4, 64, 8, 0, 3, 2, True, "adaln")   # E = 128 / D = 64: the bf16 plan takes the fused launches (exchange tail, Linear + norm)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_every_plan_passes_the_audit_at_every_bind(dtype, monkeypatch):
    from sea_amd.utils.train_utils import initialize_optimizer, rollout

    monkeypatch.setenv("SEA_CHECK_PTRS", "1")
    m = build(CFG, dtype)
    x, tgt, ib = (t.cuda() for t in recipe_inputs(2, 40, CFG, seed=4))
    eng = m.engine()
    with torch.no_grad():
        for _ in range(3):
            m(x, ib)                                       # a fresh output tensor per call: re-bound, re-audited
    p = eng.plan(2, 40, "full")
    assert p._audited and p.audit() > 50
    m.train()
    opt = initialize_optimizer(m, {"learning_rate": 1e-3})
    for _ in range(2):
        eng.train_step(x, tgt, ib, opt)                    # forward + backward lists, dout bound by address
    tp = eng.train_plan(2, 40)
    assert tp._audited and tp.audit(owners=(tp._bound_dout,)) > 200
    m.eval()
    rollout(m, x[:, :1].contiguous(), ib, 12, mode="kv")                       # sea_kv_rollout: the condition plan
    monkeypatch.setenv("SEA_KV", "fast=0")
    rollout(m, x[:, :1].contiguous(), ib, 12, mode="kv")                       # the generic step plan, bound by raw address
    steps = [p for k, p in eng._plans.items() if k[:3] == (2, 1, "step")]
    assert steps and all(p._audited for p in steps)


def test_a_corrupted_pointer_is_refused_before_any_launch():
    m = build(CFG, "fp32")
    x, _, ib = (t.cuda() for t in recipe_inputs(1, 16, CFG, seed=5))
    eng = m.engine()
    with torch.no_grad():
        ref = m(x, ib).clone()
    p = eng.plan(1, 16, "full")
    rec = next(r for r in p.records if r.name == "self.out_proj")
    good = rec.args[0][0].W
    rec.args[0][0].W = good + (1 << 44)                    # far outside every allocation
    with pytest.raises(RuntimeError, match=r"pointer audit .*self\.out_proj.*SeaGemmGroup\.W"):
        p.audit()
    rec.args[0][0].W = good
    rec.args[0][0].M = 10 ** 6                             # the operand would run past its buffer
    with pytest.raises(RuntimeError, match=r"past the end of its buffer"):
        p.audit()
    rec.args[0][0].M = 16
    assert p.audit() > 0
    with torch.no_grad():
        assert torch.equal(m(x, ib), ref)


def test_a_corrupted_chain_pointer_is_refused():
    """The row-chain launches (sea_row_chain / sea_row_chain_riders, bf16 plans) are audited like every other record: the chain's operands with their extents
    (SeaRowChain), the projection entries (SeaQkvGroup) and the rider groups (SeaGemmGroup)."""
    cfg = O.OracleConfig(1, 128, 4, 64, 8, 0, 3, 2, True, "adaln")
    m = build(cfg, "bf16")
    x, _, ib = (t.cuda() for t in recipe_inputs(1, 40, cfg, seed=5))
    eng = m.engine()
    with torch.no_grad():
        ref = m(x, ib).clone()
    p = eng.plan(1, 40, "full")
    rec = next(r for r in p.records if r.name == "self.out_proj_down_qkv")
    ch = rec.args[0][0]
    good = ch.W2
    ch.W2 = good + (1 << 44)
    with pytest.raises(RuntimeError, match=r"pointer audit .*self\.out_proj_down_qkv.*SeaRowChain\.W2"):
        p.audit()
    ch.W2 = good
    ch.M = 10 ** 6
    with pytest.raises(RuntimeError, match=r"past the end of its buffer"):
        p.audit()
    ch.M = 40
    q = ch.proj[0]
    good_q = q.W
    q.W = good_q + (1 << 44)
    with pytest.raises(RuntimeError, match=r"pointer audit .*SeaQkvGroup\.W"):
        p.audit()
    q.W = good_q
    tail = next(r for r in p.records if r.name == "cross0.tail")
    if len(tail.args) > 5:   # a rider host: its SeaGemmGroup array
        g0 = tail.args[3][0]
        good_a = g0.A
        g0.A = good_a + (1 << 44)
        with pytest.raises(RuntimeError, match=r"pointer audit .*SeaGemmGroup\.A"):
            p.audit()
        g0.A = good_a
    assert p.audit() > 0
    with torch.no_grad():
        assert torch.equal(m(x, ib), ref)


def test_plan_holds_its_bound_output_until_the_next_bind():
    m = build(CFG, "bf16")
    x, _, ib = (t.cuda() for t in recipe_inputs(1, 24, CFG, seed=6))
    eng = m.engine()
    with torch.no_grad():
        out = eng.forward(x, ib)
    p = eng.plan(1, 24, "full")
    addr = out.data_ptr()
    ref = out.clone()
    del out
    gc.collect()
    torch.cuda.empty_cache()                               # a dropped output block would be unmapped here
    assert p._bound_tensors[2].data_ptr() == addr          # ... but the plan still owns it
    p.run()                                                # replay into the same buffer: no fault, same values
    torch.cuda.synchronize()
    assert torch.equal(p._bound_tensors[2], ref)
    with torch.no_grad():
        g1 = eng.forward_graphed(x, ib).clone()            # capture right behind plain replays (the stream is drained first)
    assert torch.equal(g1, ref)
