"""sea_gemm_fewrows / sea_qkv_rope_fewrows (sea_amd/csrc/gemv.hip): the Linear layers of a KV-cache rollout step at the shipped widths (one row per
trajectory and field), with the row norm in front of the layer evaluated as a prologue.
Checked against plain fp32 torch on the bf16-rounded operands, against the launches they replace (sea_rownorm -> sea_gemm_grouped, sea_qkv_rope_grouped) and — a step plan built from them — against the generic step plan and the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def _r(*shape, scale=1.0, dtype=torch.float32, seed=[0]):
    seed[0] += 1
    g = torch.Generator().manual_seed(1000 + seed[0])
    return (torch.randn(*shape, generator=g) * scale).cuda().to(dtype)


def _ln_ref(x, gamma, beta=None, mod=None, eps=1e-5):
    d = x.shape[-1]
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).mean(-1, keepdim=True)
    xh = (x - mean) / torch.sqrt(var + eps)
    g = gamma + (1.0 + mod[:, :d].float() if mod is not None else 0.0)
    b = (beta if beta is not None else 0.0) + (mod[:, d:].float() if mod is not None else 0.0)
    return xh * g + b


@pytest.mark.parametrize("M,N,K", [(1, 512, 512), (2, 1024, 1024), (1, 3072, 1024), (4, 1000, 2048), (3, 516, 4096), (1, 1024, 8192), (2, 260, 16384)])
@pytest.mark.parametrize("act", [0, 1])
def test_gemm_fewrows_matches_torch_and_the_tiled_launch(M, N, K, act):
    from sea_amd import ops

    groups, refs = [], []
    for gi in range(2):
        A, W = _r(M, K, dtype=BF), _r(N, K, scale=K ** -0.5, dtype=BF)
        bias, R = _r(N), _r(M, N)
        C32, Cact, Z = torch.zeros(M, N, device="cuda"), torch.zeros(M, N, device="cuda", dtype=BF), torch.zeros(M, N, device="cuda", dtype=BF)
        d = dict(A=A, W=W, bias=bias, R=R, C32=C32, Cact=Cact, act=act, bias_scale=(2.0 if gi else 1.0))
        if act:
            d["Z"] = Z
        groups.append(d)
        v = A.float() @ W.float().t() + bias * d["bias_scale"]
        pre_act = v.clone()
        if act:
            v = torch.nn.functional.gelu(v)
        refs.append((v + R, pre_act))
    ops.gemm_fewrows(groups, BF)
    torch.cuda.synchronize()
    for d, (ref, pre_act) in zip(groups, refs):
        assert torch.allclose(d["C32"], ref, rtol=1e-4, atol=2e-4), (d["C32"] - ref).abs().max()
        assert torch.allclose(d["Cact"].float(), ref, rtol=1e-2, atol=1e-2)
        if act:
            assert torch.allclose(d["Z"].float(), pre_act, rtol=1e-2, atol=1e-2)
    # the launch it replaces on the same operands (MFMA, another summation order)
    tiled = [dict(d, C32=torch.zeros_like(d["C32"]), Cact=torch.zeros_like(d["Cact"])) for d in groups]
    ops.gemm_grouped(tiled, BF)
    for d, t in zip(groups, tiled):
        assert torch.allclose(d["C32"], t["C32"], rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("M,K,adaln,addend", [(1, 1024, True, True), (2, 2048, False, True), (4, 512, True, False), (1, 2048, False, False)])
def test_gemm_fewrows_norm_prologue(M, K, adaln, addend):
    """pre: the A operand is sea_rownorm of fp32 rows (AdaLN or LayerNorm, optionally x + addend with x + addend written out once)."""
    from sea_amd import ops

    N = 1536
    groups, pres, refs, chain = [], [], [], []
    for gi in range(2):
        X, W, bias = _r(M, K, scale=1.5) + 0.3, _r(N, K, scale=K ** -0.5, dtype=BF), _r(N)
        gamma, beta = 1.0 + 0.1 * _r(K), 0.1 * _r(K)
        mod = _r(M, 2 * K, scale=0.2, dtype=BF) if adaln else None
        add = _r(M, K, scale=0.5) if addend else None
        xout = torch.zeros(M, K, device="cuda") if addend else None
        C32 = torch.zeros(M, N, device="cuda")
        groups.append(dict(W=W, bias=bias, C32=C32))
        sp = dict(X=X, gamma=gamma, beta=(beta if adaln else None), mod=mod)
        if addend:
            sp.update(addend=add, Xout=xout)
        pres.append(sp)
        xs = X + add if addend else X
        a_ref = _ln_ref(xs, gamma, beta if adaln else None, mod).to(BF).float()
        refs.append((a_ref @ W.float().t() + bias, xs))
        # the two launches it replaces
        n_e, x2, c2 = torch.zeros(M, K, device="cuda", dtype=BF), (torch.zeros(M, K, device="cuda") if addend else None), torch.zeros(M, N, device="cuda")
        g = dict(X=X, gamma=gamma, beta=(beta if adaln else None), mod=mod, Yact=n_e)
        if addend:
            g.update(addend=add, Xout=x2)
        ops.rownorm([g], M, K, False, False, 1e-5, BF)
        ops.gemm_grouped([dict(A=n_e, W=W, bias=bias, C32=c2)], BF)
        chain.append(c2)
    ops.gemm_fewrows(groups, BF, pre=pres)
    torch.cuda.synchronize()
    for d, sp, (ref, xs), c2 in zip(groups, pres, refs, chain):
        assert torch.allclose(d["C32"], ref, rtol=2e-3, atol=4e-3), (d["C32"] - ref).abs().max()   # a bf16 ulp of a normalised operand here and there (summation order of the statistics)
        assert torch.allclose(d["C32"], c2, rtol=2e-3, atol=4e-3)
        if addend:
            assert torch.equal(sp["Xout"], xs)


@pytest.mark.parametrize("M,N,K,gelu", [(1, 1024, 8192, True), (2, 512, 4096, True), (4, 1024, 2048, False), (3, 260, 8192, True)])
def test_gemm_fewrows_layernorm_gelu_prologue_on_activation_rows(M, N, K, gelu):
    """pre_x_is_act: nn.LayerNorm (+ GELU) of hidden rows held in bf16 (padded row stride) as the prologue of the Linear that consumes them."""
    from sea_amd import ops

    groups, pres, refs, chain = [], [], [], []
    for gi in range(2):
        Hb = torch.zeros(M, K + 64, device="cuda", dtype=BF)[:, :K]
        Hb.copy_(_r(M, K, scale=1.2) + 0.2)
        W, bias, R = _r(N, K, scale=K ** -0.5, dtype=BF), _r(N), _r(M, N)
        lnw, lnb = 1.0 + 0.1 * _r(K), 0.1 * _r(K)
        Cact = torch.zeros(M, N, device="cuda", dtype=BF)
        groups.append(dict(W=W, bias=bias, R=R, Cact=Cact))
        pres.append(dict(X=Hb, gamma=lnw, beta=lnb))
        a = _ln_ref(Hb.float(), lnw, lnb)
        a = (torch.nn.functional.gelu(a) if gelu else a).to(BF).float()
        refs.append(a @ W.float().t() + bias + R)
        hg, c2 = torch.zeros(M, K, device="cuda", dtype=BF), torch.zeros(M, N, device="cuda", dtype=BF)
        ops.rownorm([dict(X=Hb, gamma=lnw, beta=lnb, Yact=hg)], M, K, True, gelu, 1e-5, BF)
        ops.gemm_grouped([dict(A=hg, W=W, bias=bias, R=R, Cact=c2)], BF)
        chain.append(c2)
    ops.gemm_fewrows(groups, BF, pre=pres, pre_x_is_act=True, pre_gelu=gelu)
    torch.cuda.synchronize()
    for d, ref, c2 in zip(groups, refs, chain):
        assert torch.allclose(d["Cact"].float(), ref, rtol=2e-2, atol=2e-2), (d["Cact"].float() - ref).abs().max()
        assert torch.allclose(d["Cact"].float(), c2.float(), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("M,K,H,hd,pre", [(1, 1024, 8, 128, True), (2, 2048, 8, 256, False), (4, 512, 8, 64, True), (1, 512, 8, 64, False)])
def test_qkv_rope_fewrows_matches_the_tiled_launch(M, K, H, hd, pre):
    """Self form (one group, [q | k | v]) and cross form (a q group and a k,v group with another operand): same cache rows as sea_qkv_rope_grouped."""
    from sea_amd import ops

    Ea, cap, pos0 = H * hd, 16, 5
    rope = torch.randn(cap, hd // 2, 2).cuda()
    rope = rope / rope.norm(dim=-1, keepdim=True)
    B = M

    def outs():
        return dict(Q=torch.zeros(B, H, 1, hd, device="cuda", dtype=BF), K=torch.zeros(B, H, cap, hd, device="cuda", dtype=BF),
                    Vt=torch.zeros(B, H, hd, cap, device="cuda", dtype=BF))

    for form in ("self", "cross"):
        X1, X2 = _r(M, K, scale=1.3), _r(M, K, scale=0.7) + 0.2
        gamma = 1.0 + 0.1 * _r(K)
        n1, n2 = torch.zeros(M, K, device="cuda", dtype=BF), torch.zeros(M, K, device="cuda", dtype=BF)
        ops.rownorm([dict(X=X1, gamma=gamma, Yact=n1), dict(X=X2, gamma=gamma, Yact=n2)], M, K, False, False, 1e-5, BF)
        if form == "self":
            W, bias = _r(3 * Ea, K, scale=K ** -0.5, dtype=BF), _r(3 * Ea)
            o_ref, o_new = outs(), outs()
            ref_groups = [dict(A=n1, W=W, bias=bias, col0=0, **o_ref)]
            new_groups = [dict(A=(None if pre else n1), W=W, bias=bias, col0=0, **o_new)]
            specs = [dict(X=X1, gamma=gamma)] if pre else None
        else:
            Wq, bq, Wkv, bkv = _r(Ea, K, scale=K ** -0.5, dtype=BF), _r(Ea), _r(2 * Ea, K, scale=K ** -0.5, dtype=BF), _r(2 * Ea)
            o_ref, o_new = outs(), outs()
            ref_groups = [dict(A=n1, W=Wq, bias=bq, col0=0, Q=o_ref["Q"]), dict(A=n2, W=Wkv, bias=bkv, col0=Ea, K=o_ref["K"], Vt=o_ref["Vt"])]
            new_groups = [dict(A=(None if pre else n1), W=Wq, bias=bq, col0=0, Q=o_new["Q"]),
                          dict(A=(None if pre else n2), W=Wkv, bias=bkv, col0=Ea, K=o_new["K"], Vt=o_new["Vt"])]
            specs = [dict(X=X1, gamma=gamma), dict(X=X2, gamma=gamma)] if pre else None
        ops.qkv_rope_grouped(ref_groups, rope, H, hd, 1, pos0, cap, ops.q_scale(hd), BF)
        ops.qkv_rope_fewrows(new_groups, rope, H, hd, 1, pos0, cap, ops.q_scale(hd), BF, pre=specs)
        torch.cuda.synchronize()
        for k in ("Q", "K", "Vt"):
            a, b = o_new[k].float(), o_ref[k].float()
            assert torch.allclose(a, b, rtol=2e-2, atol=2e-2), (form, k, (a - b).abs().max())
            assert b.abs().sum() > 0
        # nothing but position pos0 of the caches is written
        mask = torch.ones(cap, dtype=torch.bool)
        mask[pos0] = False
        assert float(o_new["K"][:, :, mask.cuda()].abs().sum()) == 0 and float(o_new["Vt"][:, :, :, mask.cuda()].abs().sum()) == 0


@pytest.mark.parametrize("cfg_args,B", [((1, 1024, 8, 40, 8, 0, 2, 2, True, "adaln"), 1), ((1, 1024, 8, 40, 8, 0, 3, 2, True, "ln"), 2),
                                        ((1, 1024, 8, 40, 8, 0, 2, 2, False, "adaln"), 4), ((2, 1024, 8, 24, 8, 0, 1, 2, True, "adaln"), 1)])
def test_step_plan_of_fewrow_launches_equals_generic_step_plan_and_oracle(cfg_args, B, monkeypatch):
    """The KV-cache step plan at a shipped width (embed_dim 1024: cylinder_flow's; 2 and 3 field groups, AdaLN / LayerNorm, the info-bottleneck add behind and in
    front of the exchange, two layers, 1-4 trajectories) built from the few-row launches: 17 launches per layer instead of 22 (no launch of its own for AdaLN_0,
    ln_cross, the info-bottleneck add + AdaLN_2 or the LayerNorm + GELU of the hidden rows), the rollout of the generic
    step plan within bf16 rounding and the CPU oracle's recompute rollout within the bf16 rollout tolerance."""
    from oracle import sea_oracle as O
    from oracle.recipe import recipe_inputs, recipe_params
    from sea_amd.utils.train_utils import rollout
    from tests.test_model_gpu import build, rel_l2

    cfg = O.OracleConfig(*cfg_args)
    n = 12
    x, _, ib = recipe_inputs(B, n, cfg, seed=5)
    x0, ibg = x[:, :1].cuda().contiguous(), ib.cuda().contiguous()
    with torch.no_grad():
        ref = O.rollout(x[:, :1], ib, n, recipe_params(cfg), cfg).numpy()
    m = build(cfg, "bf16")
    eng = m.engine()
    monkeypatch.setenv("SEA_KV", "gemv=1")
    a = rollout(m, x0, ibg, n, mode="kv")
    plans = [p for k, p in eng._plans.items() if k[:3] == (B, 1, "step")]
    assert len(plans) == 1 and plans[0]._few
    names = [r.name for r in plans[0].records]
    assert not any("adaln" in nm or "norm_old" in nm or "norm_new" in nm or nm == "mlp.ln_gelu" for nm in names), names
    assert any(r.fn is eng_lib().sea_gemm_fewrows for r in plans[0].records) and any(r.fn is eng_lib().sea_qkv_rope_fewrows for r in plans[0].records)
    n_few = len(names)
    monkeypatch.setenv("SEA_KV", "gemv=1,loop=python")
    assert torch.equal(rollout(m, x0, ibg, n, mode="kv"), a)       # the native step loop patches what the Python loop binds
    eng._plans.clear()
    monkeypatch.setenv("SEA_KV", "gemv=0")
    b = rollout(m, x0, ibg, n, mode="kv")
    plans = [p for k, p in eng._plans.items() if k[:3] == (B, 1, "step")]
    assert len(plans) == 1 and not plans[0]._few and len(plans[0].records) > n_few
    e_ab, e_a, e_b = rel_l2(a.cpu().numpy(), b.cpu().numpy()), rel_l2(a.cpu().numpy(), ref), rel_l2(b.cpu().numpy(), ref)
    print(f"few-row step plan {n_few} launches vs generic {len(plans[0].records)}: rel-L2 between them {e_ab:.3e}; vs oracle {e_a:.3e} / {e_b:.3e}")
    assert e_ab < 2e-2 and e_a < 3e-2 and e_b < 3e-2


@pytest.mark.parametrize("cfg_args,B", [((1, 1024, 8, 32, 8, 0, 2, 2, True, "adaln"), 3), ((1, 1024, 8, 32, 8, 0, 2, 2, True, "ln"), 1)])
def test_fewrow_step_plan_without_the_hoisted_condition_work(cfg_args, B, monkeypatch):
    """SEA_KV=hoist=0: the AdaLN modulations / info-bottleneck rows are evaluated inside every step (silu + condition GEMM launches, the add as a launch of its
    own for LayerNorm models) and the few-row launches read them from the step plan's own buffers — same rollout as with the hoisted buffers (bf16 rounding of
    the batched against the per-step condition GEMM); 3 trajectories take the 4-row instantiations."""
    from oracle import sea_oracle as O
    from oracle.recipe import recipe_inputs
    from sea_amd.utils.train_utils import rollout
    from tests.test_model_gpu import build, rel_l2

    cfg = O.OracleConfig(*cfg_args)
    n = 10
    x, _, ib = recipe_inputs(B, n, cfg, seed=8)
    x0, ibg = x[:, :1].cuda().contiguous(), ib.cuda().contiguous()
    m = build(cfg, "bf16")
    eng = m.engine()
    monkeypatch.setenv("SEA_KV", "hoist=1")
    a = rollout(m, x0, ibg, n, mode="kv")
    eng._plans.clear()
    monkeypatch.setenv("SEA_KV", "hoist=0")
    b = rollout(m, x0, ibg, n, mode="kv")
    plans = [p for k, p in eng._plans.items() if k[:3] == (B, 1, "step")]
    assert len(plans) == 1 and plans[0]._few and not plans[0]._hoisted
    assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 2e-2


def eng_lib():
    from sea_amd import _native as N

    return N.lib()
