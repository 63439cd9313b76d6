"""Counter-based dropout (training mode): the kernels' masks are reproduced with sea_dropout_mask and fed to a plain fp32 PyTorch
restatement, so forward AND backward with dropout are checked exactly (the reference's torch.nn.Dropout stream itself cannot be
reproduced; parity with it is distributional: keep rate and 1/(1-p) scaling)."""
import math

import pytest
import torch

from oracle import sea_oracle as O
from oracle.recipe import recipe_inputs
from tests.test_bwd_ops_gpu import _rope, _rope_table, dev, gelu, rel, rnd

pytestmark = pytest.mark.gpu


def mask(rows, cols, seed, stream, thr):
    from sea_amd import _native as N

    out = torch.empty(rows, cols, device=dev())
    N.check(N.lib().sea_dropout_mask(out.data_ptr(), rows, cols, seed, stream, thr, N.stream_ptr()), "mask")
    return out


def test_mask_statistics():
    thr = 26  # p = 0.1 -> 26/256
    m = mask(4096, 1024, 123, 7, thr)
    keep = float((m > 0).float().mean())
    assert abs(keep - (1 - thr / 256)) < 2e-3
    assert torch.all((m == 0) | (m == 256.0 / (256 - thr)))
    assert abs(float(m.mean()) - 1.0) < 3e-3  # unbiased
    m2, m3 = mask(4096, 1024, 124, 7, thr), mask(4096, 1024, 123, 8, thr)
    for other in (m2, m3):  # different seed / stream: independent masks
        agree = float(((m > 0) == (other > 0)).float().mean())
        assert abs(agree - ((1 - thr / 256) ** 2 + (thr / 256) ** 2)) < 3e-3
    rows = (m > 0).float().mean(1)
    cols = (m > 0).float().mean(0)
    assert float(rows.std()) < 0.02 and float(cols.std()) < 0.01  # no row/column structure


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_epilogue_dropout(dtype):
    from sea_amd import _native as N, ops

    M, K, N_ = 200, 64, 96
    A, W, bias, R = rnd(M, K, dtype=dtype, seed=1), rnd(N_, K, dtype=dtype, seed=2), rnd(N_, seed=3), rnd(M, N_, seed=4)
    seed, stream, thr = 99, 5, 51
    mk = mask(M, N_, seed, stream, thr)
    pre = A.float() @ W.float().t() + bias
    for mode in (1, 2, 3):
        arr = (N.SeaGemmGroup * 1)()
        C32, Cact = torch.empty(M, N_, device=dev()), torch.empty(M, N_, device=dev(), dtype=dtype)
        ops.fill_gemm_group(arr[0], A, W, bias, R, C32, Cact)
        arr[0].drop.seed, arr[0].drop.stream, arr[0].drop.thr, arr[0].drop.mode = seed, stream, thr, mode
        N.check(N.lib().sea_gemm_grouped(arr, 1, N.dtype_code(dtype), N.stream_ptr()), "gemm")
        if mode == 1:
            assert rel(C32, pre * mk + R) < 2e-5
            assert rel(Cact.float(), pre * mk + R) < (2e-5 if dtype == torch.float32 else 6e-3)
        elif mode == 3:   # the complete value, residual included (PositionalEncoding's dropout(x + pe)), in both outputs
            assert rel(C32, (pre + R) * mk) < 2e-5
            assert rel(Cact.float(), (pre + R) * mk) < (2e-5 if dtype == torch.float32 else 6e-3)
        else:
            assert rel(C32, pre + R) < 2e-5
            assert rel(Cact.float(), (pre + R) * mk) < (2e-5 if dtype == torch.float32 else 6e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("hd,T", [(16, 70), (32, 130)])
def test_attention_dropout_forward_backward(dtype, hd, T):
    from sea_amd import _native as N, ops
    import ctypes as C

    B, H, nprob = 2, 3, 2
    E, cap, scale = H * hd, (T + 7) // 8 * 8, ops.q_scale(hd)   # hd^-1/2 * log2(e): the attention kernels score in log2 units
    seed, stream0, thr = 4242, 11, 64  # p = 0.25
    table = _rope_table(hd, cap)
    probs, bprobs, refs = [], [], []
    for pi in range(nprob):
        q0, k0, v0 = (rnd(B, T, H, hd, seed=10 * pi + s).requires_grad_(True) for s in (1, 2, 3))
        dO = rnd(B, T, E, dtype=dtype, seed=10 * pi + 4)
        Q = (_rope(q0, table) * scale).permute(0, 2, 1, 3).contiguous().to(dtype).detach()
        K = torch.zeros(B, H, cap, hd, device=dev(), dtype=dtype)
        V = torch.zeros(B, H, cap, hd, device=dev(), dtype=dtype)
        K[:, :, :T] = _rope(k0, table).permute(0, 2, 1, 3).to(dtype).detach()
        V[:, :, :T] = v0.permute(0, 2, 1, 3).to(dtype).detach()
        Ot, LSE = torch.empty(B, T, E, device=dev(), dtype=dtype), torch.empty(B, H, T, device=dev())
        probs.append(dict(Q=Q, K=K, Vt=V.transpose(2, 3).contiguous(), O=Ot, LSE=LSE))
        dQ, dKV = torch.empty(B * T, E, device=dev(), dtype=dtype), torch.empty(B * T, 2 * E, device=dev(), dtype=dtype)
        bprobs.append(dict(Q=Q, K=K, V=V, O=Ot, dO=dO, LSE=LSE, delta=torch.empty(B, H, T, device=dev()), dQ=dQ, dK=dKV[:, :E], dV=dKV[:, E:]))
        refs.append((q0, k0, v0, dO))
    # forward with dropout (struct filled by hand to set the dropout fields)
    P = N.SeaAttnParams()
    P.n_problems = nprob
    for i, d in enumerate(probs):
        P.p[i].Q, P.p[i].K, P.p[i].Vt, P.p[i].O, P.p[i].LSE = (d[k].data_ptr() for k in ("Q", "K", "Vt", "O", "LSE"))
    P.B, P.H, P.hd, P.Tq, P.Tk, P.cap, P.q_pos0, P.src_len, P.ldo = B, H, hd, T, T, cap, 0, 0, E
    P.drop.seed, P.drop.stream, P.drop.thr = seed, stream0, thr
    N.check(N.lib().sea_attention_fwd(C.byref(P), N.dtype_code(dtype), N.stream_ptr()), "attn fwd")
    PB = N.SeaAttnBwdParams()
    PB.n_problems = nprob
    for i, d in enumerate(bprobs):
        q = PB.p[i]
        q.Q, q.K, q.V, q.O, q.dO, q.LSE, q.delta, q.dQ, q.dK, q.dV = (d[k].data_ptr() for k in ("Q", "K", "V", "O", "dO", "LSE", "delta", "dQ", "dK", "dV"))
    PB.rope = table.data_ptr()
    PB.B, PB.H, PB.hd, PB.Tq, PB.Tk, PB.cap, PB.q_pos0, PB.src_len = B, H, hd, T, T, cap, 0, 0
    PB.ldo, PB.lddo, PB.lddq, PB.lddk, PB.lddv, PB.q_scale = E, E, E, 2 * E, 2 * E, scale
    PB.drop.seed, PB.drop.stream, PB.drop.thr = seed, stream0, thr
    N.check(N.lib().sea_attention_bwd(C.byref(PB), N.dtype_code(dtype), N.stream_ptr()), "attn bwd")
    t = 2e-5 if dtype == torch.float32 else 2.5e-2
    for pi, (q0, k0, v0, dO) in enumerate(refs):
        # the kernels' mask for (problem pi, b, h) is stream (stream0 + pi) * B*H + b*H + h over the (query, key) grid
        mk = torch.stack([torch.stack([mask(T, T, seed, (stream0 + pi) * B * H + b * H + h, thr) for h in range(H)]) for b in range(B)])
        qr = (_rope(q0, table) * scale).permute(0, 2, 1, 3)
        kr = _rope(k0, table).permute(0, 2, 1, 3)
        vr = v0.permute(0, 2, 1, 3)
        if dtype != torch.float32:
            qr, kr, vr = (t_ + (t_.detach().to(dtype).float() - t_.detach()) for t_ in (qr, kr, vr))
        S = (qr @ kr.transpose(-1, -2)) * 0.6931471805599453
        i = torch.arange(T, device=dev())
        S = S.masked_fill(i[None, :] > i[:, None], float("-inf"))
        Oref = ((torch.softmax(S, -1) * mk) @ vr).transpose(1, 2).reshape(B, T, E)
        Oref.backward(dO.float())
        assert rel(probs[pi]["O"].float(), Oref.detach()) < (2e-5 if dtype == torch.float32 else 1e-2)
        assert rel(bprobs[pi]["dQ"].float().view(B, T, H, hd), q0.grad) < t
        assert rel(bprobs[pi]["dK"].float().reshape(B, T, H, hd), k0.grad) < t
        assert rel(bprobs[pi]["dV"].float().reshape(B, T, H, hd), v0.grad) < t


def test_ib_dropout_forward_backward():
    from sea_amd import _native as N
    import ctypes as C

    M, E, h, F = 150, 256, 8, 3
    seed, stream0, thr = 77, 3, 26
    c = torch.rand(M, device=dev())
    ps = [rnd(h, seed=61), rnd(h, seed=62), 1 + 0.1 * rnd(h, seed=63), 0.1 * rnd(h, seed=64), rnd(E, h, seed=65), rnd(E, seed=66)]
    xs = [rnd(M, E, seed=70 + f) for f in range(F)]
    x0 = [x.clone() for x in xs]
    P = N.SeaIbParams()
    for f in range(F):
        P.X[f] = xs[f].data_ptr()
    P.n_fields, P.ldx, P.c = F, E, c.data_ptr()
    P.w1, P.b1, P.lnw, P.lnb, P.w2, P.b2 = (t.data_ptr() for t in ps)
    P.M, P.E, P.h = M, E, h
    P.drop.seed, P.drop.stream, P.drop.thr = seed, stream0, thr
    N.check(N.lib().sea_ib_add(C.byref(P), N.stream_ptr()), "ib")
    rs = [p.clone().requires_grad_(True) for p in ps]
    ib = gelu(torch.nn.functional.layer_norm(c[:, None] * rs[0][None, :] + rs[1], (h,), rs[2], rs[3], 1e-5)) @ rs[4].t() + rs[5]
    masks = [mask(M, E, seed, stream0 + f, thr) for f in range(F)]
    for f in range(F):
        assert rel(xs[f], x0[f] + ib.detach() * masks[f]) < 1e-6
    dxs = [rnd(M, E, seed=80 + f) for f in range(F)]
    sum(((ib * masks[f]) * dxs[f]).sum() for f in range(F)).backward()
    grads = [torch.zeros_like(p) for p in ps]
    B_ = N.SeaIbBwdParams()
    for f in range(F):
        B_.dX[f] = dxs[f].data_ptr()
    B_.n_fields, B_.ldx, B_.c = F, E, c.data_ptr()
    B_.w1, B_.b1, B_.lnw, B_.lnb, B_.w2 = (t.data_ptr() for t in ps[:5])
    B_.dw1, B_.db1, B_.dlnw, B_.dlnb, B_.dw2, B_.db2 = (t.data_ptr() for t in grads)
    B_.M, B_.E, B_.h = M, E, h
    B_.drop.seed, B_.drop.stream, B_.drop.thr = seed, stream0, thr
    N.check(N.lib().sea_ib_bwd(C.byref(B_), N.stream_ptr()), "ib bwd")
    for g, r, name in zip(grads, rs, ["w1", "b1", "lnw", "lnb", "w2", "b2"]):
        assert rel(g, r.grad) < 5e-5, name


@pytest.mark.parametrize("variant", [("sea", "add", True, "adaln", 3), ("pool", "add", True, "ln", 2), ("pool", "add", True, "adaln", 3),
                                     ("sea", "attention", True, "adaln", 2), ("sea", "attention", False, "ln", 3)])
def test_model_training_with_dropout_is_consistent(variant):
    """Whole model, dropout 0.1 (the shipped cylinder setting), fp32: with the step's seed held fixed the loss is a deterministic smooth
    function of the parameters, so the hand-written backward must match a central finite difference along a random direction — this
    checks that every forward mask is regenerated identically in the backward.  Also: eval() ignores dropout; train() re-keys per step.
    Variants: the shipped structure; exchange_mode 'pool', whose position-encoded rows are dropped too (PositionalEncoding, models/base_blocks.py:370-372:
    mode 3 of the GEMM epilogue, the same mask on the rows' gradient); ib_addition_mode 'attention', where the reference evaluates the info-bottleneck
    MLP — and its dropout — once per field (models/temporal.py:110-118: F row sets with their own masks) and drops the attention probabilities."""
    from sea_amd.models.temporal import TemporalModel
    from oracle.recipe import recipe_params

    xm, ibm, after, ln, F = variant
    cfg = O.OracleConfig(1, 64, 4, 48, 8, 0, F, 2, after, ln, xm, ibm)
    m = TemporalModel(1, 64, 4, 48, 8, 0, F, 2, 0.1, xm, "learnable", "mlp", ibm, 1, 1, after, ln)
    p = recipe_params(cfg)
    with torch.no_grad():
        for k, prm in m.named_parameters():
            prm.copy_(p[k])
    m = m.to("cuda:0")
    x, tgt, ib = (t.cuda() for t in recipe_inputs(2, 40, cfg, seed=3))
    eng = m.engine()
    m.eval()
    with torch.no_grad():
        e1, e2 = m(x, ib), m(x, ib)
    assert torch.equal(e1, e2)
    m.train()
    with torch.no_grad():
        t1, t2 = m(x, ib).clone(), m(x, ib).clone()
    assert not torch.equal(t1, t2) and not torch.equal(t1, e1)  # a new mask every step
    assert rel(t1, e1) < 0.5  # same function up to the dropout noise

    def loss_at(delta_scale, direction):
        eng._drop_step = 1000  # hold the seed: forward_train increments it to 1001 every time
        with torch.no_grad():
            eng.params.flat32[: eng.params.n_live].add_(direction, alpha=delta_scale)
            out, plan = eng.forward_train(x, ib)
            loss, dout = eng.mse_loss_and_grad(out, tgt)
            eng.params.flat32[: eng.params.n_live].add_(direction, alpha=-delta_scale)
        return float(loss.double()), plan, dout

    torch.manual_seed(0)
    direction = torch.randn(eng.params.n_live, device="cuda:0")
    direction /= direction.norm()
    l0, plan, dout = loss_at(0.0, direction)
    eng.zero_grads()
    eng.backward(plan, dout)
    analytic = float((eng.grads[: eng.params.n_live].double() * direction.double()).sum())
    eps = 2e-2
    lp, _, _ = loss_at(eps, direction)
    lm, _, _ = loss_at(-eps, direction)
    numeric = (lp - lm) / (2 * eps)
    assert abs(analytic - numeric) < 3e-2 * max(abs(numeric), 1e-3), (analytic, numeric)


def test_standalone_module_forwards_with_dropout():
    """The stand-alone module mirrors in train() with dropout > 0 (cylinder ships 0.1): the MLP output is zeroed with probability thr / 256 and the kept
    values are the eval-mode values scaled by 256 / (256 - thr) — exactly, element by element (nn.Dropout's keep / scale semantics); the attention
    modules drop probabilities (fresh masks per call, mean over calls -> the eval output); eval() is untouched."""
    from sea_amd.models.base_blocks import MLP, MaskedMultiHeadAttention, MaskedMultiHeadCrossAttention

    torch.manual_seed(11)
    p = 0.25
    thr = round(256 * p)
    mlp = MLP(64, p, 8).to(dev())
    x = rnd(3, 40, 64, seed=5)
    mlp.eval()
    with torch.no_grad():
        y_eval = mlp(x)
        mlp.train()
        y1, y2 = mlp(x), mlp(x)
    zero = y1 == 0
    assert abs(float(zero.float().mean()) - thr / 256) < 0.03
    assert rel(y1[~zero], y_eval[~zero] * (256.0 / (256 - thr))) < 1e-5
    assert not torch.equal(y1, y2)                      # fresh masks per call
    att = MaskedMultiHeadAttention(4, 64, 48, 0, p).to(dev())
    cross = MaskedMultiHeadCrossAttention(4, 64, 48, 0, p).to(dev())
    for mod, args in ((att, (x,)), (cross, (x, rnd(3, 40, 64, seed=6)))):
        mod.eval()
        with torch.no_grad():
            ref = mod(*args)
            mod.train()
            outs = [mod(*args) for _ in range(48)]
        assert all(torch.isfinite(o).all() for o in outs) and not torch.equal(outs[0], outs[1])
        mean = torch.stack(outs).mean(0)
        assert rel(mean[:, 8:], ref[:, 8:]) < 0.12     # unbiased (rows with several keys; 48 draws)
        mod.eval()
        with torch.no_grad():
            assert torch.equal(mod(*args), ref)


def test_pool_position_rows_are_dropped_like_the_reference():
    """exchange_mode='pool' in train(): the rows the pool MLP, the queries and the GELU sum read are dropout(ln_cross(cross_down(x)) + pe) (PositionalEncoding,
    models/base_blocks.py:370-372) — a fraction thr / 256 of their elements is exactly zero, the others are the eval rows scaled by 256 / (256 - thr)."""
    from sea_amd.models.temporal import TemporalModel

    m = TemporalModel(1, 64, 4, 48, 8, 0, 2, 2, 0.25, "pool", "learnable", "mlp", "add", 1, 1, True, "ln").to("cuda:0")
    cfg = O.OracleConfig(1, 64, 4, 48, 8, 0, 2, 2, True, "ln", "pool")
    x, _, ib = (t.cuda() for t in recipe_inputs(2, 20, cfg, seed=4))
    eng = m.engine()
    m.train()
    out, plan = eng.forward_train(x, ib)
    assert torch.isfinite(out).all()
    torch.cuda.synchronize()
    big = plan.saved[0]["big"][:, : 2 * 32].float()      # [n_0 | n_1], D = 32
    zero = big == 0
    assert abs(float(zero.float().mean()) - 64 / 256) < 0.03
    m.eval()
    with torch.no_grad():
        assert torch.isfinite(m(x, ib)).all()
