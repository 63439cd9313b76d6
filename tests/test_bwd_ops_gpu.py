"""Backward kernels (through the C ABI) against torch.autograd on the same fp32 formulas, on the MI355X."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DTYPES = [torch.float32, torch.bfloat16]


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def rnd(*shape, dtype=torch.float32, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev()).to(dtype)


def gelu(x):
    return 0.5 * x * (1 + torch.erf(x / math.sqrt(2)))


def tol(dtype, f32=2e-5, bf16=1e-2):
    return f32 if dtype == torch.float32 else bf16


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N_,K", [(2024, 256, 256), (300, 64, 128), (1000, 2048, 256), (77, 16, 8), (4096, 128, 2048)])
def test_wgrad(dtype, M, N_, K):
    from sea_amd import ops

    dYb = rnd(M, N_ + 16, dtype=dtype, seed=1)
    dY = dYb[:, 8:8 + N_]  # strided
    X = rnd(M, K, dtype=dtype, seed=2)
    dW = rnd(N_, K, seed=3)  # accumulated into existing content
    db = rnd(N_, seed=4)
    dW0, db0 = dW.clone(), db.clone()
    ops.wgrad_grouped([dict(dY=dY, X=X, dW=dW, db=db)], dtype)
    ref_w = dW0 + dY.float().t() @ X.float()
    ref_b = db0 + dY.float().sum(0)
    assert rel(dW, ref_w) < 2e-5  # fp32 accumulation of exactly representable products in both modes
    assert rel(db, ref_b) < 2e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shapes", [[(2048, 256)] * 3, [(256, 2048)] * 3, [(512, 512)] * 5 + [(256, 256)] * 2, [(768, 256)] * 3 + [(1024, 512)]])
def test_wgrad_cfg3_launch_shapes(dtype, shapes):
    """The grouped launches of the cfg3 backward (MLP, condition, QKV matrices: the 128 tile, XCD-contiguous work order, whole-round splits) at a ragged row
    count: the plain path of interior stages and the clamped path of the last one.  Strided operands, bias sums on the matrix core, accumulation into existing content."""
    from sea_amd import ops

    M = 2048 + 333
    gs, refs = [], []
    for i, (n, k) in enumerate(shapes):
        dY = rnd(M, n + 64, dtype=dtype, seed=100 + i)[:, :n]
        X = rnd(M, k + 64, dtype=dtype, seed=200 + i)[:, 64:]
        dW, db = rnd(n, k, seed=300 + i), rnd(n, seed=400 + i)
        refs.append((dW + dY.float().t() @ X.float(), db + dY.float().sum(0)))
        gs.append(dict(dY=dY, X=X, dW=dW, db=db))
    ops.wgrad_grouped(gs, dtype)
    for g, (rw, rb) in zip(gs, refs):
        assert rel(g["dW"], rw) < 2e-5
        assert rel(g["db"], rb) < 2e-5


@pytest.mark.parametrize("dtype", DTYPES)
def test_wgrad_grouped_shared_output(dtype):
    """Several groups accumulating into the same dW (the multi-segment cross_up case)."""
    from sea_amd import ops

    M, N_, K = 500, 256, 128
    dY = rnd(M, N_, dtype=dtype, seed=5)
    Xs = [rnd(M, K, dtype=dtype, seed=6 + i) for i in range(2)]
    dW = torch.zeros(N_, K, device=dev())
    db = torch.zeros(N_, device=dev())
    ops.wgrad_grouped([dict(dY=dY, X=Xs[0], dW=dW, db=db), dict(dY=dY, X=Xs[1], dW=dW)], dtype)
    assert rel(dW, dY.float().t() @ (Xs[0].float() + Xs[1].float())) < 2e-5
    assert rel(db, dY.float().sum(0)) < 2e-5


def test_transpose_weights():
    from sea_amd import ops

    shapes = [(64, 32), (33, 70), (256, 128), (8, 1)]
    n = sum(a * b for a, b in shapes)
    src = rnd(n, seed=9)
    offs, o = [], 0
    for a, b in shapes:
        offs.append(o)
        o += a * b
    desc = torch.tensor([[offs[i], offs[i], a, b] for i, (a, b) in enumerate(shapes)], dtype=torch.int64, device=dev())
    tiles = [((a + 31) // 32) * ((b + 31) // 32) for a, b in shapes]
    ts = torch.tensor([0] + list(torch.tensor(tiles).cumsum(0)), dtype=torch.int32, device=dev())
    for dt in DTYPES:
        dst = torch.zeros(n, device=dev(), dtype=dt)
        ops.transpose_weights(src, dst, desc, ts)
        for i, (a, b) in enumerate(shapes):
            ref = src[offs[i]:offs[i] + a * b].view(a, b).t().contiguous().to(dt)
            assert torch.equal(dst[offs[i]:offs[i] + a * b].view(b, a), ref)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("d", [64, 256])
def test_rownorm_bwd_adaln(dtype, d):
    from sea_amd import ops

    M = 301
    x = (rnd(M, d, seed=10) * 1.5 + 0.2)
    mod = rnd(M, 2 * d, dtype=dtype, scale=0.5, seed=11)
    gamma, beta = 1 + 0.1 * rnd(d, seed=12), 0.1 * rnd(d, seed=13)
    dy = rnd(M, d, seed=14)
    mean, rstd = torch.empty(M, device=dev()), torch.empty(M, device=dev())
    y = torch.empty(M, d, device=dev())
    ops.rownorm([dict(X=x, mod=mod, gamma=gamma, beta=beta, Y32=y, mean=mean, rstd=rstd)], M, d, False, False, 1e-5, dtype)
    dx = rnd(M, d, seed=15)  # accumulate on top of an existing residual gradient
    dx0 = dx.clone()
    dmod = torch.empty(M, 2 * d, device=dev(), dtype=dtype)
    dgamma, dbeta = torch.zeros(d, device=dev()), torch.zeros(d, device=dev())
    ops.rownorm_bwd([dict(dY=dy, X=x, mod=mod, gamma=gamma, beta=beta, mean=mean, rstd=rstd, dX32=dx, dmod=dmod, dgamma=dgamma, dbeta=dbeta)],
                    M, d, False, False, False, True, dtype)
    xr, mr, gr, br = x.clone().requires_grad_(True), mod.float().clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    mu = xr.mean(-1, keepdim=True)
    var = ((xr - mu) ** 2).mean(-1, keepdim=True)
    yr = (xr - mu) / torch.sqrt(var + 1e-5) * (gr + 1 + mr[:, :d]) + (br + mr[:, d:])
    yr.backward(dy)
    assert rel(dx - dx0, xr.grad) < 3e-5
    assert rel(dmod.float(), mr.grad) < tol(dtype)
    assert rel(dgamma, gr.grad) < 3e-5 and rel(dbeta, br.grad) < 3e-5


@pytest.mark.parametrize("dtype", DTYPES)
def test_rownorm_bwd_ln_gelu_act(dtype):
    """MLP's LayerNorm + GELU: act-dtype h in, act-dtype dY in, dX as act; and the plain gamma-only LayerNorm."""
    from sea_amd import ops

    M, S = 130, 2048
    h = (rnd(M, S, seed=20) * 1.3 + 0.1).to(dtype)
    gamma, beta = 1 + 0.1 * rnd(S, seed=21), 0.1 * rnd(S, seed=22)
    dy = rnd(M, S, dtype=dtype, seed=23)
    mean, rstd = torch.empty(M, device=dev()), torch.empty(M, device=dev())
    hg = torch.empty(M, S, device=dev(), dtype=dtype)
    is_act = dtype != torch.float32
    ops.rownorm([dict(X=h, gamma=gamma, beta=beta, Yact=hg, mean=mean, rstd=rstd)], M, S, is_act, True, 1e-5, dtype)
    dh = torch.empty(M, S, device=dev(), dtype=dtype)
    dgamma, dbeta = torch.zeros(S, device=dev()), torch.zeros(S, device=dev())
    ops.rownorm_bwd([dict(dY=dy, X=h, gamma=gamma, beta=beta, mean=mean, rstd=rstd, dXact=dh, dgamma=dgamma, dbeta=dbeta)],
                    M, S, is_act, is_act, True, False, dtype)
    hr, gr, br = h.float().clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    gelu(torch.nn.functional.layer_norm(hr, (S,), gr, br, 1e-5)).backward(dy.float())
    assert rel(dh.float(), hr.grad) < tol(dtype)
    assert rel(dgamma, gr.grad) < 5e-5 and rel(dbeta, br.grad) < 5e-5
    # gamma-only LayerNorm (LN_type='ln'), fp32 x
    x = rnd(M, 128, seed=24)
    g2 = 1 + 0.1 * rnd(128, seed=25)
    y = torch.empty(M, 128, device=dev())
    ops.rownorm([dict(X=x, gamma=g2, Y32=y, mean=mean, rstd=rstd)], M, 128, False, False, 1e-5, dtype)
    dy2 = rnd(M, 128, seed=26)
    dx = torch.empty(M, 128, device=dev())
    dg2 = torch.zeros(128, device=dev())
    ops.rownorm_bwd([dict(dY=dy2, X=x, gamma=g2, mean=mean, rstd=rstd, dX32=dx, dgamma=dg2)], M, 128, False, False, False, False, dtype)
    xr, gr = x.clone().requires_grad_(True), g2.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr, (128,), gr, None, 1e-5).backward(dy2)
    assert rel(dx, xr.grad) < 3e-5 and rel(dg2, gr.grad) < 3e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("S,M,two_stage", [(1536, 70, False), (2048, 260, True), (4096, 90, True), (8192, 70, False), (8192, 130, True), (16384, 40, True), (12288, 33, False)])
def test_rownorm_bwd_wide_rows(dtype, S, M, two_stage):
    """The workgroup-per-row backward of LayerNorm + GELU at the hidden widths of every shipped model (2048 at embed_dim 256; 8192 / 16384 at the shipped
    cylinder / multiphase widths: 256 / 512 / 1024 threads per row, the 16384-column form re-reading the row in its second pass), two groups per launch,
    column sums through the two-stage workspace and through atomics, against autograd."""
    from sea_amd import ops

    is_act = dtype != torch.float32
    groups, refs = [], []
    for gi in range(2):
        h = (rnd(M, S, seed=120 + gi) * 1.3 + 0.1).to(dtype)
        gamma, beta = 1 + 0.1 * rnd(S, seed=122 + gi), 0.1 * rnd(S, seed=124 + gi)
        dy = rnd(M, S, dtype=dtype, seed=126 + gi)
        mean, rstd = torch.empty(M, device=dev()), torch.empty(M, device=dev())
        hg = torch.empty(M, S, device=dev(), dtype=dtype)
        ops.rownorm([dict(X=h, gamma=gamma, beta=beta, Yact=hg, mean=mean, rstd=rstd)], M, S, is_act, True, 1e-5, dtype)
        dh = torch.full((M, S), float("nan"), device=dev(), dtype=dtype)
        dgamma, dbeta = torch.zeros(S, device=dev()), torch.zeros(S, device=dev())
        groups.append(dict(dY=dy, X=h, gamma=gamma, beta=beta, mean=mean, rstd=rstd, dXact=dh, dgamma=dgamma, dbeta=dbeta))
        refs.append((h, gamma, beta, dy))
    ws = torch.empty(2 * 512 * 2 * S, device=dev()) if two_stage else None
    ops.rownorm_bwd(groups, M, S, is_act, is_act, True, False, dtype, ws=ws)
    for g, (h, gamma, beta, dy) in zip(groups, refs):
        hr, gr, br = h.float().clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        gelu(torch.nn.functional.layer_norm(hr, (S,), gr, br, 1e-5)).backward(dy.float())
        assert rel(g["dXact"].float(), hr.grad) < tol(dtype)
        assert rel(g["dgamma"], gr.grad) < 5e-5 and rel(g["dbeta"], br.grad) < 5e-5


@pytest.mark.parametrize("dtype", DTYPES)
def test_silu_outer_bwd(dtype):
    from sea_amd import ops

    M = 257
    c = torch.rand(M, device=dev())
    groups, refs = [], []
    for i, K2 in enumerate((512, 128)):
        w1, b1 = rnd(K2, seed=30 + i), rnd(K2, seed=40 + i)
        dH = rnd(M, K2, dtype=dtype, seed=50 + i)
        dw1, db1 = torch.zeros(K2, device=dev()), torch.zeros(K2, device=dev())
        groups.append(dict(dHid=dH, w1=w1, b1=b1, dw1=dw1, db1=db1))
        wr, br = w1.clone().requires_grad_(True), b1.clone().requires_grad_(True)
        torch.nn.functional.silu(c[:, None] * wr[None, :] + br[None, :]).backward(dH.float())
        refs.append((wr.grad, br.grad))
    ops.silu_outer_bwd(groups, c, M, dtype)
    for g, (rw, rb) in zip(groups, refs):
        assert rel(g["dw1"], rw) < 3e-5 and rel(g["db1"], rb) < 3e-5


def test_ib_bwd():
    from sea_amd import ops

    M, E, h, F = 203, 256, 8, 3
    dxb = rnd(M, F * E, seed=60)
    dxs = [dxb[:, i * E:(i + 1) * E] for i in range(F)]
    c = torch.rand(M, device=dev())
    ps = [rnd(h, seed=61), rnd(h, seed=62), 1 + 0.1 * rnd(h, seed=63), 0.1 * rnd(h, seed=64), rnd(E, h, seed=65), rnd(E, seed=66)]
    grads = [torch.zeros_like(p) for p in ps]
    ops.ib_bwd(dxs, c, ps[0], ps[1], ps[2], ps[3], ps[4], *grads)
    rs = [p.clone().requires_grad_(True) for p in ps]
    pre = c[:, None] * rs[0][None, :] + rs[1]
    ib = gelu(torch.nn.functional.layer_norm(pre, (h,), rs[2], rs[3], 1e-5)) @ rs[4].t() + rs[5]
    ib.backward(sum(dxs))
    for g, r, name in zip(grads, rs, ["w1", "b1", "lnw", "lnb", "w2", "b2"]):
        assert rel(g, r.grad) < 5e-5, name


@pytest.mark.parametrize("M,E,h,F,splits", [(203, 256, 8, 3, 13), (798, 1024, 8, 2, 64), (40, 64, 4, 1, 3), (1000, 2048, 8, 2, 7), (17, 320, 8, 2, 1)])
def test_ib_bwd_column_block_form(M, E, h, F, splits):
    """The three-launch column-block backward of the info-bottleneck MLP (bwd.hip: ib_bwd_cols / ib_bwd_rows / ib_bwd_finish) against autograd, at
    the benchmark's width, the shipped cylinder and multiphase widths (several column blocks adding into dhid), a partial last column block, one row
    split; run twice on the same workspaces (dhid must come back zero)."""
    from sea_amd import ops

    dxb = rnd(M, F * E, seed=160)
    dxs = [dxb[:, i * E:(i + 1) * E] for i in range(F)]
    c = torch.rand(M, device=dev())
    ps = [rnd(h, seed=161), rnd(h, seed=162), 1 + 0.1 * rnd(h, seed=163), 0.1 * rnd(h, seed=164), rnd(E, h, seed=165), rnd(E, seed=166)]
    ws = torch.full((splits * E * (1 + h),), float("nan"), device=dev())
    dhid = torch.zeros(M, 8, device=dev())
    rs = [p.clone().requires_grad_(True) for p in ps]
    pre = c[:, None] * rs[0][None, :] + rs[1]
    ib = gelu(torch.nn.functional.layer_norm(pre, (h,), rs[2], rs[3], 1e-5)) @ rs[4].t() + rs[5]
    ib.backward(sum(dxs))
    for _ in range(2):
        grads = [torch.zeros_like(p) for p in ps]
        ops.ib_bwd(dxs, c, ps[0], ps[1], ps[2], ps[3], ps[4], *grads, ws=ws, dhid=dhid)
        for g, r, name in zip(grads, rs, ["w1", "b1", "lnw", "lnb", "w2", "b2"]):
            assert rel(g, r.grad) < 5e-5, name
        assert float(dhid.abs().max()) == 0 or E <= 256   # (one column block stores: nothing to re-zero)


def _rope_table(hd, n):
    freqs = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
    ang = torch.outer(torch.arange(n, dtype=torch.float32), freqs)
    return torch.stack((torch.cos(ang), torch.sin(ang)), dim=-1).contiguous().to(dev())


def _rope(x, table):  # x [B,T,H,hd]
    c, s = table[: x.shape[1], :, 0][None, :, None, :], table[: x.shape[1], :, 1][None, :, None, :]
    xe, xo = x[..., 0::2], x[..., 1::2]
    return torch.stack((xe * c - xo * s, xe * s + xo * c), dim=-1).flatten(-2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hd", [8, 16, 32, 64, 128])
@pytest.mark.parametrize("T,src_len", [(1, 0), (50, 0), (130, 0), (200, 3)])
def test_attention_backward(dtype, hd, T, src_len):
    _attention_backward_case(dtype, hd, T, src_len, B=2, H=3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T,src_len", [(1, 0), (70, 0), (200, 2)])
def test_attention_backward_head_dim_256(dtype, T, src_len):
    """Head dim 256 (the shipped multiphase dims: embed_dim 2048 / 8 heads).  bf16: the four [64, 256] tiles of the kernels fill 136 KB of LDS;
    f32 (the 1e-4 parity path): one buffer pair (BwdCfg::NBUF = 1), two barriers per tile."""
    _attention_backward_case(dtype, 256, T, src_len, B=2, H=2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hd", [16, 32])
def test_attention_backward_full_length(dtype, hd):
    """T = 2024 (cfg3's sequence length: 32 key tiles per query tile, the multi-tile dK/dV walk) at the two head dims of the cfg2 / cfg3 model."""
    _attention_backward_case(dtype, hd, 2024, 0, B=1, H=2)


@pytest.mark.parametrize("mode", [1, 2, 3])
@pytest.mark.parametrize("dtype,hd,T,src_len,B,H", [(torch.bfloat16, 32, 330, 0, 2, 4), (torch.bfloat16, 16, 257, 2, 1, 8), (torch.float32, 32, 130, 0, 2, 3),
                                                  (torch.bfloat16, 32, 64, 0, 1, 8), (torch.float32, 16, 200, 3, 4, 2)])
def test_attention_backward_paired_tiles_and_xcd_local_order(monkeypatch, mode, dtype, hd, T, src_len, B, H):
    """The long-launch forms of the backward kernels (attention_bwd.hip: ATTNB_XCD = 1, ATTNB_PAIRED = 2; chosen by themselves from 4096 workgroups), forced
    on short launches: odd and even tile counts (the middle tile has no partner), one tile, B * H a multiple of 8 (the XCD-local order) and not (its fallback)."""
    monkeypatch.setenv("SEA_TUNE", f"attnb_mode={mode}")
    _attention_backward_case(dtype, hd, T, src_len, B=B, H=H)


def test_attention_backward_long_launch_takes_the_paired_form_and_equals_the_plain_one():
    """B * H = 16 at T = 2024 is 4096 workgroups: the launch picks the paired, XCD-local form itself; bitwise equal to the plain order (no atomics either way)."""
    from sea_amd import ops

    B, H, hd, T = 2, 8, 32, 2024
    dt = torch.bfloat16
    cap, E = T, H * hd
    Q = (rnd(B, H, T, hd, seed=1) * hd ** -0.25).to(dt)
    K = (rnd(B, H, cap, hd, seed=2) * hd ** -0.25).to(dt)
    V = rnd(B, H, cap, hd, seed=3).to(dt)
    O = torch.empty(B, T, E, device=dev(), dtype=dt)
    LSE = torch.empty(B, H, T, device=dev())
    ops.attention_fwd([dict(Q=Q, K=K, Vt=V.transpose(2, 3).contiguous(), O=O, LSE=LSE)], B, H, hd, T, T, cap, 0, 0, dt)
    dO = rnd(B, T, E, dtype=dt, seed=4)
    table = _rope_table(hd, cap)
    outs = []
    for tune in (None, "attnb_mode=0"):
        if tune is None:
            os.environ.pop("SEA_TUNE", None)
        else:
            os.environ["SEA_TUNE"] = tune
        try:
            g = [torch.full((B * T, E), float("nan"), device=dev(), dtype=dt) for _ in range(3)]
            delta = torch.empty(B, H, T, device=dev())
            ops.attention_bwd([dict(Q=Q, K=K, V=V, O=O, dO=dO, LSE=LSE, delta=delta, dQ=g[0], dK=g[1], dV=g[2])], table, B, H, hd, T, T, cap, 0, 0, ops.q_scale(hd), dt)
            torch.cuda.synchronize()
            outs.append(g)
        finally:
            os.environ.pop("SEA_TUNE", None)
    for a, b in zip(*outs):
        assert torch.isfinite(a.float()).all() and torch.equal(a, b)


def _attention_backward_case(dtype, hd, T, src_len, B, H):
    """dQ/dK/dV of the q/k/v projection outputs (RoPE and scale undone) against autograd through rope -> scale -> masked softmax."""
    from sea_amd import ops

    E = H * hd
    cap = (T + 7) // 8 * 8
    scale = hd ** -0.5
    scale2 = ops.q_scale(hd)      # what the QKV epilogue puts on q: hd^-1/2 * log2(e) — the attention kernels score in log2 units
    table = _rope_table(hd, cap)
    q0 = rnd(B, T, H, hd, seed=70).requires_grad_(True)
    k0 = rnd(B, T, H, hd, seed=71).requires_grad_(True)
    v0 = rnd(B, T, H, hd, seed=72).requires_grad_(True)
    dO = rnd(B, T, E, dtype=dtype, seed=73)
    # what the QKV epilogue would have written (rounded to the activation dtype)
    Q = (_rope(q0, table) * scale2).permute(0, 2, 1, 3).contiguous().to(dtype)
    K = torch.zeros(B, H, cap, hd, device=dev(), dtype=dtype)
    V = torch.zeros(B, H, cap, hd, device=dev(), dtype=dtype)
    K[:, :, :T] = _rope(k0, table).permute(0, 2, 1, 3).to(dtype)
    V[:, :, :T] = v0.permute(0, 2, 1, 3).to(dtype)
    Vt = V.transpose(2, 3).contiguous()
    O = torch.empty(B, T, E, device=dev(), dtype=dtype)
    LSE = torch.empty(B, H, T, device=dev())
    ops.attention_fwd([dict(Q=Q, K=K.detach(), Vt=Vt.detach(), O=O, LSE=LSE)], B, H, hd, T, T, cap, 0, src_len, dtype)
    dQ = torch.full((B * T, E), float("nan"), device=dev(), dtype=dtype)
    dKV = torch.full((B * T, 2 * E), float("nan"), device=dev(), dtype=dtype)
    delta = torch.empty(B, H, T, device=dev())
    ops.attention_bwd([dict(Q=Q.detach(), K=K.detach(), V=V.detach(), O=O, dO=dO, LSE=LSE, delta=delta, dQ=dQ, dK=dKV[:, :E], dV=dKV[:, E:])],
                      table, B, H, hd, T, T, cap, 0, src_len, scale2, dtype)
    # autograd reference in fp32 on the same (rounded) operands: natural-log softmax of the log2-unit scores times ln 2
    qr = (_rope(q0, table) * scale2).permute(0, 2, 1, 3)
    kr = _rope(k0, table).permute(0, 2, 1, 3)
    vr = v0.permute(0, 2, 1, 3)
    if dtype != torch.float32:  # straight-through rounding so that the reference sees the operands the kernel saw
        qr = qr + (qr.detach().to(dtype).float() - qr.detach())
        kr = kr + (kr.detach().to(dtype).float() - kr.detach())
        vr = vr + (vr.detach().to(dtype).float() - vr.detach())
    S = (qr @ kr.transpose(-1, -2)) * 0.6931471805599453
    i = torch.arange(T, device=dev())
    S = S.masked_fill(i[None, :] > i[:, None] + src_len, float("-inf"))
    Oref = (torch.softmax(S, -1) @ vr).transpose(1, 2).reshape(B, T, E)
    Oref.backward(dO.float())
    t = 2e-5 if dtype == torch.float32 else 2e-2
    assert torch.isfinite(dQ.float()).all() and torch.isfinite(dKV.float()).all()
    def close(a, b):  # relative to the reference, with a floor for the T=1 case where dQ = dK = 0 exactly
        return float((a.double() - b.double()).norm()) <= t * max(float(b.double().norm()), 0.1 * float(dO.float().norm()))

    assert close(dQ.float().view(B, T, H, hd), q0.grad)
    assert close(dKV[:, :E].float().reshape(B, T, H, hd), k0.grad)
    assert close(dKV[:, E:].float().reshape(B, T, H, hd), v0.grad)
