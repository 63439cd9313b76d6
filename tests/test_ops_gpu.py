"""Operator-level parity of the HIP kernels (through the C ABI) against plain fp32 PyTorch restatements of the same
formulas, on the MI355X.  fp32 mode must agree to accumulation-order noise; bf16 mode is checked against the same
formula evaluated in fp32 on the bf16-rounded operands (tolerance = bf16 output rounding)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def dev():
    return torch.device("cuda:0")


def tol(dtype, f32=2e-5, bf16=6e-3):
    return f32 if dtype == torch.float32 else bf16


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def rnd(*shape, dtype=torch.float32, scale=1.0, seed=None):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed if seed is not None else (hash(shape) & 0xFFFF) + 17)
    return (torch.randn(*shape, generator=g) * scale).to(dev()).to(dtype)


def gelu(x):
    return 0.5 * x * (1 + torch.erf(x / math.sqrt(2)))


def test_mfma_fragment_maps():
    from sea_amd import _native as N

    rc = N.lib().sea_selftest_mfma()
    assert rc == 0, N.lib().sea_last_error()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N_,K", [(2024, 256, 256), (100, 48, 40), (1, 16, 8), (2024, 2048, 256), (300, 256, 2048), (64, 128, 128)])
def test_gemm_bias_residual(dtype, M, N_, K):
    from sea_amd import ops

    A = rnd(M, K, dtype=dtype, seed=1)
    W = rnd(N_, K, dtype=dtype, scale=K ** -0.5, seed=2)
    bias = rnd(N_, seed=3)
    Rbig = rnd(M, 3 * N_, seed=4)
    R = Rbig[:, N_: 2 * N_]  # strided residual, like one field of x[B,T,F,E]
    C32 = torch.full((M, N_), float("nan"), device=dev())
    Cact = torch.empty(M, N_, device=dev(), dtype=dtype)
    ops.gemm_grouped([dict(A=A, W=W, bias=bias, R=R, C32=C32, Cact=Cact)], dtype)
    ref = A.float() @ W.float().t() + bias + R
    assert rel(C32, ref) < tol(dtype, bf16=1e-5)  # operands are already rounded; accumulation is fp32 in both modes
    assert rel(Cact.float(), ref) < tol(dtype)


def test_gemm256_coalesced_residual_epilogue(monkeypatch):
    """gemm256.hip's path for a bf16 output with a fp32 residual (the training forward's mlp.fc2: x + MLP(x)): the residual is added in the coalesced copy-out.  cfg3's
    shape at a sixth of its rows per group — ragged last row tile, strided residual (one field of [M, F E]), N below the 256-column tile in one group — against fp32
    torch and against the 128 x 128 kernel."""
    from sea_amd import ops

    bf = torch.bfloat16
    K = 2048
    groups, refs = [], []
    for i, (M, N_) in enumerate([(2700, 256), (2700, 256), (2100, 200)]):
        A = rnd(M, K, dtype=bf, seed=110 + i)
        W = rnd(N_, K, dtype=bf, scale=K ** -0.5, seed=120 + i)
        bias = rnd(N_, seed=130 + i)
        Rbig = rnd(M, 3 * N_, seed=140 + i)
        R = Rbig[:, N_:2 * N_]
        groups.append(dict(A=A, W=W, bias=bias, R=R, Cact=torch.full((M, N_), float("nan"), device=dev(), dtype=bf)))
        refs.append(A.float() @ W.float().t() + bias + R)
    monkeypatch.setenv("SEA_TUNE", "gemm256=1")
    ops.gemm_grouped(groups, bf)
    torch.cuda.synchronize()
    first = [g["Cact"].clone() for g in groups]
    for c, ref in zip(first, refs):
        assert torch.isfinite(c.float()).all() and rel(c.float(), ref) < 6e-3
    monkeypatch.setenv("SEA_TUNE", "gemm256=0")
    ops.gemm_grouped(groups, bf)
    torch.cuda.synchronize()
    for g, c in zip(groups, first):
        assert rel(g["Cact"].float(), c.float()) < 4e-3


@pytest.mark.parametrize("K", [256, 512])
def test_gemm_weight_stationary_streaming_kernel(monkeypatch, K):
    """gemm_ws.hip (plain bf16 launches with K = 256 / 512: weights in registers, activation rows streamed, the tile's store under the next tile's MFMAs),
    forced on short launches (SEA_TUNE=gemm_ws=2) to reach every edge in one launch of four groups: a ragged last row tile, one tile, fewer rows than a tile,
    N not a multiple of the 256-column panel (and of 32: a wave with no column), strided A and C, with and without bias, bias_scale; against fp32 torch and
    bitwise against the tiled kernels (both accumulate a bf16 x bf16 product in fp32 in the same k order per MFMA ... not the same order across k: tolerance)."""
    from sea_amd import ops

    bf = torch.bfloat16
    shapes = [(1000, 768, True), (32, 256, False), (5, 40, True), (333, 264, True)]
    groups, refs, outs = [], [], []
    for i, (M, N_, has_bias) in enumerate(shapes):
        Abig = rnd(M, K + 64, dtype=bf, seed=10 + i)
        A = Abig[:, 32:32 + K]                      # lda = K + 64, 64-byte offset
        W = rnd(N_, K, dtype=bf, scale=K ** -0.5, seed=20 + i)
        bias = rnd(N_, seed=30 + i) if has_bias else None
        Cbig = torch.full((M, N_ + 24), float("nan"), device=dev(), dtype=bf)
        C = Cbig[:, 8:8 + N_]
        g = dict(A=A, W=W, Cact=C)
        if bias is not None:
            g["bias"] = bias
            g["bias_scale"] = 2.0 if i == 3 else 1.0
        groups.append(g)
        outs.append((Cbig, C))
        refs.append(A.float() @ W.float().t() + (bias * g.get("bias_scale", 1.0) if bias is not None else 0.0))
    monkeypatch.setenv("SEA_TUNE", "gemm_ws=2")
    ops.gemm_grouped(groups, bf)
    torch.cuda.synchronize()
    for (Cbig, C), ref, (M, N_, _) in zip(outs, refs, shapes):
        assert torch.isfinite(C.float()).all()
        assert rel(C.float(), ref) < 6e-3
        assert torch.isnan(Cbig[:, :8].float()).all() and torch.isnan(Cbig[:, 8 + N_:].float()).all()   # nothing outside the group's columns
    # the tiled kernels on the same launch: the same bf16 values up to the rounding of a differently ordered fp32 sum
    keep = [C.clone() for _, C in outs]
    for _, C in outs:
        C.fill_(float("nan"))
    monkeypatch.setenv("SEA_TUNE", "gemm_ws=0")
    ops.gemm_grouped(groups, bf)
    torch.cuda.synchronize()
    for (_, C), k in zip(outs, keep):
        assert rel(C.float(), k.float()) < 4e-3


def test_gemm_weight_stationary_mixed_contraction_lengths(monkeypatch):
    """K = 256 and K = 512 groups in ONE call (cfg3's condition GEMMs: nine modules of width 512, three of width 256) go as two launches of the streaming kernel."""
    from sea_amd import ops

    bf = torch.bfloat16
    groups, refs = [], []
    for i, (M, N_, K) in enumerate([(700, 512, 512), (700, 256, 256), (90, 512, 512), (1000, 256, 256)]):
        A = rnd(M, K, dtype=bf, seed=70 + i)
        W = rnd(N_, K, dtype=bf, scale=K ** -0.5, seed=80 + i)
        bias = rnd(N_, seed=90 + i)
        groups.append(dict(A=A, W=W, bias=bias, Cact=torch.full((M, N_), float("nan"), device=dev(), dtype=bf)))
        refs.append(A.float() @ W.float().t() + bias)
    monkeypatch.setenv("SEA_TUNE", "gemm_ws=2")
    ops.gemm_grouped(groups, bf)
    torch.cuda.synchronize()
    for g, ref in zip(groups, refs):
        assert torch.isfinite(g["Cact"].float()).all() and rel(g["Cact"].float(), ref) < 6e-3


def test_gemm_weight_stationary_kernel_is_taken_by_long_launches(monkeypatch):
    """cfg3's mlp.fc1 shape at a quarter of its rows (3 fields x 4096 rows, N = 2048, K = 256: 3072 iterations, 12 per CU) picks the streaming kernel by itself;
    every workgroup's chunk crosses panels and groups.  Against fp32 torch and against the tiled kernels."""
    from sea_amd import ops

    bf = torch.bfloat16
    M, N_, K = 4096, 2048, 256
    groups, refs = [], []
    for i in range(3):
        A = rnd(M, K, dtype=bf, seed=40 + i)
        W = rnd(N_, K, dtype=bf, scale=K ** -0.5, seed=50 + i)
        bias = rnd(N_, seed=60 + i)
        groups.append(dict(A=A, W=W, bias=bias, Cact=torch.full((M, N_), float("nan"), device=dev(), dtype=bf)))
        refs.append(A.float() @ W.float().t() + bias)
    monkeypatch.delenv("SEA_TUNE", raising=False)
    ops.gemm_grouped(groups, bf)
    torch.cuda.synchronize()
    first = [g["Cact"].clone() for g in groups]
    for c, ref in zip(first, refs):
        assert torch.isfinite(c.float()).all() and rel(c.float(), ref) < 6e-3
    monkeypatch.setenv("SEA_TUNE", "gemm_ws=0")
    ops.gemm_grouped(groups, bf)
    torch.cuda.synchronize()
    for g, c in zip(groups, first):
        assert rel(g["Cact"].float(), c.float()) < 4e-3


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_grouped_gelu_multiseg(dtype):
    from sea_amd import ops

    M, D, E = 333, 128, 256
    # three GELU(projection) groups of different shapes + one multi-segment accumulate group in a second launch
    As = [rnd(M, D, dtype=dtype, seed=10 + i) for i in range(3)]
    Ws = [rnd(D, D, dtype=dtype, scale=D ** -0.5, seed=20 + i) for i in range(3)]
    seg = torch.empty(3, M, D, device=dev(), dtype=dtype)
    ops.gemm_grouped([dict(A=As[i], W=Ws[i], Cact=seg[i], act=1) for i in range(3)], dtype)
    for i in range(3):
        assert rel(seg[i].float(), gelu(As[i].float() @ Ws[i].float().t())) < tol(dtype)
    Wup = rnd(E, D, dtype=dtype, scale=D ** -0.5, seed=30)
    bup = rnd(E, seed=31)
    x = rnd(M, E, seed=32)
    x0 = x.clone()
    ops.gemm_grouped([dict(A=seg[0], W=Wup, bias=bup, R=x, C32=x, n_seg=3, a_seg_stride=M * D, bias_scale=3.0)], dtype)
    ref = x0 + sum(seg[i].float() @ Wup.float().t() + bup for i in range(3))
    assert rel(x, ref) < tol(dtype, bf16=1e-5)


@pytest.mark.parametrize("M", [1, 3, 16])
@pytest.mark.parametrize("K,N_", [(128, 256), (256, 2048), (512, 512), (2048, 256), (256, 100), (384, 48)])
def test_gemm_few_rows_kernel(M, K, N_):
    """M <= 16 rows in bf16 (the GEMMs of a KV-cache rollout step) run the one-workgroup-per-16-columns kernel with the contraction split over
    its waves: bias (scaled), in-place fp32 residual, fp32 and bf16 outputs, strided A, three groups of different widths in one launch."""
    from sea_amd import ops

    dt = torch.bfloat16
    groups, refs = [], []
    for i, (k, n) in enumerate(((K, N_), (K, 64), (256, N_))):
        Abig = rnd(M, 2 * k, dtype=dt, seed=1500 + i)
        A = Abig[:, k:]
        W = rnd(n, k, dtype=dt, scale=k ** -0.5, seed=1510 + i)
        b = rnd(n, seed=1520 + i)
        x = rnd(M, n, seed=1530 + i)
        x0 = x.clone()
        act = torch.full((M, n), float("nan"), device=dev(), dtype=dt)
        groups.append(dict(A=A, W=W, bias=b, R=x, C32=x, Cact=act, bias_scale=2.0))
        refs.append(x0 + A.float() @ W.float().t() + 2.0 * b)
    ops.gemm_grouped(groups, dt)
    for g_, ref in zip(groups, refs):
        assert rel(g_["C32"], ref) < 2e-5
        assert rel(g_["Cact"].float(), ref) < 4e-3


@pytest.mark.parametrize("M", [1, 2, 16])
@pytest.mark.parametrize("K,N_", [(8192, 1024), (16384, 2048), (2048, 16384), (4096, 100)])
def test_gemm_few_rows_long_contraction_and_gelu(M, K, N_):
    """The few-row kernels at the SHIPPED widths (fc2 / fc1 of a KV-cache step at embed_dim 1024 / 2048: contractions of 8192 / 16384 over eight waves in
    rounds) and with the GELU epilogue (act 1: the pre-activation kept in Z) the cross-attention projection of a step uses."""
    from sea_amd import ops

    dt = torch.bfloat16
    A = rnd(M, K, dtype=dt, seed=1600)
    W = rnd(N_, K, dtype=dt, scale=K ** -0.5, seed=1601)
    b = rnd(N_, seed=1602)
    x = rnd(M, N_, seed=1603)
    x0 = x.clone()
    act = torch.full((M, N_), float("nan"), device=dev(), dtype=dt)
    ops.gemm_grouped([dict(A=A, W=W, bias=b, R=x, C32=x, Cact=act)], dt)
    ref = x0 + A.float() @ W.float().t() + b
    assert rel(x, ref) < 3e-5 and rel(act.float(), ref) < 4e-3
    # GELU epilogue, two groups of different contraction lengths in one launch (both whole rounds when one is long)
    z = torch.full((M, N_), float("nan"), device=dev(), dtype=dt)
    g = torch.full((M, N_), float("nan"), device=dev(), dtype=dt)
    K2 = 2048
    A2, W2 = rnd(M, K2, dtype=dt, seed=1604), rnd(64, K2, dtype=dt, scale=K2 ** -0.5, seed=1605)
    g2 = torch.empty(M, 64, device=dev())
    ops.gemm_grouped([dict(A=A, W=W, bias=b, Cact=g, Z=z, act=1), dict(A=A2, W=W2, C32=g2, act=1)], dt)
    pre = A.float() @ W.float().t() + b
    assert rel(z.float(), pre) < 4e-3
    assert rel(g.float(), torch.nn.functional.gelu(pre)) < 6e-3
    assert rel(g2, torch.nn.functional.gelu(A2.float() @ W2.float().t())) < 3e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,d", [(1, 8192), (2, 16384), (5, 4096), (3, 2052), (2, 2048), (3, 1024), (16, 1028), (32, 4100)])
def test_rownorm_few_long_rows(dtype, M, d):
    """A few long rows (the MLP's LayerNorm + GELU of a KV-cache step at the shipped widths): a workgroup per row.  LayerNorm + GELU of act rows, and the
    modulated form on fp32 rows with both outputs and the statistics."""
    from sea_amd import ops

    x = rnd(M, d, dtype=dtype, seed=1700)
    lnw, lnb = 1 + 0.1 * rnd(d, seed=1701), 0.1 * rnd(d, seed=1702)
    y = torch.full((M, d), float("nan"), device=dev(), dtype=dtype)
    ops.rownorm([dict(X=x, gamma=lnw, beta=lnb, Yact=y)], M, d, True, True, 1e-5, dtype)
    xf = x.float()
    mu, var = xf.mean(1, keepdim=True), xf.var(1, unbiased=False, keepdim=True)
    ref = torch.nn.functional.gelu((xf - mu) / torch.sqrt(var + 1e-5) * lnw + lnb)
    assert rel(y.float(), ref) < tol(dtype, f32=1e-5, bf16=6e-3)
    x32 = rnd(M, d, seed=1703)
    mod = (0.3 * rnd(M, 2 * d, seed=1704)).to(dtype)
    y32, ya = torch.empty(M, d, device=dev()), torch.empty(M, d, device=dev(), dtype=dtype)
    mean, rstd = torch.empty(M, device=dev()), torch.empty(M, device=dev())
    ops.rownorm([dict(X=x32, mod=mod, gamma=lnw, beta=lnb, Y32=y32, Yact=ya, mean=mean, rstd=rstd)], M, d, False, False, 1e-5, dtype)
    mu, var = x32.mean(1, keepdim=True), x32.var(1, unbiased=False, keepdim=True)
    ref = (x32 - mu) / torch.sqrt(var + 1e-5) * (lnw + 1 + mod[:, :d].float()) + lnb + mod[:, d:].float()
    assert rel(y32, ref) < 1e-5 and rel(ya.float(), ref) < tol(dtype, f32=1e-5, bf16=6e-3)
    assert rel(mean, mu[:, 0]) < 1e-4 and rel(rstd, 1 / torch.sqrt(var[:, 0] + 1e-5)) < 1e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,d", [(2, 2048), (3, 1024)])
def test_rownorm_few_rows_with_addend(dtype, M, d):
    """The one-round-trip form (d <= 4096) with the info-bottleneck addend: x + addend is written back exactly and normalised (plain LayerNorm, no beta)."""
    from sea_amd import ops

    x, add = rnd(M, d, seed=1710), 0.2 * rnd(M, d, seed=1711)
    g = 1 + 0.1 * rnd(d, seed=1712)
    xio, ya = x.clone(), torch.empty(M, d, device=dev(), dtype=dtype)
    ops.rownorm([dict(X=xio, addend=add, Xout=xio, gamma=g, Yact=ya)], M, d, False, False, 1e-5, dtype)
    xs = x + add
    assert torch.equal(xio, xs)
    ref = torch.nn.functional.layer_norm(xs, (d,), g, None, 1e-5)
    assert rel(ya.float(), ref) < tol(dtype, f32=1e-5, bf16=6e-3)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_strided_a(dtype):
    from sea_amd import ops

    M, K, N_ = 130, 64, 96
    Abig = rnd(M, 3 * K, dtype=dtype, seed=5)
    A = Abig[:, K: 2 * K]
    W = rnd(N_, K, dtype=dtype, seed=6)
    out = torch.empty(M, N_, device=dev())
    ops.gemm_grouped([dict(A=A, W=W, C32=out)], dtype)
    assert rel(out, A.float() @ W.float().t()) < 1e-5


def rope_ref(x, cos, sin):
    xe, xo = x[..., 0::2], x[..., 1::2]
    c, s = cos[None, :, None, :], sin[None, :, None, :]
    return torch.stack((xe * c - xo * s, xe * s + xo * c), dim=-1).flatten(-2)


def rope_table(hd, n):
    freqs = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
    ang = torch.outer(torch.arange(n, dtype=torch.float32), freqs)
    return torch.stack((torch.cos(ang), torch.sin(ang)), dim=-1).contiguous().to(dev())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,T,H,hd,pos0,cap", [(2, 100, 8, 32, 0, 104), (1, 2024, 8, 32, 0, 2024), (3, 1, 4, 16, 5, 16), (2, 9, 4, 8, 0, 16),
                                                (3, 1, 8, 32, 5, 16), (1, 1, 8, 16, 100, 104), (2, 4, 8, 32, 7, 16), (16, 1, 8, 32, 2023, 2024)])   # the last four: the few-row kernel in bf16
def test_qkv_rope_self(dtype, B, T, H, hd, pos0, cap):
    from sea_amd import ops

    E = H * hd
    M = B * T
    x = rnd(M, E, dtype=dtype, seed=40)
    W = rnd(3 * E, E, dtype=dtype, scale=E ** -0.5, seed=41)
    bias = rnd(3 * E, seed=42)
    table = rope_table(hd, pos0 + T)
    Q = torch.zeros(B, H, T, hd, device=dev(), dtype=dtype)
    K = torch.zeros(B, H, cap, hd, device=dev(), dtype=dtype)
    Vt = torch.zeros(B, H, hd, cap, device=dev(), dtype=dtype)
    scale = hd ** -0.5
    ops.qkv_rope_grouped([dict(A=x, W=W, bias=bias, col0=0, Q=Q, K=K, Vt=Vt)], table, H, hd, T, pos0, cap, scale, dtype)
    y = x.float() @ W.float().t() + bias
    q, k, v = (y[:, i * E:(i + 1) * E].view(B, T, H, hd) for i in range(3))
    cos, sin = table[pos0:pos0 + T, :, 0], table[pos0:pos0 + T, :, 1]
    q = rope_ref(q, cos, sin) * scale
    k = rope_ref(k, cos, sin)
    t = tol(dtype)
    assert rel(Q.float(), q.permute(0, 2, 1, 3)) < t
    assert rel(K[:, :, pos0:pos0 + T].float(), k.permute(0, 2, 1, 3)) < t
    assert rel(Vt[:, :, :, pos0:pos0 + T].float(), v.permute(0, 2, 3, 1)) < t
    if cap > pos0 + T:  # nothing outside the written window is touched
        assert float(K[:, :, pos0 + T:].abs().max()) == 0 and float(Vt[:, :, :, pos0 + T:].abs().max()) == 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,T", [(2, 50), (3, 1), (2, 5)])
def test_qkv_rope_cross_groups(dtype, B, T):
    from sea_amd import ops

    H, hd = 8, 16
    D = H * hd
    M = B * T
    xi, xj = rnd(M, D, dtype=dtype, seed=50), rnd(M, D, dtype=dtype, seed=51)
    Wq = rnd(D, D, dtype=dtype, scale=D ** -0.5, seed=52)
    Wkv = rnd(2 * D, D, dtype=dtype, scale=D ** -0.5, seed=53)
    bq, bkv = rnd(D, seed=54), rnd(2 * D, seed=55)
    table = rope_table(hd, T)
    Q = torch.zeros(B, H, T, hd, device=dev(), dtype=dtype)
    K = torch.zeros(B, H, T + 6, hd, device=dev(), dtype=dtype)
    Vt = torch.zeros(B, H, hd, T + 6, device=dev(), dtype=dtype)
    ops.qkv_rope_grouped([dict(A=xi, W=Wq, bias=bq, col0=0, Q=Q), dict(A=xj, W=Wkv, bias=bkv, col0=D, K=K, Vt=Vt)],
                         table, H, hd, T, 0, T + 6, hd ** -0.5, dtype)
    cos, sin = table[:T, :, 0], table[:T, :, 1]
    q = rope_ref((xi.float() @ Wq.float().t() + bq).view(B, T, H, hd), cos, sin) * hd ** -0.5
    kv = xj.float() @ Wkv.float().t() + bkv
    k = rope_ref(kv[:, :D].view(B, T, H, hd), cos, sin)
    v = kv[:, D:].view(B, T, H, hd)
    t = tol(dtype)
    assert rel(Q.float(), q.permute(0, 2, 1, 3)) < t
    assert rel(K[:, :, :T].float(), k.permute(0, 2, 1, 3)) < t
    assert rel(Vt[:, :, :, :T].float(), v.permute(0, 2, 3, 1)) < t


LN2 = 0.6931471805599453


def attention_ref(Q, K, Vt, q_pos0, src_len, Tk):
    """Q [B,H,Tq,hd] (scaled), K [B,H,cap,hd], Vt [B,H,hd,cap] -> O [B,Tq,H*hd], LSE [B,H,Tq].  The kernels' contract (include/sea_hip.h): scores are in LOG2
    units — the QKV epilogue hands over q * hd^-1/2 * log2(e) — so P = 2^(Q K^T - max) = softmax(ln2 * Q K^T), and LSE is the base-2 log-sum-exp."""
    Q, K, V = Q.float(), K[:, :, :Tk].float(), Vt[:, :, :, :Tk].float().transpose(2, 3)
    B, H, Tq, hd = Q.shape
    S = (Q @ K.transpose(-1, -2)) * LN2
    i = torch.arange(Tq, device=Q.device)[:, None] + q_pos0 + src_len
    j = torch.arange(Tk, device=Q.device)[None, :]
    S = S.masked_fill(j > i, float("-inf"))
    lse = torch.logsumexp(S, dim=-1) / LN2
    O = torch.softmax(S, dim=-1) @ V
    return O.transpose(1, 2).reshape(B, Tq, H * hd), lse


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hd", [8, 16, 32, 64, 128, 256])
@pytest.mark.parametrize("T,src_len", [(1, 0), (7, 0), (65, 0), (200, 3), (2024, 0)])
def test_attention_prefill(dtype, hd, T, src_len):
    """Head dim 256 (the shipped multiphase width) runs in both dtypes: bf16 on the double-buffered ring, f32 on one LDS buffer (133 KiB per K / V^T pair)."""
    from sea_amd import ops

    if T == 2024 and hd not in (16, 32):
        pytest.skip("full-length case only at the benchmark head dims")
    B, H = (2, 4) if hd < 256 else (2, 2)
    cap = (T + 7) // 8 * 8
    Q = rnd(B, H, T, hd, dtype=dtype, scale=hd ** -0.25, seed=60)
    K = rnd(B, H, cap, hd, dtype=dtype, scale=hd ** -0.25, seed=61)
    Vt = rnd(B, H, hd, cap, dtype=dtype, seed=62)
    if cap > T:  # poison the padding: it must never leak into the result
        K[:, :, T:] = float("nan")
        Vt[:, :, :, T:] = float("nan")
    O = torch.full((B, T, H * hd), float("nan"), device=dev(), dtype=dtype)
    LSE = torch.empty(B, H, T, device=dev())
    ops.attention_fwd([dict(Q=Q, K=K, Vt=Vt, O=O, LSE=LSE)], B, H, hd, T, T, cap, 0, src_len, dtype)
    Oref, lse_ref = attention_ref(Q, K, Vt, 0, src_len, T)
    assert torch.isfinite(O.float()).all()
    assert rel(O.float(), Oref) < tol(dtype, f32=2e-5, bf16=8e-3)
    assert rel(LSE, lse_ref) < tol(dtype, f32=1e-5, bf16=1e-3)


@pytest.mark.parametrize("dtype,hd,T,src_len,B,H", [(torch.bfloat16, 32, 330, 0, 2, 4), (torch.bfloat16, 64, 257, 2, 1, 8), (torch.float32, 32, 130, 0, 2, 3),
                                                  (torch.bfloat16, 32, 64, 0, 1, 8), (torch.float32, 32, 1100, 3, 1, 8)])
def test_attention_prefill_paired_tiles(monkeypatch, dtype, hd, T, src_len, B, H):
    """The long-launch form of the forward (attention.hip: a workgroup takes query tile n - 1 - t, then tile t; (trajectory, head) pairs XCD-local when their
    number is a multiple of 8), forced on short launches: odd and even tile counts, one tile, the fallback order; against the reference AND bitwise against the plain form."""
    from sea_amd import ops

    cap = (T + 7) // 8 * 8
    Q = rnd(B, H, T, hd, dtype=dtype, scale=hd ** -0.25, seed=60)
    K = rnd(B, H, cap, hd, dtype=dtype, scale=hd ** -0.25, seed=61)
    Vt = rnd(B, H, hd, cap, dtype=dtype, seed=62)
    outs = []
    for paired in (1, 0):
        monkeypatch.setenv("SEA_TUNE", f"attn_paired={paired},attn_split4=0")
        O = torch.full((B, T, H * hd), float("nan"), device=dev(), dtype=dtype)
        LSE = torch.full((B, H, T), float("nan"), device=dev())
        ops.attention_fwd([dict(Q=Q, K=K, Vt=Vt, O=O, LSE=LSE)], B, H, hd, T, T, cap, 0, src_len, dtype)
        torch.cuda.synchronize()
        outs.append((O, LSE))
    Oref, lse_ref = attention_ref(Q, K, Vt, 0, src_len, T)
    assert rel(outs[0][0].float(), Oref) < tol(dtype, f32=2e-5, bf16=8e-3)
    assert rel(outs[0][1], lse_ref) < tol(dtype, f32=1e-5, bf16=1e-3)
    if (B * H * ((T + 63) // 64)) > 1024 or T < 256:   # (short launches with long key ranges run two wave groups per tile: never paired, nothing to compare)
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_decode_and_multiproblem(dtype):
    """Tq = 1 rows at absolute position q_pos0 against a longer cache; three problems in one launch; strided O."""
    from sea_amd import ops

    B, H, hd, cap = 3, 8, 16, 136
    for q_pos0 in (0, 63, 64, 130):
        Tk = q_pos0 + 1
        probs, refs = [], []
        Obig = torch.zeros(3, B, 1, 2 * H * hd, device=dev(), dtype=dtype)
        for p in range(3):
            Q = rnd(B, H, 1, hd, dtype=dtype, seed=70 + p)
            K = rnd(B, H, cap, hd, dtype=dtype, seed=80 + p)
            Vt = rnd(B, H, hd, cap, dtype=dtype, seed=90 + p)
            O = Obig[p, :, :, H * hd:]
            probs.append(dict(Q=Q, K=K, Vt=Vt, O=O))
            refs.append(attention_ref(Q, K, Vt, q_pos0, 0, Tk)[0])
        ops.attention_fwd(probs, B, H, hd, 1, Tk, cap, q_pos0, 0, dtype)
        for p in range(3):
            assert rel(probs[p]["O"].float(), refs[p]) < tol(dtype, bf16=8e-3)
        assert float(Obig[..., : H * hd].abs().max()) == 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hd", [8, 16, 32, 64, 128, 256])
@pytest.mark.parametrize("q_pos0,src_len,Tk", [(0, 0, 1), (6, 0, 7), (510, 0, 511), (511, 0, 512), (2023, 0, 2024), (100, 3, 104), (100, 3, 90), (5000, 0, 5001)])
def test_attention_single_query_row_kernel(dtype, hd, q_pos0, src_len, Tk):
    """The KV-cache step (Tq = 1, no LSE): the one-row kernels (a workgroup per (trajectory, head), fp32 probabilities; head dims 128 / 256 — the shipped
    widths — with several lanes per key) against the formula; cache padding and everything past the visible keys poisoned with NaN; keys limited by the
    causal rule or by Tk, whichever is smaller."""
    from sea_amd import ops

    B, H = 2, 4
    cap = (max(Tk, q_pos0 + src_len + 1) + 15) // 8 * 8
    Q = rnd(B, H, 1, hd, dtype=dtype, scale=hd ** -0.25, seed=260)
    K = rnd(B, H, cap, hd, dtype=dtype, scale=hd ** -0.25, seed=261)
    Vt = rnd(B, H, hd, cap, dtype=dtype, seed=262)
    nk = min(Tk, q_pos0 + src_len + 1)
    K[:, :, nk:] = float("nan")
    Vt[:, :, :, nk:] = float("nan")
    O = torch.full((B, 1, H * hd), float("nan"), device=dev(), dtype=dtype)
    ops.attention_fwd([dict(Q=Q, K=K, Vt=Vt, O=O)], B, H, hd, 1, Tk, cap, q_pos0, src_len, dtype)
    Kc, Vc = K.clone(), Vt.clone()
    Kc[:, :, nk:] = 0
    Vc[:, :, :, nk:] = 0
    Oref, _ = attention_ref(Q, Kc, Vc, q_pos0, src_len, Tk)
    assert torch.isfinite(O.float()).all()
    assert rel(O.float(), Oref) < tol(dtype, f32=2e-5, bf16=8e-3)


def test_attention_softmax_rescale_branch():
    """Force the running max to jump late in the key sequence (cdna guide rule 26): one key far above the others."""
    from sea_amd import ops

    dtype = torch.float32
    B, H, hd, T = 1, 2, 32, 300
    Q = rnd(B, H, T, hd, seed=100)
    K = rnd(B, H, 304, hd, seed=101) * 0.1
    K[:, :, 250] = Q[:, :, 280] * 4.0  # spike for late queries at key 250 (4th key tile)
    Vt = rnd(B, H, hd, 304, seed=102)
    O = torch.empty(B, T, H * hd, device=dev())
    ops.attention_fwd([dict(Q=Q, K=K, Vt=Vt, O=O)], B, H, hd, T, T, 304, 0, 0, dtype)
    assert rel(O, attention_ref(Q, K, Vt, 0, 0, T)[0]) < 2e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("d", [64, 128, 256, 1024])
def test_rownorm_adaln_and_ln(dtype, d):
    from sea_amd import ops

    M = 203
    xb = rnd(M, 3 * d, seed=110) * 1.7 + 0.3
    x = xb[:, d:2 * d]
    mod = rnd(M, 2 * d, dtype=dtype, scale=0.5, seed=111)
    gamma, beta = 1 + 0.1 * rnd(d, seed=112), 0.1 * rnd(d, seed=113)
    y = torch.empty(M, d, device=dev(), dtype=dtype)
    y32b = torch.zeros(M, 3 * d, device=dev())
    mean, rstd = torch.empty(M, device=dev()), torch.empty(M, device=dev())
    ops.rownorm([dict(X=x, mod=mod, gamma=gamma, beta=beta, Yact=y, Y32=y32b[:, d:2 * d], mean=mean, rstd=rstd)], M, d, False, False, 1e-5, dtype)
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    xh = (x - mu) / torch.sqrt(var + 1e-5)
    ref = xh * (gamma + 1 + mod[:, :d].float()) + (beta + mod[:, d:].float())
    assert rel(y32b[:, d:2 * d], ref) < 1e-5
    assert rel(y.float(), ref) < tol(dtype, f32=1e-5)
    assert rel(mean, mu[:, 0]) < 1e-5 and rel(rstd, 1 / torch.sqrt(var[:, 0] + 1e-5)) < 1e-5
    assert float(y32b[:, :d].abs().max()) == 0 and float(y32b[:, 2 * d:].abs().max()) == 0
    # plain LayerNorm without bias (custom LayerNorm of the reference)
    y2 = torch.empty(M, d, device=dev(), dtype=dtype)
    ops.rownorm([dict(X=x, gamma=gamma, Yact=y2)], M, d, False, False, 1e-5, dtype)
    assert rel(y2.float(), xh * gamma) < tol(dtype, f32=1e-5)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N_,K", [(128, 256), (256, 256), (64, 72), (48, 512)])
def test_gemm_rownorm_matches_gemm_then_rownorm(dtype, N_, K):
    """sea_gemm_rownorm = Linear (+ residual) followed by the AdaLN / LayerNorm row pass, in one launch: against the fp32 formula, and
    against the two-launch form it replaces (cross_down + ln_cross, proj + final norm), ragged M, strided fp32 output."""
    from sea_amd import ops

    M = 203
    groups, refs, keep = [], [], []
    for gi in range(3):
        A = rnd(M, K, dtype=dtype, seed=500 + gi)
        W = rnd(N_, K, dtype=dtype, scale=0.2, seed=510 + gi)
        bias = 0.3 * rnd(N_, seed=520 + gi) + 0.5
        R = rnd(M, N_, seed=530 + gi) if gi == 1 else None
        mod = rnd(M, 2 * N_, dtype=dtype, scale=0.5, seed=540 + gi) if gi != 2 else None
        gamma, beta = 1 + 0.1 * rnd(N_, seed=550 + gi), (0.1 * rnd(N_, seed=560 + gi) if gi != 2 else None)
        y = torch.empty(M, N_, device=dev(), dtype=dtype)
        y32b = torch.zeros(M, 3 * N_, device=dev())
        c32 = torch.empty(M, N_, device=dev())
        mean, rstd = torch.empty(M, device=dev()), torch.empty(M, device=dev())
        groups.append(dict(A=A, W=W, bias=bias, R=R, mod=mod, gamma=gamma, beta=beta, Yact=y, Y32=y32b[:, N_:2 * N_], C32=c32, mean=mean, rstd=rstd))
        v = A.float() @ W.float().t() + bias + (R if R is not None else 0)
        mu = v.mean(-1, keepdim=True)
        var = ((v - mu) ** 2).mean(-1, keepdim=True)
        xh = (v - mu) / torch.sqrt(var + 1e-5)
        ref = xh * (gamma + 1 + mod[:, :N_].float()) + (beta + mod[:, N_:].float()) if mod is not None else xh * gamma
        refs.append((v, mu[:, 0], 1 / torch.sqrt(var[:, 0] + 1e-5), ref))
        keep.append((y, y32b, c32, mean, rstd))
    ops.gemm_rownorm(groups, 1e-5, dtype)
    for (v, mu, rs, ref), (y, y32b, c32, mean, rstd) in zip(refs, keep):
        assert rel(c32, v) < 2e-5
        assert rel(y32b[:, N_:2 * N_], ref) < 3e-5
        assert rel(y.float(), ref) < tol(dtype, f32=3e-5)
        assert rel(mean, mu) < 2e-5 and rel(rstd, rs) < 2e-5
        assert float(y32b[:, :N_].abs().max()) == 0 and float(y32b[:, 2 * N_:].abs().max()) == 0
    # the two-launch form on the same operands: identical up to the fp32 rounding of the intermediate
    g0 = groups[0]
    c = torch.empty(M, N_, device=dev())
    y2 = torch.empty(M, N_, device=dev())
    ops.gemm_grouped([dict(A=g0["A"], W=g0["W"], bias=g0["bias"], C32=c)], dtype)
    ops.rownorm([dict(X=c, mod=g0["mod"], gamma=g0["gamma"], beta=g0["beta"], Y32=y2)], M, N_, False, False, 1e-5, dtype)
    assert rel(keep[0][1][:, N_:2 * N_], y2) < 1e-6


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M", [203, 2500])
def test_gemm_rownorm_segments_ib_addend(dtype, M):
    """The exchange tail of one field in one launch: x += cross_up(sum_j g_j) (segments, bias counted per segment, in-place residual),
    the copy cross_down reads is written BEFORE the info-bottleneck addend, AdaLN_2 is taken after it.  M = 2500 runs the 64-row tiles
    (forced through SEA_TUNE=gemm_norm_rows is not needed: the launch is longer than 512 row tiles only at M > 32768, so both shapes are forced by env in
    test_gemm_rownorm_tile_shapes_agree)."""
    from sea_amd import ops

    D, E, h, S = 128, 256, 8, 2
    gp = rnd(S, M, D, dtype=dtype, seed=600)
    W = rnd(E, D, dtype=dtype, scale=0.2, seed=601)
    bias = 0.2 * rnd(E, seed=602)
    x = rnd(M, E, seed=603)
    x0 = x.clone()
    mod = rnd(M, 2 * E, dtype=dtype, scale=0.5, seed=604)
    gamma, beta = 1 + 0.1 * rnd(E, seed=605), 0.1 * rnd(E, seed=606)
    c = torch.rand(M, device=dev())
    w1, b1 = rnd(h, seed=607), rnd(h, seed=608)
    lnw, lnb = 1 + 0.1 * rnd(h, seed=609), 0.1 * rnd(h, seed=610)
    w2, b2 = rnd(E, h, scale=0.3, seed=611), 0.1 * rnd(E, seed=612)
    xa = torch.empty(M, E, device=dev(), dtype=dtype)
    ne = torch.empty(M, E, device=dev(), dtype=dtype)
    ops.gemm_rownorm([dict(A=gp[0], W=W, bias=bias, bias_scale=float(S), n_seg=S, a_seg_stride=M * D, R=x, C32=x, Cact=xa, mod=mod, gamma=gamma, beta=beta,
                           Yact=ne, ib=dict(c=c, w1=w1, b1=b1, lnw=lnw, lnb=lnb, w2=w2, b2=b2, h=h))], 1e-5, dtype)
    pre = x0 + gp.float().sum(0) @ W.float().t() + S * bias
    hid = gelu(torch.nn.functional.layer_norm(c[:, None] * w1 + b1, (h,), lnw, lnb, 1e-5))
    v = pre + hid @ w2.t() + b2
    assert rel(xa.float(), pre) < tol(dtype, f32=2e-5)
    assert rel(x, v) < 2e-5
    ref = torch.nn.functional.layer_norm(v, (E,), None, None, 1e-5) * (gamma + 1 + mod[:, :E].float()) + (beta + mod[:, E:].float())
    assert rel(ne.float(), ref) < tol(dtype, f32=3e-5)


@pytest.mark.parametrize("D,E,S", [(128, 256, 2), (128, 256, 1), (64, 128, 2), (64, 128, 4)])
@pytest.mark.parametrize("M,has_down", [(203, True), (203, False), (2024, True)])
def test_exchange_tail_matches_three_launch_form(D, E, S, M, has_down):
    """sea_exchange_tail (cross-attention projections + GELU, up-projection of their sum + residual, down-projection + AdaLN of the updated rows) against
    the fp32 formula on the bf16 operands, and against the three launches it replaces (same kernels' arithmetic: agreement to bf16 rounding of g and x)."""
    from sea_amd import ops

    dt = torch.bfloat16
    att = [rnd(M, D, dtype=dt, seed=700 + s) for s in range(S)]
    Wp = [rnd(D, D, dtype=dt, scale=0.15, seed=710 + s) for s in range(S)]
    Wup, bup = rnd(E, D, dtype=dt, scale=0.1, seed=720), 0.2 * rnd(E, seed=721)
    Wd, bd = rnd(D, E, dtype=dt, scale=0.1, seed=722), 0.2 * rnd(D, seed=723)
    x = rnd(M, E, seed=724)
    x0 = x.clone()
    mod = rnd(M, 2 * D, dtype=dt, scale=0.5, seed=725)
    gamma, beta = 1 + 0.1 * rnd(D, seed=726), 0.1 * rnd(D, seed=727)
    xa = torch.empty(M, E, device=dev(), dtype=dt)
    nd = torch.empty(M, D, device=dev(), dtype=dt)
    assert ops.exchange_tail_supported(dt, D, E, S)
    ops.exchange_tail(att, Wp, Wup, bup, float(S), x, Xact=xa, down=dict(W=Wd, bias=bd, gamma=gamma, beta=beta, mod=mod, Yact=nd) if has_down else None)
    # cross_up is linear and shared by the segments: the kernel sums the GELU outputs in fp32 and rounds the SUM to bf16 (MFMA operand of the next layer)
    gs = sum(gelu(att[s].float() @ Wp[s].float().t()) for s in range(S)).to(dt).float()
    xn = x0 + gs @ Wup.float().t() + S * bup
    assert rel(x, xn) < 5e-5
    assert rel(xa.float(), xn) < 6e-3
    if has_down:
        v = xn.to(dt).float() @ Wd.float().t() + bd                                          # likewise the new x as operand of the down-projection
        ref = torch.nn.functional.layer_norm(v, (D,), None, None, 1e-5) * (gamma + 1 + mod[:, :D].float()) + (beta + mod[:, D:].float())
        assert rel(nd.float(), ref) < 6e-3
    # the three-launch form on the same operands
    gp = torch.empty(S, M, D, device=dev(), dtype=dt)
    ops.gemm_grouped([dict(A=att[s], W=Wp[s], Cact=gp[s], act=1) for s in range(S)], dt)
    x2, xa2 = x0.clone(), torch.empty_like(xa)
    ops.gemm_grouped([dict(A=gp[0], W=Wup, bias=bup, bias_scale=float(S), n_seg=S, a_seg_stride=M * D, R=x2, C32=x2, Cact=xa2)], dt)
    assert rel(x, x2) < 5e-3      # (that form rounds every g_s to bf16 and sums inside the MFMA; the update is as large as x here)
    if has_down:
        nd2 = torch.empty_like(nd)
        ops.gemm_rownorm([dict(A=xa2, W=Wd, bias=bd, mod=mod, gamma=gamma, beta=beta, Yact=nd2)], 1e-5, dt)
        assert rel(nd.float(), nd2.float()) < 2e-2


@pytest.mark.parametrize("D,E", [(128, 256), (64, 128)])
@pytest.mark.parametrize("B,T,form,rows", [(1, 203, "A", 0), (1, 203, "B", 0), (1, 2024, "A", 0), (1, 2024, "B", 0), (2, 1012, "A", 32), (3, 117, "B", 32), (2, 100, "A", 16)])
def test_row_chain_matches_the_separate_launches(D, E, B, T, form, rows, monkeypatch):
    """sea_row_chain against the launches it replaces AND the fp32 formulas on the bf16 operands: form A = self-attention out-projection + residual, form B = a
    field's exchange tail (projections + GELU, up-projection of the sum + residual); then cross_down + AdaLN of the updated rows and — from the NORMALISED rows,
    without a round trip through HBM — q for two pairs and k / v for two pairs with the rotary epilogue, written in the attention layouts (reference
    models/temporal.py:176-192, models/base_blocks.py:271-280).  16- and 32-row workgroups (`rows`; 0 = the launcher's choice); tiles that straddle
    trajectories (T not a multiple of 16); one group without projections and one without the down-projection beside the full one."""
    from sea_amd import ops

    if rows:
        monkeypatch.setenv("SEA_TUNE", f"chain_rows={rows}")   # (read once per process by the library: only the first forced value of a process takes effect)
    dt = torch.bfloat16
    M, H = B * T, D // 16
    hd = D // H
    cap = T + 8
    table = rope_table(hd, T)
    S = 2 if form == "B" else 0
    att = [rnd(M, D, dtype=dt, seed=800 + s) for s in range(S)]
    Wp = [rnd(D, D, dtype=dt, scale=0.15, seed=810 + s) for s in range(S)]
    K2 = D if form == "B" else E
    a2 = rnd(M, E, dtype=dt, seed=805) if form == "A" else None
    W2, b2 = rnd(E, K2, dtype=dt, scale=0.1, seed=820), (0.2 * rnd(E, seed=821) if form == "B" else None)
    Wd, bd = rnd(D, E, dtype=dt, scale=0.1, seed=822), 0.2 * rnd(D, seed=823)
    xin = rnd(M, 3 * E, seed=824)[:, E:2 * E]          # the residual rows inside a wider tensor (row stride 3 E: the caller's [B, T, F, E] layout)
    mod = rnd(M, 2 * D, dtype=dt, scale=0.5, seed=825)
    gamma, beta = 1 + 0.1 * rnd(D, seed=826), 0.1 * rnd(D, seed=827)
    Wq = [rnd(D, D, dtype=dt, scale=D ** -0.5, seed=830 + k) for k in range(2)]
    bq = [rnd(D, seed=834 + k) for k in range(2)]
    Wkv = [rnd(2 * D, D, dtype=dt, scale=D ** -0.5, seed=840 + k) for k in range(2)]
    bkv = [rnd(2 * D, seed=844 + k) for k in range(2)]
    new = lambda *shape: torch.zeros(*shape, device=dev(), dtype=dt)   # noqa: E731
    Q = [new(B, H, T, hd) for _ in range(2)]
    Kc, Vt = [new(B, H, cap, hd) for _ in range(2)], [new(B, H, hd, cap) for _ in range(2)]
    x, xa, nd = torch.empty(M, E, device=dev()), new(M, E), new(M, D)
    x_b, x_c, nd_b = torch.empty(M, E, device=dev()), torch.empty(M, E, device=dev()), new(M, D)
    qs = ops.q_scale(hd)
    proj = [dict(W=Wq[0], bias=bq[0], col0=0, Q=Q[0]), dict(W=Wkv[0], bias=bkv[0], col0=D, K=Kc[0], Vt=Vt[0]),
            dict(W=Wq[1], bias=bq[1], col0=0, Q=Q[1]), dict(W=Wkv[1], bias=bkv[1], col0=D, K=Kc[1], Vt=Vt[1])]
    common = dict(W2=W2, b2=b2, bias_scale=float(max(S, 1)), Xin=xin, att=att, Wp=Wp, a2=a2)
    down = dict(W=Wd, bias=bd, gamma=gamma, beta=beta, mod=mod)
    assert ops.row_chain_supported(dt, D, E, S, hd)
    ops.row_chain([dict(common, X=x, Xact=xa, down=dict(down, Yact=nd), proj=proj),       # everything
                   dict(common, X=x_b, down=dict(down, Yact=nd_b)),                        # no projections
                   dict(common, X=x_c)],                                                    # no down-projection either (the last field's tail)
                  rope=table, H=H, hd=hd, T=T, pos0=0, cap=cap, q_scale_=qs)
    # fp32 formulas on the bf16 operands (intermediates rounded where the kernel rounds them: g, x and y are MFMA operands of the next layer)
    if form == "B":
        gs = sum(gelu(att[s].float() @ Wp[s].float().t()) for s in range(S)).to(dt).float()
        xn = xin + gs @ W2.float().t() + S * b2
    else:
        xn = xin + a2.float() @ W2.float().t()
    for got in (x, x_b, x_c):
        assert rel(got, xn) < 5e-5
    assert rel(xa.float(), xn) < 6e-3
    v = xn.to(dt).float() @ Wd.float().t() + bd
    y = torch.nn.functional.layer_norm(v, (D,), None, None, 1e-5) * (gamma + 1 + mod[:, :D].float()) + (beta + mod[:, D:].float())
    assert rel(nd.float(), y) < 6e-3 and rel(nd_b.float(), y) < 6e-3
    yb = y.to(dt).float()
    cos, sin = table[:T, :, 0], table[:T, :, 1]
    for k in range(2):
        q = rope_ref((yb @ Wq[k].float().t() + bq[k]).view(B, T, H, hd), cos, sin) * qs
        kv = yb @ Wkv[k].float().t() + bkv[k]
        kk = rope_ref(kv[:, :D].view(B, T, H, hd), cos, sin)
        vv = kv[:, D:].view(B, T, H, hd)
        assert rel(Q[k].float(), q.permute(0, 2, 1, 3)) < 1.2e-2, k
        assert rel(Kc[k][:, :, :T].float(), kk.permute(0, 2, 1, 3)) < 1.2e-2, k
        assert rel(Vt[k][:, :, :, :T].float(), vv.permute(0, 2, 3, 1)) < 1.2e-2, k
        assert float(Kc[k][:, :, T:].abs().max()) == 0 and float(Vt[k][:, :, :, T:].abs().max()) == 0   # nothing beyond the rows of this call
    # the launches the chain replaces, on the same operands
    x2, xa2 = torch.empty(M, E, device=dev()), new(M, E)
    if form == "B":
        ops.exchange_tail(att, Wp, W2, b2, float(S), x2.copy_(xin), Xact=xa2)
    else:
        ops.gemm_grouped([dict(A=a2, W=W2, R=xin.contiguous(), C32=x2, Cact=xa2)], dt)
    assert rel(x, x2) < 2e-5
    nd2 = new(M, D)
    ops.gemm_rownorm([dict(A=xa2, W=Wd, bias=bd, mod=mod, gamma=gamma, beta=beta, Yact=nd2)], 1e-5, dt)
    assert rel(nd.float(), nd2.float()) < 1e-2
    Q2, K2_, V2 = new(B, H, T, hd), new(B, H, cap, hd), new(B, H, hd, cap)
    ops.qkv_rope_grouped([dict(A=nd2, W=Wq[1], bias=bq[1], col0=0, Q=Q2), dict(A=nd2, W=Wkv[1], bias=bkv[1], col0=D, K=K2_, Vt=V2)], table, H, hd, T, 0, cap, qs, dt)
    assert rel(Q[1].float(), Q2.float()) < 1.5e-2 and rel(Kc[1].float(), K2_.float()) < 1.5e-2 and rel(Vt[1].float(), V2.float()) < 1.5e-2


def test_row_chain_unsupported_shapes_are_refused():
    from sea_amd import ops

    assert not ops.row_chain_supported(torch.float32, 128, 256, 2, 16) and not ops.row_chain_supported(torch.bfloat16, 128, 256, 3, 16)
    assert not ops.row_chain_supported(torch.bfloat16, 256, 512, 0, 32)
    with pytest.raises(RuntimeError, match="unsupported"):
        ops.row_chain([dict(W2=rnd(64, 64, dtype=torch.bfloat16), Xin=rnd(8, 64), X=rnd(8, 64), a2=rnd(8, 64, dtype=torch.bfloat16))])


@pytest.mark.parametrize("M,d", [(203, 256), (2024, 256), (140, 128), (300, 64), (65, 512), (16192, 256), (4100, 512)])   # the last two: 128-row tiles
def test_gemm_adaln_matches_the_two_launch_form(M, d):
    """sea_gemm_adaln (AdaLN as the epilogue of cond_mlp.2's GEMM: the [M, 2 d] modulation matrix is never stored) against the fp32 formula of
    models/base_blocks.py:343-350 on the bf16 operands, against sea_gemm_grouped + sea_rownorm on the same operands, and a plain group (modulation stored) in
    the same launch; X is read through a row stride (the caller's [B, T, F, E] layout), statistics are saved."""
    from sea_amd import ops

    dt = torch.bfloat16
    K = 2 * d
    hid = rnd(M, K, dtype=dt, scale=0.5, seed=900)
    W = rnd(2 * d, K, dtype=dt, scale=K ** -0.5, seed=901)
    b2 = 0.3 * rnd(2 * d, seed=902)
    xw = rnd(M, 3 * d, seed=903)
    x = xw[:, d:2 * d]
    gamma, beta = 1 + 0.1 * rnd(d, seed=904), 0.1 * rnd(d, seed=905)
    y = torch.empty(M, d, device=dev(), dtype=dt)
    y32 = torch.empty(M, d, device=dev())
    mod_out = torch.empty(M, 2 * d, device=dev(), dtype=dt)
    mean, rstd = torch.empty(M, device=dev()), torch.empty(M, device=dev())
    ops.gemm_adaln([dict(A=hid, W=W, bias=b2, X=x, gamma=gamma, beta=beta, Yact=y, Y32=y32, mean=mean, rstd=rstd),
                    dict(A=hid, W=W, bias=b2, Yact=mod_out)])
    mod = hid.float() @ W.float().t() + b2
    xhat = torch.nn.functional.layer_norm(x, (d,), None, None, 1e-5)
    ref = xhat * (gamma + 1 + mod[:, :d]) + (beta + mod[:, d:])
    assert rel(y32, ref) < 2e-5                      # (the modulation stays fp32 here: closer to the formula than the two-launch form, which rounds it to bf16)
    assert rel(y.float(), ref) < 6e-3
    assert rel(mod_out.float(), mod) < 6e-3
    assert rel(mean, x.mean(dim=1)) < 1e-5 and rel(rstd, 1.0 / torch.sqrt(x.var(dim=1, unbiased=False) + 1e-5)) < 1e-5
    # the two launches it replaces
    mod2 = torch.empty(M, 2 * d, device=dev(), dtype=dt)
    ops.gemm_grouped([dict(A=hid, W=W, bias=b2, Cact=mod2)], dt)
    assert rel(mod_out.float(), mod2.float()) < 1e-5
    y2 = torch.empty(M, d, device=dev(), dtype=dt)
    ops.rownorm([dict(X=x.contiguous(), mod=mod2, gamma=gamma, beta=beta, Yact=y2)], M, d, False, False, 1e-5, dt)
    assert rel(y.float(), y2.float()) < 1e-2


def test_gemm_adaln_is_bf16_only():
    from sea_amd import ops

    with pytest.raises(RuntimeError, match="bf16 only"):
        ops.gemm_adaln([dict(A=rnd(8, 64), W=rnd(64, 64), Yact=rnd(8, 64))], dtype=torch.float32)


def test_exchange_tail_unsupported_shapes_are_refused():
    from sea_amd import ops

    assert not ops.exchange_tail_supported(torch.float32, 128, 256, 2) and not ops.exchange_tail_supported(torch.bfloat16, 128, 256, 3)
    att, Wp = [rnd(8, 32, dtype=torch.bfloat16)], [rnd(32, 32, dtype=torch.bfloat16)]
    with pytest.raises(RuntimeError, match="unsupported"):
        ops.exchange_tail(att, Wp, rnd(64, 32, dtype=torch.bfloat16), None, 1.0, rnd(8, 64))


def test_gemm_rownorm_rejects_bad_shapes():
    from sea_amd import ops

    A, W = rnd(8, 64), rnd(260, 64)
    with pytest.raises(RuntimeError, match="N a multiple of 16 up to 256"):
        ops.gemm_rownorm([dict(A=A, W=W, gamma=rnd(260), Y32=torch.empty(8, 260, device=dev()))], 1e-5, torch.float32)


@pytest.mark.parametrize("dtype", DTYPES)
def test_silu_launch_evaluates_ib_and_rownorm_adds_it(dtype):
    """The info-bottleneck add without a launch of its own: sea_silu_outer_ib stores ib(c) (two blocks' MLPs here) beside the silu rows, and
    sea_rownorm's addend adds it, writes x + ib back and normalises that — against sea_ib_add followed by the plain row norm (bitwise) and the formula."""
    import ctypes as C
    from sea_amd import _native as N, ops

    M, E, h, K2 = 203, 256, 8, 512
    c = torch.rand(M, device=dev())
    w1s, b1s, hid = rnd(K2, seed=900), rnd(K2, seed=901), torch.empty(M, K2, device=dev(), dtype=dtype)
    arr = (N.SeaSiluGroup * 1)()
    arr[0].w1, arr[0].b1, arr[0].Hid, arr[0].K2, arr[0].ld = w1s.data_ptr(), b1s.data_ptr(), hid.data_ptr(), K2, K2
    ibs = (N.SeaIbParams * 2)()
    prm, bufs = [], []
    for k in range(2):
        q = dict(w1=rnd(h, seed=910 + k), b1=rnd(h, seed=920 + k), lnw=1 + 0.1 * rnd(h, seed=930 + k), lnb=0.1 * rnd(h, seed=940 + k),
                 w2=rnd(E, h, scale=0.3, seed=950 + k), b2=0.1 * rnd(E, seed=960 + k))
        buf = torch.full((M, E), float("nan"), device=dev())
        ibs[k].X[0], ibs[k].n_fields, ibs[k].ldx, ibs[k].M, ibs[k].E, ibs[k].h = buf.data_ptr(), 1, E, M, E, h
        ibs[k].w1, ibs[k].b1, ibs[k].lnw, ibs[k].lnb, ibs[k].w2, ibs[k].b2 = (q[n].data_ptr() for n in ("w1", "b1", "lnw", "lnb", "w2", "b2"))
        prm.append(q)
        bufs.append(buf)
    N.check(N.lib().sea_silu_outer_ib(arr, 1, c.data_ptr(), M, N.dtype_code(dtype), ibs, 2, N.stream_ptr()), "sea_silu_outer_ib")
    assert rel(hid.float(), torch.nn.functional.silu(c[:, None] * w1s + b1s)) < tol(dtype, f32=1e-5)
    for q, buf in zip(prm, bufs):
        ref = gelu(torch.nn.functional.layer_norm(c[:, None] * q["w1"] + q["b1"], (h,), q["lnw"], q["lnb"], 1e-5)) @ q["w2"].t() + q["b2"]
        assert rel(buf, ref) < 2e-5
    # the add rides in the row norm
    x = rnd(M, E, seed=970) * 1.3
    mod = rnd(M, 2 * E, dtype=dtype, scale=0.5, seed=971)
    gamma, beta = 1 + 0.1 * rnd(E, seed=972), 0.1 * rnd(E, seed=973)
    xa, ya = x.clone(), torch.empty(M, E, device=dev(), dtype=dtype)
    ops.rownorm([dict(X=xa, addend=bufs[0], Xout=xa, mod=mod, gamma=gamma, beta=beta, Yact=ya)], M, E, False, False, 1e-5, dtype)
    xb, yb = x.clone(), torch.empty(M, E, device=dev(), dtype=dtype)
    q = prm[0]
    ops.ib_add([xb], c, q["w1"], q["b1"], q["lnw"], q["lnb"], q["w2"], q["b2"])
    ops.rownorm([dict(X=xb, mod=mod, gamma=gamma, beta=beta, Yact=yb)], M, E, False, False, 1e-5, dtype)
    assert rel(xa, xb) < 1e-6 and rel(ya.float(), yb.float()) < tol(dtype, f32=1e-6, bf16=4e-3)
    assert rel(xa, x + bufs[0]) < 1e-7


@pytest.mark.parametrize("E,S", [(256, 2048), (128, 1024)])
@pytest.mark.parametrize("M", [77, 2024])
def test_mlp_fc1_ln_gelu_matches_two_launch_form(E, S, M):
    """sea_mlp_fc1_ln_gelu (Linear + nn.LayerNorm + GELU with 32 complete hidden rows per workgroup) against the fp32 formula on the bf16 operands and
    against the two launches it replaces (sea_gemm_grouped, sea_rownorm with the GELU epilogue — which round the pre-activation to bf16 in between)."""
    from sea_amd import ops

    dt = torch.bfloat16
    groups, refs, keep = [], [], []
    for i in range(3):
        A = rnd(M, E, dtype=dt, seed=1300 + i)
        W1, b1 = rnd(S, E, dtype=dt, scale=0.08, seed=1310 + i), 0.3 * rnd(S, seed=1320 + i)
        lnw, lnb = 1 + 0.1 * rnd(S, seed=1330 + i), 0.1 * rnd(S, seed=1340 + i)
        Hg = torch.full((M, S), float("nan"), device=dev(), dtype=dt)
        groups.append(dict(A=A, W1=W1, b1=b1, lnw=lnw, lnb=lnb, Hg=Hg))
        refs.append(gelu(torch.nn.functional.layer_norm(A.float() @ W1.float().t() + b1, (S,), lnw, lnb, 1e-5)))
        keep.append(Hg)
    assert ops.mlp_fc1_supported(dt, E, S)
    ops.mlp_fc1_ln_gelu(groups)
    for ref, Hg, d in zip(refs, keep, groups):
        assert rel(Hg.float(), ref) < 6e-3
        h = torch.empty(M, S, device=dev(), dtype=dt)
        hg2 = torch.empty(M, S, device=dev(), dtype=dt)
        ops.gemm_grouped([dict(A=d["A"], W=d["W1"], bias=d["b1"], Cact=h)], dt)
        ops.rownorm([dict(X=h, gamma=d["lnw"], beta=d["lnb"], Yact=hg2)], M, S, True, True, 1e-5, dt)
        assert rel(Hg.float(), hg2.float()) < 8e-3


@pytest.mark.parametrize("E,S", [(256, 2048), (128, 1024)])
@pytest.mark.parametrize("M,adaln,with_add", [(77, True, True), (2024, True, True), (333, False, False), (64, True, False)])
def test_mlp_fc1_ln_gelu_norm_prologue(E, S, M, adaln, with_add):
    """The operand rows produced inside the launch (x + addend, AdaLN_2 / LayerNorm of the fp32 residual stream) against the row pass + the launch on its
    ready-made rows: x + addend written back exactly, hidden rows within the bf16 rounding of the normalised rows (statistics are summed in another order)."""
    from sea_amd import ops

    dt = torch.bfloat16
    groups, groups2, keep = [], [], []
    for i in range(3):
        x = rnd(M, E, seed=1400 + i) * 1.7 + 0.3
        add = 0.2 * rnd(M, E, seed=1410 + i) if with_add else None
        mod = (0.3 * rnd(M, 2 * E, seed=1420 + i)).to(dt) if adaln else None
        gamma, beta = 1 + 0.1 * rnd(E, seed=1430 + i), (0.1 * rnd(E, seed=1440 + i) if adaln else None)
        W1, b1 = rnd(S, E, dtype=dt, scale=0.08, seed=1450 + i), 0.3 * rnd(S, seed=1460 + i)
        lnw, lnb = 1 + 0.1 * rnd(S, seed=1470 + i), 0.1 * rnd(S, seed=1480 + i)
        Hg = torch.full((M, S), float("nan"), device=dev(), dtype=dt)
        x_in = x.clone()
        groups.append(dict(W1=W1, b1=b1, lnw=lnw, lnb=lnb, Hg=Hg,
                           norm=dict(X32=x_in, addend=add, Xout=(x_in if with_add else None), mod=mod, gamma=gamma, beta=beta)))
        # the two-launch form on the same inputs
        x2, n2, Hg2 = x.clone(), torch.empty(M, E, device=dev(), dtype=dt), torch.empty(M, S, device=dev(), dtype=dt)
        nd = dict(X=x2, Yact=n2, gamma=gamma, beta=beta, mod=mod)
        if with_add:
            nd.update(addend=add, Xout=x2)
        ops.rownorm([nd], M, E, False, False, 1e-5, dt)
        groups2.append(dict(A=n2, W1=W1, b1=b1, lnw=lnw, lnb=lnb, Hg=Hg2))
        keep.append((x_in, x2, Hg, Hg2))
    ops.mlp_fc1_ln_gelu(groups)
    ops.mlp_fc1_ln_gelu(groups2)
    for x_in, x2, Hg, Hg2 in keep:
        assert torch.equal(x_in, x2)
        assert bool(torch.isfinite(Hg.float()).all())
        assert rel(Hg.float(), Hg2.float()) < 8e-3


@pytest.mark.parametrize("E,S", [(256, 2048), (128, 1024)])
@pytest.mark.parametrize("M,norm", [(77, "adaln"), (2024, "adaln"), (333, "ln"), (64, None), (9000, "adaln")])
def test_mlp_fc2_proj_norm_matches_three_launch_form(E, S, M, norm):
    """sea_mlp_fc2_proj_norm (fc2 + residual, proj, the final norm: 32 complete rows per workgroup through both layers) against the fp32 formula on the
    bf16 operands and against the three launches it replaces (which round x3 to bf16 between the layers, as this kernel does)."""
    from sea_amd import ops

    dt = torch.bfloat16
    groups, refs, keep = [], [], []
    for i in range(3):
        Hg = rnd(M, S, dtype=dt, seed=1500 + i)
        W2, b2 = rnd(E, S, dtype=dt, scale=0.03, seed=1510 + i), 0.2 * rnd(E, seed=1520 + i)
        R = rnd(M, E, seed=1530 + i)
        Wp, bp = rnd(E, E, dtype=dt, scale=0.08, seed=1540 + i), 0.2 * rnd(E, seed=1550 + i)
        gamma = 1 + 0.1 * rnd(E, seed=1560 + i) if norm else None
        beta = 0.1 * rnd(E, seed=1570 + i) if norm == "adaln" else None
        mod = (0.3 * rnd(M, 2 * E, seed=1580 + i)).to(dt) if norm == "adaln" else None
        out = torch.full((M, 3 * E), float("nan"), device=dev())                 # the caller's [M, F, E] output, this field's column range
        y32 = out[:, i * E:(i + 1) * E]
        groups.append(dict(Hg=Hg, W2=W2, b2=b2, R=R, Wproj=Wp, bproj=bp, Y32=y32, gamma=gamma, beta=beta, mod=mod))
        x3 = (Hg.float() @ W2.float().t() + b2 + R).to(dt).float()
        y = x3 @ Wp.float().t() + bp
        if norm:
            mu, var = y.mean(1, keepdim=True), y.var(1, unbiased=False, keepdim=True)
            yh = (y - mu) / torch.sqrt(var + 1e-5)
            if norm == "adaln":
                y = yh * (gamma + 1 + mod[:, :E].float()) + beta + mod[:, E:].float()
            else:
                y = yh * gamma
        refs.append(y)
        keep.append(y32)
    ops.mlp_fc2_proj_norm(groups)
    for ref, y32, d in zip(refs, keep, groups):
        assert bool(torch.isfinite(y32).all())
        assert rel(y32, ref) < 6e-3
        # the three launches
        xm = torch.empty(M, E, device=dev(), dtype=dt)
        y2 = torch.empty(M, E, device=dev())
        ops.gemm_grouped([dict(A=d["Hg"], W=d["W2"], bias=d["b2"], R=d["R"], Cact=xm)], dt)
        ops.gemm_grouped([dict(A=xm, W=d["Wproj"], bias=d["bproj"], C32=y2)], dt)
        if norm:
            nd = dict(X=y2, Y32=y2, gamma=d["gamma"], beta=d["beta"], mod=d["mod"])
            ops.rownorm([nd], M, E, False, False, 1e-5, dt)
        assert rel(y32, y2) < 2e-3


@pytest.mark.parametrize("third", [False, True])
@pytest.mark.parametrize("B,T,H,pos0,cap", [(1, 2024, 8, 0, 2024), (3, 70, 8, 0, 72), (2, 45, 16, 8, 64), (1, 33, 8, 3, 40)])
def test_adaln_qkv_matches_the_three_launches(B, T, H, pos0, cap, third):
    """sea_adaln_qkv (the front of a block in one launch: condition MLP with generated hidden rows, AdaLN_0, q / k / v + rotary epilogue) against
    sea_silu_outer + sea_gemm_adaln + sea_qkv_rope_grouped on the same operands and against the fp32 formulas (models/base_blocks.py:337-350, 176-190, 300-324);
    three fields per launch, X read through the caller's row stride, two rider groups (plain GEMMs of a later launch) beside them; rows that cross trajectories,
    an unaligned cache position (element stores of V^T) and both head dims."""
    from sea_amd import ops

    dt = torch.bfloat16
    E, M = 256, B * T
    hd = E // H
    table = rope_table(hd, pos0 + T)
    cond = torch.rand(M, device=dev())
    xw = rnd(M, 3 * E, seed=2000)
    groups, refs, keep, thirds = [], [], [], []
    for i in range(3):
        w1, b1 = rnd(2 * E, seed=2010 + i), rnd(2 * E, seed=2020 + i)
        W2c, b2c = rnd(2 * E, 2 * E, dtype=dt, scale=(2 * E) ** -0.5, seed=2030 + i), 0.3 * rnd(2 * E, seed=2040 + i)
        gamma, beta = 1 + 0.1 * rnd(E, seed=2050 + i), 0.1 * rnd(E, seed=2060 + i)
        Wqkv, bqkv = rnd(3 * E, E, dtype=dt, scale=E ** -0.5, seed=2070 + i), rnd(3 * E, seed=2080 + i)
        x = xw[:, i * E:(i + 1) * E]
        Q = torch.zeros(B, H, T, hd, device=dev(), dtype=dt)
        K = torch.zeros(B, H, cap, hd, device=dev(), dtype=dt)
        Vt = torch.zeros(B, H, hd, cap, device=dev(), dtype=dt)
        groups.append(dict(X=x, cond=cond, w1=w1, b1=b1, W2c=W2c, b2c=b2c, gamma=gamma, beta=beta, Wqkv=Wqkv, bqkv=bqkv, Q=Q, K=K, Vt=Vt))
        if third:   # the modulation of a second module of the same rows (ln_cross: 256 columns), stored
            w13, b13 = rnd(256, seed=2110 + i), rnd(256, seed=2120 + i)
            W3, b3 = rnd(256, 256, dtype=dt, scale=1 / 16, seed=2130 + i), 0.3 * rnd(256, seed=2140 + i)
            o3 = torch.full((M, 256), float("nan"), device=dev(), dtype=dt)
            groups[-1]["third"] = dict(w1=w13, b1=b13, W=W3, bias=b3, out=o3)
            h3 = torch.nn.functional.silu(cond[:, None] * w13 + b13).to(dt).float()
            thirds.append((o3, h3 @ W3.float().t() + b3))
        # the three launches
        hid = torch.empty(M, 2 * E, device=dev(), dtype=dt)
        ops.silu_outer([dict(w1=w1, b1=b1, Hid=hid)], cond, M, dt)
        n_e = torch.empty(M, E, device=dev(), dtype=dt)
        ops.gemm_adaln([dict(A=hid, W=W2c, bias=b2c, X=x, gamma=gamma, beta=beta, Yact=n_e)])
        Q2, K2, Vt2 = torch.zeros_like(Q), torch.zeros_like(K), torch.zeros_like(Vt)
        ops.qkv_rope_grouped([dict(A=n_e, W=Wqkv, bias=bqkv, col0=0, Q=Q2, K=K2, Vt=Vt2)], table, H, hd, T, pos0, cap, ops.q_scale(hd), dt)
        keep.append((Q, K, Vt, Q2, K2, Vt2))
        # fp32 formula on the bf16 operands (hidden rows, modulation and normalised rows rounded to bf16 where the launches round them)
        h = torch.nn.functional.silu(cond[:, None] * w1 + b1).to(dt).float()
        mod = (h @ W2c.float().t() + b2c).to(dt).float()
        y = (torch.nn.functional.layer_norm(x, (E,), None, None, 1e-5) * (gamma + 1 + mod[:, :E]) + beta + mod[:, E:]).to(dt).float()
        qkv = y @ Wqkv.float().t() + bqkv
        q, k, v = (qkv[:, j * E:(j + 1) * E].view(B, T, H, hd) for j in range(3))
        cos, sin = table[pos0:pos0 + T, :, 0], table[pos0:pos0 + T, :, 1]
        refs.append((rope_ref(q, cos, sin) * ops.q_scale(hd), rope_ref(k, cos, sin), v))
    rA, rW, rb = rnd(M, 512, dtype=dt, seed=2100), rnd(256, 512, dtype=dt, scale=0.05, seed=2101), rnd(256, seed=2102)
    rC = [torch.full((M, 256), float("nan"), device=dev(), dtype=dt) for _ in range(2)]
    sw = [(rnd(512, seed=2150 + k), rnd(512, seed=2160 + k), torch.full((M, 512), float("nan"), device=dev(), dtype=dt)) for k in range(2)]   # row riders: hidden rows of two other modules
    ops.adaln_qkv(groups, table, H, hd, T, pos0, cap, ops.q_scale(hd), riders=[dict(A=rA, W=rW, bias=rb, Cact=rC[0]), dict(A=rA, W=rW, Cact=rC[1])],
                  silu=[dict(w1=w_, b1=b_, Hid=h_) for w_, b_, h_ in sw], silu_c=cond)
    for w_, b_, h_ in sw:
        assert rel(h_.float(), torch.nn.functional.silu(cond[:, None] * w_ + b_)) < 6e-3
    for o3, ref3 in thirds:
        assert rel(o3.float(), ref3) < 6e-3
    for (Q, K, Vt, Q2, K2, Vt2), (q, k, v) in zip(keep, refs):
        assert rel(Q.float(), Q2.float()) < 1e-2 and rel(K.float(), K2.float()) < 1e-2 and rel(Vt.float(), Vt2.float()) < 1e-2
        assert rel(Q.float(), q.permute(0, 2, 1, 3)) < 1.5e-2
        assert rel(K[:, :, pos0:pos0 + T].float(), k.permute(0, 2, 1, 3)) < 1.5e-2
        assert rel(Vt[:, :, :, pos0:pos0 + T].float(), v.permute(0, 2, 3, 1)) < 1.5e-2
        if cap > pos0 + T:   # nothing outside the written window is touched
            assert float(K[:, :, pos0 + T:].abs().max()) == 0 and float(Vt[:, :, :, pos0 + T:].abs().max()) == 0
        if pos0 > 0:
            assert float(K[:, :, :pos0].abs().max()) == 0 and float(Vt[:, :, :, :pos0].abs().max()) == 0
    ref_r = rA.float() @ rW.float().t()
    assert rel(rC[0].float(), ref_r + rb) < 6e-3 and rel(rC[1].float(), ref_r) < 6e-3


@pytest.mark.parametrize("dtype", DTYPES)
def test_splitk_finish_sums_partials_in_order(dtype):
    """sea_splitk_finish against torch: S partial matrices (a strided view), bias with its scale, residual, both outputs; and K-slices of a product through
    sea_gemm_grouped + the finish against the un-split launch."""
    from sea_amd import ops

    S, M, N = 4, 203, 512
    Pw = rnd(S, M, N + 64, seed=2200)
    P = Pw[:, :, :N]
    bias, R = rnd(N, seed=2201), rnd(M, N, seed=2202)
    C32, Cact = torch.empty(M, N, device=dev()), torch.empty(M, N, device=dev(), dtype=dtype)
    ops.splitk_finish([dict(P=P, bias=bias, bias_scale=2.0, R=R, C32=C32, Cact=Cact)], dtype)
    ref = ((P[0] + P[1]) + P[2]) + P[3] + bias * 2.0 + R
    assert rel(C32, ref) < 1e-6 and rel(Cact.float(), ref) < tol(dtype, f32=1e-6)
    if dtype == torch.bfloat16:
        K = 8192
        A, W = rnd(M, K, dtype=dtype, seed=2203), rnd(N, K, dtype=dtype, scale=K ** -0.5, seed=2204)
        whole = torch.empty(M, N, device=dev())
        ops.gemm_grouped([dict(A=A, W=W, bias=bias, R=R, C32=whole)], dtype)
        parts = torch.empty(S, M, N, device=dev())
        ks = K // S
        ops.gemm_grouped([dict(A=A[:, q * ks:(q + 1) * ks], W=W[:, q * ks:(q + 1) * ks], C32=parts[q]) for q in range(S)], dtype)
        out = torch.empty(M, N, device=dev())
        ops.splitk_finish([dict(P=parts, bias=bias, R=R, C32=out)], dtype)
        assert rel(out, whole) < 1e-5


def test_adaln_qkv_refuses_other_shapes():
    from sea_amd import ops

    assert ops.adaln_qkv_supported(torch.bfloat16, 256, 8) and not ops.adaln_qkv_supported(torch.bfloat16, 128, 8) and not ops.adaln_qkv_supported(torch.float32, 256, 8)
    assert not ops.adaln_qkv_supported(torch.bfloat16, 256, 4)   # head dim 64


@pytest.mark.parametrize("E,S,M", [(256, 2048, 2024), (256, 2048, 33), (128, 1024, 1000), (256, 2048, 9000)])
@pytest.mark.parametrize("prologue", [None, "adaln_add", "ln"])
@pytest.mark.parametrize("norm", [None, "adaln"])
def test_mlp_block_matches_the_two_fused_launches(E, S, M, prologue, norm):
    """sea_mlp_block (the field MLP, proj and the final norm in one launch; the activated hidden rows stay in registers) against sea_mlp_fc1_ln_gelu followed by
    sea_mlp_fc2_proj_norm on the same operands — the same arithmetic up to the summation order of the second layer's contraction (eight partial tiles, added
    in wave order) — and against the fp32 formula of models/base_blocks.py:22-25 / models/temporal.py:139-146, 412-415.  Three fields per launch, the output
    written into the caller's strided [M, F, E] layout; with a norm prologue the residual is x + ib formed inside the launch (R = None)."""
    from sea_amd import ops

    if M == 9000 and (prologue == "ln" or norm is None):
        pytest.skip("the several-round case once per output form")
    dt = torch.bfloat16
    groups, two1, two2, outs, outs2, refs = [], [], [], [], [], []
    out = torch.full((M, 3 * E), float("nan"), device=dev())
    out2 = torch.full((M, 3 * E), float("nan"), device=dev())
    for i in range(3):
        W1, b1 = rnd(S, E, dtype=dt, scale=0.08, seed=1600 + i), 0.2 * rnd(S, seed=1610 + i)
        lnw, lnb = 1 + 0.1 * rnd(S, seed=1620 + i), 0.1 * rnd(S, seed=1630 + i)
        W2, b2 = rnd(E, S, dtype=dt, scale=0.03, seed=1640 + i), 0.2 * rnd(E, seed=1650 + i)
        Wp, bp = rnd(E, E, dtype=dt, scale=0.08, seed=1660 + i), 0.2 * rnd(E, seed=1670 + i)
        x = rnd(M, E, seed=1680 + i)
        gamma = 1 + 0.1 * rnd(E, seed=1690 + i) if norm else None
        beta = 0.1 * rnd(E, seed=1700 + i) if norm else None
        mod = (0.3 * rnd(M, 2 * E, seed=1710 + i)).to(dt) if norm else None
        first = dict(W1=W1, b1=b1, lnw=lnw, lnb=lnb)
        if prologue is None:
            A = rnd(M, E, dtype=dt, seed=1720 + i)
            first["A"] = A
            a_rows, res, R = A.float(), x, x
        else:
            add = 0.5 * rnd(M, E, seed=1730 + i) if prologue == "adaln_add" else None
            pg, pb = 1 + 0.1 * rnd(E, seed=1740 + i), (0.1 * rnd(E, seed=1750 + i) if prologue == "adaln_add" else None)
            pmod = (0.3 * rnd(M, 2 * E, seed=1760 + i)).to(dt) if prologue == "adaln_add" else None
            first["norm"] = dict(X32=x, gamma=pg, beta=pb, mod=pmod, addend=add)
            res = x + add if add is not None else x
            xh = torch.nn.functional.layer_norm(res, (E,), None, None, 1e-5)
            a_rows = (xh * (pg + 1 + pmod[:, :E].float()) + pb + pmod[:, E:].float()) if pmod is not None else xh * pg
            a_rows = a_rows.to(dt).float()
            R = None
        y32 = out[:, i * E:(i + 1) * E]
        groups.append(dict(**first, W2=W2, b2=b2, R=R, Wproj=Wp, bproj=bp, Y32=y32, gamma=gamma, beta=beta, mod=mod))
        # the two launches: (the prologue form writes x + ib to rows of its own, which the second launch reads as its residual)
        hg = torch.empty(M, S, device=dev(), dtype=dt)
        f1 = dict(first, Hg=hg)
        if prologue is not None:
            xq = torch.empty(M, E, device=dev())
            f1["norm"] = dict(first["norm"], Xout=xq)
        two1.append(f1)
        two2.append(dict(Hg=hg, W2=W2, b2=b2, R=(xq if prologue is not None else x), Wproj=Wp, bproj=bp, Y32=out2[:, i * E:(i + 1) * E], gamma=gamma, beta=beta, mod=mod))
        h = gelu(torch.nn.functional.layer_norm(a_rows @ W1.float().t() + b1, (S,), lnw, lnb, 1e-5)).to(dt).float()
        x3 = (h @ W2.float().t() + b2 + res).to(dt).float()
        y = x3 @ Wp.float().t() + bp
        if norm:
            yh = torch.nn.functional.layer_norm(y, (E,), None, None, 1e-5)
            y = yh * (gamma + 1 + mod[:, :E].float()) + beta + mod[:, E:].float()
        refs.append(y)
    ops.mlp_block(groups)
    ops.mlp_fc1_ln_gelu(two1)
    ops.mlp_fc2_proj_norm(two2)
    for i, ref in enumerate(refs):
        y32, y2 = out[:, i * E:(i + 1) * E], out2[:, i * E:(i + 1) * E]
        assert bool(torch.isfinite(y32).all())
        assert rel(y32, y2) < 3e-3, (i, rel(y32, y2))       # (x3 is rounded to bf16 in both; the fp32 sums in front of that rounding differ in order)
        assert rel(y32, ref) < 8e-3, (i, rel(y32, ref))


def test_mlp_block_refuses_what_it_cannot_do():
    from sea_amd import ops

    dt = torch.bfloat16
    g = dict(A=rnd(40, 256, dtype=dt), W1=rnd(2048, 256, dtype=dt), b1=rnd(2048), lnw=rnd(2048), lnb=rnd(2048), W2=rnd(256, 2048, dtype=dt), b2=rnd(256),
             R=None, Wproj=rnd(256, 256, dtype=dt), bproj=rnd(256), Y32=torch.empty(40, 256, device=dev()))
    with pytest.raises(RuntimeError, match="R may be NULL with the norm prologue"):
        ops.mlp_block([g])
    with pytest.raises(RuntimeError, match="unsupported dtype / shape"):
        ops.mlp_block([dict(g, R=rnd(40, 256), W1=rnd(1024, 256, dtype=dt), b1=rnd(1024), lnw=rnd(1024), lnb=rnd(1024), W2=rnd(256, 1024, dtype=dt))])


@pytest.mark.parametrize("dtype", DTYPES)
def test_rownorm_ln_gelu_act_input(dtype):
    from sea_amd import ops

    M, S = 150, 2048
    hs = [rnd(M, S, dtype=dtype, seed=120 + i) * 1.3 + 0.2 for i in range(3)]
    gs = [1 + 0.1 * rnd(S, seed=130 + i) for i in range(3)]
    bs = [0.1 * rnd(S, seed=140 + i) for i in range(3)]
    outs = [torch.empty(M, S, device=dev(), dtype=dtype) for _ in range(3)]
    ops.rownorm([dict(X=hs[i], gamma=gs[i], beta=bs[i], Yact=outs[i]) for i in range(3)], M, S, True, True, 1e-5, dtype)
    for i in range(3):
        ref = gelu(torch.nn.functional.layer_norm(hs[i].float(), (S,), gs[i], bs[i], 1e-5))
        assert rel(outs[i].float(), ref) < tol(dtype, f32=1e-5)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M", [77, 2024])
def test_cond_gemm_with_generated_operand(dtype, M):
    """AdaLN's condition MLP in ONE launch: cond_mlp.0 + SiLU evaluated inside the GEMM of cond_mlp.2 (generated A operand), against the
    two-launch form (sea_silu_outer, then the GEMM on its output — identical arithmetic) and the fp32 formula; groups of different widths."""
    from sea_amd import ops

    c = torch.rand(M, device=dev())
    gen, two, refs, outs, outs2 = [], [], [], [], []
    silu_groups = []
    for i, K2 in enumerate((512, 256, 512, 64)):
        w1, b1 = rnd(K2, seed=800 + i), rnd(K2, seed=810 + i)
        W, bias = rnd(K2, K2, dtype=dtype, scale=0.1, seed=820 + i), 0.3 * rnd(K2, seed=830 + i)
        o1, o2 = torch.empty(M, K2, device=dev(), dtype=dtype), torch.empty(M, K2, device=dev(), dtype=dtype)
        hid = torch.empty(M, K2, device=dev(), dtype=dtype)
        gen.append(dict(W=W, bias=bias, Cact=o1, silu=dict(c=c, w1=w1, b1=b1)))
        silu_groups.append(dict(w1=w1, b1=b1, Hid=hid))
        two.append(dict(A=hid, W=W, bias=bias, Cact=o2))
        h = torch.nn.functional.silu(c[:, None] * w1 + b1).to(dtype).float()
        refs.append(h @ W.float().t() + bias)
        outs.append(o1)
        outs2.append(o2)
    ops.gemm_grouped(gen, dtype)
    ops.silu_outer(silu_groups, c, M, dtype)
    ops.gemm_grouped(two, dtype)
    for o1, o2, ref in zip(outs, outs2, refs):
        assert rel(o1.float(), ref) < tol(dtype, f32=2e-5)
        assert torch.equal(o1, o2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_silu_outer_and_cond_gemm(dtype):
    from sea_amd import ops

    M = 77
    c = torch.rand(M, device=dev())
    groups, refs = [], []
    for i, K2 in enumerate((512, 256, 64)):
        w1, b1 = rnd(K2, seed=150 + i), rnd(K2, seed=160 + i)
        Hid = torch.empty(M, K2, device=dev(), dtype=dtype)
        groups.append(dict(w1=w1, b1=b1, Hid=Hid))
        refs.append(torch.nn.functional.silu(c[:, None] * w1[None, :] + b1[None, :]))
    ops.silu_outer(groups, c, M, dtype)
    for g, r in zip(groups, refs):
        assert rel(g["Hid"].float(), r) < tol(dtype, f32=1e-6)


@pytest.mark.parametrize("M,E,h,F", [(101, 256, 8, 3), (1, 2048, 8, 2), (3, 1024, 8, 3), (2, 1028, 8, 1), (16, 4096, 8, 4), (2, 2048, 12, 2), (17, 1024, 8, 2)])
def test_ib_add(M, E, h, F):
    """Many rows (a wave per row), and a few rows of a wide model: the one-round-trip kernel (h = 8, 1024 <= E <= 4096, up to 4 fields), the
    wave-per-quarter-row form otherwise."""
    from sea_amd import ops

    xb = rnd(M, F * E, seed=170)
    xs = [xb[:, i * E:(i + 1) * E] for i in range(F)]
    x0 = xb.clone()
    c = torch.rand(M, device=dev())
    w1, b1, lnw, lnb = rnd(h, seed=171), rnd(h, seed=172), 1 + 0.1 * rnd(h, seed=173), 0.1 * rnd(h, seed=174)
    w2, b2 = rnd(E, h, seed=175), rnd(E, seed=176)
    ops.ib_add(xs, c, w1, b1, lnw, lnb, w2, b2)
    pre = c[:, None] * w1[None, :] + b1
    hid = gelu(torch.nn.functional.layer_norm(pre, (h,), lnw, lnb, 1e-5))
    ib = hid @ w2.t() + b2
    for i in range(F):
        assert rel(xs[i], x0[:, i * E:(i + 1) * E] + ib) < 1e-6


def test_convert():
    from sea_amd import ops

    src = rnd(37, 3 * 64, seed=180)
    dst = torch.zeros(37, 64, device=dev(), dtype=torch.bfloat16)
    ops.convert(src[:, 64:128], dst)
    assert torch.equal(dst, src[:, 64:128].to(torch.bfloat16))


def test_bad_arguments_raise():
    from sea_amd import ops

    A = rnd(10, 12)  # K = 12 not a multiple of 8
    W = rnd(16, 12)
    with pytest.raises(RuntimeError, match="sea_gemm_grouped"):
        ops.gemm_grouped([dict(A=A, W=W, C32=torch.empty(10, 16, device=dev()))], torch.float32)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gemm_grouped([dict(A=torch.zeros(8, 8), W=torch.zeros(8, 8), C32=torch.zeros(8, 8))], torch.float32)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_gelu_preactivation_and_backward_epilogue(dtype):
    """act=1 keeps the pre-activation in Z; act=2 multiplies by GELU'(Z) (backward of the fused GELU)."""
    from sea_amd import ops

    M, K, N_ = 200, 128, 96
    A = rnd(M, K, dtype=dtype, seed=200)
    W = rnd(N_, K, dtype=dtype, scale=K ** -0.5, seed=201)
    Z = torch.empty(M, N_, device=dev(), dtype=dtype)
    G = torch.empty(M, N_, device=dev(), dtype=dtype)
    ops.gemm_grouped([dict(A=A, W=W, Cact=G, Z=Z, act=1)], dtype)
    pre = A.float() @ W.float().t()
    assert rel(Z.float(), pre) < tol(dtype) and rel(G.float(), gelu(pre)) < tol(dtype)
    # backward: dA' = (dG . W2) * gelu'(Z)
    dG = rnd(M, 64, dtype=dtype, seed=202)
    W2t = rnd(N_, 64, dtype=dtype, seed=203)  # plays W^T: [N_out = N_, K = 64]
    out = torch.empty(M, N_, device=dev())
    ops.gemm_grouped([dict(A=dG, W=W2t, C32=out, Z=Z, act=2)], dtype)
    z = Z.float().requires_grad_(True)
    (gelu(z)).backward(dG.float() @ W2t.float().t())
    assert rel(out, z.grad) < tol(dtype, f32=2e-5, bf16=2e-5)


@pytest.mark.parametrize("dtype", DTYPES)
def test_qkv_rope_row_major_v_copy(dtype):
    from sea_amd import ops

    B, T, H, hd = 2, 33, 4, 16
    E, M, cap = H * hd, B * T, 40
    x = rnd(M, E, dtype=dtype, seed=210)
    W = rnd(3 * E, E, dtype=dtype, scale=E ** -0.5, seed=211)
    bias = rnd(3 * E, seed=212)
    Q = torch.zeros(B, H, T, hd, device=dev(), dtype=dtype)
    K = torch.zeros(B, H, cap, hd, device=dev(), dtype=dtype)
    Vt = torch.zeros(B, H, hd, cap, device=dev(), dtype=dtype)
    V = torch.zeros(B, H, cap, hd, device=dev(), dtype=dtype)
    ops.qkv_rope_grouped([dict(A=x, W=W, bias=bias, col0=0, Q=Q, K=K, Vt=Vt, V=V)], rope_table(hd, T), H, hd, T, 0, cap, hd ** -0.5, dtype)
    assert torch.equal(V.transpose(2, 3), Vt)
    v = (x.float() @ W.float().t() + bias)[:, 2 * E:].view(B, T, H, hd).permute(0, 2, 1, 3)
    assert rel(V[:, :, :T].float(), v) < tol(dtype)
