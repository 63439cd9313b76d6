"""sea_rowchain (fused row-local operator chains) against a plain torch fp32 evaluation of the same stage list."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from sea_amd import ops  # noqa: E402


def _rope_tables(T, hd):
    f = 10000.0 ** (-torch.arange(0, hd, 2, dtype=torch.float32) / hd)
    ang = torch.outer(torch.arange(T, dtype=torch.float32), f)
    return torch.stack([ang.cos(), ang.sin()], -1).contiguous()  # [T, hd/2, 2]


def _gelu(x):
    return 0.5 * x * (1 + torch.erf(x / math.sqrt(2)))


def _ln(v, gamma, beta, mod, eps=1e-5):
    mean = v.mean(-1, keepdim=True)
    var = ((v - mean) ** 2).mean(-1, keepdim=True)
    xh = (v - mean) / torch.sqrt(var + eps)
    n = v.shape[-1]
    if mod is not None:
        return xh * (gamma + 1 + mod[:, :n]) + ((beta if beta is not None else 0) + mod[:, n:])
    return xh * gamma + (beta if beta is not None else 0)


def _rope_apply(v, tab, pos, hd):
    # v [M, n] with n % hd == 0; pairs (2k, 2k+1) rotate by tab[pos, k]
    M, n = v.shape
    x = v.view(M, n // hd, hd // 2, 2)
    c, s = tab[pos][:, None, :, 0], tab[pos][:, None, :, 1]
    return torch.stack([x[..., 0] * c - x[..., 1] * s, x[..., 0] * s + x[..., 1] * c], -1).view(M, n)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1.5e-2)])
@pytest.mark.parametrize("B,T", [(1, 70), (2, 45), (1, 1)])
def test_exchange_chain_matches_torch(dtype, tol, B, T):
    torch.manual_seed(5)
    dev = torch.device("cuda:0")
    E, D, H, h = 256, 128, 8, 8
    hd = D // H
    M, cap, pos0 = B * T, (T + 7) // 8 * 8 + 8, 3
    rnd = lambda *s, sc=1.0: (torch.randn(*s) * sc)
    att0, att1 = rnd(M, D), rnd(M, D)
    Wp0, Wp1 = rnd(D, D, sc=D ** -0.5), rnd(D, D, sc=D ** -0.5)
    Wup, bup = rnd(E, D, sc=D ** -0.5), rnd(E, sc=0.1)
    xr = rnd(M, E)
    cond = torch.rand(M)
    ib = dict(w1=rnd(h), b1=rnd(h, sc=0.1), lnw=1 + rnd(h, sc=0.1), lnb=rnd(h, sc=0.1), w2=rnd(E, h, sc=0.3), b2=rnd(E, sc=0.1))
    g2, b2_, mod2 = 1 + rnd(E, sc=0.1), rnd(E, sc=0.1), rnd(M, 2 * E, sc=0.2)
    Wd, bd = rnd(D, E, sc=E ** -0.5), rnd(D, sc=0.1)
    gd, bd2, modd = 1 + rnd(D, sc=0.1), rnd(D, sc=0.1), rnd(M, 2 * D, sc=0.2)
    Wkv, bkv = rnd(2 * D, D, sc=D ** -0.5), rnd(2 * D, sc=0.1)
    Wq, bq = rnd(D, D, sc=D ** -0.5), rnd(D, sc=0.1)
    tab = _rope_tables(pos0 + T + 4, hd)

    # ---- torch evaluation with the kernel's rounding points (LDS tiles and act-dtype tensors are `dtype`)
    q = lambda t: t.to(dtype).float()
    a0, a1, m2q, mdq = q(att0), q(att1), q(mod2), q(modd)
    gp = q(_gelu(a0 @ q(Wp0).T) + _gelu(a1 @ q(Wp1).T))
    x_new = gp @ q(Wup).T + 2 * bup + xr
    pre = ib["w1"] * cond[:, None] + ib["b1"]
    hid = _gelu(_ln(pre, ib["lnw"], ib["lnb"], None))
    x_ib = x_new + hid @ ib["w2"].T + ib["b2"]
    n_e = _ln(x_ib, g2, b2_, m2q)
    dn = q(x_new) @ q(Wd).T + bd
    nd = _ln(dn, gd, bd2, mdq)
    pos = pos0 + torch.arange(M) % T
    kv = q(nd) @ q(Wkv).T + bkv
    k_ref = _rope_apply(kv[:, :D].contiguous(), tab, pos, hd)
    v_ref = kv[:, D:]
    qq = q(nd) @ q(Wq).T + bq
    q_ref = _rope_apply(qq, tab, pos, hd) * hd ** -0.5

    # ---- device
    d = lambda t, dt=torch.float32: t.to(dev).to(dt).contiguous()
    att0_d, att1_d = d(att0, dtype), d(att1, dtype)
    xr_d, xout = d(xr), torch.empty(M, E, device=dev)
    n_e_d, nd_d = torch.empty(M, E, device=dev, dtype=dtype), torch.empty(M, D, device=dev, dtype=dtype)
    Qo = torch.zeros(B, H, T, hd, device=dev, dtype=dtype)
    Ko = torch.zeros(B, H, cap, hd, device=dev, dtype=dtype)
    Vt = torch.zeros(B, H, hd, cap, device=dev, dtype=dtype)
    ibd = {k: d(v) for k, v in ib.items()}
    tab_d = d(tab)
    stages = [
        dict(kind=2, N=D, X=att0_d, raw_slot=0) if B == 1 else dict(kind=1, N=D, X=att0_d, x_is_act=1, raw_slot=0),
        dict(kind=2, N=D, X=att1_d, raw_slot=1) if B == 1 else dict(kind=1, N=D, X=att1_d, x_is_act=1, raw_slot=1),
        dict(a_slot=0, N=D, K=D, W=d(Wp0, dtype), act=1, sum_op=1),
        dict(a_slot=1, N=D, K=D, W=d(Wp1, dtype), act=1, sum_op=2, raw_slot=2),
        dict(a_slot=2, N=E, K=D, W=d(Wup, dtype), bias=d(bup), bias_scale=2.0, R=xr_d, raw_slot=0,
             ib_w1=ibd["w1"], ib_b1=ibd["b1"], ib_lnw=ibd["lnw"], ib_lnb=ibd["lnb"], ib_w2=ibd["w2"], ib_b2=ibd["b2"], ib_h=h,
             C32=xout, norm=1, gamma=d(g2), beta=d(b2_), mod=d(mod2, dtype), Nact=n_e_d),
        dict(a_slot=0, N=D, K=E, W=d(Wd, dtype), bias=d(bd), norm=1, gamma=d(gd), beta=d(bd2), mod=d(modd, dtype), norm_slot=1, Nact=nd_d),
        dict(a_slot=1, N=2 * D, K=D, W=d(Wkv, dtype), bias=d(bkv), qkv=1, col0=D, hd=hd, rope=tab_d, Kout=Ko, Vtout=Vt),
        dict(a_slot=1, N=D, K=D, W=d(Wq, dtype), bias=d(bq), qkv=1, col0=0, hd=hd, rope=tab_d, q_scale=hd ** -0.5, Qout=Qo),
    ]
    ops.rowchain([stages], M, T, pos0, cap, H, dtype, cond=d(cond))
    torch.cuda.synchronize()

    def rel(a, b):
        a, b = a.float().cpu(), b.float().cpu()
        return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()

    assert rel(xout, x_ib) < tol
    assert rel(n_e_d, n_e) < tol
    assert rel(nd_d, nd) < tol
    bt = lambda t: t.view(B, T, H, hd).permute(0, 2, 1, 3)
    assert rel(Qo, bt(q_ref)) < tol
    assert rel(Ko[:, :, pos0:pos0 + T], bt(k_ref)) < tol
    assert rel(Vt[:, :, :, pos0:pos0 + T], bt(v_ref).transpose(2, 3)) < tol
    assert Ko[:, :, :pos0].abs().max().item() == 0 and Ko[:, :, pos0 + T:].abs().max().item() == 0  # nothing outside the written positions


def test_rejects_bad_programs():
    dev = torch.device("cuda:0")
    x = torch.zeros(64, 128, device=dev)
    w = torch.zeros(128, 128, device=dev)
    with pytest.raises(RuntimeError):  # A slot never written
        ops.rowchain([[dict(a_slot=0, N=128, K=128, W=w, C32=x)]], 64, 64, 0, 64, 8, torch.float32)
    with pytest.raises(RuntimeError):  # N not a multiple of 64
        ops.rowchain([[dict(kind=1, N=32, X=x, raw_slot=0)]], 64, 64, 0, 64, 8, torch.float32)
    with pytest.raises(RuntimeError):  # stage overwrites the tile it reads
        ops.rowchain([[dict(kind=1, N=128, X=x, raw_slot=0), dict(a_slot=0, N=128, K=128, W=w, raw_slot=0)]], 64, 64, 0, 64, 8, torch.float32)
