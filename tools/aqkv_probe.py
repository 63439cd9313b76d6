"""Development aid (round 4): sea_adaln_qkv at the cfg2 shape (3 fields x 2024 rows, E = 256, H = 8; three ln_cross riders) timed stand-alone, beside the launches it
replaces.  SEA_TUNE=aqkv_probe=n ends the kernel after stage n (see adaln_qkv.hip): python tools/aqkv_probe.py [riders=1]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sea_amd import ops
from tools.bench_ops import timeit

dev = torch.device("cuda:0")
dt = torch.bfloat16
with_riders = (sys.argv[1] if len(sys.argv) > 1 else "1") != "0"
B, T, E, H, F = 1, 2024, 256, 8, 3
hd, M, cap = E // H, B * T, 2024


def rnd(*shape, dtype=torch.float32, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(dtype)


ang = torch.outer(torch.arange(T, dtype=torch.float32), 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd)))
table = torch.stack((torch.cos(ang), torch.sin(ang)), dim=-1).contiguous().to(dev)
cond = torch.rand(M, device=dev)
xw = rnd(M, F * E)
groups, sil, ada, qkv = [], [], [], []
for i in range(F):
    w1, b1 = rnd(2 * E), rnd(2 * E)
    W2c, b2c = rnd(2 * E, 2 * E, dtype=dt, scale=0.04), rnd(2 * E)
    gamma, beta = 1 + 0.1 * rnd(E), 0.1 * rnd(E)
    Wqkv, bqkv = rnd(3 * E, E, dtype=dt, scale=0.06), rnd(3 * E)
    x = xw[:, i * E:(i + 1) * E]
    Q, K, Vt = torch.zeros(B, H, T, hd, device=dev, dtype=dt), torch.zeros(B, H, cap, hd, device=dev, dtype=dt), torch.zeros(B, H, hd, cap, device=dev, dtype=dt)
    groups.append(dict(X=x, cond=cond, w1=w1, b1=b1, W2c=W2c, b2c=b2c, gamma=gamma, beta=beta, Wqkv=Wqkv, bqkv=bqkv, Q=Q, K=K, Vt=Vt))
    hid, n_e = torch.empty(M, 2 * E, device=dev, dtype=dt), torch.empty(M, E, device=dev, dtype=dt)
    sil.append(dict(w1=w1, b1=b1, Hid=hid))
    ada.append(dict(A=hid, W=W2c, bias=b2c, X=x, gamma=gamma, beta=beta, Yact=n_e))
    qkv.append(dict(A=n_e, W=Wqkv, bias=bqkv, col0=0, Q=Q, K=K, Vt=Vt))
riders = [dict(A=rnd(M, 512, dtype=dt), W=rnd(256, 512, dtype=dt, scale=0.05), bias=rnd(256), Cact=torch.empty(M, 256, device=dev, dtype=dt)) for _ in range(3)] if with_riders else []
t_new = timeit(lambda: ops.adaln_qkv(groups, table, H, hd, T, 0, cap, ops.q_scale(hd), riders=riders))
t_s = timeit(lambda: ops.silu_outer(sil, cond, M, dt))
t_a = timeit(lambda: ops.gemm_adaln(ada + [dict(A=r["A"], W=r["W"], bias=r["bias"], Yact=r["Cact"]) for r in riders]))
t_q = timeit(lambda: ops.qkv_rope_grouped(qkv, table, H, hd, T, 0, cap, ops.q_scale(hd), dt))
if os.environ.get("AQKV_STAMPS"):
    import ctypes as C
    from sea_amd import _native as N
    L = N.lib()
    L.sea_aqkv_debug_stamps.argtypes = [C.c_void_p]
    L.sea_aqkv_debug_stamps.restype = None
    n_wg = 3 * ((M + 31) // 32) + (96 if with_riders else 0)
    buf = torch.zeros(n_wg * 16, device=dev, dtype=torch.int64)
    L.sea_aqkv_debug_stamps(buf.data_ptr())
    for _ in range(3):
        buf.zero_()
        ops.adaln_qkv(groups, table, H, hd, T, 0, cap, ops.q_scale(hd), riders=riders)
        torch.cuda.synchronize()
    L.sea_aqkv_debug_stamps(None)
    st = buf.view(n_wg, 16)[:3 * ((M + 31) // 32)].cpu().double()
    t0 = st[:, 0].min()
    names = ["entry", "hid", "loopA", "modx", "rowpass", "loopB", "loopB.all", "staged", "staged.all", "QK", "V"]
    print("stamps (us from the first workgroup's entry; mean / max over workgroups):")
    for k, nm in enumerate(names):
        v = (st[:, k] - t0) / 100.0
        print(f"  {k:2d} {nm:10s} {v.mean():7.2f} {v.max():7.2f}")
print(f"riders={int(with_riders)} SEA_TUNE={os.environ.get('SEA_TUNE', '')}: adaln_qkv {t_new:.1f} us; silu(3) {t_s:.1f} + gemm_adaln {t_a:.1f} + qkv_rope {t_q:.1f} = {t_s + t_a + t_q:.1f} us", flush=True)
