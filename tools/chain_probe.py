"""Development aid (round 4): the cfg2 forward replayed back to back (for a rocprofv3 kernel trace), and sea_row_chain alone in variants.
    python tools/chain_probe.py replay [n]      n plain replays of the default plan (SEA_PLAN applies)
    python tools/chain_probe.py op              back-to-back launches of the chain op: form A / B, with and without down / projections, 16 / 32 rows"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from sea_amd import ops

dev = torch.device("cuda:0")


def replay(n):
    model = bench.build_model(dev, "bf16").eval()
    x, _, ib = bench.inputs(1, 2024, 3, 256, 0, dev)
    eng = model.engine(dev)
    with torch.no_grad():
        for _ in range(10):
            out = eng.forward(x, ib)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = eng.forward(x, ib)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    print(f"replay: {dt * 1e3:.4f} ms/step over {n} steps; launches {[r.name for r in eng.plan(1, 2024, 'full').records]}", flush=True)
    if os.environ.get("AQKV_STAMPS"):   # the front launch's phase stamps inside the step (sea_aqkv_debug_stamps)
        import ctypes as C
        from sea_amd import _native as N
        L = N.lib()
        L.sea_aqkv_debug_stamps.argtypes = [C.c_void_p]
        L.sea_aqkv_debug_stamps.restype = None
        n_chain = 3 * ((2024 + 31) // 32)
        buf = torch.zeros(1024 * 16, device=dev, dtype=torch.int64)
        acc = None
        for _ in range(20):
            buf.zero_()
            L.sea_aqkv_debug_stamps(buf.data_ptr())
            with torch.no_grad():
                eng.forward(x, ib)
            torch.cuda.synchronize()
            st = buf.view(1024, 16)[:n_chain].cpu().double()
            v = (st[:, :11] - st[:, 0].min()) / 100.0
            acc = v if acc is None else acc + v
        L.sea_aqkv_debug_stamps(None)
        acc /= 20
        names = ["entry", "hid", "loopA", "modx", "rowpass", "loopB", "loopB.all", "staged", "staged.all", "QK", "V"]
        print("front launch inside the step: stamps (us from the first workgroup's entry; mean / max over the chain workgroups, 20 steps):")
        for k, nm in enumerate(names):
            print(f"  {k:2d} {nm:10s} {acc[:, k].mean():7.2f} {acc[:, k].max():7.2f}")


def rnd(*shape, dtype=torch.float32, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(dtype)


def op():
    dt = torch.bfloat16
    D, E, H, T = 128, 256, 8, 2024
    M, hd = T, D // H
    cap = T + 8
    ang = torch.outer(torch.arange(T, dtype=torch.float32), 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd)))
    table = torch.stack((torch.cos(ang), torch.sin(ang)), dim=-1).contiguous().to(dev)
    F = 3

    def mk(form, has_down, n_q, n_kv):
        gs = []
        for f in range(F):
            S = 2 if form == "B" else 0
            g = dict(W2=rnd(E, D if S else E, dtype=dt, scale=0.1), Xin=rnd(M, E), X=torch.empty(M, E, device=dev),
                     att=[rnd(M, D, dtype=dt) for _ in range(S)], Wp=[rnd(D, D, dtype=dt, scale=0.1) for _ in range(S)],
                     a2=None if S else rnd(M, E, dtype=dt), b2=rnd(E) if S else None, bias_scale=2.0 if S else 1.0)
            if has_down:
                g["down"] = dict(W=rnd(D, E, dtype=dt, scale=0.1), bias=rnd(D), gamma=rnd(D), beta=rnd(D), mod=rnd(M, 2 * D, dtype=dt), Yact=torch.empty(M, D, device=dev, dtype=dt))
                proj = []
                for _ in range(n_q):
                    proj.append(dict(W=rnd(D, D, dtype=dt, scale=0.1), bias=rnd(D), col0=0, Q=torch.empty(1, H, T, hd, device=dev, dtype=dt)))
                for _ in range(n_kv):
                    proj.append(dict(W=rnd(2 * D, D, dtype=dt, scale=0.1), bias=rnd(2 * D), col0=D, K=torch.empty(1, H, cap, hd, device=dev, dtype=dt),
                                     Vt=torch.empty(1, H, hd, cap, device=dev, dtype=dt)))
                g["proj"] = proj
            gs.append(g)
        return gs

    def run(name, groups, rows):
        os.environ["SEA_TUNE"] = f"chain_rows={rows}" if rows else ""
        kw = dict(rope=table, H=H, hd=hd, T=T, pos0=0, cap=cap, q_scale_=ops.q_scale(hd))
        for _ in range(5):
            ops.row_chain(groups, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.row_chain(groups, **kw)
        e1.record()
        torch.cuda.synchronize()
        print(f"{name:46s} rows/wg {rows or 'auto':>4}  {e0.elapsed_time(e1) / 20 * 1e3:7.2f} us", flush=True)

    for rows in (16, 32):
        for nf in (1, 3):
            tag = f"{nf} field(s)"
            run(f"A  x only                      {tag}", mk("A", False, 0, 0)[:nf], rows)
            run(f"A  + down/norm                 {tag}", mk("A", True, 0, 0)[:nf], rows)
            run(f"A  + down/norm + 2q            {tag}", mk("A", True, 2, 0)[:nf], rows)
            run(f"A  + down/norm + 2q + 2kv      {tag}", mk("A", True, 2, 2)[:nf], rows)
            run(f"B  x only                      {tag}", mk("B", False, 0, 0)[:nf], rows)
            run(f"B  + down/norm                 {tag}", mk("B", True, 0, 0)[:nf], rows)
            run(f"B  + down/norm + 2kv           {tag}", mk("B", True, 0, 2)[:nf], rows)


def stamps():
    """Phase timeline of one launch (100 MHz stamps of thread 0 of every workgroup): median over workgroups of each phase boundary, relative to the earliest entry."""
    import ctypes as C
    from sea_amd import _native as N
    dt = torch.bfloat16
    D, E, H, T = 128, 256, 8, 2024
    M, hd, cap = T, 16, T + 8
    ang = torch.outer(torch.arange(T, dtype=torch.float32), 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd)))
    table = torch.stack((torch.cos(ang), torch.sin(ang)), dim=-1).contiguous().to(dev)
    L = N.lib()
    L.sea_chain_debug_stamps.argtypes = [C.c_void_p]
    L.sea_chain_debug_stamps.restype = None
    form2 = len(sys.argv) > 2 and sys.argv[2] == "2"
    for form, nf, rows in ((("A", 1, 64), ("A", 3, 64), ("B", 1, 64)) if form2 else (("A", 1, 16), ("A", 3, 32), ("B", 1, 16))):
        gs = []
        for f in range(nf):
            S = 2 if form == "B" else 0
            g = dict(W2=rnd(E, D if S else E, dtype=dt, scale=0.1), Xin=rnd(M, E), X=torch.empty(M, E, device=dev), att=[rnd(M, D, dtype=dt) for _ in range(S)],
                     Wp=[rnd(D, D, dtype=dt, scale=0.1) for _ in range(S)], a2=None if S else rnd(M, E, dtype=dt), b2=rnd(E) if S else None, bias_scale=2.0 if S else 1.0)
            g["down"] = dict(W=rnd(D, E, dtype=dt, scale=0.1), bias=rnd(D), gamma=rnd(D), beta=rnd(D), mod=rnd(M, 2 * D, dtype=dt), Yact=torch.empty(M, D, device=dev, dtype=dt))
            g["proj"] = [dict(W=rnd(D, D, dtype=dt, scale=0.1), bias=rnd(D), col0=0, Q=torch.empty(1, H, T, hd, device=dev, dtype=dt)) for _ in range(2)] + \
                        [dict(W=rnd(2 * D, D, dtype=dt, scale=0.1), bias=rnd(2 * D), col0=D, K=torch.empty(1, H, cap, hd, device=dev, dtype=dt), Vt=torch.empty(1, H, hd, cap, device=dev, dtype=dt)) for _ in range(2)]
            gs.append(g)
        os.environ["SEA_TUNE"] = f"chain_rows={rows}"
        kw = dict(rope=table, H=H, hd=hd, T=T, pos0=0, cap=cap, q_scale_=ops.q_scale(hd))
        nwg = nf * ((M + rows - 1) // rows)
        buf = torch.zeros(nwg * 16, dtype=torch.int64, device=dev)
        for _ in range(3):
            ops.row_chain(gs, **kw)
        torch.cuda.synchronize()
        L.sea_chain_debug_stamps(buf.data_ptr())
        ops.row_chain(gs, **kw)
        torch.cuda.synchronize()
        L.sea_chain_debug_stamps(None)
        st = buf.view(nwg, 16).cpu().double()
        t0 = st[:, 0].min()
        rel = (st - t0) / 100.0   # microseconds
        rel[st == 0] = float("nan")
        med = torch.nanmedian(rel, dim=0).values
        mx = torch.from_numpy(__import__("numpy").nanmax(rel.numpy(), axis=0))
        print(f"form {form} {nf} field(s) {rows} rows/wg ({nwg} workgroups): phase stamps, us after the first workgroup's entry (median | max over workgroups)")
        if form2:
            names2 = ["entry", "own rows landed", "stage 2 multiplied", "x stored", "stage 3 multiplied", "y done", "entry 0 done", "entry 1 done", "entry 2 done", "entry 3 done",
                      "loader 0 starts", "loader 0 first publish", "loader 0 last publish", "", "", ""]
        names = names2 if form2 else ["entry", "burst0 requested", "operands requested", "burst0 landed", "stage 2 multiplied", "Wd + proj burst landed", "x stored", "stage 3 done",
                 "burst A landed", "burst A computed", "burst B landed", "burst B computed", "burst C landed", "burst C computed", "", ""]
        for k in range(16):
            if not torch.isnan(med[k]):
                print(f"   {k:2d} {names[k]:24s} {float(med[k]):7.2f} | {float(mx[k]):7.2f}")


if __name__ == "__main__":
    if sys.argv[1] == "stamps":
        stamps()
        sys.exit(0)
    if sys.argv[1] == "replay":
        replay(int(sys.argv[2]) if len(sys.argv) > 2 else 30)
    else:
        op()
