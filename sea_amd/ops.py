"""Tensor-level wrappers over the C ABI (include/sea_hip.h).

Each function fills the ABI structs from torch tensors (device memory owned by PyTorch) and launches on the current
stream.  They are the un-fused building blocks used by the module mirrors in sea_amd/models/base_blocks.py and by the
operator tests; sea_amd/engine.py pre-builds the same structs once per plan for the whole-model path.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence

import torch

from . import _native as N


def _mat(t: torch.Tensor, name: str) -> torch.Tensor:
    N.require_gpu(t, name)
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{name}: need a 2-D tensor with unit inner stride, got shape {tuple(t.shape)} strides {t.stride()}")
    return t


def fill_gemm_group(g: N.SeaGemmGroup, A, W, bias=None, R=None, C32=None, Cact=None, n_seg=1, a_seg_stride=0, act=0,
                    bias_scale=1.0, M=None, N_=None, K=None, Z=None, silu=None, drop=None) -> None:
    if silu is not None:   # generated A operand: dict(c f32 [M], w1 f32 [K], b1 f32 [K]) — A[m, k] = silu(w1[k] c[m] + b1[k])
        g.silu_c, g.silu_w1, g.silu_b1 = silu["c"].data_ptr(), silu["w1"].data_ptr(), silu["b1"].data_ptr()
        g.A, g.lda, g.W, g.ldw = None, 0, W.data_ptr(), W.stride(0)
        g.bias, g.R, g.C32, g.Cact, g.Z = N.ptr(bias), None, N.ptr(C32), N.ptr(Cact), None
        g.ldc32 = C32.stride(0) if C32 is not None else 0
        g.ldcact = Cact.stride(0) if Cact is not None else 0
        g.M, g.N, g.K, g.n_seg, g.act, g.bias_scale = silu["c"].numel(), W.shape[0], W.shape[1], 1, 0, bias_scale
        return
    g.A, g.W = A.data_ptr(), W.data_ptr()
    g.bias = N.ptr(bias)
    g.R, g.C32, g.Cact = N.ptr(R), N.ptr(C32), N.ptr(Cact)
    g.Z, g.ldz = N.ptr(Z), (Z.stride(0) if Z is not None else 0)
    g.a_seg_stride = a_seg_stride
    g.lda, g.ldw = A.stride(0), W.stride(0)
    g.ldr = R.stride(0) if R is not None else 0
    g.ldc32 = C32.stride(0) if C32 is not None else 0
    g.ldcact = Cact.stride(0) if Cact is not None else 0
    g.M = A.shape[0] if M is None else M
    g.N = W.shape[0] if N_ is None else N_
    g.K = W.shape[1] if K is None else K
    g.n_seg = n_seg
    g.act = act
    g.bias_scale = bias_scale
    if drop is not None:   # (seed, stream, thr, mode): counter-based dropout of the epilogue (include/sea_hip.h, SeaDropout)
        g.drop.seed, g.drop.stream, g.drop.thr, g.drop.mode = drop


def gemm_grouped(groups: Sequence[Dict], dtype: torch.dtype) -> None:
    """groups: dicts with A [M,K] act, W [N,K] act, optional bias f32 [N], R f32 [M,N], C32 f32 [M,N], Cact act [M,N],
    n_seg, a_seg_stride, act (0|1), bias_scale."""
    n = len(groups)
    arr = (N.SeaGemmGroup * n)()
    for i, d in enumerate(groups):
        if d.get("silu") is not None:
            W = _mat(d["W"], "W")
            if W.dtype != dtype:
                raise ValueError(f"gemm group {i}: W dtype {W.dtype} != activation dtype {dtype}")
            fill_gemm_group(arr[i], None, W, d.get("bias"), None, d.get("C32"), d.get("Cact"), bias_scale=d.get("bias_scale", 1.0), silu=d["silu"])
            continue
        A, W = _mat(d["A"], "A"), _mat(d["W"], "W")
        if A.dtype != dtype or W.dtype != dtype:
            raise ValueError(f"gemm group {i}: A/W dtype {A.dtype}/{W.dtype} != activation dtype {dtype}")
        for k in ("bias", "R", "C32"):
            if d.get(k) is not None and d[k].dtype != torch.float32:
                raise ValueError(f"gemm group {i}: {k} must be float32")
        if d.get("Cact") is not None and d["Cact"].dtype != dtype:
            raise ValueError(f"gemm group {i}: Cact dtype mismatch")
        fill_gemm_group(arr[i], A, W, d.get("bias"), d.get("R"), d.get("C32"), d.get("Cact"), d.get("n_seg", 1),
                        d.get("a_seg_stride", 0), d.get("act", 0), d.get("bias_scale", 1.0), K=d.get("K"), Z=d.get("Z"), drop=d.get("drop"))
    N.check(N.lib().sea_gemm_grouped(arr, n, N.dtype_code(dtype), N.stream_ptr()), "sea_gemm_grouped")


LOG2E = 1.4426950408889634


def q_scale(hd: int) -> float:
    """The factor the QKV epilogue puts on (rotated) q for the attention kernels: hd^-1/2 (reference models/base_blocks.py:191) times log2(e) — the
    attention kernels work in log2 units (P = 2^(S - max): one v_exp_f32 per probability, the subtraction in the MFMA's C operand), see sea_hip.h."""
    return float(hd) ** -0.5 * LOG2E


def qkv_rope_grouped(groups: Sequence[Dict], rope: torch.Tensor, H: int, hd: int, T: int, pos0: int, cap: int,
                     q_scale: float, dtype: torch.dtype) -> None:
    """groups: dicts with A [M,K], W [N,K], bias f32 [N], col0, Q/K/Vt output tensors (see sea_hip.h)."""
    n = len(groups)
    arr = (N.SeaQkvGroup * n)()
    for i, d in enumerate(groups):
        A, W = _mat(d["A"], "A"), _mat(d["W"], "W")
        if A.dtype != dtype or W.dtype != dtype:
            raise ValueError(f"qkv group {i}: dtype mismatch")
        g = arr[i]
        g.A, g.W, g.bias = A.data_ptr(), W.data_ptr(), N.ptr(d.get("bias"))
        g.Qout, g.Kout, g.Vtout, g.Vout = N.ptr(d.get("Q")), N.ptr(d.get("K")), N.ptr(d.get("Vt")), N.ptr(d.get("V"))
        g.lda, g.ldw = A.stride(0), W.stride(0)
        g.M, g.N, g.K = A.shape[0], W.shape[0], W.shape[1]
        g.col0 = d.get("col0", 0)
    N.require_gpu(rope, "rope")
    assert rope.dtype == torch.float32 and rope.is_contiguous() and rope.shape[-1] == 2 and rope.shape[-2] == hd // 2
    assert rope.shape[0] >= pos0 + T, "rope table shorter than pos0 + T"
    common = N.SeaQkvCommon(rope.data_ptr(), H, hd, T, pos0, cap, q_scale)
    N.check(N.lib().sea_qkv_rope_grouped(arr, n, C.byref(common), N.dtype_code(dtype), N.stream_ptr()), "sea_qkv_rope_grouped")


def attention_fwd(problems: Sequence[Dict], B: int, H: int, hd: int, Tq: int, Tk: int, cap: int, q_pos0: int,
                  src_len: int, dtype: torch.dtype, drop=None) -> None:
    """problems: dicts with Q [B,H,Tq,hd], K [B,H,cap,hd], Vt [B,H,hd,cap], O [B,Tq,H*hd] (row stride = O.stride(1)),
    optional LSE f32 [B,H,Tq]."""
    P = N.SeaAttnParams()
    P.n_problems = len(problems)
    ldo = None
    for i, d in enumerate(problems):
        for k in ("Q", "K", "Vt", "O"):
            N.require_gpu(d[k], k)
            assert d[k].dtype == dtype, f"attention problem {i}: {k} dtype"
        assert d["Q"].is_contiguous() and d["K"].is_contiguous() and d["Vt"].is_contiguous()
        O = d["O"]
        assert O.dim() == 3 and O.stride(2) == 1 and O.stride(0) == Tq * O.stride(1)
        ldo = O.stride(1) if ldo is None else ldo
        assert O.stride(1) == ldo
        P.p[i].Q, P.p[i].K, P.p[i].Vt, P.p[i].O = d["Q"].data_ptr(), d["K"].data_ptr(), d["Vt"].data_ptr(), O.data_ptr()
        P.p[i].LSE = N.ptr(d.get("LSE"))
    P.B, P.H, P.hd, P.Tq, P.Tk, P.cap, P.q_pos0, P.src_len, P.ldo = B, H, hd, Tq, Tk, cap, q_pos0, src_len, ldo
    if drop is not None:   # (seed, first stream, thr): dropout of the attention probabilities, problem i on stream + i
        P.drop.seed, P.drop.stream, P.drop.thr = drop
    N.check(N.lib().sea_attention_fwd(C.byref(P), N.dtype_code(dtype), N.stream_ptr()), "sea_attention_fwd")


def fill_norm_group(g: N.SeaNormGroup, gd: Dict) -> None:
    """gd: X, optional mod act [M, 2d], gamma f32 [d], optional beta f32 [d], Y32 and/or Yact, optional mean / rstd f32 [M], addend / Xout (include/sea_hip.h, SeaNormGroup)."""
    X = gd["X"]
    g.X, g.ldx = X.data_ptr(), gd.get("ldx", X.stride(0) if X.dim() == 2 else 0)
    mod = gd.get("mod")
    g.mod, g.ldmod = N.ptr(mod), (mod.stride(0) if mod is not None else 0)
    g.gamma, g.beta = gd["gamma"].data_ptr(), N.ptr(gd.get("beta"))
    y32, yact = gd.get("Y32"), gd.get("Yact")
    g.Y32, g.ldy32 = N.ptr(y32), gd.get("ldy32", y32.stride(0) if y32 is not None else 0)
    g.Yact, g.ldyact = N.ptr(yact), (yact.stride(0) if yact is not None else 0)
    g.mean, g.rstd = N.ptr(gd.get("mean")), N.ptr(gd.get("rstd"))
    add, xout = gd.get("addend"), gd.get("Xout")
    g.addend, g.ldadd = N.ptr(add), (add.stride(0) if add is not None else 0)
    g.Xout, g.ldxout = N.ptr(xout), (xout.stride(0) if xout is not None else 0)


def rownorm(groups: Sequence[Dict], M: int, d: int, x_is_act: bool, gelu: bool, eps: float, dtype: torch.dtype) -> None:
    """groups: dicts with X [M,d], optional mod act [M,2d], gamma f32 [d], optional beta f32 [d], Y32 and/or Yact,
    optional mean/rstd f32 [M]."""
    n = len(groups)
    arr = (N.SeaNormGroup * n)()
    for i, gd in enumerate(groups):
        X = _mat(gd["X"], "X")
        assert X.dtype == (dtype if x_is_act else torch.float32)
        fill_norm_group(arr[i], gd)
    N.check(N.lib().sea_rownorm(arr, n, M, d, int(x_is_act), int(gelu), eps, N.dtype_code(dtype), N.stream_ptr()), "sea_rownorm")


def fewrows_supported(dtype: torch.dtype, M: int, Ks: Sequence[int], qkv: bool = False, pre: bool = False) -> bool:
    """Do sea_gemm_fewrows / sea_qkv_rope_fewrows (gemv.hip) cover a launch of these groups?  bf16, M <= 4 rows per group, one K per launch out of N.FEW_K
    (<= 2048 for the q/k/v form and for a norm prologue)."""
    Ks = list(Ks)
    return (dtype == torch.bfloat16 and 1 <= M <= 4 and 1 <= len(Ks) <= N.FEW_MAX_GROUPS and all(k == Ks[0] for k in Ks) and Ks[0] in N.FEW_K
            and (Ks[0] <= 2048 or not (qkv or pre)))


def _norm_array(specs: Optional[Sequence[Optional[Dict]]], n: int):
    if specs is None or all(sp is None for sp in specs):
        return None
    arr = (N.SeaNormGroup * n)()
    for g, sp in zip(arr, specs):
        if sp is not None:
            fill_norm_group(g, sp)
    return arr


def gemm_fewrows(groups: Sequence[Dict], dtype: torch.dtype, pre=None, pre_x_is_act: bool = False, pre_gelu: bool = False, eps: float = 1e-5) -> None:
    """sea_gemm_fewrows: groups as gemm_grouped (A may be None where pre[i] is given); pre: list of rownorm group dicts or None per group (fp32 rows, or —
    pre_x_is_act — rows in the activation dtype with a plain LayerNorm, optionally followed by GELU)."""
    n = len(groups)
    arr = (N.SeaGemmGroup * n)()
    for i, d in enumerate(groups):
        W = _mat(d["W"], "W")
        A = d.get("A")
        if A is None:
            X = pre[i]["X"]
            fill_gemm_group(arr[i], W, W, d.get("bias"), d.get("R"), d.get("C32"), d.get("Cact"), act=d.get("act", 0), bias_scale=d.get("bias_scale", 1.0), M=X.shape[0], Z=d.get("Z"))
            arr[i].A, arr[i].lda = None, 0
        else:
            fill_gemm_group(arr[i], _mat(A, "A"), W, d.get("bias"), d.get("R"), d.get("C32"), d.get("Cact"), act=d.get("act", 0), bias_scale=d.get("bias_scale", 1.0), Z=d.get("Z"))
    N.check(N.lib().sea_gemm_fewrows(arr, _norm_array(pre, n), n, int(pre_x_is_act), int(pre_gelu), eps, N.dtype_code(dtype), N.stream_ptr()), "sea_gemm_fewrows")


def qkv_rope_fewrows(groups: Sequence[Dict], rope: torch.Tensor, H: int, hd: int, T: int, pos0: int, cap: int, q_scale: float, dtype: torch.dtype, pre=None,
                     eps: float = 1e-5) -> None:
    """sea_qkv_rope_fewrows: groups as qkv_rope_grouped (A may be None where pre[i] is given, then 'M' gives the rows)."""
    n = len(groups)
    arr = (N.SeaQkvGroup * n)()
    for i, d in enumerate(groups):
        W = _mat(d["W"], "W")
        g = arr[i]
        A = d.get("A")
        g.A, g.lda = (N.ptr(A), A.stride(0)) if A is not None else (None, 0)
        g.W, g.bias, g.ldw = W.data_ptr(), N.ptr(d.get("bias")), W.stride(0)
        g.Qout, g.Kout, g.Vtout, g.Vout = N.ptr(d.get("Q")), N.ptr(d.get("K")), N.ptr(d.get("Vt")), N.ptr(d.get("V"))
        g.M, g.N, g.K = (A.shape[0] if A is not None else pre[i]["X"].shape[0]), W.shape[0], W.shape[1]
        g.col0 = d.get("col0", 0)
    common = N.SeaQkvCommon(rope.data_ptr(), H, hd, T, pos0, cap, q_scale)
    N.check(N.lib().sea_qkv_rope_fewrows(arr, _norm_array(pre, n), n, C.byref(common), eps, N.dtype_code(dtype), N.stream_ptr()), "sea_qkv_rope_fewrows")


def fill_gemm_norm_group(g: N.SeaGemmNormGroup, A, W, gamma, bias=None, R=None, C32=None, mod=None, beta=None, Y32=None, Yact=None,
                         mean=None, rstd=None, ldr=None, ldy32=None, n_seg=1, a_seg_stride=0, bias_scale=1.0, Cact=None, ib=None, K=None) -> None:
    """ib: dict(c, w1, b1, lnw, lnb, w2 [N, h], b2, h) — the info-bottleneck addend (sea_hip.h, SeaGemmNormGroup)."""
    g.A, g.W, g.bias, g.R, g.C32 = A.data_ptr(), W.data_ptr(), N.ptr(bias), N.ptr(R), N.ptr(C32)
    g.mod, g.gamma, g.beta = N.ptr(mod), gamma.data_ptr(), N.ptr(beta)
    g.Y32, g.Yact, g.mean, g.rstd = N.ptr(Y32), N.ptr(Yact), N.ptr(mean), N.ptr(rstd)
    g.lda, g.ldw = A.stride(0), W.stride(0)
    g.ldr = (ldr if ldr is not None else R.stride(0)) if R is not None else 0
    g.ldc32 = C32.stride(0) if C32 is not None else 0
    g.ldmod = mod.stride(0) if mod is not None else 0
    g.ldy32 = (ldy32 if ldy32 is not None else Y32.stride(0)) if Y32 is not None else 0
    g.ldyact = Yact.stride(0) if Yact is not None else 0
    g.M, g.N, g.K = A.shape[0], W.shape[0], (W.shape[1] if K is None else K)
    g.n_seg, g.a_seg_stride, g.bias_scale = n_seg, a_seg_stride, bias_scale
    g.Cact, g.ldcact = N.ptr(Cact), (Cact.stride(0) if Cact is not None else 0)
    if ib is not None:
        g.ib_c, g.ib_w1, g.ib_b1, g.ib_lnw, g.ib_lnb = N.ptr(ib.get("c")), ib["w1"].data_ptr(), ib["b1"].data_ptr(), ib["lnw"].data_ptr(), ib["lnb"].data_ptr()
        g.ib_w2, g.ib_b2, g.ib_h = ib["w2"].data_ptr(), ib["b2"].data_ptr(), ib["h"]


def gemm_rownorm(groups: Sequence[Dict], eps: float, dtype: torch.dtype) -> None:
    """Linear + row normalisation in one launch (sea_gemm_rownorm).  groups: dicts with A [M,K] act, W [N,K] act (N <= 256, N % 16 == 0),
    gamma f32 [N], optional bias f32 [N], R f32 [M,N], C32 f32 [M,N] (pre-norm), mod act [M,2N], beta f32 [N], Y32 / Yact, mean / rstd."""
    n = len(groups)
    arr = (N.SeaGemmNormGroup * n)()
    for i, d in enumerate(groups):
        A, W = _mat(d["A"], "A"), _mat(d["W"], "W")
        if A.dtype != dtype or W.dtype != dtype:
            raise ValueError(f"gemm_rownorm group {i}: A/W dtype {A.dtype}/{W.dtype} != activation dtype {dtype}")
        if d.get("Yact") is not None and d["Yact"].dtype != dtype:
            raise ValueError(f"gemm_rownorm group {i}: Yact dtype mismatch")
        if d.get("mod") is not None and d["mod"].dtype != dtype:
            raise ValueError(f"gemm_rownorm group {i}: mod dtype mismatch")
        fill_gemm_norm_group(arr[i], A, W, d["gamma"], d.get("bias"), d.get("R"), d.get("C32"), d.get("mod"), d.get("beta"), d.get("Y32"),
                             d.get("Yact"), d.get("mean"), d.get("rstd"), n_seg=d.get("n_seg", 1), a_seg_stride=d.get("a_seg_stride", 0),
                             bias_scale=d.get("bias_scale", 1.0), Cact=d.get("Cact"), ib=d.get("ib"))
    N.check(N.lib().sea_gemm_rownorm(arr, n, eps, N.dtype_code(dtype), N.stream_ptr()), "sea_gemm_rownorm")


def fill_exchange_tail(P: N.SeaExchangeTail, att: Sequence[torch.Tensor], Wp: Sequence[torch.Tensor], Wup, bup, bias_scale: float, X, Xact=None,
                       down: Optional[Dict] = None) -> None:
    """att: n_seg act [M, D] matrices, Wp: n_seg act [D, D]; Wup act [E, D]; X f32 [M, E] (in place); down: dict(W [D, E], bias, gamma, beta, mod, Yact, Y32)."""
    for s, a in enumerate(att):
        P.att[s] = a.data_ptr()
        P.Wp[s] = Wp[s].data_ptr()
    P.n_seg, P.ldatt, P.ldwp = len(att), att[0].stride(0), Wp[0].stride(0)
    P.Wup, P.ldwup, P.bup, P.bias_scale = Wup.data_ptr(), Wup.stride(0), N.ptr(bup), bias_scale
    P.X, P.ldx = X.data_ptr(), X.stride(0)
    P.Xact, P.ldxact = N.ptr(Xact), (Xact.stride(0) if Xact is not None else 0)
    P.M, P.E, P.D = X.shape[0], Wup.shape[0], Wup.shape[1]
    P.has_down = int(down is not None)
    if down is not None:
        g = P.down
        g.W, g.ldw, g.bias = down["W"].data_ptr(), down["W"].stride(0), N.ptr(down.get("bias"))
        g.gamma, g.beta = down["gamma"].data_ptr(), N.ptr(down.get("beta"))
        mod, y32, yact = down.get("mod"), down.get("Y32"), down.get("Yact")
        g.mod, g.ldmod = N.ptr(mod), (mod.stride(0) if mod is not None else 0)
        g.Y32, g.ldy32 = N.ptr(y32), (y32.stride(0) if y32 is not None else 0)
        g.Yact, g.ldyact = N.ptr(yact), (yact.stride(0) if yact is not None else 0)
        g.mean, g.rstd = N.ptr(down.get("mean")), N.ptr(down.get("rstd"))


def exchange_tail_supported(dtype: torch.dtype, D: int, E: int, n_seg: int) -> bool:
    """Shapes sea_exchange_tail instantiates (include/sea_hip.h)."""
    return dtype == torch.bfloat16 and (D, E) in ((128, 256), (64, 128)) and 1 <= n_seg and n_seg * D <= 256


def exchange_tail(att, Wp, Wup, bup, bias_scale, X, Xact=None, down=None, eps: float = 1e-5, dtype: torch.dtype = torch.bfloat16) -> None:
    """One field's exchange tail in one launch: X += sum_s gelu(att_s Wp_s^T) Wup^T + bias_scale * bup; optionally the down-projection + row norm
    of the updated rows (sea_exchange_tail)."""
    for t in list(att) + list(Wp) + [Wup, X]:
        N.require_gpu(t, "exchange_tail operand")
    P = (N.SeaExchangeTail * 1)()
    fill_exchange_tail(P[0], att, Wp, Wup, bup, bias_scale, X, Xact, down)
    N.check(N.lib().sea_exchange_tail(P, 1, eps, N.dtype_code(dtype), N.stream_ptr()), "sea_exchange_tail")


def exchange_tail_grouped(groups: Sequence[Dict], eps: float = 1e-5, dtype: torch.dtype = torch.bfloat16) -> None:
    """Several problems of one shape in one launch (grid row per group): dicts with the arguments of exchange_tail."""
    P = (N.SeaExchangeTail * len(groups))()
    for p_, d in zip(P, groups):
        fill_exchange_tail(p_, d["att"], d["Wp"], d["Wup"], d.get("bup"), d.get("bias_scale", 1.0), d["X"], d.get("Xact"), d.get("down"))
    N.check(N.lib().sea_exchange_tail(P, len(groups), eps, N.dtype_code(dtype), N.stream_ptr()), "sea_exchange_tail")


def fill_adaln_group(g: N.SeaAdalnGroup, A, W, bias=None, X=None, gamma=None, beta=None, Yact=None, Y32=None, mean=None, rstd=None, ldx=None, ldy32=None) -> None:
    """One group of sea_gemm_adaln: A act [M, K] (silu rows), W act [2d, K] (cond_mlp.2.weight); X f32 [M, d] + gamma (+ beta) -> Yact / Y32 [M, d], or — X None —
    Yact act [M, 2d] receives the modulation itself."""
    g.A, g.lda, g.W, g.ldw, g.bias = A.data_ptr(), A.stride(0), W.data_ptr(), W.stride(0), N.ptr(bias)
    g.X, g.ldx = N.ptr(X), ((ldx if ldx is not None else X.stride(0)) if X is not None else 0)
    g.gamma, g.beta = N.ptr(gamma), N.ptr(beta)
    g.Yact, g.ldyact = N.ptr(Yact), (Yact.stride(0) if Yact is not None else 0)
    g.Y32, g.ldy32 = N.ptr(Y32), ((ldy32 if ldy32 is not None else Y32.stride(0)) if Y32 is not None else 0)
    g.mean, g.rstd = N.ptr(mean), N.ptr(rstd)
    g.M, g.d, g.K = A.shape[0], W.shape[0] // 2, W.shape[1]


def gemm_adaln(groups: Sequence[Dict], eps: float = 1e-5, dtype: torch.dtype = torch.bfloat16) -> None:
    """sea_gemm_adaln: AdaLN as the epilogue of cond_mlp.2's GEMM (dict keys: the arguments of fill_adaln_group)."""
    arr = (N.SeaAdalnGroup * len(groups))()
    for g, d in zip(arr, groups):
        fill_adaln_group(g, **d)
    N.check(N.lib().sea_gemm_adaln(arr, len(groups), eps, N.dtype_code(dtype), N.stream_ptr()), "sea_gemm_adaln")


def adaln_qkv_supported(dtype: torch.dtype, E: int, H: int) -> bool:
    """Shapes sea_adaln_qkv instantiates (include/sea_hip.h): bf16, E = 256, head dim 16 or 32."""
    return dtype == torch.bfloat16 and E == 256 and H > 0 and E % H == 0 and E // H in (16, 32)


def fill_adaln_qkv(g: N.SeaAdalnQkv, X, cond, w1, b1, W2c, b2c, gamma, beta, Wqkv, bqkv, Q, K, Vt, ldx=None, third=None) -> None:
    """One group of sea_adaln_qkv: X f32 [M, E] (row stride ldx), cond f32 [M], cond_mlp.0 (w1, b1 f32 [2E]), cond_mlp.2 (W2c act [2E, 2E], b2c), AdaLN_0's gamma / beta,
    [Wq; Wk; Wv] act [3E, E] + bias, the attention operands Q [B,H,T,hd] / K [B,H,cap,hd] / Vt [B,H,hd,cap]."""
    g.X, g.ldx, g.cond = X.data_ptr(), (ldx if ldx is not None else X.stride(0)), N.ptr(cond)
    g.w1, g.b1, g.W2c, g.ldw2c, g.b2c = w1.data_ptr(), b1.data_ptr(), W2c.data_ptr(), W2c.stride(0), N.ptr(b2c)
    g.gamma, g.beta, g.Wqkv, g.ldw, g.bqkv = gamma.data_ptr(), N.ptr(beta), Wqkv.data_ptr(), Wqkv.stride(0), N.ptr(bqkv)
    g.Q, g.K, g.Vt = Q.data_ptr(), K.data_ptr(), Vt.data_ptr()
    g.M, g.E = X.shape[0], Wqkv.shape[1]
    if third is not None:   # dict(w1, b1 f32 [N3], W act [N3, N3], bias f32 [N3] or None, out act [M, N3]): the modulation of another AdaLN module of the same rows
        W3, o3 = third["W"], third["out"]
        g.w13, g.b13, g.W3, g.ldw3, g.b3 = third["w1"].data_ptr(), third["b1"].data_ptr(), W3.data_ptr(), W3.stride(0), N.ptr(third.get("bias"))
        g.mod3, g.ldmod3, g.N3 = o3.data_ptr(), o3.stride(0), W3.shape[0]


def adaln_qkv(groups: Sequence[Dict], rope: torch.Tensor, H: int, hd: int, T: int, pos0: int, cap: int, q_scale: float, riders: Sequence[Dict] = (), silu: Sequence[Dict] = (),
              silu_c: Optional[torch.Tensor] = None, eps: float = 1e-5, dtype: torch.dtype = torch.bfloat16) -> None:
    """sea_adaln_qkv: the front of a block in one launch (dict keys: the arguments of fill_adaln_qkv); `riders`: plain GEMM groups (A, W, bias, Cact) as sea_gemm_grouped's;
    `silu`: row riders as sea_silu_outer's groups (w1, b1, Hid) evaluated on silu_c."""
    arr = (N.SeaAdalnQkv * len(groups))()
    for g, d in zip(arr, groups):
        fill_adaln_qkv(g, **d)
    c = N.SeaQkvCommon()
    c.rope, c.H, c.hd, c.T, c.pos0, c.cap, c.q_scale = rope.data_ptr(), H, hd, T, pos0, cap, q_scale
    rarr = None
    if riders:
        rarr = (N.SeaGemmGroup * len(riders))()
        for g, d in zip(rarr, riders):
            fill_gemm_group(g, **d)
    sarr = None
    if silu:
        sarr = (N.SeaSiluGroup * len(silu))()
        for g, gd in zip(sarr, silu):
            g.w1, g.b1, g.Hid, g.K2, g.ld = gd["w1"].data_ptr(), gd["b1"].data_ptr(), gd["Hid"].data_ptr(), gd["Hid"].shape[1], gd["Hid"].stride(0)
    N.check(N.lib().sea_adaln_qkv(arr, len(groups), C.byref(c), rarr, len(riders), sarr, len(silu), (silu_c.data_ptr() if silu_c is not None else None),
                                  (silu_c.numel() if silu_c is not None else 0), None, eps, N.dtype_code(dtype), N.stream_ptr()), "sea_adaln_qkv")


def row_chain_supported(dtype: torch.dtype, D: int, E: int, n_seg: int, hd: int) -> bool:
    """Shapes sea_row_chain instantiates (include/sea_hip.h): bf16, (D, E) in {(128, 256), (64, 128)}, n_seg * D <= E, cross head dim 16 or 32."""
    return dtype == torch.bfloat16 and (D, E) in ((128, 256), (64, 128)) and 0 <= n_seg and n_seg * D <= E and hd in (16, 32)


def fill_row_chain(P: N.SeaRowChain, W2, Xin, X, att: Sequence[torch.Tensor] = (), Wp: Sequence[torch.Tensor] = (), a2=None, b2=None, bias_scale: float = 1.0, Xact=None,
                   down: Optional[Dict] = None, proj: Sequence[Dict] = (), ldxin: Optional[int] = None) -> None:
    """One group of sea_row_chain.  Form A: a2 act [M, E], W2 act [E, E]; form B: att / Wp lists (n_seg act [M, D] / [D, D]), W2 act [E, D].  Xin f32 [M, E] (row
    stride ldxin when given: the caller's strided tensor), X f32 [M, E]; down: dict(W [D, E], bias, gamma, beta, mod, Yact, Y32); proj: dicts(W [N, D], bias, col0, Q / K / Vt / V)."""
    for s_, a in enumerate(att):
        P.att[s_] = a.data_ptr()
        P.Wp[s_] = Wp[s_].data_ptr()
    P.n_seg = len(att)
    if att:
        P.ldatt, P.ldwp = att[0].stride(0), Wp[0].stride(0)
    if a2 is not None:
        P.a2, P.lda2 = a2.data_ptr(), a2.stride(0)
    P.W2, P.ldw2, P.b2, P.bias_scale = W2.data_ptr(), W2.stride(0), N.ptr(b2), bias_scale
    P.Xin, P.ldxin = Xin.data_ptr(), (ldxin if ldxin is not None else Xin.stride(0))
    P.X, P.ldx = X.data_ptr(), X.stride(0)
    P.Xact, P.ldxact = N.ptr(Xact), (Xact.stride(0) if Xact is not None else 0)
    P.M, P.E = X.shape[0], W2.shape[0]
    P.D = P.E // 2
    P.has_down = int(down is not None)
    if down is not None:
        g = P.down
        g.W, g.ldw, g.bias = down["W"].data_ptr(), down["W"].stride(0), N.ptr(down.get("bias"))
        g.gamma, g.beta = down["gamma"].data_ptr(), N.ptr(down.get("beta"))
        mod, y32, yact = down.get("mod"), down.get("Y32"), down.get("Yact")
        g.mod, g.ldmod = N.ptr(mod), (mod.stride(0) if mod is not None else 0)
        g.Y32, g.ldy32 = N.ptr(y32), (y32.stride(0) if y32 is not None else 0)
        g.Yact, g.ldyact = N.ptr(yact), (yact.stride(0) if yact is not None else 0)
        g.mean, g.rstd = N.ptr(down.get("mean")), N.ptr(down.get("rstd"))
        g.M, g.N, g.K, g.n_seg = P.M, P.D, P.E, 1   # (the launcher sets them again; here for the pointer audit's extents)
    if len(proj) > N.CHAIN_MAX_PROJ:
        raise ValueError(f"row_chain: {len(proj)} projection entries (at most {N.CHAIN_MAX_PROJ})")
    P.n_proj = len(proj)
    for q, d in zip(P.proj, proj):
        W = d["W"]
        q.A, q.lda, q.M = None, 0, P.M
        q.W, q.ldw, q.bias = W.data_ptr(), W.stride(0), N.ptr(d.get("bias"))
        q.N, q.K, q.col0 = W.shape[0], W.shape[1], d.get("col0", 0)
        q.Qout, q.Kout, q.Vtout, q.Vout = N.ptr(d.get("Q")), N.ptr(d.get("K")), N.ptr(d.get("Vt")), N.ptr(d.get("V"))


def row_chain(groups: Sequence[Dict], rope: Optional[torch.Tensor] = None, H: int = 1, hd: int = 4, T: int = 1, pos0: int = 0, cap: int = 0, q_scale_: float = 1.0,
              eps: float = 1e-5, dtype: torch.dtype = torch.bfloat16) -> None:
    """sea_row_chain: the row-local chain between two attention launches for up to three groups (fields) of one shape (dict keys: the arguments of fill_row_chain)."""
    P = (N.SeaRowChain * len(groups))()
    for p_, d in zip(P, groups):
        fill_row_chain(p_, **d)
    common = N.SeaQkvCommon(N.ptr(rope), H, hd, T, pos0, cap, q_scale_) if rope is not None else None
    N.check(N.lib().sea_row_chain(P, len(groups), C.byref(common) if common is not None else None, eps, N.dtype_code(dtype), N.stream_ptr()), "sea_row_chain")


def silu_outer(groups: Sequence[Dict], c: torch.Tensor, M: int, dtype: torch.dtype) -> None:
    """groups: dicts with w1 f32 [K2], b1 f32 [K2], Hid act [M,K2]."""
    n = len(groups)
    arr = (N.SeaSiluGroup * n)()
    for i, gd in enumerate(groups):
        g = arr[i]
        H = _mat(gd["Hid"], "Hid")
        g.w1, g.b1, g.Hid, g.K2, g.ld = gd["w1"].data_ptr(), gd["b1"].data_ptr(), H.data_ptr(), H.shape[1], H.stride(0)
    N.require_gpu(c, "c")
    assert c.dtype == torch.float32 and c.is_contiguous() and c.numel() == M
    N.check(N.lib().sea_silu_outer(arr, n, c.data_ptr(), M, N.dtype_code(dtype), N.stream_ptr()), "sea_silu_outer")


def ib_add(xs: Sequence[torch.Tensor], c: torch.Tensor, w1, b1, lnw, lnb, w2, b2, drop=None) -> None:
    """xs: f32 [M,E] matrices (equal row stride) updated in place: x += W2 gelu(LN(w1 c + b1)) + b2; drop = (seed, first stream, thr): dropout of the MLP output,
    field i on stream + i."""
    P = N.SeaIbParams()
    if drop is not None:
        P.drop.seed, P.drop.stream, P.drop.thr = drop
    M, E = xs[0].shape
    for i, x in enumerate(xs):
        _mat(x, "x")
        assert x.dtype == torch.float32 and x.shape == (M, E) and x.stride(0) == xs[0].stride(0)
        P.X[i] = x.data_ptr()
    P.n_fields, P.ldx = len(xs), xs[0].stride(0)
    for t in (c, w1, b1, lnw, lnb, w2, b2):
        N.require_gpu(t, "ib parameter")
        assert t.dtype == torch.float32 and t.is_contiguous()
    P.c, P.w1, P.b1, P.lnw, P.lnb, P.w2, P.b2 = (t.data_ptr() for t in (c, w1, b1, lnw, lnb, w2, b2))
    P.M, P.E, P.h = M, E, w1.numel()
    assert c.numel() == M and w2.shape == (E, P.h)
    N.check(N.lib().sea_ib_add(C.byref(P), N.stream_ptr()), "sea_ib_add")


def convert(src: torch.Tensor, dst: torch.Tensor) -> None:
    """dst[r, c] = (dst dtype) src[r, c]; src float32, 2-D with unit inner stride."""
    src, dst = _mat(src, "src"), _mat(dst, "dst")
    assert src.dtype == torch.float32 and src.shape == dst.shape
    N.check(N.lib().sea_convert_f32_to_act(src.data_ptr(), src.stride(0), dst.data_ptr(), dst.stride(0), src.shape[0],
                                           src.shape[1], N.dtype_code(dst.dtype), N.stream_ptr()), "sea_convert_f32_to_act")


# ------------------------------------------------------------------------------------------------ backward wrappers
def wgrad_grouped(groups: Sequence[Dict], dtype: torch.dtype) -> None:
    """groups: dicts with dY act [M,N], X act [M,K], dW f32 [N,K] (accumulated), optional db f32 [N] (accumulated)."""
    n = len(groups)
    arr = (N.SeaWgradGroup * n)()
    for g, d in zip(arr, groups):
        dY, X, dW = _mat(d["dY"], "dY"), _mat(d["X"], "X"), _mat(d["dW"], "dW")
        assert dY.dtype == dtype and X.dtype == dtype and dW.dtype == torch.float32
        g.dY, g.X, g.dW, g.db = dY.data_ptr(), X.data_ptr(), dW.data_ptr(), N.ptr(d.get("db"))
        g.lddy, g.ldx, g.lddw = dY.stride(0), X.stride(0), dW.stride(0)
        g.M, g.N, g.K = dY.shape[0], dY.shape[1], X.shape[1]
        assert X.shape[0] == g.M and dW.shape == (g.N, g.K)
    N.check(N.lib().sea_wgrad_grouped(arr, n, N.dtype_code(dtype), N.stream_ptr()), "sea_wgrad_grouped")


def rownorm_bwd(groups: Sequence[Dict], M: int, d: int, dy_is_act: bool, x_is_act: bool, gelu: bool, accumulate: bool,
                dtype: torch.dtype, ws: Optional[torch.Tensor] = None) -> None:
    n = len(groups)
    arr = (N.SeaNormBwdGroup * n)()
    for g, gd in zip(arr, groups):
        dY, X = _mat(gd["dY"], "dY"), _mat(gd["X"], "X")
        g.dY, g.lddy, g.X, g.ldx = dY.data_ptr(), dY.stride(0), X.data_ptr(), X.stride(0)
        mod, dmod = gd.get("mod"), gd.get("dmod")
        g.mod, g.ldmod = N.ptr(mod), (mod.stride(0) if mod is not None else 0)
        g.dmod, g.lddmod = N.ptr(dmod), (dmod.stride(0) if dmod is not None else 0)
        g.gamma, g.beta = gd["gamma"].data_ptr(), N.ptr(gd.get("beta"))
        g.mean, g.rstd = gd["mean"].data_ptr(), gd["rstd"].data_ptr()
        dx32, dxa = gd.get("dX32"), gd.get("dXact")
        g.dX32, g.lddx32 = N.ptr(dx32), (dx32.stride(0) if dx32 is not None else 0)
        g.dXact, g.lddxact = N.ptr(dxa), (dxa.stride(0) if dxa is not None else 0)
        g.dgamma, g.dbeta = N.ptr(gd.get("dgamma")), N.ptr(gd.get("dbeta"))
    N.check(N.lib().sea_rownorm_bwd(arr, n, M, d, int(dy_is_act), int(x_is_act), int(gelu), int(accumulate), N.dtype_code(dtype),
                                    N.ptr(ws), 0 if ws is None else ws.numel(), N.stream_ptr()), "sea_rownorm_bwd")


def silu_outer_bwd(groups: Sequence[Dict], c: torch.Tensor, M: int, dtype: torch.dtype, ws: Optional[torch.Tensor] = None) -> None:
    n = len(groups)
    arr = (N.SeaSiluBwdGroup * n)()
    for g, gd in zip(arr, groups):
        dH = _mat(gd["dHid"], "dHid")
        g.dHid, g.w1, g.b1, g.dw1, g.db1 = dH.data_ptr(), gd["w1"].data_ptr(), gd["b1"].data_ptr(), gd["dw1"].data_ptr(), gd["db1"].data_ptr()
        g.K2, g.ld = dH.shape[1], dH.stride(0)
    N.check(N.lib().sea_silu_outer_bwd(arr, n, c.data_ptr(), M, N.dtype_code(dtype), N.ptr(ws), 0 if ws is None else ws.numel(),
                                       N.stream_ptr()), "sea_silu_outer_bwd")


def ib_bwd(dxs: Sequence[torch.Tensor], c, w1, b1, lnw, lnb, w2, dw1, db1, dlnw, dlnb, dw2, db2, ws: Optional[torch.Tensor] = None,
           dhid: Optional[torch.Tensor] = None) -> None:
    """ws (f32, >= E (1 + h) floats per row split) and dhid (f32 [M, 8], zero on entry) select the column-block form (h <= 8)."""
    P = N.SeaIbBwdParams()
    M, E = dxs[0].shape
    if ws is not None and dhid is not None:
        P.ws, P.ws_floats, P.dhid = ws.data_ptr(), ws.numel(), dhid.data_ptr()
    for i, x in enumerate(dxs):
        _mat(x, "dx")
        assert x.dtype == torch.float32 and x.stride(0) == dxs[0].stride(0)
        P.dX[i] = x.data_ptr()
    P.n_fields, P.ldx = len(dxs), dxs[0].stride(0)
    P.c, P.w1, P.b1, P.lnw, P.lnb, P.w2 = (t.data_ptr() for t in (c, w1, b1, lnw, lnb, w2))
    P.dw1, P.db1, P.dlnw, P.dlnb, P.dw2, P.db2 = (t.data_ptr() for t in (dw1, db1, dlnw, dlnb, dw2, db2))
    P.M, P.E, P.h = M, E, w1.numel()
    N.check(N.lib().sea_ib_bwd(C.byref(P), N.stream_ptr()), "sea_ib_bwd")


def transpose_weights(src_flat: torch.Tensor, dst_flat: torch.Tensor, desc: torch.Tensor, tile_start: torch.Tensor) -> None:
    """desc int64 [n,4] (src_off, dst_off, rows, cols) and tile_start int32 [n+1] live on the device."""
    n = desc.shape[0]
    total = int(tile_start[-1].item()) if not hasattr(tile_start, "_sea_total") else tile_start._sea_total
    N.check(N.lib().sea_transpose_weights(src_flat.data_ptr(), dst_flat.data_ptr(), N.dtype_code(dst_flat.dtype), desc.data_ptr(),
                                          tile_start.data_ptr(), n, total, N.stream_ptr()), "sea_transpose_weights")


def attention_bwd(problems: Sequence[Dict], rope: torch.Tensor, B: int, H: int, hd: int, Tq: int, Tk: int, cap: int, q_pos0: int,
                  src_len: int, q_scale: float, dtype: torch.dtype) -> None:
    """problems: dicts with Q, K, V (row-major), O, dO [B,Tq,H*hd], LSE, delta f32 [B,H,Tq], dQ/dK/dV act [B*T, >= H*hd]."""
    P = N.SeaAttnBwdParams()
    P.n_problems = len(problems)
    for i, d in enumerate(problems):
        q = P.p[i]
        for k in ("Q", "K", "V", "O", "dO", "dQ", "dK", "dV"):
            N.require_gpu(d[k], k)
            assert d[k].dtype == dtype, k
        q.Q, q.K, q.V, q.O, q.dO = (d[k].data_ptr() for k in ("Q", "K", "V", "O", "dO"))
        q.LSE, q.delta = d["LSE"].data_ptr(), d["delta"].data_ptr()
        q.dQ, q.dK, q.dV = d["dQ"].data_ptr(), d["dK"].data_ptr(), d["dV"].data_ptr()
    d0 = problems[0]
    P.rope = rope.data_ptr()
    P.B, P.H, P.hd, P.Tq, P.Tk, P.cap, P.q_pos0, P.src_len = B, H, hd, Tq, Tk, cap, q_pos0, src_len
    P.ldo, P.lddo = d0["O"].stride(-2), d0["dO"].stride(-2)
    P.lddq, P.lddk, P.lddv = d0["dQ"].stride(-2), d0["dK"].stride(-2), d0["dV"].stride(-2)
    P.q_scale = q_scale
    N.check(N.lib().sea_attention_bwd(C.byref(P), N.dtype_code(dtype), N.stream_ptr()), "sea_attention_bwd")


def unpatchify(fields: torch.Tensor, layout: str, index_map: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, n_points: int,
               point_slot: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[b, index_map[p, c], f] = fields[b, p, f|c ...] * scale[f] + shift[f] (sea_unpatchify).  `layout` names the order of the last two
    axes of `fields`: "BPFC" (the decoder's output) or "BPCF" (the reference's inverse_scale_and_unpatch argument)."""
    N.require_gpu(fields, "fields")
    assert fields.dtype == torch.float32 and fields.dim() == 4 and layout in ("BPFC", "BPCF")
    assert index_map.dtype == torch.int32 and index_map.is_cuda and index_map.is_contiguous()
    B, P = fields.shape[0], fields.shape[1]
    F, Cc = (fields.shape[2], fields.shape[3]) if layout == "BPFC" else (fields.shape[3], fields.shape[2])
    sb, sp = fields.stride(0), fields.stride(1)
    sf, sc = (fields.stride(2), fields.stride(3)) if layout == "BPFC" else (fields.stride(3), fields.stride(2))
    assert tuple(index_map.shape) == (P, Cc) and scale.numel() == F and shift.numel() == F
    out = torch.empty(B, n_points, F, device=fields.device, dtype=torch.float32)
    if point_slot is not None:
        assert point_slot.dtype == torch.int32 and point_slot.is_cuda and point_slot.numel() == n_points
    N.check(N.lib().sea_unpatchify(fields.data_ptr(), sb, sp, sf, sc, index_map.data_ptr(), N.ptr(point_slot), scale.data_ptr(), shift.data_ptr(),
                                   out.data_ptr(), B, P, F, Cc, n_points, N.stream_ptr()), "sea_unpatchify")
    return out


def patchify(fields: torch.Tensor, index_map: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, layout: str = "BPFC", c_out: Optional[int] = None,
             pad_value: float = 0.0) -> torch.Tensor:
    """out[b, p, f, c] = fields[b, index_map[p, c], f] * scale[f] + shift[f], pad_value at empty / padded slots (sea_patchify).  fields f32
    [B, n_points, F]; `layout` of the result: "BPFC" (the encoder's input, C padded to c_out) or "BPCF" (the reference's stacked fields)."""
    N.require_gpu(fields, "fields")
    assert fields.dtype == torch.float32 and fields.dim() == 3 and fields.is_contiguous() and layout in ("BPFC", "BPCF")
    assert index_map.dtype == torch.int32 and index_map.is_cuda and index_map.is_contiguous()
    B, n_points, F = fields.shape
    P, C_map = index_map.shape
    C_out = C_map if c_out is None else c_out
    assert C_out >= C_map and scale.numel() == F and shift.numel() == F
    out = torch.empty((B, P, F, C_out) if layout == "BPFC" else (B, P, C_out, F), device=fields.device, dtype=torch.float32)
    sb, sp = out.stride(0), out.stride(1)
    sf, sc = (out.stride(2), out.stride(3)) if layout == "BPFC" else (out.stride(3), out.stride(2))
    N.check(N.lib().sea_patchify(fields.data_ptr(), index_map.data_ptr(), scale.data_ptr(), shift.data_ptr(), out.data_ptr(), sb, sp, sf, sc, B, P, F, C_map, C_out,
                                 n_points, pad_value, N.stream_ptr()), "sea_patchify")
    return out


def splitk_finish(groups: Sequence[Dict], dtype: torch.dtype) -> None:
    """sea_splitk_finish: out = sum_s P[s] + bias * bias_scale + R; dicts with P f32 [S, M, N], optional bias / bias_scale / R, outputs C32 and / or Cact."""
    arr = (N.SeaSplitkGroup * len(groups))()
    for g, d in zip(arr, groups):
        P, R, C32, Cact = d["P"], d.get("R"), d.get("C32"), d.get("Cact")
        g.P, g.p_stride, g.S, g.M, g.N, g.ldp = P.data_ptr(), P.stride(0), P.shape[0], P.shape[1], P.shape[2], P.stride(1)
        g.bias, g.bias_scale = N.ptr(d.get("bias")), d.get("bias_scale", 1.0)
        g.R, g.ldr = N.ptr(R), (R.stride(0) if R is not None else 0)
        g.C32, g.ldc32 = N.ptr(C32), (C32.stride(0) if C32 is not None else 0)
        g.Cact, g.ldcact = N.ptr(Cact), (Cact.stride(0) if Cact is not None else 0)
    N.check(N.lib().sea_splitk_finish(arr, len(groups), N.dtype_code(dtype), N.stream_ptr()), "sea_splitk_finish")


def mlp_fc1_supported(dtype: torch.dtype, E: int, S: int) -> bool:
    """Shapes sea_mlp_fc1_ln_gelu instantiates (include/sea_hip.h)."""
    return dtype == torch.bfloat16 and (E, S) in ((256, 2048), (128, 1024))


def fill_mlp_group(g: N.SeaMlpGroup, A, W1, b1, lnw, lnb, Hg, norm: Optional[Dict] = None) -> None:
    """norm (optional): dict(X32 f32 [M,E], gamma, beta=None, mod=None, addend=None, Xout=None, eps=1e-5) — the operand rows are normalised inside
    the launch from the fp32 residual stream (A is then None)."""
    g.W1, g.b1, g.lnw, g.lnb, g.Hg = W1.data_ptr(), b1.data_ptr(), lnw.data_ptr(), lnb.data_ptr(), N.ptr(Hg)   # (Hg None: sea_mlp_block)
    g.ldw, g.ldh = W1.stride(0), (Hg.stride(0) if Hg is not None else 0)
    g.E, g.S = W1.shape[1], W1.shape[0]
    if norm is None:
        g.A, g.lda, g.M = A.data_ptr(), A.stride(0), A.shape[0]
        g.X32 = None
        return
    X = norm["X32"]
    g.A, g.lda, g.M = None, 0, X.shape[0]
    g.X32, g.ldx32 = X.data_ptr(), X.stride(0)
    add, xout, mod = norm.get("addend"), norm.get("Xout"), norm.get("mod")
    g.addend, g.ldadd = N.ptr(add), (add.stride(0) if add is not None else 0)
    g.Xout, g.ldxout = N.ptr(xout), (xout.stride(0) if xout is not None else 0)
    g.mod, g.ldmod = N.ptr(mod), (mod.stride(0) if mod is not None else 0)
    g.gamma, g.beta = norm["gamma"].data_ptr(), N.ptr(norm.get("beta"))
    g.norm_eps = norm.get("eps", 1e-5)


def fill_mlp2_group(g: N.SeaMlp2Group, Hg, W2, b2, R, Wproj, bproj, Y32=None, Yact=None, gamma=None, beta=None, mod=None, ldy32=None, M=None) -> None:
    """Hg / R may be None for sea_mlp_block (hidden rows in registers; residual = the norm prologue's x + addend): M is then given."""
    g.Hg, g.W2, g.b2, g.R, g.Wproj, g.bproj = N.ptr(Hg), W2.data_ptr(), b2.data_ptr(), N.ptr(R), Wproj.data_ptr(), bproj.data_ptr()
    g.ldh, g.ldw2, g.ldr, g.ldwp = (Hg.stride(0) if Hg is not None else 0), W2.stride(0), (R.stride(0) if R is not None else 0), Wproj.stride(0)
    g.gamma, g.beta, g.mod, g.ldmod = N.ptr(gamma), N.ptr(beta), N.ptr(mod), (mod.stride(0) if mod is not None else 0)
    g.Y32, g.ldy32 = N.ptr(Y32), (ldy32 if ldy32 is not None else (Y32.stride(0) if Y32 is not None else 0))
    g.Yact, g.ldyact = N.ptr(Yact), (Yact.stride(0) if Yact is not None else 0)
    g.M, g.E, g.S = (Hg.shape[0] if Hg is not None else M), W2.shape[0], W2.shape[1]


def mlp_fc2_proj_norm(groups: Sequence[Dict], eps: float = 1e-5, dtype: torch.dtype = torch.bfloat16) -> None:
    """out = norm(proj(Hg W2^T + b2 + R)) in one launch (sea_mlp_fc2_proj_norm): dicts with Hg [M,S], W2 [E,S], b2, R f32 [M,E], Wproj [E,E], bproj,
    Y32 and / or Yact, optional gamma / beta / mod (no gamma: no norm)."""
    arr = (N.SeaMlp2Group * len(groups))()
    for g, d in zip(arr, groups):
        for k in ("Hg", "W2", "Wproj", "R"):
            _mat(d[k], k)
        fill_mlp2_group(g, d["Hg"], d["W2"], d["b2"], d["R"], d["Wproj"], d["bproj"], d.get("Y32"), d.get("Yact"), d.get("gamma"), d.get("beta"), d.get("mod"),
                        d.get("ldy32"))
    N.check(N.lib().sea_mlp_fc2_proj_norm(arr, len(groups), eps, N.dtype_code(dtype), N.stream_ptr()), "sea_mlp_fc2_proj_norm")


def mlp_block(groups: Sequence[Dict], eps: float = 1e-5, dtype: torch.dtype = torch.bfloat16) -> None:
    """The whole field MLP, proj and the final norm in one launch (sea_mlp_block): dicts with the keys of mlp_fc1_ln_gelu (A or norm, W1, b1, lnw, lnb) and of
    mlp_fc2_proj_norm (W2, b2, R — or None with a norm prologue —, Wproj, bproj, Y32 / Yact, optional gamma / beta / mod); no Hg."""
    a1, a2 = (N.SeaMlpGroup * len(groups))(), (N.SeaMlp2Group * len(groups))()
    for g1, g2, d in zip(a1, a2, groups):
        fill_mlp_group(g1, d.get("A"), d["W1"], d["b1"], d["lnw"], d["lnb"], None, d.get("norm"))
        fill_mlp2_group(g2, None, d["W2"], d["b2"], d.get("R"), d["Wproj"], d["bproj"], d.get("Y32"), d.get("Yact"), d.get("gamma"), d.get("beta"), d.get("mod"), d.get("ldy32"), M=g1.M)
    N.check(N.lib().sea_mlp_block(a1, a2, len(groups), eps, N.dtype_code(dtype), N.stream_ptr()), "sea_mlp_block")


def mlp_fc1_ln_gelu(groups: Sequence[Dict], eps: float = 1e-5, dtype: torch.dtype = torch.bfloat16) -> None:
    """Hg = gelu(LayerNorm(A W1^T + b1) * lnw + lnb) in one launch (sea_mlp_fc1_ln_gelu): dicts with A [M,E], W1 [S,E], b1, lnw, lnb f32 [S], Hg [M,S]."""
    arr = (N.SeaMlpGroup * len(groups))()
    for g, d in zip(arr, groups):
        for k in ("W1", "Hg") + (("A",) if d.get("norm") is None else ()):
            _mat(d[k], k)
        fill_mlp_group(g, d.get("A"), d["W1"], d["b1"], d["lnw"], d["lnb"], d["Hg"], d.get("norm"))
    N.check(N.lib().sea_mlp_fc1_ln_gelu(arr, len(groups), eps, N.dtype_code(dtype), N.stream_ptr()), "sea_mlp_fc1_ln_gelu")
