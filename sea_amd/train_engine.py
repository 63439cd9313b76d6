"""Training plan: forward with saved activations + hand-written backward of TemporalModel as two fixed launch lists.

The reference trains with `loss.backward()` through eager autograd (train/train_temporal.py:255-257).  Here the backward is a
second pre-built launch list over the activations the forward list saved: data gradients run through the same grouped GEMM
kernel against the W^T shadow, parameter gradients accumulate (fp32) into ONE flat buffer laid out like the parameters, so the
data-parallel all-reduce and the AdamW kernel each touch a single contiguous range.
"""
from __future__ import annotations

import os

import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch

from . import _native as N
from . import _switches
from . import ops
from .engine import Plan, _Rec, blk_pe


class TrainPlan(Plan):
    def __init__(self, eng, B: int, T: int, drop_thr: int = 0, dp: bool = False):
        self.drop_thr = drop_thr
        self.dp = bool(dp)   # data-parallel layout of the backward (the condition MLPs' backward per phase, two early slices): decided ONCE per engine (TemporalEngine.dp_overlap)
        self._next_stream = 0
        self.bwd: List[_Rec] = []
        self._buckets = None
        self._dout_patches: List[Tuple[object, str, int]] = []
        eng.params.enable_transposed_shadow()
        eng.ensure_grads()
        super().__init__(eng, B, T, "full")

    def _all_records(self):
        return list(self.records) + list(self.bwd)

    def _colsum_ws(self, n_floats: int) -> torch.Tensor:
        """The shared fp32 workspace of the two-stage column sums (launches on one stream never overlap), sized once for the largest
        launch of the plan."""
        ws = getattr(self, "_ws_colsum", None)
        if ws is None:
            nb512, nb256 = min((self.M + 3) // 4, 512), min((self.M + 3) // 4, 256)
            bound = max(N.MAX_NORM_BWD_GROUPS * nb512 * 2 * self.E, self.F * nb512 * 2 * self.S, N.MAX_SILU_BWD_GROUPS * nb256 * 4 * self.E)
            ws = self._ws_colsum = torch.empty(bound, device=self.eng.device, dtype=torch.float32)
        assert n_floats <= ws.numel(), (n_floats, ws.numel())
        return ws

    def _streams(self, n: int) -> int:
        """Reserve n consecutive dropout stream ids; returns the first."""
        s = self._next_stream
        self._next_stream += n
        return s

    # ------------------------------------------------------------------ record builders (backward)
    def _wgrad(self, groups: List[dict], name: str, sole: bool = False) -> None:
        """`sole`: this launch is the ONLY contribution to its dW tensors in a backward (the field MLP's matrices): when the gradient buffer is known to hold
        zeros (engine.backward sets SeaWgradGroup.overwrite per run from grads_dirty) the kernel may store instead of adding atomically."""
        L = N.lib()
        for s in range(0, len(groups), N.MAX_WGRAD_GROUPS):
            chunk = groups[s:s + N.MAX_WGRAD_GROUPS]
            arr = (N.SeaWgradGroup * len(chunk))()
            for g, d in zip(arr, chunk):
                dY, X, dW = d["dY"], d["X"], d["dW"]
                g.dY, g.X, g.dW, g.db = dY.data_ptr(), X.data_ptr(), dW.data_ptr(), N.ptr(d.get("db"))
                g.lddy, g.ldx, g.lddw = dY.stride(0), X.stride(0), dW.stride(0)
                g.M, g.N, g.K = self.M, dW.shape[0], dW.shape[1]
                assert dY.shape[1] == g.N and X.shape[1] == g.K, (name, dY.shape, X.shape, dW.shape)
                if sole:
                    self.__dict__.setdefault("_sole_wgrads", []).append(g)
            self._cur.append(_Rec(L.sea_wgrad_grouped, [arr, len(chunk), self.code], name, arr))

    def _norm_bwd(self, groups: List[dict], d: int, name: str, dy_is_act: bool, x_is_act: bool, gelu: bool, accumulate: bool) -> None:
        L = N.lib()
        for s in range(0, len(groups), N.MAX_NORM_BWD_GROUPS):
            chunk = groups[s:s + N.MAX_NORM_BWD_GROUPS]
            arr = (N.SeaNormBwdGroup * len(chunk))()
            for g, gd in zip(arr, chunk):
                dY, X = gd["dY"], gd["X"]
                g.dY, g.lddy = dY.data_ptr(), gd.get("lddy", dY.stride(0))
                g.X, g.ldx = X.data_ptr(), gd.get("ldx", X.stride(0))
                mod, dmod = gd.get("mod"), gd.get("dmod")
                g.mod, g.ldmod = N.ptr(mod), (mod.stride(0) if mod is not None else 0)
                g.dmod, g.lddmod = N.ptr(dmod), (dmod.stride(0) if dmod is not None else 0)
                g.gamma, g.beta = gd["gamma"].data_ptr(), N.ptr(gd.get("beta"))
                g.mean, g.rstd = gd["mean"].data_ptr(), gd["rstd"].data_ptr()
                dx32, dxa = gd.get("dX32"), gd.get("dXact")
                g.dX32, g.lddx32 = N.ptr(dx32), (dx32.stride(0) if dx32 is not None else 0)
                g.dXact, g.lddxact = N.ptr(dxa), (dxa.stride(0) if dxa is not None else 0)
                g.dgamma, g.dbeta = N.ptr(gd.get("dgamma")), N.ptr(gd.get("dbeta"))
                if gd.get("X_is_x") is not None:
                    self._x_patches.append((g, "X", gd["X_is_x"]))
                if gd.get("dY_is_dout") is not None:
                    self._dout_patches.append((g, "dY", gd["dY_is_dout"]))
            ws = self._colsum_ws(len(chunk) * min((self.M + 3) // 4, 512) * 2 * d)
            self._cur.append(_Rec(L.sea_rownorm_bwd, [arr, len(chunk), self.M, d, int(dy_is_act), int(x_is_act), int(gelu), int(accumulate),
                                                      self.code, ws.data_ptr(), ws.numel()], name, arr))

    def _attn_bwd(self, problems: List[dict], hd: int, rope: torch.Tensor, name: str, drop=None, src_len: Optional[int] = None) -> None:
        L = N.lib()
        P = N.SeaAttnBwdParams()
        P.n_problems = len(problems)
        if drop is not None:
            P.drop.thr, P.drop.stream = drop
            self._drop_structs.append(P)
        for i, d in enumerate(problems):
            q = P.p[i]
            q.Q, q.K, q.V, q.O, q.dO = (d[k].data_ptr() for k in ("Q", "K", "V", "O", "dO"))
            q.LSE, q.delta = d["LSE"].data_ptr(), d["delta"].data_ptr()
            q.dQ, q.dK, q.dV = d["dQ"].data_ptr(), d["dK"].data_ptr(), d["dV"].data_ptr()
        d0 = problems[0]
        P.rope = rope.data_ptr()
        P.B, P.H, P.hd, P.Tq, P.Tk, P.cap, P.q_pos0, P.src_len = self.B, self.H, hd, self.T, self.T, self.cap, 0, (self.eng.model.src_len if src_len is None else src_len)
        P.ldo, P.lddo = d0["O"].stride(0), d0["dO"].stride(0)
        P.lddq, P.lddk, P.lddv = d0["dQ"].stride(0), d0["dK"].stride(0), d0["dV"].stride(0)
        P.q_scale = ops.q_scale(hd)   # what the QKV epilogue put on q (the kernel undoes it, and the log2 units of the scores)
        self._cur.append(_Rec(L.sea_attention_bwd, [C.byref(P), self.code], name, P))

    def _silu_bwd(self, groups: List[dict], name: str) -> None:
        L = N.lib()
        for s in range(0, len(groups), N.MAX_SILU_BWD_GROUPS):
            chunk = groups[s:s + N.MAX_SILU_BWD_GROUPS]
            arr = (N.SeaSiluBwdGroup * len(chunk))()
            for g, d in zip(arr, chunk):
                dH = d["dHid"]
                g.dHid, g.w1, g.b1, g.dw1, g.db1 = dH.data_ptr(), d["w1"].data_ptr(), d["b1"].data_ptr(), d["dw1"].data_ptr(), d["db1"].data_ptr()
                g.K2, g.ld = dH.shape[1], dH.stride(0)
            ws = self._colsum_ws(len(chunk) * min((self.M + 3) // 4, 256) * 2 * max(d["dHid"].shape[1] for d in chunk))
            rec = _Rec(L.sea_silu_outer_bwd, [arr, len(chunk), None, self.M, self.code, ws.data_ptr(), ws.numel()], name, arr)
            self._c_patches.append((rec.args, 2))
            self._cur.append(rec)

    def _ib_attn_fwd(self, pre: str, xs: List[torch.Tensor], sv: dict) -> None:
        """ib_addition_mode 'attention' with everything the backward needs kept (the inference form: engine.Plan._ib_attn): xs[i] += proj_i(attention(q_i(xs[i]),
        k_i / v_i(ib rows))), un-masked and un-rotated (models/temporal.py:117-118, models/base_blocks.py:205-243)."""
        eng, P, B, H, T, E, M, cap = self.eng, self.eng.params, self.B, self.H, self.T, self.E, self.M, self.cap
        F, hd, buf, f32 = len(xs), self.E // self.H, self._buf, torch.float32
        thr = self.drop_thr
        if thr:
            # the reference evaluates self.ib(x_additional_info) inside _add_info, once per field (models/temporal.py:110-118): in train() every field's
            # key / value rows come from their own dropout mask of the info-bottleneck MLP's output — F row sets, streams s0 .. s0 + F - 1 of ONE sea_ib_add
            s0 = self._streams(F)
            sv["ia_ib_drop"] = (thr, s0)
            sv["ia_ib_rows"] = self._ib_rows_multi(pre, F, (thr, s0))
            sv["ia_attn_drop"] = (thr, self._streams(F))
        else:
            sv["ia_ib_drop"] = sv["ia_attn_drop"] = None
            sv["ia_ib_rows"] = [self._ib_rows(pre)[1]] * F
        sv["ia_ib"] = sv["ia_ib_rows"][0]
        sv["ia_xq"] = [self._act_copy(xs[i], "ib.attn.x_act", keep=True) for i in range(F)]
        sv["ia"] = [dict(Q=buf(B, H, T, hd), K=buf(B, H, cap, hd, zero=True), V=buf(B, H, cap, hd, zero=True), Vt=buf(B, H, hd, cap, zero=True), O=buf(M, E),
                         LSE=buf(B, H, T, dtype=f32)) for _ in range(F)]
        qg = []
        for i in range(F):
            ca, pr = f"{pre}cross_attn_ib.{i}.", sv["ia"][i]
            qg.append(dict(A=sv["ia_xq"][i], W=P.act(ca + "q.weight"), bias=P.f32_vec(ca + "q.bias"), col0=0, Q=pr["Q"]))
            qg.append(dict(A=sv["ia_ib_rows"][i], W=P.act(ca + "k.weight", 2 * E), bias=P.f32_vec(ca + "k.bias", 2 * E), col0=E, K=pr["K"], Vt=pr["Vt"], V=pr["V"]))
        self._qkv(qg, eng.rope_identity(hd), hd, "ib.attn.qkv")
        self._attn([dict(Q=pr["Q"], K=pr["K"], Vt=pr["Vt"], O=pr["O"], LSE=pr["LSE"]) for pr in sv["ia"]], hd, E, "ib.attn.attention", drop=sv["ia_attn_drop"], src_len=cap)
        self._gemm([dict(A=sv["ia"][i]["O"], W=P.act(f"{pre}cross_attn_ib.{i}.projection.weight"), R=xs[i], C32=xs[i]) for i in range(F)], "ib.attn.proj")

    def _ib_attn_bwd(self, pre: str, sv: dict, dx: List[torch.Tensor], ga: List[torch.Tensor]) -> None:
        """Backward of _ib_attn_fwd: dx[i] (fp32) is the gradient of the rows it produced and ga[i] their activation-dtype copy; on return both hold the
        gradient of its INPUT rows (the residual passes through, the query path is added), and the info-bottleneck layer has received the sum of the fields'
        key / value paths."""
        eng, P, G2, Gv = self.eng, self.eng.params, self.eng.grad_mat, self.eng.grad_vec
        B, H, T, E, M, F = self.B, self.H, self.T, self.E, self.M, len(dx)
        hd, buf, f32 = E // H, self._buf, torch.float32
        self._wgrad([dict(dY=ga[i], X=sv["ia"][i]["O"], dW=G2(f"{pre}cross_attn_ib.{i}.projection.weight")) for i in range(F)], "bwd.ib.attn.proj.wgrad")
        datt = [buf(M, E) for _ in range(F)]
        self._gemm([dict(A=ga[i], W=P.actT(f"{pre}cross_attn_ib.{i}.projection.weight"), Cact=datt[i]) for i in range(F)], "bwd.ib.attn.proj.dgrad")
        dq, dkv = [buf(M, E) for _ in range(F)], [buf(M, 2 * E) for _ in range(F)]
        delta = [buf(B, H, T, dtype=f32) for _ in range(F)]
        self._attn_bwd([dict(Q=pr["Q"], K=pr["K"], V=pr["V"], O=pr["O"], dO=datt[i], LSE=pr["LSE"], delta=delta[i], dQ=dq[i], dK=dkv[i][:, :E], dV=dkv[i][:, E:])
                        for i, pr in enumerate(sv["ia"])], hd, eng.rope_identity(hd), "bwd.ib.attn.attention", drop=sv["ia_attn_drop"], src_len=self.cap)
        wg = []
        for i in range(F):
            ca = f"{pre}cross_attn_ib.{i}."
            wg.append(dict(dY=dq[i], X=sv["ia_xq"][i], dW=G2(ca + "q.weight"), db=Gv(ca + "q.bias")))
            wg.append(dict(dY=dkv[i], X=sv["ia_ib_rows"][i], dW=G2(ca + "k.weight", 2 * E), db=Gv(ca + "k.bias", 2 * E)))
        self._wgrad(wg, "bwd.ib.attn.qkv.wgrad")
        self._gemm([dict(A=dq[i], W=P.actT(f"{pre}cross_attn_ib.{i}.q.weight"), R=dx[i], C32=dx[i], Cact=ga[i]) for i in range(F)], "bwd.ib.attn.q.dgrad")
        if sv["ia_ib_drop"] is not None:   # every field's keys / values came from its own (dropped) rows: F gradients, masked per field by the ib backward
            dibs = [buf(M, E, dtype=f32) for _ in range(F)]
            self._gemm([dict(A=dkv[i], W=P.actT(f"{pre}cross_attn_ib.{i}.k.weight", 2 * E), C32=dibs[i]) for i in range(F)], "bwd.ib.attn.kv.dgrad")
            self._ib_bwd(pre, dibs, drop=sv["ia_ib_drop"])
            return
        dib = buf(M, E, dtype=f32)
        for i in range(F):   # sequential: every field's keys / values came from the same info-bottleneck rows
            g = dict(A=dkv[i], W=P.actT(f"{pre}cross_attn_ib.{i}.k.weight", 2 * E), C32=dib)
            if i > 0:
                g["R"] = dib
            self._gemm([g], f"bwd.ib.attn.kv.dgrad{i}")
        self._ib_bwd(pre, [dib])

    def _ib_bwd(self, pre: str, dxs: List[torch.Tensor], drop=None) -> None:
        P, G, mode = self.eng.params, self.eng.grad_view, self.eng.ib_mode
        if mode == 2:      # GaussianFourierProjection: its matrix is fixed (requires_grad=False in the reference): nothing to accumulate
            return
        ib = N.SeaIbBwdParams()
        for i, x in enumerate(dxs):
            ib.dX[i] = x.data_ptr()
        ib.n_fields, ib.ldx = len(dxs), dxs[0].stride(0)
        ib.M, ib.E, ib.mode = self.M, self.ib_dim, mode
        if mode == 1:      # nn.Linear(1, E): weight [E, 1], bias [E]
            ib.dw1, ib.db1 = G(pre + "ib.weight").data_ptr(), G(pre + "ib.bias").data_ptr()
        else:
            if drop is not None:
                ib.drop.thr, ib.drop.stream = drop
                self._drop_structs.append(ib)
            names = ("ib.layers.0.weight", "ib.layers.0.bias", "ib.layers.1.weight", "ib.layers.1.bias", "ib.layers.3.weight")
            ib.w1, ib.b1, ib.lnw, ib.lnb, ib.w2 = (P.f32(pre + n).data_ptr() for n in names)
            ib.dw1, ib.db1, ib.dlnw, ib.dlnb, ib.dw2 = (G(pre + n).data_ptr() for n in names)
            ib.db2 = G(pre + "ib.layers.3.bias").data_ptr()
            ib.h = self.eng.model.ib_hidden
            if ib.h <= 8:   # the column-block form (bwd.hip): partial sums per row split, d hidden per row (zeroed once: the launch leaves it zero)
                n_cb = (self.ib_dim + 255) // 256
                rs = max(1, min((1024 + n_cb - 1) // n_cb, (self.M + 15) // 16))
                ws = self._buf(rs * self.ib_dim * (1 + ib.h), dtype=torch.float32)
                dhid = self._buf(self.M, 8, dtype=torch.float32, zero=True)
                ib.ws, ib.ws_floats, ib.dhid = ws.data_ptr(), ws.numel(), dhid.data_ptr()
        self._c_patches.append((ib, "c"))
        self._cur.append(_Rec(N.lib().sea_ib_bwd, [C.byref(ib)], "bwd.ib", ib))

    def _convert(self, src32: torch.Tensor, dst: torch.Tensor, name: str) -> None:
        self._cur.append(_Rec(N.lib().sea_convert_f32_to_act, [src32.data_ptr(), src32.stride(0), dst.data_ptr(), dst.stride(0), src32.shape[0],
                                                               src32.shape[1], self.code], name))

    # ------------------------------------------------------------------ the plan
    def _build(self) -> None:
        eng, P, G = self.eng, self.eng.params, self.eng.grad_view
        F, E, D, S, M, B, T, H, NL = self.F, self.E, self.D, self.S, self.M, self.B, self.T, self.H, self.L
        L = N.lib()
        hd_s, hd_c = E // H, D // H
        Eo, concat = self.Eo, self.concat          # E: the row width inside a block; Eo: model input / output and proj output rows ('concat': E = Eo + 64)
        cap, FE, f32 = self.cap, F * Eo, torch.float32
        model = eng.model
        rope_s, rope_c = eng.rope_self, eng.rope_cross
        buf = self._buf
        # variants of the block (reference models/temporal.py:285-312, 103-116): exchange 'sea' (Gauss-Seidel cross-attention) | 'addition' (Jacobi sum of the
        # normalised down-projections) | 'simple' (none); info-bottleneck layer 'mlp' | 'linear' | 'fourier' (fixed random features: no parameter gradient),
        # added ('add') or not ('none': the layer then has no gradient at all)
        xmode = model.exchange_mode
        has_ib = model.ib_addition_mode.lower() == "add"
        ib_attn = model.ib_addition_mode.lower() == "attention"   # x_i += cross_attn_ib_i(x_i, ib rows) (models/temporal.py:117-118)
        ib_mode = eng.ib_mode

        def Gv(name, n=None):  # flat fp32 gradient of a vector parameter (optionally fused over n elements)
            return eng.grad_vec(name, n)

        def G2(name, rows=None):
            return eng.grad_mat(name, rows)

        # ================================================================ forward
        hid: Dict[str, torch.Tensor] = {}
        mods: Dict[str, torch.Tensor] = {}
        if self.adaln:
            prefixes = []
            for l in range(NL):
                pre = f"blocks.{l}."
                for i in range(F):
                    prefixes += [(f"{pre}ln.exp.{i}.0.", E), (f"{pre}ln.exp.{i}.2.", E)]
                if (F > 1 and xmode == "sea") or xmode in ("addition", "pool"):   # ('addition' / 'pool' exchange a single field with itself too)
                    for i in range(F):
                        prefixes.append((f"{pre}ln_cross.{i}.", D))
            for i in range(F):
                prefixes.append((f"ln.{i}.", Eo))
            silu_groups, gemm_groups = [], []
            for pre, d in prefixes:
                hid[pre], mods[pre] = buf(M, 2 * d), buf(M, 2 * d)
                silu_groups.append((P.f32_vec(pre + "cond_mlp.0.weight", 2 * d), P.f32_vec(pre + "cond_mlp.0.bias"), hid[pre]))
                gemm_groups.append(dict(A=hid[pre], W=P.act(pre + "cond_mlp.2.weight"), bias=P.f32_vec(pre + "cond_mlp.2.bias"), Cact=mods[pre]))
            for s in range(0, len(silu_groups), N.MAX_SILU_GROUPS):
                chunk = silu_groups[s:s + N.MAX_SILU_GROUPS]
                arr = (N.SeaSiluGroup * len(chunk))()
                for g, (w1, b1, hb) in zip(arr, chunk):
                    g.w1, g.b1, g.Hid, g.K2, g.ld = w1.data_ptr(), b1.data_ptr(), hb.data_ptr(), hb.shape[1], hb.stride(0)
                rec = _Rec(L.sea_silu_outer, [arr, len(chunk), None, M, self.code], "adaln.silu", arr)
                self._c_patches.append((rec.args, 2))
                self._cur.append(rec)
            self._gemm(gemm_groups, "adaln.cond_gemm")

        def npar(pre):
            if self.adaln:
                return dict(mod=mods[pre], gamma=P.f32_vec(pre + "weight"), beta=P.f32_vec(pre + "bias"))
            return dict(gamma=P.f32_vec(pre + "weight"))

        def stats():
            return buf(M, dtype=f32), buf(M, dtype=f32)

        thr = self.drop_thr
        Sv: List[dict] = []  # saved tensors per layer
        x_prev: Optional[List[torch.Tensor]] = None
        others_of = [[j for j in range(F) if j != i] for i in range(F)]
        for l in range(NL):
            pre = f"blocks.{l}."
            sv: dict = {}
            Sv.append(sv)
            first = l == 0
            sv["xr"] = [buf(M, E, dtype=f32) for _ in range(F)]
            sv["x5"] = [buf(M, Eo, dtype=f32) for _ in range(F)]
            sv["x_in"] = x_prev
            if not model.add_info_after_cross and (has_ib or ib_attn or concat):
                # the info-bottleneck add precedes everything and must not touch the caller's tensor: x_in := copy + ib
                xin = [buf(M, E, dtype=f32) for _ in range(F)]
                for i in range(F):
                    if first:
                        rec = _Rec(L.sea_convert_f32_to_act, [None, FE, xin[i].data_ptr(), E, M, Eo, N.SEA_F32], "x.copy")
                        self._x_patches.append((rec.args, 0, i * Eo * 4))
                    else:
                        rec = _Rec(L.sea_convert_f32_to_act, [x_prev[i].data_ptr(), Eo, xin[i].data_ptr(), E, M, Eo, N.SEA_F32], "x.copy")
                    self._cur.append(rec)
                if concat:
                    # x_in := [x | ib rows] (models/temporal.py:115-116): the info-bottleneck columns are reset, then the usual add runs on them
                    if self._zero_ib is None:
                        self._zero_ib = buf(M, self.ib_dim, dtype=f32, zero=True)
                    for i in range(F):
                        self._cur.append(_Rec(L.sea_convert_f32_to_act, [self._zero_ib.data_ptr(), self.ib_dim, xin[i][:, Eo:].data_ptr(), E, M, self.ib_dim, N.SEA_F32],
                                              "ib.concat.zero"))
                    sv["ib_drop"] = (thr, self._streams(F)) if thr else None
                    self._ib(pre, [t[:, Eo:] for t in xin], drop=sv["ib_drop"])
                elif ib_attn:
                    self._ib_attn_fwd(pre, xin, sv)
                else:
                    sv["ib_drop"] = (thr, self._streams(F)) if thr else None
                    self._ib(pre, xin, drop=sv["ib_drop"])
                sv["x_in"] = xin
                first = False
            x_in = sv["x_in"]
            # ---- self attention
            sv["st0"] = [stats() for _ in range(F)]
            sv["n0"] = [buf(M, E) for _ in range(F)]
            groups = []
            for i in range(F):
                g = dict(Yact=sv["n0"][i], mean=sv["st0"][i][0], rstd=sv["st0"][i][1], **npar(f"{pre}ln.exp.{i}.0."))
                g.update(dict(X=sv["xr"][i], ldx=FE, X_is_x=i * Eo * 4) if first else dict(X=x_in[i]))
                groups.append(g)
            self._norm(groups, E, "self.adaln0")
            sv["Q"] = [buf(B, H, T, hd_s) for _ in range(F)]
            sv["K"] = [buf(B, H, cap, hd_s, zero=True) for _ in range(F)]
            sv["V"] = [buf(B, H, cap, hd_s, zero=True) for _ in range(F)]
            sv["Vt"] = [buf(B, H, hd_s, cap, zero=True) for _ in range(F)]
            sv["att"] = [buf(M, E) for _ in range(F)]
            sv["LSE"] = [buf(B, H, T, dtype=f32) for _ in range(F)]
            self._qkv([dict(A=sv["n0"][i], W=P.act(f"{pre}attn.self.{i}.q.weight", 3 * E), bias=P.f32_vec(f"{pre}attn.self.{i}.q.bias", 3 * E),
                            col0=0, Q=sv["Q"][i], K=sv["K"][i], Vt=sv["Vt"][i], V=sv["V"][i]) for i in range(F)], rope_s, hd_s, "self.qkv_rope")
            sv["self_drop"] = (thr, self._streams(F)) if thr else None
            self._attn([dict(Q=sv["Q"][i], K=sv["K"][i], Vt=sv["Vt"][i], O=sv["att"][i], LSE=sv["LSE"][i]) for i in range(F)], hd_s, E, "self.attention",
                       drop=sv["self_drop"])
            sv["xa1"] = [buf(M, E) for _ in range(F)]
            groups = []
            for i in range(F):
                g = dict(A=sv["att"][i], W=P.act(f"{pre}attn.self.{i}.projection.weight"), C32=sv["xr"][i], Cact=sv["xa1"][i])
                g.update(dict(R=sv["xr"][i], ldr=FE, R_is_x=i * Eo * 4) if first else dict(R=x_in[i]))
                groups.append(g)
            self._gemm(groups, "self.out_proj")
            # ---- state exchange
            if xmode == "addition":
                # Jacobi: every field is read at its pre-exchange value; s = sum_j n_j is the same for every field (models/temporal.py:297-301)
                sv["dn"] = [buf(M, D, dtype=f32) for _ in range(F)]
                sv["nd"] = buf(F, M, D)
                sv["stc"] = [stats() for _ in range(F)]
                sv["s_pre"], sv["sg"] = buf(M, D), buf(M, D)
                if _switches.plan("norm", "1") != "0" and D <= 256 and D % 16 == 0:
                    self._gemm_norm([dict(A=sv["xa1"][j], W=P.act(f"{pre}cross_down.{j}.weight"), bias=P.f32_vec(f"{pre}cross_down.{j}.bias"), C32=sv["dn"][j],
                                          Yact=sv["nd"][j], mean=sv["stc"][j][0], rstd=sv["stc"][j][1], **npar(f"{pre}ln_cross.{j}.")) for j in range(F)], "add.down_norm")
                else:
                    self._gemm([dict(A=sv["xa1"][j], W=P.act(f"{pre}cross_down.{j}.weight"), bias=P.f32_vec(f"{pre}cross_down.{j}.bias"), C32=sv["dn"][j])
                                for j in range(F)], "add.down")
                    self._norm([dict(X=sv["dn"][j], Yact=sv["nd"][j], mean=sv["stc"][j][0], rstd=sv["stc"][j][1], **npar(f"{pre}ln_cross.{j}.")) for j in range(F)],
                               D, "add.norm")
                self._gemm([dict(A=sv["nd"][0], W=eng.eye(D), n_seg=F, a_seg_stride=M * D, Cact=sv["sg"], Z=sv["s_pre"], act=1)], "add.sum_gelu")
                self._gemm([dict(A=sv["sg"], W=P.act(f"{pre}cross_up.{i}.weight"), bias=P.f32_vec(f"{pre}cross_up.{i}.bias"), R=sv["xr"][i], C32=sv["xr"][i])
                            for i in range(F)], "add.up")
            if xmode == "pool":
                # 'pool' exchange (models/temporal.py:255-283, pool_update_method 'mlp'; Jacobi; runs for F = 1 too): n_j = ln_cross_j(cross_down_j(x_j)) + pe,
                # pool = W2 gelu(W0 cat_j n_j + b0) + b2, a_i = cross_attn_i(n_i, pool), x_i += cross_up_i(gelu(n_i + a_i)).  `big` = [n_0 .. n_{F-1} | a_0 .. a_{F-1}]
                # as in the inference plan (engine.py); everything a gradient needs is kept.  (pool_token / ln_pool never reach the output: dead parameters.)
                FD = F * D
                sv["dn"] = [buf(M, D, dtype=f32) for _ in range(F)]
                sv["stc"] = [stats() for _ in range(F)]
                nrm = [buf(M, D) for _ in range(F)]
                big = sv["big"] = buf(M, 2 * FD)
                pe_t = buf(M, D, dtype=f32)
                pe_t.copy_(blk_pe(eng, l)[:T].repeat(B, 1))
                if _switches.plan("norm", "1") != "0" and D <= 256 and D % 16 == 0:
                    self._gemm_norm([dict(A=sv["xa1"][j], W=P.act(f"{pre}cross_down.{j}.weight"), bias=P.f32_vec(f"{pre}cross_down.{j}.bias"), C32=sv["dn"][j],
                                          Yact=nrm[j], mean=sv["stc"][j][0], rstd=sv["stc"][j][1], **npar(f"{pre}ln_cross.{j}.")) for j in range(F)], "pool.down_norm")
                else:
                    self._gemm([dict(A=sv["xa1"][j], W=P.act(f"{pre}cross_down.{j}.weight"), bias=P.f32_vec(f"{pre}cross_down.{j}.bias"), C32=sv["dn"][j])
                                for j in range(F)], "pool.down")
                    self._norm([dict(X=sv["dn"][j], Yact=nrm[j], mean=sv["stc"][j][0], rstd=sv["stc"][j][1], **npar(f"{pre}ln_cross.{j}.")) for j in range(F)],
                               D, "pool.norm")
                # (PositionalEncoding ends in nn.Dropout, models/base_blocks.py:370-372: the position-encoded rows themselves are dropped — mode 3 of the epilogue)
                sv["pe_drop"] = self._streams(F) if thr else None
                self._gemm([dict(A=nrm[j], W=eng.eye(D), R=pe_t, Cact=big[:, j * D:(j + 1) * D], drop=((thr, sv["pe_drop"] + j, 3) if thr else None))
                            for j in range(F)], "pool.pe_add")
                sv["hp_pre"], sv["hp"], sv["pool"] = buf(M, 2 * D), buf(M, 2 * D), buf(M, D)
                self._gemm([dict(A=big[:, :FD], W=P.act(f"{pre}pool_update.0.weight"), bias=P.f32_vec(f"{pre}pool_update.0.bias"), Cact=sv["hp"], Z=sv["hp_pre"],
                                 act=1)], "pool.update0")
                self._gemm([dict(A=sv["hp"], W=P.act(f"{pre}pool_update.2.weight"), bias=P.f32_vec(f"{pre}pool_update.2.bias"), Cact=sv["pool"])], "pool.update2")
                sv["pq"] = [dict(Q=buf(B, H, T, hd_c), K=buf(B, H, cap, hd_c, zero=True), V=buf(B, H, cap, hd_c, zero=True), Vt=buf(B, H, hd_c, cap, zero=True),
                                 O=buf(M, D), LSE=buf(B, H, T, dtype=f32)) for _ in range(F)]
                qg = []
                for i in range(F):
                    ca, pr = f"{pre}cross_attn.{i}.", sv["pq"][i]
                    qg.append(dict(A=big[:, i * D:(i + 1) * D], W=P.act(ca + "q.weight"), bias=P.f32_vec(ca + "q.bias"), col0=0, Q=pr["Q"]))
                    qg.append(dict(A=sv["pool"], W=P.act(ca + "k.weight", 2 * D), bias=P.f32_vec(ca + "k.bias", 2 * D), col0=D, K=pr["K"], Vt=pr["Vt"], V=pr["V"]))
                self._qkv(qg, rope_c, hd_c, "pool.qkv_rope")
                sv["pool_drop"] = (thr, self._streams(F)) if thr else None
                self._attn([dict(Q=pr["Q"], K=pr["K"], Vt=pr["Vt"], O=pr["O"], LSE=pr["LSE"]) for pr in sv["pq"]], hd_c, D, "pool.attention", drop=sv["pool_drop"])
                self._gemm([dict(A=sv["pq"][i]["O"], W=P.act(f"{pre}cross_attn.{i}.projection.weight"), Cact=big[:, FD + i * D:FD + (i + 1) * D]) for i in range(F)],
                           "pool.proj")
                sv["ps_pre"], sv["psg"] = [buf(M, D) for _ in range(F)], [buf(M, D) for _ in range(F)]
                self._gemm([dict(A=big[:, i * D:(i + 1) * D], W=eng.eye(D), n_seg=2, a_seg_stride=FD, Cact=sv["psg"][i], Z=sv["ps_pre"][i], act=1) for i in range(F)],
                           "pool.sum_gelu")
                self._gemm([dict(A=sv["psg"][i], W=P.act(f"{pre}cross_up.{i}.weight"), bias=P.f32_vec(f"{pre}cross_up.{i}.bias"), R=sv["xr"][i], C32=sv["xr"][i])
                            for i in range(F)], "pool.up")
            if F > 1 and xmode == "sea":
                sv["dn_old"] = [buf(M, D, dtype=f32) for _ in range(F)]
                sv["nd_old"] = [buf(M, D) for _ in range(F)]
                sv["stc_old"] = [stats() for _ in range(F)]
                sv["dn_new"] = [buf(M, D, dtype=f32) for _ in range(F)]
                sv["nd_new"] = [buf(M, D) for _ in range(F)]
                sv["stc_new"] = [stats() for _ in range(F)]
                sv["xa2"] = [buf(M, E) for _ in range(F)]
                # cross_down + ln_cross in one launch (sea_gemm_rownorm); the pre-normalisation rows and the statistics are kept for the backward
                fuse_dn = _switches.plan("norm", "1") != "0" and D <= 256 and D % 16 == 0
                if fuse_dn:
                    self._gemm_norm([dict(A=sv["xa1"][j], W=P.act(f"{pre}cross_down.{j}.weight"), bias=P.f32_vec(f"{pre}cross_down.{j}.bias"), C32=sv["dn_old"][j],
                                          Yact=sv["nd_old"][j], mean=sv["stc_old"][j][0], rstd=sv["stc_old"][j][1], **npar(f"{pre}ln_cross.{j}."))
                                     for j in range(F)], "cross.down_norm_old")
                else:
                    self._gemm([dict(A=sv["xa1"][j], W=P.act(f"{pre}cross_down.{j}.weight"), bias=P.f32_vec(f"{pre}cross_down.{j}.bias"), C32=sv["dn_old"][j])
                                for j in range(F)], "cross.down_old")
                    self._norm([dict(X=sv["dn_old"][j], Yact=sv["nd_old"][j], mean=sv["stc_old"][j][0], rstd=sv["stc_old"][j][1], **npar(f"{pre}ln_cross.{j}."))
                                for j in range(F)], D, "cross.norm_old")
                sv["pair"] = {}
                sv["g"] = [buf(F - 1, M, D) for _ in range(F)]
                for i in range(F):
                    qkv_groups, probs, proj_groups = [], [], []
                    for s, j in enumerate(others_of[i]):
                        src = sv["nd_new"][j] if j < i else sv["nd_old"][j]
                        ca = f"{pre}cross_attn.{i}.{j}."
                        pr = dict(Q=buf(B, H, T, hd_c), K=buf(B, H, cap, hd_c, zero=True), V=buf(B, H, cap, hd_c, zero=True),
                                  Vt=buf(B, H, hd_c, cap, zero=True), O=buf(M, D), LSE=buf(B, H, T, dtype=f32), a=buf(M, D), src=src)
                        sv["pair"][(i, j)] = pr
                        qkv_groups.append(dict(A=sv["nd_old"][i], W=P.act(ca + "q.weight"), bias=P.f32_vec(ca + "q.bias"), col0=0, Q=pr["Q"]))
                        qkv_groups.append(dict(A=src, W=P.act(ca + "k.weight", 2 * D), bias=P.f32_vec(ca + "k.bias", 2 * D), col0=D,
                                               K=pr["K"], Vt=pr["Vt"], V=pr["V"]))
                        probs.append(dict(Q=pr["Q"], K=pr["K"], Vt=pr["Vt"], O=pr["O"], LSE=pr["LSE"]))
                        proj_groups.append(dict(A=pr["O"], W=P.act(ca + "projection.weight"), Cact=sv["g"][i][s], Z=pr["a"], act=1))
                    self._qkv(qkv_groups, rope_c, hd_c, f"cross{i}.qkv_rope")
                    sv[("cross_drop", i)] = (thr, self._streams(F - 1)) if thr else None
                    self._attn(probs, hd_c, D, f"cross{i}.attention", drop=sv[("cross_drop", i)])
                    self._gemm(proj_groups, f"cross{i}.proj_gelu")
                    self._gemm([dict(A=sv["g"][i][0], W=P.act(f"{pre}cross_up.{i}.weight"), bias=P.f32_vec(f"{pre}cross_up.{i}.bias"),
                                     bias_scale=float(F - 1), n_seg=F - 1, a_seg_stride=M * D, R=sv["xr"][i], C32=sv["xr"][i],
                                     Cact=(sv["xa2"][i] if i < F - 1 else None))], f"cross{i}.up_sum")
                    if i < F - 1 and fuse_dn:
                        self._gemm_norm([dict(A=sv["xa2"][i], W=P.act(f"{pre}cross_down.{i}.weight"), bias=P.f32_vec(f"{pre}cross_down.{i}.bias"), C32=sv["dn_new"][i],
                                              Yact=sv["nd_new"][i], mean=sv["stc_new"][i][0], rstd=sv["stc_new"][i][1], **npar(f"{pre}ln_cross.{i}."))],
                                        f"cross{i}.down_norm_new")
                    elif i < F - 1:
                        self._gemm([dict(A=sv["xa2"][i], W=P.act(f"{pre}cross_down.{i}.weight"), bias=P.f32_vec(f"{pre}cross_down.{i}.bias"),
                                         C32=sv["dn_new"][i])], f"cross{i}.down_new")
                        self._norm([dict(X=sv["dn_new"][i], Yact=sv["nd_new"][i], mean=sv["stc_new"][i][0], rstd=sv["stc_new"][i][1],
                                         **npar(f"{pre}ln_cross.{i}."))], D, f"cross{i}.norm_new")
            if model.add_info_after_cross and has_ib:
                sv["ib_drop"] = (thr, self._streams(F)) if thr else None
                self._ib(pre, sv["xr"], drop=sv["ib_drop"])
            if model.add_info_after_cross and ib_attn:
                self._ib_attn_fwd(pre, sv["xr"], sv)
            # ---- MLP + proj
            sv["st2"] = [stats() for _ in range(F)]
            sv["n2"] = [buf(M, E) for _ in range(F)]
            spad = 64 if S % 1024 == 0 else 0   # hidden rows at a power-of-two stride crowd a few memory channels (engine.Plan._build): 128 B of padding per row
            sv["h"] = [buf(M, S + spad)[:, :S] for _ in range(F)]
            sv["sth"] = [stats() for _ in range(F)]
            sv["hg"] = [buf(M, S + spad)[:, :S] for _ in range(F)]
            sv["xa4"] = [buf(M, E) for _ in range(F)]
            self._norm([dict(X=sv["xr"][i], Yact=sv["n2"][i], mean=sv["st2"][i][0], rstd=sv["st2"][i][1], **npar(f"{pre}ln.exp.{i}.2.")) for i in range(F)],
                       E, "mlp.adaln2")
            self._gemm([dict(A=sv["n2"][i], W=P.act(f"{pre}mlp.{i}.layers.0.weight"), bias=P.f32_vec(f"{pre}mlp.{i}.layers.0.bias"), Cact=sv["h"][i])
                        for i in range(F)], "mlp.fc1")
            self._norm([dict(X=sv["h"][i], gamma=P.f32_vec(f"{pre}mlp.{i}.layers.1.weight"), beta=P.f32_vec(f"{pre}mlp.{i}.layers.1.bias"),
                             Yact=sv["hg"][i], mean=sv["sth"][i][0], rstd=sv["sth"][i][1]) for i in range(F)], S, "mlp.ln_gelu", x_is_act=True, gelu=True)
            sv["mlp_drop"] = self._streams(F) if thr else None
            self._gemm([dict(A=sv["hg"][i], W=P.act(f"{pre}mlp.{i}.layers.3.weight"), bias=P.f32_vec(f"{pre}mlp.{i}.layers.3.bias"), R=sv["xr"][i],
                             Cact=sv["xa4"][i], drop=((thr, sv["mlp_drop"] + i, 1) if thr else None)) for i in range(F)], "mlp.fc2")
            self._gemm([dict(A=sv["xa4"][i], W=P.act(f"{pre}proj.{i}.weight"), bias=P.f32_vec(f"{pre}proj.{i}.bias"), C32=sv["x5"][i])
                        for i in range(F)], "proj")
            x_prev = sv["x5"]
        stf = [stats() for _ in range(F)]
        self._norm([dict(X=x_prev[i], Y32=x_prev[i], ldy32=FE, Y_is_out=i * Eo * 4, mean=stf[i][0], rstd=stf[i][1], **npar(f"ln.{i}."))
                    for i in range(F)], Eo, "final.norm")

        # ================================================================ backward
        self._cur = self.bwd
        adaln = self.adaln
        dmods: List[Tuple[str, torch.Tensor]] = []  # (prefix, dmod buffer) per USE of an AdaLN
        self._phase_marks: List[Tuple[int, int]] = []   # (number of backward records, gradient phase final after them): engine.grad_phase
        # data-parallel runs: the condition MLPs' backward is emitted per phase (the modules whose modulation gradients are complete), so that the slice of
        # a phase can be all-reduced while the later phases run; a single process keeps the one grouped launch set at the end
        dp = self.dp
        split_cond = adaln and dp

        def cond_backward(tag=""):
            """cond_mlp backward of every AdaLN use collected in `dmods` so far (shared parameters accumulate through the weight-gradient kernel)."""
            if not dmods:
                return
            dh = [buf(M, dm.shape[1]) for _, dm in dmods]
            self._wgrad([dict(dY=dm, X=hid[pre_], dW=G2(pre_ + "cond_mlp.2.weight"), db=Gv(pre_ + "cond_mlp.2.bias")) for pre_, dm in dmods], "bwd.adaln.cond.wgrad" + tag)
            self._gemm([dict(A=dm, W=P.actT(pre_ + "cond_mlp.2.weight"), Cact=dh[k]) for k, (pre_, dm) in enumerate(dmods)], "bwd.adaln.cond.dgrad" + tag)
            self._silu_bwd([dict(dHid=dh[k], w1=P.f32_vec(pre_ + "cond_mlp.0.weight", dm.shape[1]), b1=P.f32_vec(pre_ + "cond_mlp.0.bias"),
                                 dw1=Gv(pre_ + "cond_mlp.0.weight", dm.shape[1]), db1=Gv(pre_ + "cond_mlp.0.bias")) for k, (pre_, dm) in enumerate(dmods)],
                           "bwd.adaln.silu" + tag)
            dmods.clear()

        def bpar(pre, d):
            """norm-backward parameter/gradient pointers (+ a fresh dmod buffer for this use)"""
            out = dict(gamma=P.f32_vec(pre + "weight"), dgamma=Gv(pre + "weight"))
            if adaln:
                dm = buf(M, 2 * d)
                dmods.append((pre, dm))
                out.update(mod=mods[pre], beta=P.f32_vec(pre + "bias"), dbeta=Gv(pre + "bias"), dmod=dm)
            return out

        dx = [buf(M, E, dtype=f32) for _ in range(F)]      # gradient of the fp32 residual stream
        ga = [buf(M, E) for _ in range(F)]                 # act copy of the residual gradient entering a GEMM
        gb = [buf(M, E) for _ in range(F)]
        dS_ = [buf(M, S + (64 if S % 1024 == 0 else 0))[:, :S] for _ in range(F)]
        dE_ = [buf(M, E) for _ in range(F)]
        dqkv = [buf(M, 3 * E) for _ in range(F)]
        delta_s = [buf(B, H, T, dtype=f32) for _ in range(F)]
        # the gradient of a block's OUTPUT rows (Eo wide: the proj output) lives in the first Eo columns of the E-wide residual-gradient buffers
        dxo = dx if Eo == E else [t[:, :Eo] for t in dx]
        gao = ga if Eo == E else [t[:, :Eo] for t in ga]
        # final norm
        self._norm_bwd([dict(dY=dxo[i], lddy=FE, dY_is_dout=i * Eo * 4, X=x_prev[i], mean=stf[i][0], rstd=stf[i][1], dX32=dxo[i], dXact=gao[i],
                             **bpar(f"ln.{i}.", Eo)) for i in range(F)], Eo, "bwd.final_norm", False, False, False, False)
        for l in reversed(range(NL)):
            pre = f"blocks.{l}."
            sv = Sv[l]
            first = l == 0 and (model.add_info_after_cross or not (has_ib or ib_attn or concat))
            # ---- proj:  x5 = Wp xa4 + bp                       (gao = d x5 in act dtype)
            self._wgrad([dict(dY=gao[i], X=sv["xa4"][i], dW=G2(f"{pre}proj.{i}.weight"), db=Gv(f"{pre}proj.{i}.bias")) for i in range(F)], "bwd.proj.wgrad")
            # (MLP-output dropout: the residual path C32 stays whole, the copy feeding fc2's backward is masked)
            self._gemm([dict(A=gao[i], W=P.actT(f"{pre}proj.{i}.weight"), C32=dx[i], Cact=gb[i],
                             drop=((thr, sv["mlp_drop"] + i, 2) if thr else None)) for i in range(F)], "bwd.proj.dgrad")
            # ---- fc2:   x4 = x3 + hg W2^T + b2                 (gb = d x4)
            self._wgrad([dict(dY=gb[i], X=sv["hg"][i], dW=G2(f"{pre}mlp.{i}.layers.3.weight"), db=Gv(f"{pre}mlp.{i}.layers.3.bias")) for i in range(F)],
                        "bwd.fc2.wgrad", sole=True)
            self._gemm([dict(A=gb[i], W=P.actT(f"{pre}mlp.{i}.layers.3.weight"), Cact=dS_[i]) for i in range(F)], "bwd.fc2.dgrad")
            # ---- LayerNorm + GELU of the MLP (in place: d hg -> d h)
            self._norm_bwd([dict(dY=dS_[i], X=sv["h"][i], gamma=P.f32_vec(f"{pre}mlp.{i}.layers.1.weight"), beta=P.f32_vec(f"{pre}mlp.{i}.layers.1.bias"),
                                 mean=sv["sth"][i][0], rstd=sv["sth"][i][1], dXact=dS_[i], dgamma=Gv(f"{pre}mlp.{i}.layers.1.weight"),
                                 dbeta=Gv(f"{pre}mlp.{i}.layers.1.bias")) for i in range(F)], S, "bwd.mlp.ln_gelu", True, True, True, False)
            # ---- fc1
            self._wgrad([dict(dY=dS_[i], X=sv["n2"][i], dW=G2(f"{pre}mlp.{i}.layers.0.weight"), db=Gv(f"{pre}mlp.{i}.layers.0.bias")) for i in range(F)],
                        "bwd.fc1.wgrad", sole=True)
            self._phase_marks.append((len(self.bwd), (NL - 1 - l) * 3))            # the field MLPs + proj of this layer are final
            self._gemm([dict(A=dS_[i], W=P.actT(f"{pre}mlp.{i}.layers.0.weight"), Cact=dE_[i]) for i in range(F)], "bwd.fc1.dgrad")
            # ---- AdaLN_2: accumulates the norm branch onto the residual gradient
            self._norm_bwd([dict(dY=dE_[i], X=sv["xr"][i], mean=sv["st2"][i][0], rstd=sv["st2"][i][1], dX32=dx[i], dXact=ga[i],
                                 **bpar(f"{pre}ln.exp.{i}.2.", E)) for i in range(F)], E, "bwd.mlp.adaln2", True, False, False, True)
            if model.add_info_after_cross and has_ib:
                self._ib_bwd(pre, dx, drop=sv["ib_drop"])
            if model.add_info_after_cross and ib_attn:
                self._ib_attn_bwd(pre, sv, dx, ga)
            if xmode == "addition":
                # x2_i = x1_i + Wu_i g + bu_i, g = gelu(s), s = sum_j n_j, n_j = ln_cross_j(Wd_j x1_j + bd_j): ga[i] = act copy of d x2_i
                self._wgrad([dict(dY=ga[i], X=sv["sg"], dW=G2(f"{pre}cross_up.{i}.weight"), db=Gv(f"{pre}cross_up.{i}.bias")) for i in range(F)], "bwd.add.up.wgrad")
                dgp = buf(F, M, D)
                self._gemm([dict(A=ga[i], W=P.actT(f"{pre}cross_up.{i}.weight"), Cact=dgp[i]) for i in range(F)], "bwd.add.up.dgrad")
                ds = buf(M, D)      # d s = (sum_i d g_i) * gelu'(s): the sum over the fields as operand segments of an identity GEMM, the derivative as its epilogue
                self._gemm([dict(A=dgp[0], W=eng.eye(D), n_seg=F, a_seg_stride=M * D, Z=sv["s_pre"], act=2, Cact=ds)], "bwd.add.sum_gelu")
                ddn = [buf(M, D) for _ in range(F)]
                self._norm_bwd([dict(dY=ds, X=sv["dn"][j], mean=sv["stc"][j][0], rstd=sv["stc"][j][1], dXact=ddn[j], **bpar(f"{pre}ln_cross.{j}.", D)) for j in range(F)],
                               D, "bwd.add.norm", True, False, False, False)
                self._wgrad([dict(dY=ddn[j], X=sv["xa1"][j], dW=G2(f"{pre}cross_down.{j}.weight"), db=Gv(f"{pre}cross_down.{j}.bias")) for j in range(F)], "bwd.add.down.wgrad")
                self._gemm([dict(A=ddn[j], W=P.actT(f"{pre}cross_down.{j}.weight"), R=dx[j], C32=dx[j], Cact=ga[j]) for j in range(F)], "bwd.add.down.dgrad")
            if xmode == "pool":
                # x2_i = x1_i + Wu_i gelu(s_i) + bu_i, s_i = n_i + a_i: ga[i] = act copy of d x2_i; dnd[j] collects d n_j (from s_j, from q_j and through the pool)
                FD, big = F * D, sv["big"]
                self._wgrad([dict(dY=ga[i], X=sv["psg"][i], dW=G2(f"{pre}cross_up.{i}.weight"), db=Gv(f"{pre}cross_up.{i}.bias")) for i in range(F)], "bwd.pool.up.wgrad")
                ds = [buf(M, D) for _ in range(F)]
                dnd = [buf(M, D, dtype=f32) for _ in range(F)]
                self._gemm([dict(A=ga[i], W=P.actT(f"{pre}cross_up.{i}.weight"), Z=sv["ps_pre"][i], act=2, Cact=ds[i], C32=dnd[i]) for i in range(F)], "bwd.pool.up.dgrad")
                self._wgrad([dict(dY=ds[i], X=sv["pq"][i]["O"], dW=G2(f"{pre}cross_attn.{i}.projection.weight")) for i in range(F)], "bwd.pool.proj.wgrad")
                datt = [buf(M, D) for _ in range(F)]
                self._gemm([dict(A=ds[i], W=P.actT(f"{pre}cross_attn.{i}.projection.weight"), Cact=datt[i]) for i in range(F)], "bwd.pool.proj.dgrad")
                dqp, dkvp = [buf(M, D) for _ in range(F)], [buf(M, 2 * D) for _ in range(F)]
                delta_p = [buf(B, H, T, dtype=f32) for _ in range(F)]
                self._attn_bwd([dict(Q=sv["pq"][i]["Q"], K=sv["pq"][i]["K"], V=sv["pq"][i]["V"], O=sv["pq"][i]["O"], dO=datt[i], LSE=sv["pq"][i]["LSE"],
                                     delta=delta_p[i], dQ=dqp[i], dK=dkvp[i][:, :D], dV=dkvp[i][:, D:]) for i in range(F)], hd_c, rope_c, "bwd.pool.attention",
                               drop=sv["pool_drop"])
                wg = []
                for i in range(F):
                    ca = f"{pre}cross_attn.{i}."
                    wg.append(dict(dY=dqp[i], X=big[:, i * D:(i + 1) * D], dW=G2(ca + "q.weight"), db=Gv(ca + "q.bias")))
                    wg.append(dict(dY=dkvp[i], X=sv["pool"], dW=G2(ca + "k.weight", 2 * D), db=Gv(ca + "k.bias", 2 * D)))
                self._wgrad(wg, "bwd.pool.qkv.wgrad")
                self._gemm([dict(A=dqp[i], W=P.actT(f"{pre}cross_attn.{i}.q.weight"), R=dnd[i], C32=dnd[i]) for i in range(F)], "bwd.pool.q.dgrad")
                dpool32, dpool = buf(M, D, dtype=f32), buf(M, D)
                for i in range(F):   # sequential: every field's keys / values came from the same pool rows
                    g = dict(A=dkvp[i], W=P.actT(f"{pre}cross_attn.{i}.k.weight", 2 * D), C32=dpool32)
                    if i > 0:
                        g["R"] = dpool32
                    if i == F - 1:
                        g["Cact"] = dpool
                    self._gemm([g], f"bwd.pool.kv.dgrad{i}")
                self._wgrad([dict(dY=dpool, X=sv["hp"], dW=G2(f"{pre}pool_update.2.weight"), db=Gv(f"{pre}pool_update.2.bias"))], "bwd.pool.update2.wgrad")
                dhp = buf(M, 2 * D)
                self._gemm([dict(A=dpool, W=P.actT(f"{pre}pool_update.2.weight"), Z=sv["hp_pre"], act=2, Cact=dhp)], "bwd.pool.update2.dgrad")
                self._wgrad([dict(dY=dhp, X=big[:, :FD], dW=G2(f"{pre}pool_update.0.weight"), db=Gv(f"{pre}pool_update.0.bias"))], "bwd.pool.update0.wgrad")
                w0t = P.actT(f"{pre}pool_update.0.weight")          # [F D, 2 D]: rows j D .. (j+1) D map d hp onto d n_j
                # the last of the three contributions to d (dropped rows): with dropout the gradient of the rows BEFORE the mask is mask * (the sum)
                self._gemm([dict(A=dhp, W=w0t[j * D:(j + 1) * D], R=dnd[j], C32=dnd[j], drop=((thr, sv["pe_drop"] + j, 3) if thr else None))
                            for j in range(F)], "bwd.pool.update0.dgrad")
                ddn = [buf(M, D) for _ in range(F)]
                self._norm_bwd([dict(dY=dnd[j], X=sv["dn"][j], mean=sv["stc"][j][0], rstd=sv["stc"][j][1], dXact=ddn[j], **bpar(f"{pre}ln_cross.{j}.", D))
                                for j in range(F)], D, "bwd.pool.norm", False, False, False, False)
                self._wgrad([dict(dY=ddn[j], X=sv["xa1"][j], dW=G2(f"{pre}cross_down.{j}.weight"), db=Gv(f"{pre}cross_down.{j}.bias")) for j in range(F)],
                            "bwd.pool.down.wgrad")
                self._gemm([dict(A=ddn[j], W=P.actT(f"{pre}cross_down.{j}.weight"), R=dx[j], C32=dx[j], Cact=ga[j]) for j in range(F)], "bwd.pool.down.dgrad")
            # ---- state exchange (reverse Gauss-Seidel order); ga[i] = act copy of d x2_i when field i is reached
            if F > 1 and xmode == "sea":
                dnd_old = [buf(M, D, dtype=f32) for _ in range(F)]
                dnd_new = [buf(M, D, dtype=f32) for _ in range(F)]
                init_old, init_new = [False] * F, [False] * F
                # Weight gradients are leaves of the backward graph: nothing waits for them.  The exchange's are small matrices ([D or E] x [D or E]
                # outputs over M rows: 1-3 GFLOP each, 16-24 us per launch of two or three of them, most of it launch, pipeline fill and the atomic pass)
                # and there are 5 launches of them per field — they are collected in `xw` and run as ONE grouped launch behind the loop (cfg3: 17 launches
                # of ~19 us -> one; their dY operands therefore get a buffer per field instead of one reused by every field).
                xw: List[dict] = []
                ddn_f = [buf(M, D) for _ in range(F)]
                da_f = [[buf(M, D) for _ in range(F - 1)] for _ in range(F)]
                datt = [buf(M, D) for _ in range(F - 1)]
                dqc_f = [[buf(M, D) for _ in range(F - 1)] for _ in range(F)]
                dkvc_f = [[buf(M, 2 * D) for _ in range(F - 1)] for _ in range(F)]
                delta_c = [buf(B, H, T, dtype=f32) for _ in range(F - 1)]
                for i in reversed(range(F)):
                    ddn, da, dqc, dkvc = ddn_f[i], da_f[i], dqc_f[i], dkvc_f[i]
                    if i < F - 1:
                        # a. nd_new[i] was the k/v source of the later fields: back through ln_cross and cross_down onto d x2_i
                        assert init_new[i]
                        cp = f"{pre}ln_cross.{i}."
                        self._norm_bwd([dict(dY=dnd_new[i], X=sv["dn_new"][i], mean=sv["stc_new"][i][0], rstd=sv["stc_new"][i][1], dXact=ddn, **bpar(cp, D))],
                                       D, f"bwd.cross{i}.norm_new", False, False, False, False)
                        xw.append(dict(dY=ddn, X=sv["xa2"][i], dW=G2(f"{pre}cross_down.{i}.weight"), db=Gv(f"{pre}cross_down.{i}.bias")))
                        self._gemm([dict(A=ddn, W=P.actT(f"{pre}cross_down.{i}.weight"), R=dx[i], C32=dx[i], Cact=ga[i])], f"bwd.cross{i}.down_new.dgrad")
                    # b. cross_up (shared by the F-1 partners) with the GELU derivative fused: da_s = (d x2_i Wu) * gelu'(a_ij)
                    others = others_of[i]
                    xw += [dict(dY=ga[i], X=sv["g"][i][s], dW=G2(f"{pre}cross_up.{i}.weight"), db=Gv(f"{pre}cross_up.{i}.bias")) for s in range(F - 1)]   # ga[i] is next written in f.
                    self._gemm([dict(A=ga[i], W=P.actT(f"{pre}cross_up.{i}.weight"), Z=sv["pair"][(i, j)]["a"], act=2, Cact=da[s])
                                for s, j in enumerate(others)], f"bwd.cross{i}.up.dgrad")
                    # c. projection of each pair
                    xw += [dict(dY=da[s], X=sv["pair"][(i, j)]["O"], dW=G2(f"{pre}cross_attn.{i}.{j}.projection.weight")) for s, j in enumerate(others)]
                    self._gemm([dict(A=da[s], W=P.actT(f"{pre}cross_attn.{i}.{j}.projection.weight"), Cact=datt[s]) for s, j in enumerate(others)],
                               f"bwd.cross{i}.proj.dgrad")
                    # d. attention
                    self._attn_bwd([dict(Q=sv["pair"][(i, j)]["Q"], K=sv["pair"][(i, j)]["K"], V=sv["pair"][(i, j)]["V"], O=sv["pair"][(i, j)]["O"],
                                         dO=datt[s], LSE=sv["pair"][(i, j)]["LSE"], delta=delta_c[s], dQ=dqc[s], dK=dkvc[s][:, :D], dV=dkvc[s][:, D:])
                                    for s, j in enumerate(others)], hd_c, rope_c, f"bwd.cross{i}.attention", drop=sv[("cross_drop", i)])
                    # e. q / k,v projections
                    for s, j in enumerate(others):
                        ca = f"{pre}cross_attn.{i}.{j}."
                        xw.append(dict(dY=dqc[s], X=sv["nd_old"][i], dW=G2(ca + "q.weight"), db=Gv(ca + "q.bias")))
                        xw.append(dict(dY=dkvc[s], X=sv["pair"][(i, j)]["src"], dW=G2(ca + "k.weight", 2 * D), db=Gv(ca + "k.bias", 2 * D)))
                    for s, j in enumerate(others):
                        ca = f"{pre}cross_attn.{i}.{j}."
                        groups = []
                        gq = dict(A=dqc[s], W=P.actT(ca + "q.weight"), C32=dnd_old[i])
                        if init_old[i]:
                            gq["R"] = dnd_old[i]
                        init_old[i] = True
                        groups.append(gq)
                        tgt, flags = (dnd_new, init_new) if j < i else (dnd_old, init_old)
                        gk = dict(A=dkvc[s], W=P.actT(ca + "k.weight", 2 * D), C32=tgt[j])
                        if flags[j]:
                            gk["R"] = tgt[j]
                        flags[j] = True
                        groups.append(gk)
                        self._gemm(groups, f"bwd.cross{i}.qkv.dgrad{s}")  # sequential over s: both s accumulate into dnd_old[i]
                # f. the pre-exchange ln_cross / cross_down of every field
                for j in range(F):
                    assert init_old[j]
                ddo = [buf(M, D) for _ in range(F)]
                self._norm_bwd([dict(dY=dnd_old[j], X=sv["dn_old"][j], mean=sv["stc_old"][j][0], rstd=sv["stc_old"][j][1], dXact=ddo[j],
                                     **bpar(f"{pre}ln_cross.{j}.", D)) for j in range(F)], D, "bwd.cross.norm_old", False, False, False, False)
                xw += [dict(dY=ddo[j], X=sv["xa1"][j], dW=G2(f"{pre}cross_down.{j}.weight"), db=Gv(f"{pre}cross_down.{j}.bias")) for j in range(F)]
                self._wgrad(xw, "bwd.cross.wgrad")   # before the data gradient below overwrites ga
                self._gemm([dict(A=ddo[j], W=P.actT(f"{pre}cross_down.{j}.weight"), R=dx[j], C32=dx[j], Cact=ga[j]) for j in range(F)], "bwd.cross.down_old.dgrad")
            if split_cond:
                cond_backward(f".l{l}a")   # the modules used so far: (final norms,) the norm in front of the MLP, ln_cross
            if dp:
                self._phase_marks.append((len(self.bwd), (NL - 1 - l) * 3 + 1))    # exchange, ln_cross, norm 2, info-bottleneck step behind the exchange(, final norms)
            # ---- self attention: x1 = x0 + att Wo^T            (ga = d x1)
            self._wgrad([dict(dY=ga[i], X=sv["att"][i], dW=G2(f"{pre}attn.self.{i}.projection.weight")) for i in range(F)], "bwd.self.out_proj.wgrad")
            self._gemm([dict(A=ga[i], W=P.actT(f"{pre}attn.self.{i}.projection.weight"), Cact=dE_[i]) for i in range(F)], "bwd.self.out_proj.dgrad")
            self._attn_bwd([dict(Q=sv["Q"][i], K=sv["K"][i], V=sv["V"][i], O=sv["att"][i], dO=dE_[i], LSE=sv["LSE"][i], delta=delta_s[i],
                                 dQ=dqkv[i][:, :E], dK=dqkv[i][:, E:2 * E], dV=dqkv[i][:, 2 * E:]) for i in range(F)], hd_s, rope_s, "bwd.self.attention",
                           drop=sv["self_drop"])
            self._wgrad([dict(dY=dqkv[i], X=sv["n0"][i], dW=G2(f"{pre}attn.self.{i}.q.weight", 3 * E), db=Gv(f"{pre}attn.self.{i}.q.bias", 3 * E))
                         for i in range(F)], "bwd.self.qkv.wgrad")
            self._gemm([dict(A=dqkv[i], W=P.actT(f"{pre}attn.self.{i}.q.weight", 3 * E), Cact=dE_[i]) for i in range(F)], "bwd.self.qkv.dgrad")
            groups = []
            for i in range(F):
                g = dict(dY=dE_[i], mean=sv["st0"][i][0], rstd=sv["st0"][i][1], dX32=dx[i], dXact=ga[i], **bpar(f"{pre}ln.exp.{i}.0.", E))
                g.update(dict(X=dx[i], ldx=FE, X_is_x=i * E * 4) if first else dict(X=sv["x_in"][i]))
                groups.append(g)
            self._norm_bwd(groups, E, "bwd.self.adaln0", True, False, False, True)
            if not model.add_info_after_cross and has_ib:
                self._ib_bwd(pre, dx, drop=sv["ib_drop"])
            if not model.add_info_after_cross and ib_attn:
                self._ib_attn_bwd(pre, sv, dx, ga)
            if concat:   # the info-bottleneck columns of the widened rows; columns 0 .. Eo-1 of dx / ga are the gradient of the previous block's output
                self._ib_bwd(pre, [t[:, Eo:] for t in dx], drop=sv["ib_drop"])
            if split_cond:
                cond_backward(f".l{l}b")   # the norm in front of the self-attention
            if dp and l > 0:
                self._phase_marks.append((len(self.bwd), (NL - 1 - l) * 3 + 2))
        # ---- AdaLN condition MLPs: every USE contributes dmod; parameters are shared through the atomically accumulated gradients (data-parallel runs
        # have emitted them per phase above: nothing is left here)
        if adaln:
            cond_backward()
        self._cur = self.records
        self.saved = Sv

    # ------------------------------------------------------------------ replay
    def bind_dout(self, dout_ptr: int) -> None:
        for tgt, field, off in self._dout_patches:
            setattr(tgt, field, dout_ptr + off)

    def grad_buckets(self):
        """[(number of backward records after which the slice is final, flat start, flat end)] in backward order.  The flat buffers are laid out by gradient
        phase (engine.grad_phase): the field MLPs + proj of a layer (45 % of the parameters at cfg3) are one contiguous slice, complete right after that
        layer's `bwd.fc1.wgrad`; in a data-parallel run the exchange / norm-2 / ln_cross / final-norm slice (with its condition MLPs, whose backward is then
        emitted per phase) is complete before the self-attention backward starts.  A data-parallel step reduces each slice while the rest of the backward
        runs (sea_amd/parallel.py); what the last phase holds goes after the backward."""
        if self._buckets is None:
            P = self.eng.params
            size = lambda n: int(torch.Size(P.offsets[n][1]).numel())
            last_phase = max(P.phase_of.values()) if P.phase_of else 0
            out = []
            for idx, ph in self._phase_marks:
                names = [n for n, q in P.phase_of.items() if q == ph]
                if not names or ph == last_phase:
                    continue
                lo = min(P.offsets[n][0] for n in names)
                hi = max(P.offsets[n][0] + size(n) for n in names)
                inside = {n for n, (o, _) in P.offsets.items() if lo <= o < hi}
                assert inside == set(names) and hi <= P.n_live, ("gradient phase is not one contiguous live slice", ph)
                out.append((idx, lo, hi))
            self._buckets = out
        return self._buckets

    def set_grads_fresh(self, fresh: bool) -> None:
        """The gradient buffer holds zeros (nothing has accumulated since the last zero_grads): the launches that are the sole contribution to their
        tensors may store instead of adding."""
        for g in self.__dict__.get("_sole_wgrads", ()):
            g.overwrite = int(fresh)

    def run_backward(self, on_bucket=None) -> None:
        """Replay the backward launch list; `on_bucket(lo, hi)` is called as soon as grads[lo:hi] is final (grad_buckets)."""
        stream = N.stream_ptr()
        marks = {n: (lo, hi) for n, lo, hi in self.grad_buckets()} if on_bucket is not None else {}
        for k, r in enumerate(self.bwd):
            rc = r.fn(*r.args, stream)
            if rc != 0:
                N.check(rc, r.name)
            if k + 1 in marks:
                on_bucket(*marks[k + 1])
