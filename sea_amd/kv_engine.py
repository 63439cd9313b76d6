"""Fast exact KV-cache rollout for small models (sea_kv_rollout, sea_amd/csrc/kvstep.hip): the loop of utils/train_utils.py:202-209 with
seven launches per layer and step instead of the generic step plan's twenty-two.

What depends on the condition only — AdaLN's cond_mlp output of every module (models/base_blocks.py:337-344) and the info-bottleneck term
(models/temporal.py:103-114) — is evaluated for ALL steps by one batched pass (`CondPlan`: the same silu / grouped-GEMM / ib launches the
full-context plan uses, on n_steps * B rows) before the native step loop starts.

`supported(eng, B)` says whether a model is inside the kernels' limits; everything else keeps the generic step plan (`engine.rollout_kv`).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional

import torch

from . import _native as N
from . import _switches
from . import ptrcheck
from .engine import Plan, _round_up


class CondPlan(Plan):
    """Launch list that evaluates, for M = n_steps * B condition rows: mods[prefix] = cond_mlp output [M, 2d] (act dtype) of every AdaLN module and
    ibufs[l] = the info-bottleneck term [M, E] (fp32, accumulated into a zeroed buffer) of every layer."""

    def __init__(self, eng, M: int):
        self.mods: Dict[str, torch.Tensor] = {}
        self.ibufs: List[torch.Tensor] = []
        super().__init__(eng, M, 1, "full")

    def _build(self) -> None:
        m = self.eng.model
        self.mods = self._cond_mods(split=False)
        if m.ib_addition_mode.lower() == "add":
            for l in range(self.L):
                ibuf = self._buf(self.M, self.E, dtype=torch.float32, zero=True)
                self.ibufs.append(ibuf)
                self._ib(f"blocks.{l}.", [ibuf])


def cond_plan_for(eng, M: int) -> CondPlan:
    """The engine's condition plan for M = n_steps * B rows (one row count at a time: the modulation buffers are ~12 M x 2d elements)."""
    cache = eng.__dict__.setdefault("_cond_plans", {})
    cp = cache.get(M)
    if cp is None:
        cache.clear()
        cp = cache[M] = CondPlan(eng, M)
    return cp


def _chunks_ok(K: int, epc: int) -> bool:
    if K % epc:
        return False
    kc = K // epc
    return (2 <= kc <= 64 and kc & (kc - 1) == 0) or kc in (128, 256, 512)


def supported(eng, B: int) -> bool:
    """Is (model, batch) inside what sea_kv_rollout covers?  SEA_KV=fast=0 keeps the generic step plan (A/B measurements, parity tests)."""
    m = eng.model
    if _switches.kv("fast", "1") == "0":
        return False
    if m.exchange_mode not in ("sea", "simple") or m.src_len != 0 or m.ib_addition_mode.lower() in ("attention", "concat"):
        return False   # ('concat': rows of two widths; the generic step plan covers it)
    F, E, D, S, H = m.num_variables, m.embed_dim, m.down_dim, m.mlp_hidden, m.n_heads
    epc = 8 if eng.act_dtype == torch.bfloat16 else 4
    if not (1 <= F <= N.KV_MAX_FIELDS and 1 <= B <= 64 and E <= 512 and S <= 4096 and S % 4 == 0 and _chunks_ok(E, epc) and _chunks_ok(S, epc)):
        return False
    if E % H or E // H not in (8, 16, 32, 64):
        return False
    if m.exchange_mode == "sea" and F > 1:
        if D % H or D // H not in (8, 16, 32, 64) or D > 512 or not _chunks_ok(D, epc):
            return False
    return _round_up(m.max_len, 8) <= 8192


class KvFast:
    """Workspace + argument structs of sea_kv_rollout for one (engine, B)."""

    def __init__(self, eng, B: int):
        m = eng.model
        self.eng, self.B = eng, B
        F, E, D, S, H, L = m.num_variables, m.embed_dim, m.down_dim, m.mlp_hidden, m.n_heads, m.num_layers
        self.F, self.E, self.D, self.S, self.H, self.L = F, E, D, S, H, L
        self.cap = _round_up(m.max_len, 8)
        self.exchange = m.exchange_mode == "sea" and F > 1
        self.adaln = m.LN_type.lower() == "adaln"
        self.has_ib = m.ib_addition_mode.lower() == "add"
        dev, dt, f32 = eng.device, eng.act_dtype, torch.float32
        self._keep: List[torch.Tensor] = []

        def buf(*shape, dtype=f32, zero=False):
            t = (torch.zeros if zero else torch.empty)(*shape, device=dev, dtype=dtype)
            self._keep.append(t)
            return t

        P = eng.params
        hd_s, hd_c = E // H, (D // H if self.exchange else 0)
        G = self.G = N.SeaKvGlobal()
        G.F, G.E, G.D, G.S, G.H, G.B, G.L, G.cap = F, E, D, S, H, B, L, self.cap
        G.exchange, G.ib_after_cross = int(self.exchange), int(bool(m.add_info_after_cross))
        G.rope_self = eng.rope_self.data_ptr()
        G.rope_cross = eng.rope_cross.data_ptr() if self.exchange else None
        G.xl[0], G.xl[1] = buf(B, F, E).data_ptr(), buf(B, F, E).data_ptr()
        G.att_e, G.xr, G.xq, G.x3 = (buf(B, F, E).data_ptr() for _ in range(4))
        G.hbuf = buf(B, F, S).data_ptr()
        if self.exchange:
            npair = F * (F - 1)
            G.nd_old = buf(B, F, D).data_ptr()
            G.oc, G.qc = buf(npair, B, D).data_ptr(), buf(npair, B, D).data_ptr()
            G.ml = buf(npair, B, H, 2).data_ptr()
        # granule words: the tails' hand-off of the seven-launch form, or — one trajectory, one layer — the whole arena of the persistent form
        words = max(int(N.lib().sea_kv_arena_words(C.byref(G))), B * F * max(D, 1), 1)
        G.handoff, G.handoff_words = buf(words, dtype=torch.int64, zero=True).data_ptr(), words
        self.err = buf(1, dtype=torch.int32, zero=True)
        G.err = self.err.data_ptr()
        self.layers = (N.SeaKvLayer * L)()
        self._norms = []   # (SeaKvNorm, prefix): mod pointers are set per rollout
        for l in range(L):
            pre = f"blocks.{l}."
            Ly = self.layers[l]
            for i in range(F):
                f = Ly.f[i]
                self._fill_norm(f.ln0, f"{pre}ln.exp.{i}.0.")
                self._fill_norm(f.ln2, f"{pre}ln.exp.{i}.2.")
                f.Wqkv, f.bqkv = P.act(f"{pre}attn.self.{i}.q.weight", 3 * E).data_ptr(), P.f32_vec(f"{pre}attn.self.{i}.q.bias", 3 * E).data_ptr()
                f.Wo = P.act(f"{pre}attn.self.{i}.projection.weight").data_ptr()
                f.W1, f.b1 = P.act(f"{pre}mlp.{i}.layers.0.weight").data_ptr(), P.f32_vec(f"{pre}mlp.{i}.layers.0.bias").data_ptr()
                f.lnw, f.lnb = P.f32_vec(f"{pre}mlp.{i}.layers.1.weight").data_ptr(), P.f32_vec(f"{pre}mlp.{i}.layers.1.bias").data_ptr()
                f.W2, f.b2 = P.act(f"{pre}mlp.{i}.layers.3.weight").data_ptr(), P.f32_vec(f"{pre}mlp.{i}.layers.3.bias").data_ptr()
                f.Wproj, f.bproj = P.act(f"{pre}proj.{i}.weight").data_ptr(), P.f32_vec(f"{pre}proj.{i}.bias").data_ptr()
                f.Ks, f.Vs = buf(B, H, self.cap, hd_s, dtype=dt, zero=True).data_ptr(), buf(B, H, self.cap, hd_s, dtype=dt, zero=True).data_ptr()
                if self.exchange:
                    self._fill_norm(f.ln_cross, f"{pre}ln_cross.{i}.")
                    f.Wdown, f.bdown = P.act(f"{pre}cross_down.{i}.weight").data_ptr(), P.f32_vec(f"{pre}cross_down.{i}.bias").data_ptr()
                    f.Wup, f.bup = P.act(f"{pre}cross_up.{i}.weight").data_ptr(), P.f32_vec(f"{pre}cross_up.{i}.bias").data_ptr()
                    for j in range(F):
                        if j == i:
                            continue
                        ca, p = f"{pre}cross_attn.{i}.{j}.", Ly.p[i][j]
                        p.Wq, p.bq = P.act(ca + "q.weight").data_ptr(), P.f32_vec(ca + "q.bias").data_ptr()
                        p.Wkv, p.bkv = P.act(ca + "k.weight", 2 * D).data_ptr(), P.f32_vec(ca + "k.bias", 2 * D).data_ptr()
                        p.Wp = P.act(ca + "projection.weight").data_ptr()
                        p.Kc, p.Vc = buf(B, H, self.cap, hd_c, dtype=dt, zero=True).data_ptr(), buf(B, H, self.cap, hd_c, dtype=dt, zero=True).data_ptr()
        for i in range(F):
            self._fill_norm(G.final_ln[i], f"ln.{i}.")
        self._cond: Dict[int, CondPlan] = {}
        self._tag = 1

    def _fill_norm(self, nm, pre: str) -> None:
        P = self.eng.params
        nm.gamma = P.f32_vec(pre + "weight").data_ptr()
        nm.beta = P.f32_vec(pre + "bias").data_ptr() if self.adaln else None
        self._norms.append((nm, pre))

    def rollout(self, x0: torch.Tensor, ib: torch.Tensor, n_steps: int) -> torch.Tensor:
        """x0 [B, 1, F, E], ib [B, >= n_steps, 1] -> [B, n_steps, F, E] (fp32)."""
        B, F, E = self.B, self.F, self.E
        eng = self.eng
        eng.params.sync()
        traj = torch.empty(n_steps + 1, B, F, E, device=eng.device, dtype=torch.float32)
        traj[0].copy_(x0[:, 0])
        if n_steps == 0:
            return traj[1:].permute(1, 0, 2, 3).contiguous()
        cond = ib[:, :n_steps, 0].t().contiguous().float()     # [n_steps, B]: row pos * B + b
        M = n_steps * B
        cp: Optional[CondPlan] = None
        if self.adaln or self.has_ib:
            cp = self._cond.get(M)
            if cp is None:
                self._cond.clear()                             # one row count at a time: the modulation buffers are ~12 M x 2d elements
                cp = self._cond[M] = CondPlan(eng, M)
            for t in cp.ibufs:
                t.zero_()
            cp.bind_ptrs(0, cond.data_ptr(), 0)
            if not cp._audited or ptrcheck.always():
                cp.audit(owners=(cond,))
            cp.run()
        for nm, pre in self._norms:
            mod = cp.mods.get(pre) if (cp is not None and self.adaln) else None
            nm.mod, nm.ldmod = (mod.data_ptr(), mod.stride(0)) if mod is not None else (None, 0)
        for l in range(self.L):
            self.layers[l].ib = cp.ibufs[l].data_ptr() if (cp is not None and self.has_ib) else None
        self.G.traj = traj.data_ptr()
        for attempt in (0, 1):
            rc = N.lib().sea_kv_rollout(C.byref(self.G), self.layers, 0, n_steps, self._tag, N.dtype_code(eng.act_dtype), N.stream_ptr())
            N.check(rc, "sea_kv_rollout")
            self._tag = (self._tag + n_steps * self.L) & 0xFFFFFFFF or 1
            out = traj[1:].permute(1, 0, 2, 3).contiguous()
            if attempt == 0 and _switches.kv("force_err") == "1" and self.G.handoff_words > B * F * max(self.D, 1):
                self.err.fill_(1)           # test hook: behave as if a hand-off wait of the persistent launch had given up (tests/test_kv_fast_gpu.py)
            if int(self.err.item()) == 0:   # (synchronises)
                return out
            # A hand-off wait gave up: the persistent form needs all its workgroups on the chip at once, which another process on the same GPU can
            # deny.  Nothing is lost but time — every spin is bounded, the trajectory is recomputed from position 0 — so fall back once, for good, to
            # the seven launches per step (their only hand-off is between workgroups of one 3-workgroup launch).
            self.err.zero_()
            if attempt == 0 and self.G.handoff_words > B * F * max(self.D, 1):
                self.G.handoff_words = B * F * max(self.D, 1)
                continue
            raise RuntimeError("sea_kv_rollout: a hand-off wait inside the exchange tails gave up (results invalid)")
        return out
