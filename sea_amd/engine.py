"""Whole-model execution engine for TemporalModel on MI355X.

The reference runs TemporalModel.forward (models/temporal.py:405-416) as ~130 eager ATen calls per layer from Python
loops.  Here the same computation is a PLAN: a fixed list of ~30 grouped kernel launches per layer over
pre-allocated HBM workspaces, with every ABI struct filled once (include/sea_hip.h).  A plan is replayed either launch by
launch (a few microseconds of ctypes per launch) or, for the hot loops, as a captured HIP graph.

Memory layout (all in HBM, allocated through torch):
  * parameters: ONE fp32 buffer `flat32`; every nn.Parameter of the model is a view into it (the reference's state_dict
    schema is unchanged).  Live parameters first, the ~12 % that never receive a gradient at the tail, so
    the optimizer and the gradient all-reduce run over one contiguous prefix.  q/k/v weights of an attention module are
    adjacent in q,k,v order, so the fused [3E, E] projection is a plain view.
  * `flat_act`: the same layout in the activation dtype (bf16 shadow; aliases flat32 in fp32 mode).
  * residual stream: fp32 [M, E] per field (M = B*T rows); matrix-operand activations in the activation dtype.
  * Q [B,H,T,hd], K [B,H,cap,hd], V^T [B,H,hd,cap] per attention problem (written by the QKV epilogue).
"""
from __future__ import annotations

import os

import ctypes as C
import re
from typing import Dict, List, Optional, Tuple

import torch

from . import _native as N
from . import _switches
from . import ops
from . import ptrcheck

_QKV = re.compile(r"^(.*\.)(k|q|v)\.(weight|bias)$")
_QKV_RANK = {("q", "weight"): 0, ("k", "weight"): 1, ("v", "weight"): 2, ("q", "bias"): 3, ("k", "bias"): 4, ("v", "bias"): 5}


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def dead_prefixes(model) -> List[str]:
    """Parameter-name prefixes that never receive a gradient on this path: the unused middle AdaLN of every field, `ln.cross`, the
    diagonal cross-attention modules, the info-bottleneck's residual projection — and, per variant, the whole info-bottleneck layer when its output is
    not added (ib_addition_mode 'none': the reference still evaluates it, models/temporal.py:112-114) and the fixed random matrix of the Fourier form
    (models/base_blocks.py:147: requires_grad=False)."""
    out = []
    ib_none = model.ib_addition_mode.lower() == "none"
    for layer in range(model.num_layers):
        pre = f"blocks.{layer}."
        out += [pre + "ln.cross.", pre + "ib.residual_projection."]
        if ib_none:
            out.append(pre + "ib.")
        elif model.ib_scale_mode.lower() == "fourier":
            out.append(pre + "ib.W")
        for i in range(model.num_variables):
            out += [f"{pre}ln.exp.{i}.1.", f"{pre}cross_attn.{i}.{i}."]
        if model.exchange_mode == "pool":   # the 'mlp' pool update ignores its token argument (models/temporal.py:244-249): the token and its norm are never used
            out += [pre + "pool_token", pre + "ln_pool."]
    return out


def grad_phase(model, name: str) -> int:
    """When a parameter's gradient is FINAL in the hand-written backward, as an ordinal (lower = earlier): per layer, last layer first —
    0: the field MLP + proj; 1: everything between them and the self-attention (exchange, ln_cross, the norm in front of the MLP, the info-bottleneck step
    behind the exchange; with the last layer: the model's final norms); 2: self-attention, the norm in front of it, an info-bottleneck step in front of
    the block.  The flat buffers are laid out in this order, so that what a data-parallel step can reduce early is ONE contiguous slice per phase
    (train_engine.TrainPlan.grad_buckets)."""
    L = model.num_layers
    m = re.match(r"^blocks\.(\d+)\.(.*)$", name)
    if not m:
        return 1   # ln.{i}.*: with the last layer's middle phase
    l, rest = int(m.group(1)), m.group(2)
    if rest.startswith("mlp.") or rest.startswith("proj."):
        sub = 0
    elif rest.startswith("attn.self.") or re.match(r"^ln\.exp\.\d+\.0\.", rest):
        sub = 2
    elif (rest.startswith("ib.") or rest.startswith("cross_attn_ib.")) and not model.add_info_after_cross:
        sub = 2
    else:
        sub = 1
    return (L - 1 - l) * 3 + sub


class FlatParams:
    """One contiguous fp32 buffer holding every parameter of the model (+ the activation-dtype shadow)."""

    def __init__(self, model: torch.nn.Module, device: torch.device, act_dtype: torch.dtype):
        named = list(model.named_parameters())
        first_idx: Dict[str, int] = {}
        keyed = []
        for idx, (name, p) in enumerate(named):
            m = _QKV.match(name)
            if m:
                base = first_idx.setdefault(m.group(1), idx)
                keyed.append(((base, _QKV_RANK[(m.group(2), m.group(3))]), name, p))
            else:
                keyed.append(((idx, 0), name, p))
        dead = dead_prefixes(model)
        is_dead = lambda n: any(n.startswith(d) for d in dead)  # noqa: E731
        keyed.sort(key=lambda t: (is_dead(t[1]), grad_phase(model, t[1]), t[0]))
        self.offsets: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        off = 0
        self.n_live = 0
        for _, name, p in keyed:
            if is_dead(name) and self.n_live == 0 and off > 0:
                self.n_live = off
            self.offsets[name] = (off, tuple(p.shape))
            off += _round_up(p.numel(), 8)
        if self.n_live == 0:
            self.n_live = off
        self.n_total = off
        self.device, self.act_dtype = device, act_dtype
        self.flat32 = torch.zeros(self.n_total, device=device, dtype=torch.float32)
        with torch.no_grad():
            for _, name, p in keyed:
                o, shp = self.offsets[name]
                view = self.flat32[o:o + p.numel()].view(shp)
                view.copy_(p.detach().to(device=device, dtype=torch.float32))
                p.data = view  # the module's parameter now aliases the flat buffer
        self.live_names = [n for _, n, _ in keyed if not is_dead(n)]
        self.phase_of = {n: grad_phase(model, n) for _, n, _ in keyed if not is_dead(n)}
        self.flat_act = self.flat32 if act_dtype == torch.float32 else torch.empty(self.n_total, device=device, dtype=act_dtype)
        self.flat_actT: Optional[torch.Tensor] = None   # transposed weight shadow, built on first training use
        self._t_desc = self._t_tiles = None
        self._t_total = 0
        self._synced_version = -1
        self._synced_T_version = -1
        self.sync()

    # -- views
    def f32(self, name: str) -> torch.Tensor:
        o, shp = self.offsets[name]
        n = 1
        for s in shp:
            n *= s
        return self.flat32[o:o + n].view(shp)

    def act(self, name: str, rows: Optional[int] = None) -> torch.Tensor:
        """Activation-dtype view of a 2-D weight; rows > shape[0] extends over the adjacent tensors (fused q|k|v)."""
        o, shp = self.offsets[name]
        r = shp[0] if rows is None else rows
        return self.flat_act[o:o + r * shp[1]].view(r, shp[1])

    def f32_vec(self, name: str, n: Optional[int] = None) -> torch.Tensor:
        o, shp = self.offsets[name]
        k = shp[0] if n is None else n
        return self.flat32[o:o + k]

    # -- transposed shadow (operand of the data-gradient GEMMs: dX = dY . W needs W^T in the K-contiguous [N,K] form)
    def enable_transposed_shadow(self) -> None:
        if self.flat_actT is not None:
            return
        rows = []
        fused_done = set()
        for name, (o, shp) in self.offsets.items():
            if len(shp) != 2 or shp[1] == 1 or ".ib." in name:
                continue
            m = _QKV.match(name)
            if m and m.group(3) == "weight":
                pre, which = m.group(1), m.group(2)
                is_self = ".attn.self." in pre
                if is_self:
                    if which == "q":
                        rows.append((o, o, 3 * shp[0], shp[1]))      # [q;k;v] as one [3E, E] matrix
                    continue
                if which == "q":
                    rows.append((o, o, shp[0], shp[1]))
                elif which == "k":
                    rows.append((o, o, 2 * shp[0], shp[1]))          # [k;v] as one [2D, D] matrix
                continue
            rows.append((o, o, shp[0], shp[1]))
        desc = torch.tensor(rows, dtype=torch.int64)
        tiles = ((desc[:, 2] + 31) // 32) * ((desc[:, 3] + 31) // 32)
        ts = torch.zeros(len(rows) + 1, dtype=torch.int32)
        ts[1:] = torch.cumsum(tiles, 0).to(torch.int32)
        self._t_total = int(ts[-1])
        self._t_desc, self._t_tiles = desc.to(self.device), ts.to(self.device)
        self.flat_actT = torch.zeros(self.n_total, device=self.device, dtype=self.act_dtype)
        self.sync_transposed(force=True)

    def sync_transposed(self, force: bool = False) -> None:
        if self.flat_actT is None:
            return
        v = self.flat32._version
        if force or v != self._synced_T_version:
            N.check(N.lib().sea_transpose_weights(self.flat32.data_ptr(), self.flat_actT.data_ptr(), N.dtype_code(self.act_dtype),
                                                  self._t_desc.data_ptr(), self._t_tiles.data_ptr(), self._t_desc.shape[0], self._t_total,
                                                  N.stream_ptr()), "weight transpose")
            self._synced_T_version = v

    def actT(self, name: str, rows: Optional[int] = None) -> torch.Tensor:
        """Activation-dtype view of W^T for the weight `name` ([rows, cols] in the model, possibly fused over adjacent tensors)."""
        o, shp = self.offsets[name]
        r = shp[0] if rows is None else rows
        return self.flat_actT[o:o + r * shp[1]].view(shp[1], r)

    def sync(self, force: bool = False) -> None:
        """Refresh the activation-dtype shadow if any parameter view was modified in place since the last sync."""
        if self.act_dtype == torch.float32:
            return
        v = self.flat32._version
        if force or v != self._synced_version:
            N.check(N.lib().sea_convert_f32_to_act(self.flat32.data_ptr(), self.n_total, self.flat_act.data_ptr(), self.n_total, 1,
                                                   self.n_total, N.dtype_code(self.act_dtype), N.stream_ptr()), "weight shadow")
            self._synced_version = v


class _Rec:
    """One pre-built launch: C function + argument list (the stream is appended at run time).  `lane` selects the stream of a
    concurrent replay (0 = the caller's stream); fn None marks a lane fork / join (name "fork" / "join")."""
    __slots__ = ("fn", "args", "name", "keep", "lane")

    def __init__(self, fn, args, name, keep=None, lane=0):
        self.fn, self.args, self.name, self.keep, self.lane = fn, list(args), name, keep, lane


class Plan:
    """Launch list for one (B, T, mode) of TemporalModel.forward."""

    def __init__(self, eng: "TemporalEngine", B: int, T: int, mode: str = "full", save_for_backward: bool = False, cond=None):
        assert mode in ("full", "step")
        self.eng, self.B, self.T, self.mode = eng, B, T, mode
        # `cond` (step mode): a kv_engine.CondPlan that has evaluated what depends on the condition only — every AdaLN modulation, the
        # info-bottleneck term — for ALL steps of a rollout (row s * B + b): the step plan then carries no condition launch at all and reads row
        # block s of those buffers, its pointers into them advanced per step (self._hoisted, step_patch_table)
        self._cond_src = cond
        self._hoisted: List[Tuple[object, str, int, int]] = []   # (struct, field, address at step 0, bytes per step)
        m = eng.model
        self.F, self.H, self.L = m.num_variables, m.n_heads, m.num_layers
        # E: the row width INSIDE a block; Eo: the width of the model's input / output rows and of every block's proj output.  They differ only with
        # ib_addition_mode 'concat' (models/temporal.py:48: rows widened by ib_dim_concat = 64 info-bottleneck columns, `ib_dim` wide layer)
        self.Eo, self.E = m.embed_dim, m.internal_embed_dim
        self.concat = m.ib_addition_mode.lower() == "concat"
        self.ib_dim = self.E - self.Eo if self.concat else self.E
        self.D = m.down_dim
        self.S = m.mlp_hidden
        self.M = B * T
        self.dt = eng.act_dtype
        self.code = N.dtype_code(self.dt)
        self.adaln = m.LN_type.lower() == "adaln"
        self.cap = _round_up(m.max_len, 8) if mode == "step" else _round_up(T, 8)
        self.pos0 = 0
        self.records: List[_Rec] = []
        self._cur: List[_Rec] = self.records   # list the record builders append to (forward or backward)
        self._keep: List[object] = []
        self._x_patches: List[Tuple[object, str, int]] = []   # (struct, field, byte offset from x base)
        self._out_patches: List[Tuple[object, str, int]] = []
        self._c_patches: List[Tuple[object, object]] = []      # (container, key) receiving the condition pointer
        self._pos_structs: List[object] = []                   # SeaQkvCommon / SeaAttnParams to update per step
        self._drop_structs: List[object] = []                  # structs whose .drop.seed is re-keyed every training step
        self._bound = (None, None, None)
        self._bound_tensors = None        # (x, ib, out) of the last bind(): held until the next one (see bind)
        self._audited = False             # the pointer audit has run on this plan (ptrcheck)
        self._lane = 0                    # lane the record builders tag new records with
        self._lane_streams: Dict[int, torch.cuda.Stream] = {}
        self._zero_ib: Optional[torch.Tensor] = None   # 'concat': the zeros the info-bottleneck columns are reset to
        self._build()
        self._find_hoisted()
        self._clist = None          # (SeaLaunchRec array, [(rec index, field, args list, args index)]) for sea_run_list
        self._compile_list()

    def _find_hoisted(self) -> None:
        """Every pointer field of the launch structs that points into row block 0 of a hoisted condition buffer (self._cond_src): (struct, field,
        address, bytes per step) — found by address, so that no record builder has to know about the hoisting."""
        src = self._cond_src
        if src is None:
            return
        spans = []   # (lo, hi, bytes per step)
        for t in list(src.mods.values()) + list(src.ibufs):
            step = self.M * t.stride(0) * t.element_size()
            spans.append((t.data_ptr(), t.data_ptr() + step, step))
        for r in self.records:
            if r.fn is None:
                continue
            roots = [a for a in r.args if isinstance(a, (C.Structure, C.Array))] + ([r.keep] if r.keep is not None else [])
            seen = set()
            for root in roots:
                for st in ptrcheck._walk_structs(root):
                    for path, pval in ptrcheck.iter_pointers(st):
                        name = path[1:]
                        if "[" in name or (C.addressof(st), name) in seen:
                            continue
                        for lo, hi, step in spans:
                            if lo <= pval < hi:
                                seen.add((C.addressof(st), name))
                                self._hoisted.append((st, name, pval, step))
                                break
        self._keep.append(src)   # the buffers live as long as this plan

    def set_hoisted_step(self, s: int) -> None:
        """Point the plan at row block s of the hoisted condition buffers (the native step loop does the same through its patch table)."""
        for st, field, base, step in self._hoisted:
            setattr(st, field, base + s * step)

    # ------------------------------------------------------------------ allocation helpers
    def _buf(self, *shape, dtype=None, zero=False) -> torch.Tensor:
        dt = self.dt if dtype is None else dtype
        t = (torch.zeros if zero else torch.empty)(*shape, device=self.eng.device, dtype=dt)
        self._keep.append(t)
        return t

    # ------------------------------------------------------------------ lanes (concurrent replay)
    def _rec(self, fn, args, name, keep=None) -> _Rec:
        return _Rec(fn, args, name, keep, self._lane)

    def _fork(self, lane: int) -> None:
        """Records built until the matching _join run on `lane`: in a concurrent replay that stream first waits for everything
        the main stream has been given so far, then runs beside it."""
        assert self._lane == 0 and lane > 0
        self._cur.append(_Rec(None, [], "fork", None, lane))
        self._lane = lane

    def _end_lane(self) -> None:
        self._lane = 0

    def _join(self, lane: int) -> None:
        """The main stream waits for everything given to `lane`."""
        assert self._lane == 0
        self._cur.append(_Rec(None, [], "join", None, lane))

    # ------------------------------------------------------------------ record builders
    def _gemm_splitk(self, chunk: List[dict], name: str) -> bool:
        """Split-K form of a launch of skinny-M Linear layers with a very long contraction (the MLP of configs/multiphase_flow.py:112-141 at M = B T = 796: fc2 forward and
        fc1's data gradient, 2048 x 16384): at 128 x 128 tiles such a launch is 224 workgroups each walking K = 16384 alone; four K-quarters as groups of ONE
        sea_gemm_grouped launch (fp32 partial matrices, two workgroups per CU on the two-stage ring) + sea_splitk_finish: 161 -> 128 + 10 us (tools/skinny_probe.py).
        bf16, plain epilogue (bias / residual / outputs only).  SEA_PLAN=splitk=0 keeps the single launch."""
        S = 4
        if self.dt != torch.bfloat16 or _switches.plan("splitk", "1") == "0" or len(chunk) * S > N.MAX_GROUPS or len(chunk) > N.MAX_SPLITK_GROUPS:
            return False
        tiles = 0
        for d in chunk:
            A, W = d.get("A"), d["W"]
            if (A is None or d.get("silu") is not None or d.get("act", 0) != 0 or d.get("drop") is not None or d.get("n_seg", 1) != 1 or d.get("Z") is not None
                    or d.get("R_is_x") is not None or d.get("ldr") is not None or d.get("ldc32") is not None or d.get("M") is not None):
                return False
            Mg, Ng, Kg = A.shape[-2], W.shape[0], W.shape[1]
            if Kg < 8192 or Kg % (S * 64) != 0 or Mg > 1024 or Ng % 4 != 0 or A.dim() != 2:
                return False
            tiles += ((Mg + 127) // 128) * ((Ng + 127) // 128)
        if tiles >= 384:
            return False
        parts, fin = [], (N.SeaSplitkGroup * len(chunk))()
        for f_, d in zip(fin, chunk):
            A, W = d["A"], d["W"]
            Mg, Ng, ks = A.shape[0], W.shape[0], W.shape[1] // S
            P = self._buf(S, Mg, Ng, dtype=torch.float32)
            for q in range(S):
                parts.append(dict(A=A[:, q * ks:(q + 1) * ks], W=W[:, q * ks:(q + 1) * ks], C32=P[q]))
            R, C32, Cact, bias = d.get("R"), d.get("C32"), d.get("Cact"), d.get("bias")
            f_.P, f_.p_stride, f_.S, f_.M, f_.N, f_.ldp = P.data_ptr(), P.stride(0), S, Mg, Ng, P.stride(1)
            f_.bias, f_.bias_scale = N.ptr(bias), d.get("bias_scale", 1.0)
            f_.R, f_.ldr = N.ptr(R), (R.stride(0) if R is not None else 0)
            f_.C32, f_.ldc32 = N.ptr(C32), (C32.stride(0) if C32 is not None else 0)
            f_.Cact, f_.ldcact = N.ptr(Cact), (Cact.stride(0) if Cact is not None else 0)
        arr = (N.SeaGemmGroup * len(parts))()
        for g, d in zip(arr, parts):
            _fill_gemm(g, **d)
        L = N.lib()
        self._cur.append(self._rec(L.sea_gemm_grouped, [arr, len(parts), self.code], name + ".splitk", arr))
        self._cur.append(self._rec(L.sea_splitk_finish, [fin, len(chunk), self.code], name, fin))
        return True

    def _gemm(self, groups: List[dict], name: str) -> None:
        L = N.lib()
        for s in range(0, len(groups), N.MAX_GROUPS):
            chunk = groups[s:s + N.MAX_GROUPS]
            if self._gemm_splitk(chunk, name):
                continue
            arr = (N.SeaGemmGroup * len(chunk))()
            for g, d in zip(arr, chunk):
                _fill_gemm(g, **d)
            self._cur.append(self._rec(L.sea_gemm_grouped, [arr, len(chunk), self.code], name, arr))
            for g, d in zip(arr, chunk):
                if d.get("silu") is not None:
                    self._c_patches.append((g, "silu_c"))
                if d.get("R_is_x") is not None:
                    self._x_patches.append((g, "R", d["R_is_x"]))
                if d.get("drop") is not None:
                    self._drop_structs.append(g)

    def _norm(self, groups: List[dict], d: int, name: str, x_is_act=False, gelu=False) -> None:
        L = N.lib()
        for s in range(0, len(groups), N.MAX_NORM_GROUPS):
            chunk = groups[s:s + N.MAX_NORM_GROUPS]
            arr = (N.SeaNormGroup * len(chunk))()
            for g, gd in zip(arr, chunk):
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
                if gd.get("X_is_x") is not None:
                    self._x_patches.append((g, "X", gd["X_is_x"]))
                if gd.get("Y_is_out") is not None:
                    self._out_patches.append((g, "Y32", gd["Y_is_out"]))
            self._cur.append(self._rec(L.sea_rownorm, [arr, len(chunk), self.M, d, int(x_is_act), int(gelu), 1e-5, self.code], name, arr))

    def _norm_specs(self, specs, n: int):
        """SeaNormGroup array of the `pre` specs of a few-row launch (None entries: no norm for that group), with the bind-time patches of _norm."""
        if specs is None or all(sp is None for sp in specs):
            return None
        arr = (N.SeaNormGroup * n)()
        for g, sp in zip(arr, specs):
            if sp is None:
                continue
            ops.fill_norm_group(g, sp)
            if sp.get("X_is_x") is not None:
                self._x_patches.append((g, "X", sp["X_is_x"]))
        return arr

    def _gemm_few(self, groups: List[dict], name: str, pre=None, pre_act: bool = False, pre_gelu: bool = False) -> None:
        """sea_gemm_fewrows (step plans at the shipped widths): group dicts as _gemm; pre[i]: the _norm group dict (or None) of the row norm in front of the layer.
        Groups with pre[i] carry no A."""
        L = N.lib()
        n = len(groups)
        assert n <= N.FEW_MAX_GROUPS
        arr = (N.SeaGemmGroup * n)()
        for g, d in zip(arr, groups):
            d = dict(d)
            if d.get("A") is None:
                d["A"] = d["W"]   # shapes only; the operand comes from pre[i]
                _fill_gemm(g, **d)
                g.A, g.lda, g.M = None, 0, self.M
            else:
                _fill_gemm(g, **d)
            if d.get("R_is_x") is not None:
                self._x_patches.append((g, "R", d["R_is_x"]))
        pa = self._norm_specs(pre, n)
        self._cur.append(self._rec(L.sea_gemm_fewrows, [arr, pa, n, int(pre_act), int(pre_gelu), 1e-5, self.code], name, (arr, pa)))

    def _qkv_few(self, groups: List[dict], rope: torch.Tensor, hd: int, name: str, pre=None) -> None:
        """sea_qkv_rope_fewrows: group dicts as _qkv; pre[i]: the _norm group dict of the row norm in front of the projection (or None)."""
        L = N.lib()
        n = len(groups)
        assert n <= N.FEW_MAX_GROUPS
        arr = (N.SeaQkvGroup * n)()
        for g, d in zip(arr, groups):
            A, W = d.get("A"), d["W"]
            g.A, g.lda = (A.data_ptr(), A.stride(0)) if A is not None else (None, 0)
            g.W, g.bias = W.data_ptr(), d["bias"].data_ptr()
            g.Qout, g.Kout, g.Vtout, g.Vout = N.ptr(d.get("Q")), N.ptr(d.get("K")), N.ptr(d.get("Vt")), N.ptr(d.get("V"))
            g.ldw = W.stride(0)
            g.M, g.N, g.K, g.col0 = self.M, W.shape[0], W.shape[1], d["col0"]
        pa = self._norm_specs(pre, n)
        common = N.SeaQkvCommon(rope.data_ptr(), self.H, hd, self.T, self.pos0, self.cap, ops.q_scale(hd))
        self._pos_structs.append(common)
        self._cur.append(self._rec(L.sea_qkv_rope_fewrows, [arr, pa, n, C.byref(common), 1e-5, self.code], name, (arr, pa, common)))

    def _gemm_norm(self, groups: List[dict], name: str) -> None:
        """Linear + row normalisation in one launch (sea_gemm_rownorm): group dicts as _gemm (A, W, bias) plus the _norm keys
        (mod, gamma, beta, Yact, Y32/ldy32/Y_is_out)."""
        L = N.lib()
        for s in range(0, len(groups), N.MAX_GEMM_NORM_GROUPS):
            chunk = groups[s:s + N.MAX_GEMM_NORM_GROUPS]
            arr = (N.SeaGemmNormGroup * len(chunk))()
            for g, d in zip(arr, chunk):
                ops.fill_gemm_norm_group(g, d["A"], d["W"], d["gamma"], bias=d.get("bias"), R=d.get("R"), C32=d.get("C32"), mod=d.get("mod"),
                                         beta=d.get("beta"), Y32=d.get("Y32"), Yact=d.get("Yact"), mean=d.get("mean"), rstd=d.get("rstd"),
                                         ldy32=d.get("ldy32"), n_seg=d.get("n_seg", 1),
                                         a_seg_stride=d.get("a_seg_stride", 0), bias_scale=d.get("bias_scale", 1.0), Cact=d.get("Cact"), ib=d.get("ib"))
                if d.get("Y_is_out") is not None:
                    self._out_patches.append((g, "Y32", d["Y_is_out"]))
                if d.get("ib") is not None:
                    self._c_patches.append((g, "ib_c"))
            self._cur.append(self._rec(L.sea_gemm_rownorm, [arr, len(chunk), 1e-5, self.code], name, arr))

    def _xtail(self, att, Wp, Wup, bup, bias_scale, X, down, name: str) -> None:
        """One field's exchange tail in one launch (sea_exchange_tail): projections + GELU, up-projection of their sum + residual, and — `down` —
        the down-projection + row norm of the updated rows."""
        P = (N.SeaExchangeTail * 1)()
        ops.fill_exchange_tail(P[0], att, Wp, Wup, bup, bias_scale, X, None, down)
        self._cur.append(self._rec(N.lib().sea_exchange_tail, [P, 1, 1e-5, self.code], name, P))

    def _adaln(self, groups: List[dict], name: str) -> None:
        """AdaLN as the epilogue of cond_mlp.2's GEMM, and plain cond_mlp.2 groups, in one launch (sea_gemm_adaln): dicts as ops.fill_adaln_group, plus `X_is_x`
        (byte offset into the caller's tensor: the rows of the first layer are read from it, strided) and `Y_is_out`."""
        for s in range(0, len(groups), N.MAX_ADALN_GROUPS):
            chunk = groups[s:s + N.MAX_ADALN_GROUPS]
            arr = (N.SeaAdalnGroup * len(chunk))()
            for g, d in zip(arr, chunk):
                d = dict(d)
                x_off, o_off = d.pop("X_is_x", None), d.pop("Y_is_out", None)
                ops.fill_adaln_group(g, **d)
                if x_off is not None:
                    self._x_patches.append((g, "X", x_off))
                if o_off is not None:
                    self._out_patches.append((g, "Y32", o_off))
            self._cur.append(self._rec(N.lib().sea_gemm_adaln, [arr, len(chunk), 1e-5, self.code], name, arr))

    def _chain(self, groups: List[dict], rope: torch.Tensor, hd: int, name: str) -> None:
        """A row-local chain between two attention launches (sea_row_chain): group dicts as ops.fill_row_chain, plus `Xin_is_x` (byte offset into the caller's
        tensor: the residual rows of the first layer are read from it, strided)."""
        for s in range(0, len(groups), N.CHAIN_MAX_GROUPS):
            chunk = groups[s:s + N.CHAIN_MAX_GROUPS]
            arr = (N.SeaRowChain * len(chunk))()
            for g, d in zip(arr, chunk):
                d = dict(d)
                x_off = d.pop("Xin_is_x", None)
                ops.fill_row_chain(g, **d)
                if x_off is not None:
                    self._x_patches.append((g, "Xin", x_off))
            common = N.SeaQkvCommon(rope.data_ptr(), self.H, hd, self.T, self.pos0, self.cap, ops.q_scale(hd))
            self._pos_structs.append(common)
            rd = self._take_riders(len(chunk), max(g.M for g in arr)) if s == 0 else None
            if rd is None:
                self._cur.append(self._rec(N.lib().sea_row_chain, [arr, len(chunk), C.byref(common), 1e-5, self.code], name, (arr, common)))
            else:
                rarr, n_r, t0, nt, ibp = rd
                self._cur.append(self._rec(N.lib().sea_row_chain_riders, [arr, len(chunk), C.byref(common), rarr, n_r, t0, nt, (C.byref(ibp) if ibp is not None else None), 1e-5, self.code],
                                           name, (arr, common, rarr, ibp)))

    def _take_riders(self, n_groups: int, m_rows: int, last: bool = False):
        """The share of the rider work (sea_row_chain_riders) the next chain launch carries: as many 128 x 128 tiles of the rider GEMM as its idle CUs take in about the
        launch's own duration (a chain workgroup owns a CU; two tile rounds under the three-field launch behind the self-attention, one under a field's tail), the
        information-bottleneck rows with the first host.  None when nothing is left."""
        arr = getattr(self, "_rider_arr", None)
        if arr is None:
            return None
        done = getattr(self, "_rider_done", 0)
        ibp = self._rider_ib if not getattr(self, "_rider_ib_done", False) else None
        left = self._rider_tiles - done
        if left <= 0 and ibp is None:
            return None
        # measured at cfg2 (tools/chain_probe.py replay, SEA_PLAN=rider_caps): 128:128:128 tiles under the three hosts 0.2153 ms per step, 192:96:96 0.2178,
        # 256:64:64 0.2201, 384:0:0 0.2233, 0:192:192 0.2250, no riders 0.2214 — equal shares (a rider tile, alone on its CU beside 127-192 chain
        # workgroups, takes ~9 us; the hosts last 24 / 16 / 12 us)
        hosts_total = getattr(self, "_rider_hosts_total", 1)
        cap = (self._rider_tiles + hosts_total - 1) // hosts_total
        caps = _switches.plan("rider_caps", "")   # tuning aid: "a:b:c" = tiles for the first, second, third host
        if caps:
            lst = [int(v) for v in caps.split(":")]
            k = getattr(self, "_rider_host_no", 0)
            self._rider_host_no = k + 1
            cap = lst[k] if k < len(lst) else 0
            take = min(left, cap)
            self._rider_done = done + take
            self._rider_ib_done = True
            return arr, len(arr), done, take, ibp
        take = left if (last or self._rider_hosts_left <= 1) else min(left, cap)
        self._rider_hosts_left -= 1
        self._rider_done = done + take
        self._rider_ib_done = True
        return arr, len(arr), done, take, ibp

    def _qkv(self, groups: List[dict], rope: torch.Tensor, hd: int, name: str) -> None:
        L = N.lib()
        arr = (N.SeaQkvGroup * len(groups))()
        for g, d in zip(arr, groups):
            A, W = d["A"], d["W"]
            g.A, g.W, g.bias = A.data_ptr(), W.data_ptr(), d["bias"].data_ptr()
            g.Qout, g.Kout, g.Vtout, g.Vout = N.ptr(d.get("Q")), N.ptr(d.get("K")), N.ptr(d.get("Vt")), N.ptr(d.get("V"))
            g.lda, g.ldw = A.stride(0), W.stride(0)
            g.M, g.N, g.K, g.col0 = self.M, W.shape[0], W.shape[1], d["col0"]
        common = N.SeaQkvCommon(rope.data_ptr(), self.H, hd, self.T, self.pos0, self.cap, ops.q_scale(hd))
        self._pos_structs.append(common)
        self._cur.append(self._rec(L.sea_qkv_rope_grouped, [arr, len(groups), C.byref(common), self.code], name, (arr, common)))

    def _attn(self, problems: List[dict], hd: int, ldo: int, name: str, drop=None, src_len: Optional[int] = None) -> None:
        L = N.lib()
        for s in range(0, len(problems), N.MAX_ATTN_PROBLEMS):
            chunk = problems[s:s + N.MAX_ATTN_PROBLEMS]
            P = N.SeaAttnParams()
            P.n_problems = len(chunk)
            if drop is not None:  # (thr, first stream of this launch)
                P.drop.thr, P.drop.stream = drop[0], drop[1] + s
                self._drop_structs.append(P)
            for i, d in enumerate(chunk):
                P.p[i].Q, P.p[i].K, P.p[i].Vt, P.p[i].O = d["Q"].data_ptr(), d["K"].data_ptr(), d["Vt"].data_ptr(), d["O"].data_ptr()
                P.p[i].LSE = N.ptr(d.get("LSE"))
            P.B, P.H, P.hd, P.Tq, P.Tk, P.cap = self.B, self.H, hd, self.T, self.pos0 + self.T, self.cap
            P.q_pos0, P.src_len, P.ldo = self.pos0, (self.eng.model.src_len if src_len is None else src_len), ldo
            self._pos_structs.append(P)
            self._cur.append(self._rec(L.sea_attention_fwd, [C.byref(P), self.code], name, P))

    def _cond_mods(self, split: bool = False, riders: bool = False) -> Dict[str, torch.Tensor]:
        """AdaLN condition MLPs for the WHOLE model: silu launch + grouped GEMM (cond_mlp.2); returns prefix -> [M, 2d] (w | b).
        `split`: the modules the first launch of the layer needs (AdaLN_0 of layer 0) go first on the main stream, all the others run
        on lane 1 beside the self-attention and are joined by the caller (self._join(1)) before their first use."""
        P, F, E, D, M, L = self.eng.params, self.F, self.E, self.D, self.M, N.lib()
        mods: Dict[str, torch.Tensor] = {}
        if not self.adaln:
            return mods
        if self._cond_src is not None:   # evaluated for all steps up front: row block 0 here, advanced per step
            return {pre: t[:M] for pre, t in self._cond_src.mods.items()}
        first, rest = [], []
        for l in range(self.L):
            pre = f"blocks.{l}."
            for i in range(F):
                (first if l == 0 else rest).append((f"{pre}ln.exp.{i}.0.", E))
                rest.append((f"{pre}ln.exp.{i}.2.", E))
            if self.eng.model.exchange_mode != "simple":
                for i in range(F):
                    rest.append((f"{pre}ln_cross.{i}.", D))
        for i in range(F):
            rest.append((f"ln.{i}.", self.Eo))

        # cond_mlp.0 + SiLU evaluated inside the GEMM of cond_mlp.2 (generated A operand): no hidden matrix, no silu launch.  Inference plans
        # only (the weight gradient of cond_mlp.2 reads the hidden matrix).  The operand is recomputed by every column tile of a row panel (4x at
        # N = 512), VALU work that pays only once the hidden matrix's HBM round trip is the larger cost: measured 0.2685 against 0.2671 ms at cfg2
        # (M = 2024: not used), 1.215 against 1.241 ms at B = 8 (used).  SEA_PLAN=silu=1|0 forces.
        gen_a = self._gen_a(first + rest)
        ib_todo = list(getattr(self, "_ib_fold", []))   # (layer prefix, ibuf): info-bottleneck MLPs evaluated by extra row passes of the first silu launch
        self._rider_arr, self._rider_ib = None, None
        if riders:
            # the silu launch as it is (hidden rows of every module + the information-bottleneck rows); cond_mlp.2 of AdaLN_0 and ln_cross of the (only) layer in
            # front; cond_mlp.2 of every other module as rider tiles of the chain launches
            front = first + [(pre_, d) for pre_, d in rest if "ln_cross." in pre_]
            later = [(pre_, d) for pre_, d in rest if (pre_, d) not in front]
            # The front of the block as ONE launch (sea_adaln_qkv, round 4): the condition MLP of AdaLN_0 with its hidden rows generated in the launch, AdaLN_0 and the
            # self-attention's q / k / v + rotary epilogue — no hidden rows, no modulation matrix and no normalised rows of these modules in memory, no QKV launch;
            # cond_mlp.2 of ln_cross rides on the CUs it leaves idle.  SEA_PLAN=front=0 keeps silu + sea_gemm_adaln + QKV.
            self._front_chain = (_switches.plan("front", "1") != "0" and _switches.plan("adaln_gemm", "1") != "0" and ops.adaln_qkv_supported(self.dt, E, self.H)
                                 and F <= N.MAX_AQKV_GROUPS and not self.concat)
            # ... and with D = 128 the ln_cross modulation of a field is that launch's third layer (hidden rows generated in the launch, 2 D = 256 columns), the hidden
            # rows of the modules further down and the information-bottleneck rows its row riders: no silu launch at all.  SEA_PLAN=front3=0 keeps the silu launch.
            self._front3 = self._front_chain and D == 128 and len(later) <= N.AQKV_MAX_SILU and len(ib_todo) <= 1 and _switches.plan("front3", "1") != "0"
            silu_groups, hids = [], {}
            for pre_, d in front + later:
                mods[pre_] = self._buf(M, 2 * d)
                if self._front_chain and ((pre_, d) in first or (self._front3 and "ln_cross." in pre_)):
                    continue   # generated inside sea_adaln_qkv
                hids[pre_] = self._buf(M, 2 * d)
                silu_groups.append((P.f32_vec(pre_ + "cond_mlp.0.weight", 2 * d), P.f32_vec(pre_ + "cond_mlp.0.bias"), hids[pre_]))
            self._front_rows = None
            if self._front3:   # the silu / ib rows as row riders of sea_adaln_qkv (emitted by _build)
                sarr = (N.SeaSiluGroup * max(len(silu_groups), 1))()
                for g, (w1, b1, hid) in zip(sarr, silu_groups):
                    g.w1, g.b1, g.Hid, g.K2, g.ld = w1.data_ptr(), b1.data_ptr(), hid.data_ptr(), hid.shape[1], hid.stride(0)
                ibp = None
                if ib_todo:
                    ibp = N.SeaIbParams()
                    lpre, ibuf = ib_todo[0]
                    ibp.X[0], ibp.n_fields, ibp.ldx = ibuf.data_ptr(), 1, ibuf.stride(0)
                    self._fill_ib(ibp, lpre)
                    ib_todo.clear()
                self._front_rows = (sarr, len(silu_groups), ibp)
                silu_groups = []
            for s_ in range(0, len(silu_groups), N.MAX_SILU_GROUPS):
                chunk = silu_groups[s_:s_ + N.MAX_SILU_GROUPS]
                sarr = (N.SeaSiluGroup * len(chunk))()
                for g, (w1, b1, hid) in zip(sarr, chunk):
                    g.w1, g.b1, g.Hid, g.K2, g.ld = w1.data_ptr(), b1.data_ptr(), hid.data_ptr(), hid.shape[1], hid.stride(0)
                ibs, n_ib = None, 0
                if ib_todo:
                    n_ib = len(ib_todo)
                    ibs = (N.SeaIbParams * n_ib)()
                    for ibp, (lpre, ibuf) in zip(ibs, ib_todo):
                        ibp.X[0], ibp.n_fields, ibp.ldx = ibuf.data_ptr(), 1, ibuf.stride(0)
                        self._fill_ib(ibp, lpre)
                    ib_todo.clear()
                rec = self._rec(L.sea_silu_outer_ib, [sarr, len(chunk), None, M, self.code, ibs, n_ib], "adaln.silu", (sarr, ibs))
                self._c_patches.append((rec.args, 2))
                self._cur.append(rec)
            # cond_mlp.2 of the front modules: AdaLN_0's as the GEMM whose epilogue IS the normalisation (sea_gemm_adaln: no modulation matrix, no norm launch),
            # ln_cross's as plain groups of the same launch — emitted by _build where the AdaLN_0 launch used to be.  SEA_PLAN=adaln_gemm=0 keeps GEMM + norm launch.
            if _switches.plan("adaln_gemm", "1") != "0":
                self._adaln_front = {pre_: (hids.get(pre_), P.act(pre_ + "cond_mlp.2.weight"), P.f32_vec(pre_ + "cond_mlp.2.bias")) for pre_, _ in front}
            else:
                self._adaln_front = None
                self._gemm([dict(A=hids[pre_], W=P.act(pre_ + "cond_mlp.2.weight"), bias=P.f32_vec(pre_ + "cond_mlp.2.bias"), Cact=mods[pre_]) for pre_, _ in front], "adaln.cond_gemm.front")
            arr = (N.SeaGemmGroup * len(later))()
            for g, (pre_, d) in zip(arr, later):
                _fill_gemm(g, A=hids[pre_], W=P.act(pre_ + "cond_mlp.2.weight"), bias=P.f32_vec(pre_ + "cond_mlp.2.bias"), Cact=mods[pre_])
            self._rider_arr = arr
            self._rider_tiles = sum(((M + 127) // 128) * ((2 * d + 127) // 128) for _, d in later)
            self._keep.append((arr, self._rider_ib))
            return mods

        def emit(inst, tag):
            silu_groups, gemm_groups = [], []
            if gen_a:
                if ib_todo:   # no silu rows to ride on: the information-bottleneck rows as a launch of their own (added by the norm pass in front of the MLP)
                    n_ib = len(ib_todo)
                    ibs = (N.SeaIbParams * n_ib)()
                    for ibp, (lpre, ibuf) in zip(ibs, ib_todo):
                        ibp.X[0], ibp.n_fields, ibp.ldx = ibuf.data_ptr(), 1, ibuf.stride(0)
                        self._fill_ib(ibp, lpre)
                    ib_todo.clear()
                    rec = self._rec(L.sea_silu_outer_ib, [None, 0, None, M, self.code, ibs, n_ib], "ib.rows", (ibs,))
                    self._c_patches.append((rec.args, 2))
                    self._cur.append(rec)
                for pre, d in inst:
                    mod = self._buf(M, 2 * d)
                    mods[pre] = mod
                    gemm_groups.append(dict(A=None, M=M, W=P.act(pre + "cond_mlp.2.weight"), bias=P.f32_vec(pre + "cond_mlp.2.bias"), Cact=mod,
                                            silu=(P.f32_vec(pre + "cond_mlp.0.weight", 2 * d), P.f32_vec(pre + "cond_mlp.0.bias"))))
                self._gemm(gemm_groups, "adaln.cond_gemm" + tag)
                return
            for pre, d in inst:
                hid = self._buf(M, 2 * d)
                mod = self._buf(M, 2 * d)
                mods[pre] = mod
                silu_groups.append((P.f32_vec(pre + "cond_mlp.0.weight", 2 * d), P.f32_vec(pre + "cond_mlp.0.bias"), hid))
                gemm_groups.append(dict(A=hid, W=P.act(pre + "cond_mlp.2.weight"), bias=P.f32_vec(pre + "cond_mlp.2.bias"), Cact=mod))
            for s in range(0, len(silu_groups), N.MAX_SILU_GROUPS):
                chunk = silu_groups[s:s + N.MAX_SILU_GROUPS]
                arr = (N.SeaSiluGroup * len(chunk))()
                for g, (w1, b1, hid) in zip(arr, chunk):
                    g.w1, g.b1, g.Hid, g.K2, g.ld = w1.data_ptr(), b1.data_ptr(), hid.data_ptr(), hid.shape[1], hid.stride(0)
                ibs, n_ib = None, 0
                if ib_todo:
                    n_ib = len(ib_todo)
                    ibs = (N.SeaIbParams * n_ib)()
                    for ibp, (lpre, ibuf) in zip(ibs, ib_todo):
                        ibp.X[0], ibp.n_fields, ibp.ldx = ibuf.data_ptr(), 1, ibuf.stride(0)
                        self._fill_ib(ibp, lpre)
                    ib_todo.clear()
                rec = self._rec(L.sea_silu_outer_ib, [arr, len(chunk), None, M, self.code, ibs, n_ib], "adaln.silu" + tag, (arr, ibs))
                self._c_patches.append((rec.args, 2))
                self._cur.append(rec)
            self._gemm(gemm_groups, "adaln.cond_gemm" + tag)

        # Long launches (B = 8): AdaLN_0 of the first layer, its condition MLP and the self-attention's q / k / v as ONE launch too (sea_adaln_qkv without riders, emitted by
        # _build) — opt-in (SEA_PLAN=front_big=1): measured at B = 8 the launch takes 171 us against 67 (condition GEMM) + 31 (norm) + 60 (QKV) as tiled launches, the forward
        # 1.070 against 1.047 ms: with several rounds of workgroups the tiled GEMMs keep three workgroups per CU busy, the row-owning workgroup one.
        self._front_big = (type(self) is Plan and self.mode == "full" and _switches.plan("front_big", "0") == "1" and ops.adaln_qkv_supported(self.dt, E, self.H) and F <= N.MAX_AQKV_GROUPS
                           and not self.concat and M >= 1024)
        if self._front_big:
            first = []
        if not split:
            emit(first + rest, "")
            return mods
        if first:
            emit(first, ".first")
        self._fork(1)
        emit(rest, ".rest")
        self._end_lane()
        return mods

    def _gen_a(self, inst) -> bool:
        """AdaLN condition MLPs with the generated GEMM operand (no silu launch)?  Long launches only, see _cond_mods."""
        want = _switches.plan("silu", "auto")
        return type(self) is Plan and all(2 * d <= 1024 for _, d in inst) and (want == "1" or (want == "auto" and self.M >= 8192))

    # ------------------------------------------------------------------ the plan
    def _build(self) -> None:
        eng, P = self.eng, self.eng.params
        F, E, D, S, M, B, T, H = self.F, self.E, self.D, self.S, self.M, self.B, self.T, self.H
        dt, L = self.dt, N.lib()
        hd_s, hd_c = E // H, D // H
        cap = self.cap
        f32 = torch.float32

        # Optional lanes (parallel graph branches) for independent work — SEA_PLAN=lanes: "cond" = the condition MLPs the first launch does
        # not need, "all" = also every finished field's MLP beside the remaining exchange stages.  At one trajectory the
        # cross-branch dependencies of a captured HIP graph cost more than the overlap gains; from 8192 rows up the condition lane pays (1 %).
        xmode = eng.model.exchange_mode                                   # 'sea' | 'addition' | 'simple' (models/temporal.py:314-324)
        has_ib = eng.model.ib_addition_mode.lower() == "add"              # 'none': _add_info returns x (models/temporal.py:113-114)
        ib_attn = eng.model.ib_addition_mode.lower() == "attention"       # x_i += cross_attn_ib_i(x_i, ib rows) (models/temporal.py:117-118)
        mode = _switches.plan("lanes", "auto") if type(self) is Plan and xmode == "sea" and has_ib else "none"
        if mode == "auto":   # with the current 21-launch plan: cfg2 0.252 ms none / 0.284 cond / 0.358 all; B=8 1.166 none / 1.154 cond / 1.245 all
            mode = "cond" if self.M >= 8192 else "none"
        lanes = mode == "all" and F >= 2 and eng.model.add_info_after_cross
        split_cond = mode in ("cond", "all") and self.adaln
        # Linear + the row norm that follows it in one launch (sea_gemm_rownorm) where a tile can span the whole output row: cross_down + ln_cross,
        # the last layer's proj + the model's final norm.  SEA_PLAN=norm=0 keeps the two-launch form (A/B measurements).
        fuse_norm = self._fuse_norm = type(self) is Plan and _switches.plan("norm", "1") != "0"
        # the info-bottleneck add without a launch of its own: its MLP depends on the condition only, so it is EVALUATED by extra row passes of the silu
        # launch (into ibuf) and ADDED by the AdaLN_2 pass that follows it anyway (SeaNormGroup.addend).  Needs the silu launch (adaln, short launches)
        # and the add after the exchange; SEA_PLAN=fold_ib=0 keeps sea_ib_add.
        # The row-local chains between the attention launches as ONE launch each (sea_row_chain, round 4): self-attention out-projection + residual ->
        # cross_down + ln_cross -> every q of the field and the k / v of the pairs that read its PRE-exchange rows; per field, its exchange tail ->
        # cross_down + ln_cross of the updated rows -> the k / v of the pairs that read them.  No cross-attention QKV launch, no out-projection launch, no
        # down + norm launch: 18 launches -> 14 at cfg2.  bf16, the widths the kernel instantiates, at most 3 fields (the segments' weights share an LDS
        # half), short launches (a workgroup owns most of a CU's LDS: beyond a round or two of workgroups the tiled launches win; SEA_PLAN=chain_max_rows).
        # SEA_PLAN=chain=0 keeps the 18-launch plan (the reference form of tests/test_model_gpu.py::test_optional_plans_match_default_plan).
        chain = self._chain_plan = (type(self) is Plan and self.mode == "full" and xmode == "sea" and 1 < F <= 3 and fuse_norm and not lanes and not self.concat
                                    and _switches.plan("chain", "1") != "0" and _switches.plan("xtail", "1") != "0" and ops.row_chain_supported(dt, D, E, F - 1, D // H)
                                    and M <= int(_switches.plan("chain_max_rows", "4096")))
        # ... and with them RIDERS (sea_row_chain_riders): the AdaLN condition MLPs are functions of the condition alone, and only AdaLN_0 / ln_cross of the first
        # layer are needed in front of the first attention — the modules the field MLP and the final norm read (62 % of the condition GEMM's work at cfg2)
        # and the information-bottleneck rows are computed by extra workgroups of the chain launches, on the CUs those leave idle (127-192 workgroups on 256
        # CUs): the condition GEMM in front of the step covers 6 of the 12 modules.  One layer, AdaLN, the ib add behind the exchange.
        # SEA_PLAN=riders=0 keeps the whole-model condition launches.
        riders = self._riders = (chain and self.adaln and self.L == 1 and not split_cond and _switches.plan("riders", "1") != "0"
                                 and (not has_ib or (eng.model.add_info_after_cross and E <= 2048)) and not ib_attn and F + F <= N.CHAIN_MAX_RIDERS)
        # the condition GEMMs generate their operand (long launches): no silu launch for the ib rows to ride on.  SEA_PLAN=fold_ib_gen=1 gives them a launch of
        # their own (sea_silu_outer_ib without silu rows) and folds the add into the norm pass — measured at B = 8: 29 + 40 us against 30 (ib_add) + 30 (norm): not the default
        gen_all = self._gen_a([(None, E), (None, D)])
        fold_ib = (type(self) is Plan and has_ib and eng.model.add_info_after_cross and self.adaln and not lanes
                   and self.L <= N.MAX_SILU_IB and E <= 2048 and _switches.plan("fold_ib", "1") != "0"
                   and (riders or (gen_all and _switches.plan("fold_ib_gen", "0") != "0") or (not gen_all and not split_cond)))
        hoist_ib = self._cond_src is not None and has_ib and eng.model.add_info_after_cross and len(self._cond_src.ibufs) == self.L and E <= 2048
        if hoist_ib:       # the info-bottleneck rows of all steps exist already: added by the norm pass in front of the MLP (AdaLN or LayerNorm alike)
            fold_ib = True
            ibufs = [t[:M] for t in self._cond_src.ibufs]
            self._ib_fold = []
        else:
            ibufs = [self._buf(M, E, dtype=f32) for _ in range(self.L)] if fold_ib else None
            self._ib_fold = [(f"blocks.{l}.", ibufs[l]) for l in range(self.L)] if fold_ib else []
        mods = self._cond_mods(split=split_cond, riders=riders)
        cond_joined = not split_cond

        def norm_params(pre, d):
            if self.adaln:
                return dict(mod=mods[pre], gamma=P.f32_vec(pre + "weight"), beta=P.f32_vec(pre + "bias"))
            return dict(gamma=P.f32_vec(pre + "weight"))

        # the exchange tail of a field (projections + GELU, up-projection + residual, down-projection + norm) as ONE launch: bf16, the widths the
        # kernel instantiates, short launches (SEA_PLAN=xtail=0 keeps the three-launch form; SEA_PLAN=xtail_max_rows bounds M)
        fuse_xtail = (fuse_norm and not lanes and xmode == "sea" and F > 1 and _switches.plan("xtail", "1") != "0"
                      and ops.exchange_tail_supported(dt, D, E, F - 1) and M <= int(_switches.plan("xtail_max_rows", "1000000000")))   # measured: cfg2 0.281 -> 0.254 ms, B = 2 0.414 -> 0.404, B = 4 0.679 -> 0.675, B = 8 a tie (1.191)
        rope_s, rope_c = eng.rope_self, eng.rope_cross
        Eo, concat = self.Eo, self.concat
        FE = F * Eo                                             # row stride of the caller's [B, T, F, Eo] tensors
        # KV-cache step at the shipped widths (one row per trajectory and field, embed_dim 1024 / 2048): the Linear layers as sea_gemm_fewrows /
        # sea_qkv_rope_fewrows launches with the row norms in front of them folded in (gemv.hip) — 18 launches instead of 22.  SEA_KV=gemv=0 keeps the generic launches.
        few = self._few = (type(self) is Plan and self.mode == "step" and T == 1 and xmode == "sea" and not concat and not ib_attn
                           and _switches.kv("gemv", "1") != "0" and 2 * (F - 1) <= N.FEW_MAX_GROUPS and F <= N.FEW_MAX_GROUPS
                           and ops.fewrows_supported(dt, M, [E], qkv=True, pre=True) and ops.fewrows_supported(dt, M, [S])
                           and (F == 1 or ops.fewrows_supported(dt, M, [D], qkv=True, pre=True)))
        xr = [self._buf(M, E, dtype=f32) for _ in range(F)]     # fp32 residual stream
        xa = [self._buf(M, E) for _ in range(F)]                # act-dtype copy (GEMM A operand)
        n_e = [self._buf(M, E) for _ in range(F)]               # normalised rows, dim E
        att_e = [self._buf(M, E) for _ in range(F)]
        Qs = [self._buf(B, H, T, hd_s) for _ in range(F)]
        Ks = [[self._buf(B, H, cap, hd_s, zero=True) for _ in range(F)] for _ in range(self.L)]
        Vs = [[self._buf(B, H, hd_s, cap, zero=True) for _ in range(F)] for _ in range(self.L)]
        dn = [self._buf(M, D, dtype=f32) for _ in range(F)]
        if xmode == "sea":   # cross-attention workspaces
            nd_old = [self._buf(M, D) for _ in range(F)]
            nd_new = [self._buf(M, D) for _ in range(F)]
            Qc = [self._buf(B, H, T, hd_c) for _ in range(max(F - 1, 1))]
            Qc2 = [[self._buf(B, H, T, hd_c) for _ in range(F - 1)] for _ in range(F)] if chain else None   # every field's q rows exist at once
            Kc = [[[self._buf(B, H, cap, hd_c, zero=True) for _ in range(F)] for _ in range(F)] for _ in range(self.L)]
            Vc = [[[self._buf(B, H, hd_c, cap, zero=True) for _ in range(F)] for _ in range(F)] for _ in range(self.L)]
            att_c = [self._buf(M, D) for _ in range(max(F - 1, 1))]
            gp = self._buf(max(F - 1, 1), M, D)
        # hidden rows of the MLP: S is a power of two (4 KiB rows at cfg2) — a 32-row workgroup's stores, and the next launch's 32-row operand tiles, would sit
        # at one 4 KiB stride and crowd a few memory channels; 128 B of padding per row spreads them (fc1 + LN + GELU 27.0 -> 25.4 us stand-alone)
        pad = 64 if (type(self) is Plan and S % 1024 == 0) else 0
        hbuf = [self._buf(M, S + pad)[:, :S] for _ in range(F)]
        hg = [self._buf(M, S + pad)[:, :S] for _ in range(F)]
        self.ws = dict(xr=xr, xa=xa, n_e=n_e, att_e=att_e, hbuf=hbuf, hg=hg)

        xm = [self._buf(M, E) for _ in range(F)] if lanes else xa
        first = True  # the residual stream still lives in the caller's x [B,T,F,E]
        for l in range(self.L):
            pre = f"blocks.{l}."
            last = l == self.L - 1
            if first and not eng.model.add_info_after_cross and (has_ib or ib_attn or concat):
                # the info-bottleneck add comes first and must not modify the caller's tensor: copy x into xr
                for i in range(F):
                    rec = _Rec(L.sea_convert_f32_to_act, [None, FE, xr[i].data_ptr(), E, M, Eo, N.SEA_F32], "x.copy")
                    self._x_patches.append((rec.args, 0, i * Eo * 4))
                    self.records.append(rec)
                first = False
            if concat:
                # x_i := [x_i | ib rows] (models/temporal.py:115-116): columns Eo .. E-1 of the residual rows are zeroed, then the usual info-bottleneck add
                # runs on them (columns 0 .. Eo-1 hold the caller's rows / the previous block's proj output)
                if self._zero_ib is None:
                    self._zero_ib = self._buf(M, self.ib_dim, dtype=f32, zero=True)
                for i in range(F):
                    self._cur.append(self._rec(L.sea_convert_f32_to_act, [self._zero_ib.data_ptr(), self.ib_dim, xr[i][:, Eo:].data_ptr(), E, M, self.ib_dim, N.SEA_F32],
                                               "ib.concat.zero"))
                self._ib(pre, [xr[i][:, Eo:] for i in range(F)])
            if not eng.model.add_info_after_cross and has_ib:
                self._ib(pre, xr)
            if not eng.model.add_info_after_cross and ib_attn:
                self._ib_attn(pre, xr)
            # -- self attention: x_i += proj(attn(AdaLN_0(x_i)))
            groups = []
            one_launch_front = l == 0 and not few and (getattr(self, "_front_big", False) or (getattr(self, "_adaln_front", None) and getattr(self, "_front_chain", False)))
            for i in range(F if not one_launch_front else 0):   # (the one-launch front reads the module's parameters itself: its modulation matrix does not exist)
                g = dict(**norm_params(f"{pre}ln.exp.{i}.0.", E)) if few else dict(Yact=n_e[i], **norm_params(f"{pre}ln.exp.{i}.0.", E))
                if first:
                    g.update(X=xr[i], ldx=FE, X_is_x=i * Eo * 4)
                else:
                    g.update(X=xr[i])
                groups.append(g)
            if few:   # AdaLN_0 / LayerNorm as the prologue of the projection
                self._qkv_few([dict(W=P.act(f"{pre}attn.self.{i}.q.weight", 3 * E), bias=P.f32_vec(f"{pre}attn.self.{i}.q.bias", 3 * E),
                                    col0=0, Q=Qs[i], K=Ks[l][i], Vt=Vs[l][i]) for i in range(F)], rope_s, hd_s, "self.qkv_rope", pre=groups)
            else:
                af = getattr(self, "_adaln_front", None) if l == 0 else None
                if l == 0 and not af and getattr(self, "_front_big", False) and not few:
                    arr = (N.SeaAdalnQkv * F)()
                    for g_, i in zip(arr, range(F)):
                        mp = f"{pre}ln.exp.{i}.0."
                        ops.fill_adaln_qkv(g_, X=xr[i], cond=None, w1=P.f32_vec(mp + "cond_mlp.0.weight", 2 * E), b1=P.f32_vec(mp + "cond_mlp.0.bias"), W2c=P.act(mp + "cond_mlp.2.weight"),
                                           b2c=P.f32_vec(mp + "cond_mlp.2.bias"), gamma=P.f32_vec(mp + "weight"), beta=P.f32_vec(mp + "bias"), Wqkv=P.act(f"{pre}attn.self.{i}.q.weight", 3 * E),
                                           bqkv=P.f32_vec(f"{pre}attn.self.{i}.q.bias", 3 * E), Q=Qs[i], K=Ks[l][i], Vt=Vs[l][i], ldx=(FE if first else None))
                        g_.M = M
                        self._c_patches.append((g_, "cond"))
                        if first:
                            self._x_patches.append((g_, "X", i * Eo * 4))
                    common = N.SeaQkvCommon(rope_s.data_ptr(), self.H, hd_s, self.T, self.pos0, self.cap, ops.q_scale(hd_s))
                    self._pos_structs.append(common)
                    self._cur.append(self._rec(L.sea_adaln_qkv, [arr, F, C.byref(common), None, 0, None, 0, None, 0, None, 1e-5, self.code], "self.cond_adaln0_qkv_rope", (arr, common, None, None, None)))
                elif af and getattr(self, "_front_chain", False):
                    front3 = getattr(self, "_front3", False)
                    arr = (N.SeaAdalnQkv * F)()
                    for g_, i in zip(arr, range(F)):
                        mp = f"{pre}ln.exp.{i}.0."
                        ops.fill_adaln_qkv(g_, X=xr[i], cond=None, w1=P.f32_vec(mp + "cond_mlp.0.weight", 2 * E), b1=P.f32_vec(mp + "cond_mlp.0.bias"), W2c=af[mp][1], b2c=af[mp][2],
                                           gamma=P.f32_vec(mp + "weight"), beta=P.f32_vec(mp + "bias"), Wqkv=P.act(f"{pre}attn.self.{i}.q.weight", 3 * E),
                                           bqkv=P.f32_vec(f"{pre}attn.self.{i}.q.bias", 3 * E), Q=Qs[i], K=Ks[l][i], Vt=Vs[l][i], ldx=(FE if first else None))
                        g_.M = M
                        self._c_patches.append((g_, "cond"))
                        if first:
                            self._x_patches.append((g_, "X", i * Eo * 4))
                        if front3:   # ln_cross_i's modulation as the launch's third layer
                            lc = f"{pre}ln_cross.{i}."
                            W3 = af[lc][1]
                            g_.w13, g_.b13 = P.f32_vec(lc + "cond_mlp.0.weight", 2 * D).data_ptr(), P.f32_vec(lc + "cond_mlp.0.bias").data_ptr()
                            g_.W3, g_.ldw3, g_.b3, g_.N3 = W3.data_ptr(), W3.stride(0), N.ptr(af[lc][2]), 2 * D
                            g_.mod3, g_.ldmod3 = mods[lc].data_ptr(), mods[lc].stride(0)
                    rg = [] if front3 else [(key, v) for key, v in af.items() if "ln_cross." in key]
                    rarr = (N.SeaGemmGroup * max(len(rg), 1))()
                    for g_, (key, (hid_, W_, b_)) in zip(rarr, rg):
                        _fill_gemm(g_, A=hid_, W=W_, bias=b_, Cact=mods[key])
                    common = N.SeaQkvCommon(rope_s.data_ptr(), self.H, hd_s, self.T, self.pos0, self.cap, ops.q_scale(hd_s))
                    self._pos_structs.append(common)
                    sarr, n_s, ibp = self._front_rows if front3 else (None, 0, None)
                    rec = self._rec(L.sea_adaln_qkv, [arr, F, C.byref(common), (rarr if rg else None), len(rg), (sarr if n_s else None), n_s, None, (M if (n_s or ibp is not None) else 0),
                                                      (C.byref(ibp) if ibp is not None else None), 1e-5, self.code], "self.cond_adaln0_qkv_rope", (arr, common, rarr, sarr, ibp))
                    if n_s or ibp is not None:
                        self._c_patches.append((rec.args, 7))
                    self._cur.append(rec)
                elif af:
                    ag = []
                    for i in range(F):
                        hid_, W_, b_ = af[f"{pre}ln.exp.{i}.0."]
                        g = dict(A=hid_, W=W_, bias=b_, X=xr[i], gamma=P.f32_vec(f"{pre}ln.exp.{i}.0.weight"), beta=P.f32_vec(f"{pre}ln.exp.{i}.0.bias"), Yact=n_e[i])
                        if first:
                            g.update(ldx=FE, X_is_x=i * Eo * 4)
                        ag.append(g)
                    for key, (hid_, W_, b_) in af.items():
                        if "ln_cross." in key:
                            ag.append(dict(A=hid_, W=W_, bias=b_, Yact=mods[key]))
                    self._adaln(ag, "self.cond_adaln0")
                else:
                    self._norm(groups, E, "self.adaln0")
                if not ((af and getattr(self, "_front_chain", False)) or (l == 0 and not af and getattr(self, "_front_big", False) and not few)):
                    self._qkv([dict(A=n_e[i], W=P.act(f"{pre}attn.self.{i}.q.weight", 3 * E), bias=P.f32_vec(f"{pre}attn.self.{i}.q.bias", 3 * E),
                                    col0=0, Q=Qs[i], K=Ks[l][i], Vt=Vs[l][i]) for i in range(F)], rope_s, hd_s, "self.qkv_rope")
            self._attn([dict(Q=Qs[i], K=Ks[l][i], Vt=Vs[l][i], O=att_e[i]) for i in range(F)], hd_s, E, "self.attention")
            groups = []
            for i in range(F):
                g = dict(A=att_e[i], W=P.act(f"{pre}attn.self.{i}.projection.weight"), C32=xr[i], Cact=xa[i])
                if first:
                    g.update(R=xr[i], ldr=FE, R_is_x=i * Eo * 4)
                else:
                    g.update(R=xr[i])
                groups.append(g)
            if chain:
                self._rider_hosts_left = self._rider_hosts_total = F   # (SEA_PLAN=rider_caps overrides) chain launches of this layer that may carry riders: this one and the tails of the fields that are not last (their results are read by the MLP)
                # out-projection + residual, cross_down + ln_cross of the PRE-exchange rows, and from those normalised rows (never stored): q_ij for every
                # j != i, and k / v of the pairs (a, i), a < i — field a runs its cross-attention before field i is updated (models/temporal.py:187-192)
                groups = []
                for i in range(F):
                    proj = []
                    for s_, j in enumerate([j for j in range(F) if j != i]):
                        ca = f"{pre}cross_attn.{i}.{j}."
                        proj.append(dict(W=P.act(ca + "q.weight"), bias=P.f32_vec(ca + "q.bias"), col0=0, Q=Qc2[i][s_]))
                    for a in range(i):
                        ca = f"{pre}cross_attn.{a}.{i}."
                        proj.append(dict(W=P.act(ca + "k.weight", 2 * D), bias=P.f32_vec(ca + "k.bias", 2 * D), col0=D, K=Kc[l][a][i], Vt=Vc[l][a][i]))
                    g = dict(a2=att_e[i], W2=P.act(f"{pre}attn.self.{i}.projection.weight"), Xin=xr[i], X=xr[i], proj=proj,
                             down=dict(W=P.act(f"{pre}cross_down.{i}.weight"), bias=P.f32_vec(f"{pre}cross_down.{i}.bias"), **norm_params(f"{pre}ln_cross.{i}.", D)))
                    if first:
                        g.update(ldxin=FE, Xin_is_x=i * Eo * 4)
                    groups.append(g)
                self._chain(groups, rope_c, hd_c, "self.out_proj_down_qkv")
            else:
                (self._gemm_few if few else self._gemm)(groups, "self.out_proj")
            first = False
            if not cond_joined:  # everything below reads modulations computed on lane 1
                self._join(1)
                cond_joined = True
            if xmode == "addition":
                # -- 'addition' exchange (models/temporal.py:291-301; Jacobi: every field read at its pre-exchange value): down-projection + norm of all
                # fields in one launch, s = GELU(sum_j n_j) as an identity-weight GEMM over F segments, x_i += cross_up_i(s) in one grouped launch
                ndall = self._buf(F, M, D)
                if fuse_norm and D <= 256 and D % 16 == 0:
                    self._gemm_norm([dict(A=xa[j], W=P.act(f"{pre}cross_down.{j}.weight"), bias=P.f32_vec(f"{pre}cross_down.{j}.bias"), Yact=ndall[j],
                                          **norm_params(f"{pre}ln_cross.{j}.", D)) for j in range(F)], "add.down_norm")
                else:
                    self._gemm([dict(A=xa[j], W=P.act(f"{pre}cross_down.{j}.weight"), bias=P.f32_vec(f"{pre}cross_down.{j}.bias"), C32=dn[j])
                                for j in range(F)], "add.down")
                    self._norm([dict(X=dn[j], Yact=ndall[j], **norm_params(f"{pre}ln_cross.{j}.", D)) for j in range(F)], D, "add.norm")
                sg = self._buf(M, D)
                self._gemm([dict(A=ndall[0], W=eng.eye(D), n_seg=F, a_seg_stride=M * D, Cact=sg, act=1)], "add.sum_gelu")
                self._gemm([dict(A=sg, W=P.act(f"{pre}cross_up.{i}.weight"), bias=P.f32_vec(f"{pre}cross_up.{i}.bias"), R=xr[i], C32=xr[i])
                            for i in range(F)], "add.up")
            if xmode == "pool":
                # -- 'pool' exchange (models/temporal.py:255-277, pool_update_method 'mlp'; Jacobi): n_j = ln_cross_j(cross_down_j(x_j)) + pe; pool =
                # MLP(cat_j n_j); a_i = cross_attn_i(n_i, pool); x_i += cross_up_i(GELU(n_i + a_i)).  `big` holds [n_0 .. n_{F-1} | a_0 .. a_{F-1}] side by
                # side, so that cat(n) is a view and (n_i, a_i) are two segments of one identity-weight GEMM
                FD = F * D
                big = self._buf(M, 2 * FD)
                nrm = [self._buf(M, D) for _ in range(F)]
                pe_t = self._buf(M, D, dtype=f32)
                pe_t.copy_(blk_pe(eng, l)[:T].repeat(B, 1))
                if fuse_norm and D <= 256 and D % 16 == 0:
                    self._gemm_norm([dict(A=xa[j], W=P.act(f"{pre}cross_down.{j}.weight"), bias=P.f32_vec(f"{pre}cross_down.{j}.bias"), Yact=nrm[j],
                                          **norm_params(f"{pre}ln_cross.{j}.", D)) for j in range(F)], "pool.down_norm")
                else:
                    self._gemm([dict(A=xa[j], W=P.act(f"{pre}cross_down.{j}.weight"), bias=P.f32_vec(f"{pre}cross_down.{j}.bias"), C32=dn[j])
                                for j in range(F)], "pool.down")
                    self._norm([dict(X=dn[j], Yact=nrm[j], **norm_params(f"{pre}ln_cross.{j}.", D)) for j in range(F)], D, "pool.norm")
                self._gemm([dict(A=nrm[j], W=eng.eye(D), R=pe_t, Cact=big[:, j * D:(j + 1) * D]) for j in range(F)], "pool.pe_add")
                hp, pool = self._buf(M, 2 * D), self._buf(M, D)
                self._gemm([dict(A=big[:, :FD], W=P.act(f"{pre}pool_update.0.weight"), bias=P.f32_vec(f"{pre}pool_update.0.bias"), Cact=hp, act=1)], "pool.update0")
                self._gemm([dict(A=hp, W=P.act(f"{pre}pool_update.2.weight"), bias=P.f32_vec(f"{pre}pool_update.2.bias"), Cact=pool)], "pool.update2")
                Qp = [self._buf(B, H, T, hd_c) for _ in range(F)]
                Kp = [self._buf(B, H, cap, hd_c, zero=True) for _ in range(F)]
                Vp = [self._buf(B, H, hd_c, cap, zero=True) for _ in range(F)]
                att_p = [self._buf(M, D) for _ in range(F)]
                qg = []
                for i in range(F):
                    ca = f"{pre}cross_attn.{i}."
                    qg.append(dict(A=big[:, i * D:(i + 1) * D], W=P.act(ca + "q.weight"), bias=P.f32_vec(ca + "q.bias"), col0=0, Q=Qp[i]))
                    qg.append(dict(A=pool, W=P.act(ca + "k.weight", 2 * D), bias=P.f32_vec(ca + "k.bias", 2 * D), col0=D, K=Kp[i], Vt=Vp[i]))
                self._qkv(qg, rope_c, hd_c, "pool.qkv_rope")
                self._attn([dict(Q=Qp[i], K=Kp[i], Vt=Vp[i], O=att_p[i]) for i in range(F)], hd_c, D, "pool.attention")
                self._gemm([dict(A=att_p[i], W=P.act(f"{pre}cross_attn.{i}.projection.weight"), Cact=big[:, FD + i * D:FD + (i + 1) * D]) for i in range(F)], "pool.proj")
                sg = [self._buf(M, D) for _ in range(F)]
                self._gemm([dict(A=big[:, i * D:(i + 1) * D], W=eng.eye(D), n_seg=2, a_seg_stride=FD, Cact=sg[i], act=1) for i in range(F)], "pool.sum_gelu")
                self._gemm([dict(A=sg[i], W=P.act(f"{pre}cross_up.{i}.weight"), bias=P.f32_vec(f"{pre}cross_up.{i}.bias"), R=xr[i], C32=xr[i])
                            for i in range(F)], "pool.up")
            # -- state exchange (Gauss-Seidel over i, models/temporal.py:187-192)
            if F > 1 and xmode == "sea" and few:
                # the same Gauss-Seidel sweep with ln_cross evaluated by its consumers: dn[j] holds cross_down_j(x_j) of the CURRENT x_j (rewritten by down_new once
                # field j has been updated), every q / k,v projection normalises the rows it reads itself
                self._gemm_few([dict(A=xa[j], W=P.act(f"{pre}cross_down.{j}.weight"), bias=P.f32_vec(f"{pre}cross_down.{j}.bias"), C32=dn[j]) for j in range(F)], "cross.down_old")
                for i in range(F):
                    others = [j for j in range(F) if j != i]
                    qkv_groups, qkv_pre, probs, proj_groups = [], [], [], []
                    for s, j in enumerate(others):
                        ca = f"{pre}cross_attn.{i}.{j}."
                        qkv_groups.append(dict(W=P.act(ca + "q.weight"), bias=P.f32_vec(ca + "q.bias"), col0=0, Q=Qc[s]))
                        qkv_pre.append(dict(X=dn[i], **norm_params(f"{pre}ln_cross.{i}.", D)))
                        qkv_groups.append(dict(W=P.act(ca + "k.weight", 2 * D), bias=P.f32_vec(ca + "k.bias", 2 * D), col0=D, K=Kc[l][i][j], Vt=Vc[l][i][j]))
                        qkv_pre.append(dict(X=dn[j], **norm_params(f"{pre}ln_cross.{j}.", D)))
                        probs.append(dict(Q=Qc[s], K=Kc[l][i][j], Vt=Vc[l][i][j], O=att_c[s]))
                        proj_groups.append(dict(A=att_c[s], W=P.act(ca + "projection.weight"), Cact=gp[s], act=1))
                    self._qkv_few(qkv_groups, rope_c, hd_c, f"cross{i}.qkv_rope", pre=qkv_pre)
                    self._attn(probs, hd_c, D, f"cross{i}.attention")
                    self._gemm_few(proj_groups, f"cross{i}.proj_gelu")
                    up = dict(A=gp[0], W=P.act(f"{pre}cross_up.{i}.weight"), bias=P.f32_vec(f"{pre}cross_up.{i}.bias"),
                              bias_scale=float(F - 1), n_seg=F - 1, a_seg_stride=M * D, R=xr[i], C32=xr[i], Cact=(xa[i] if i < F - 1 else None))
                    (self._gemm_few if F == 2 else self._gemm)([up], f"cross{i}.up_sum")   # (F > 2: the sum over the F - 1 projections needs sea_gemm_grouped's segments)
                    if i < F - 1:
                        self._gemm_few([dict(A=xa[i], W=P.act(f"{pre}cross_down.{i}.weight"), bias=P.f32_vec(f"{pre}cross_down.{i}.bias"), C32=dn[i])], f"cross{i}.down_new")
            elif F > 1 and xmode == "sea" and chain:
                for i in range(F):
                    others = [j for j in range(F) if j != i]
                    self._attn([dict(Q=Qc2[i][s_], K=Kc[l][i][j], Vt=Vc[l][i][j], O=att_c[s_]) for s_, j in enumerate(others)], hd_c, D, f"cross{i}.attention")
                    # the field's tail; from its UPDATED rows: cross_down + ln_cross, then k / v of the pairs (b, i), b > i (fields that run after it)
                    proj = []
                    for b_ in range(i + 1, F):
                        ca = f"{pre}cross_attn.{b_}.{i}."
                        proj.append(dict(W=P.act(ca + "k.weight", 2 * D), bias=P.f32_vec(ca + "k.bias", 2 * D), col0=D, K=Kc[l][b_][i], Vt=Vc[l][b_][i]))
                    down = None
                    if i < F - 1:
                        down = dict(W=P.act(f"{pre}cross_down.{i}.weight"), bias=P.f32_vec(f"{pre}cross_down.{i}.bias"), **norm_params(f"{pre}ln_cross.{i}.", D))
                    self._chain([dict(att=[att_c[s_] for s_ in range(len(others))], Wp=[P.act(f"{pre}cross_attn.{i}.{j}.projection.weight") for j in others],
                                      W2=P.act(f"{pre}cross_up.{i}.weight"), b2=P.f32_vec(f"{pre}cross_up.{i}.bias"), bias_scale=float(F - 1), Xin=xr[i], X=xr[i],
                                      down=down, proj=proj)], rope_c, hd_c, f"cross{i}.tail")
            elif F > 1 and xmode == "sea":
                if fuse_norm and D <= 256 and D % 16 == 0:
                    self._gemm_norm([dict(A=xa[j], W=P.act(f"{pre}cross_down.{j}.weight"), bias=P.f32_vec(f"{pre}cross_down.{j}.bias"), Yact=nd_old[j],
                                          **norm_params(f"{pre}ln_cross.{j}.", D)) for j in range(F)], "cross.down_norm_old")
                else:
                    self._gemm([dict(A=xa[j], W=P.act(f"{pre}cross_down.{j}.weight"), bias=P.f32_vec(f"{pre}cross_down.{j}.bias"), C32=dn[j])
                                for j in range(F)], "cross.down_old")
                    self._norm([dict(X=dn[j], Yact=nd_old[j], **norm_params(f"{pre}ln_cross.{j}.", D)) for j in range(F)], D, "cross.norm_old")
                for i in range(F):
                    others = [j for j in range(F) if j != i]
                    qkv_groups, probs, proj_groups = [], [], []
                    for s, j in enumerate(others):
                        src = nd_new[j] if j < i else nd_old[j]
                        ca = f"{pre}cross_attn.{i}.{j}."
                        qkv_groups.append(dict(A=nd_old[i], W=P.act(ca + "q.weight"), bias=P.f32_vec(ca + "q.bias"), col0=0, Q=Qc[s]))
                        qkv_groups.append(dict(A=src, W=P.act(ca + "k.weight", 2 * D), bias=P.f32_vec(ca + "k.bias", 2 * D), col0=D,
                                               K=Kc[l][i][j], Vt=Vc[l][i][j]))
                        probs.append(dict(Q=Qc[s], K=Kc[l][i][j], Vt=Vc[l][i][j], O=att_c[s]))
                        proj_groups.append(dict(A=att_c[s], W=P.act(ca + "projection.weight"), Cact=gp[s], act=1))
                    self._qkv(qkv_groups, rope_c, hd_c, f"cross{i}.qkv_rope")
                    self._attn(probs, hd_c, D, f"cross{i}.attention")
                    if fuse_xtail:
                        # everything between this field's cross-attention and the next field's: projections + GELU, up-projection of the sum + residual,
                        # down-projection + ln_cross of the updated field, in one launch (a workgroup carries 16 rows through the three layers)
                        down = None
                        if i < F - 1:
                            down = dict(W=P.act(f"{pre}cross_down.{i}.weight"), bias=P.f32_vec(f"{pre}cross_down.{i}.bias"), Yact=nd_new[i],
                                        **norm_params(f"{pre}ln_cross.{i}.", D))
                        self._xtail([att_c[s] for s in range(len(others))], [P.act(f"{pre}cross_attn.{i}.{j}.projection.weight") for j in others],
                                    P.act(f"{pre}cross_up.{i}.weight"), P.f32_vec(f"{pre}cross_up.{i}.bias"), float(F - 1), xr[i], down, f"cross{i}.tail")
                        continue
                    self._gemm(proj_groups, f"cross{i}.proj_gelu")
                    up = dict(A=gp[0], W=P.act(f"{pre}cross_up.{i}.weight"), bias=P.f32_vec(f"{pre}cross_up.{i}.bias"),
                              bias_scale=float(F - 1), n_seg=F - 1, a_seg_stride=M * D, R=xr[i], C32=xr[i],
                              Cact=(xa[i] if i < F - 1 else None))
                    self._gemm([up], f"cross{i}.up_sum")
                    if lanes:
                        # x_i is final for the exchange: its info-bottleneck add, MLP, proj (and final norm) run on their own lane beside the
                        # remaining Gauss-Seidel stages; the last field stays on the main stream.  fc2 writes xm, not xa: the main stream
                        # still reads xa[i] (down_new) while the lane runs.
                        if i < F - 1:
                            self._fork(2 + i)
                        self._ib(pre, [xr[i]])
                        self._mlp_proj(pre, [i], xr, xm, n_e, hbuf, hg, mods, last, tag=f".f{i}")
                        if i < F - 1:
                            self._end_lane()
                    if i < F - 1 and fuse_norm and D <= 256 and D % 16 == 0:
                        self._gemm_norm([dict(A=xa[i], W=P.act(f"{pre}cross_down.{i}.weight"), bias=P.f32_vec(f"{pre}cross_down.{i}.bias"), Yact=nd_new[i],
                                              **norm_params(f"{pre}ln_cross.{i}.", D))], f"cross{i}.down_norm_new")
                    elif i < F - 1:
                        self._gemm([dict(A=xa[i], W=P.act(f"{pre}cross_down.{i}.weight"), bias=P.f32_vec(f"{pre}cross_down.{i}.bias"), C32=dn[i])],
                                   f"cross{i}.down_new")
                        self._norm([dict(X=dn[i], Yact=nd_new[i], **norm_params(f"{pre}ln_cross.{i}.", D))], D, f"cross{i}.norm_new")
            if lanes:
                for i in range(F - 1):
                    self._join(2 + i)
                continue
            if eng.model.add_info_after_cross and has_ib and not fold_ib:
                self._ib(pre, xr)
            if eng.model.add_info_after_cross and ib_attn:
                self._ib_attn(pre, xr)
            self._mlp_proj(pre, list(range(F)), xr, xa, n_e, hbuf, hg, mods, last, addend=(ibufs[l] if fold_ib else None), few=few)

    def _mlp_proj(self, pre, fields, xr, xm, n_e, hbuf, hg, mods, final_norm, tag="", addend=None, few=False) -> bool:
        """x_i += W2 gelu(LN(W1 AdaLN_2(x_i))) ; x_i = proj_i(x_i) for the listed fields (models/temporal.py:143-146), optionally followed
        by the model's final per-field norm written straight into out (models/temporal.py:412-415)."""
        P, E, S, Eo, FE = self.eng.params, self.E, self.S, self.Eo, self.F * self.Eo
        xo = xr if Eo == E else [t[:, :Eo] for t in xr]   # the proj output / the next block's input rows ('concat': the first Eo columns of the widened rows)

        def norm_params(p_, d):
            if self.adaln:
                return dict(mod=mods[p_], gamma=P.f32_vec(p_ + "weight"), beta=P.f32_vec(p_ + "bias"))
            return dict(gamma=P.f32_vec(p_ + "weight"))

        if few:
            # KV-cache step at the shipped widths: [ib add + AdaLN_2 / LayerNorm] fc1 | [LayerNorm + GELU] fc2 + residual | proj | final norm.
            # x + ib goes to rows of its own (xq): the prologue runs in every workgroup of fc1, so it must not update the rows it reads
            xq = [self._buf(self.M, E, dtype=torch.float32) for _ in fields] if addend is not None else None
            res = xq if addend is not None else [xr[i] for i in fields]
            pre_n = [dict(X=xr[i], **(dict(addend=addend, Xout=xq[k]) if addend is not None else {}), **norm_params(f"{pre}ln.exp.{i}.2.", E)) for k, i in enumerate(fields)]
            self._gemm_few([dict(W=P.act(f"{pre}mlp.{i}.layers.0.weight"), bias=P.f32_vec(f"{pre}mlp.{i}.layers.0.bias"), Cact=hbuf[i]) for i in fields], "mlp.fc1" + tag, pre=pre_n)
            ln_h = [dict(X=hbuf[i], gamma=P.f32_vec(f"{pre}mlp.{i}.layers.1.weight"), beta=P.f32_vec(f"{pre}mlp.{i}.layers.1.bias")) for i in fields]
            fold_ln = S <= 8192   # LayerNorm + GELU of the hidden rows as the prologue of fc2 while re-reading the row and its gains / shifts per workgroup stays below the weight stream
            if not fold_ln:
                self._norm([dict(Yact=hg[i], **g_) for i, g_ in zip(fields, ln_h)], S, "mlp.ln_gelu" + tag, x_is_act=True, gelu=True)
            self._gemm_few([dict(A=(None if fold_ln else hg[i]), W=P.act(f"{pre}mlp.{i}.layers.3.weight"), bias=P.f32_vec(f"{pre}mlp.{i}.layers.3.bias"), R=res[k], Cact=xm[i])
                            for k, i in enumerate(fields)], "mlp.fc2" + tag, pre=(ln_h if fold_ln else None), pre_act=fold_ln, pre_gelu=fold_ln)
            self._gemm_few([dict(A=xm[i], W=P.act(f"{pre}proj.{i}.weight"), bias=P.f32_vec(f"{pre}proj.{i}.bias"), C32=xo[i]) for i in fields], "proj" + tag)
            if final_norm:
                self._norm([dict(X=xo[i], Y32=xo[i], ldy32=FE, Y_is_out=i * Eo * 4, **norm_params(f"ln.{i}.", Eo)) for i in fields], Eo, "final.norm" + tag)
            return final_norm
        # Linear + nn.LayerNorm + GELU in one launch where the kernel is instantiated (bf16; SEA_PLAN=mlp1=0 keeps the two launches, =1 forces
        # the one launch).  Every workgroup of that kernel streams the whole of W1, so it needs enough 32-row tiles to pay: with the few rows
        # of a KV-cache step the two launches are faster (0.143 vs 0.163 ms per step at cfg2), hence the row threshold.
        want = _switches.plan("mlp1", "auto")
        fused = (type(self) is Plan and want != "0" and (want == "1" or self.M >= 1024) and ops.mlp_fc1_supported(self.dt, E, S) and len(fields) <= N.MAX_MLP_GROUPS)
        # ... and the row pass in front of it (info-bottleneck add + AdaLN_2 / LayerNorm) as that launch's prologue: a workgroup owns its 32 rows from the
        # fp32 residual stream to the activated hidden rows.  Measured (graph replay): cfg2 0.2472 -> 0.2440 ms (the launch itself 27.4 -> 31.2-32.2 us: its
        # loads sit in front of the weight stream; the row pass it replaces is 7.8 us), B = 8 1.127 -> 1.138 ms (142 -> 171 us per launch against a 32 us row
        # pass that runs at HBM speed) — used for short launches only.  SEA_PLAN=mlpnorm=0 / 1 forces.
        wn = _switches.plan("mlpnorm", "auto")
        norm_in = fused and (wn == "1" or (wn == "auto" and self.M <= 4096))
        extra = (lambda i: dict(addend=addend, Xout=xr[i])) if addend is not None else (lambda i: {})   # x_i += ib rides in this pass
        # The two fused launches as ONE (sea_mlp_block: the activated hidden rows stay in the owning workgroup's registers; no hg matrix, one launch boundary less, the
        # x + ib rows are not written back — the block's residual is formed from x and ib again).  Short launches, where both halves are fused.  SEA_PLAN=mlpblock=0 keeps two launches.
        # Long launches too (B = 8, M = 16192: 262 us against ib_add 30 + AdaLN_2 30 + fc1 + LN + GELU 160 + fc2 88 + proj + norm 55), there always with the norm
        # prologue: the row pass it replaces and the hidden rows it keeps to itself are 0.6 GB of traffic.
        w2 = _switches.plan("mlp2", "auto")
        two = (type(self) is Plan and (w2 == "1" or (w2 == "auto" and 1024 <= self.M)) and ops.mlp_fc1_supported(self.dt, E, S) and Eo == E and len(fields) <= N.MAX_MLP_GROUPS)
        if fused and two and _switches.plan("mlpblock", "1") != "0":
            norm_in = norm_in or (self.M > 4096 and wn != "0")
            a1, a2 = (N.SeaMlpGroup * len(fields))(), (N.SeaMlp2Group * len(fields))()
            if not norm_in:
                self._norm([dict(X=xr[i], Yact=n_e[i], **extra(i), **norm_params(f"{pre}ln.exp.{i}.2.", E)) for i in fields], E, ("mlp.ib_adaln2" if addend is not None else "mlp.adaln2") + tag)
            for g1, g2, i in zip(a1, a2, fields):
                nrm = dict(X32=xr[i], **(dict(addend=addend) if addend is not None else {}), **norm_params(f"{pre}ln.exp.{i}.2.", E)) if norm_in else None
                ops.fill_mlp_group(g1, None if norm_in else n_e[i], P.act(f"{pre}mlp.{i}.layers.0.weight"), P.f32_vec(f"{pre}mlp.{i}.layers.0.bias"),
                                   P.f32_vec(f"{pre}mlp.{i}.layers.1.weight"), P.f32_vec(f"{pre}mlp.{i}.layers.1.bias"), None, nrm)
                fin = norm_params(f"ln.{i}.", E) if final_norm else {}
                ops.fill_mlp2_group(g2, None, P.act(f"{pre}mlp.{i}.layers.3.weight"), P.f32_vec(f"{pre}mlp.{i}.layers.3.bias"), (None if norm_in else xr[i]),
                                    P.act(f"{pre}proj.{i}.weight"), P.f32_vec(f"{pre}proj.{i}.bias"), Y32=xr[i], ldy32=(FE if final_norm else None), M=g1.M, **fin)
                if final_norm:
                    self._out_patches.append((g2, "Y32", i * E * 4))
            self._cur.append(self._rec(N.lib().sea_mlp_block, [a1, a2, len(fields), 1e-5, self.code], ("mlp.block_norm" if final_norm else "mlp.block") + tag, (a1, a2)))
            return final_norm
        if not norm_in:
            self._norm([dict(X=xr[i], Yact=n_e[i], **extra(i), **norm_params(f"{pre}ln.exp.{i}.2.", E)) for i in fields], E, ("mlp.ib_adaln2" if addend is not None else "mlp.adaln2") + tag)
        if fused:
            arr = (N.SeaMlpGroup * len(fields))()
            for g_, i in zip(arr, fields):
                nrm = dict(X32=xr[i], **extra(i), **norm_params(f"{pre}ln.exp.{i}.2.", E)) if norm_in else None
                ops.fill_mlp_group(g_, None if norm_in else n_e[i], P.act(f"{pre}mlp.{i}.layers.0.weight"), P.f32_vec(f"{pre}mlp.{i}.layers.0.bias"),
                                   P.f32_vec(f"{pre}mlp.{i}.layers.1.weight"), P.f32_vec(f"{pre}mlp.{i}.layers.1.bias"), hg[i], nrm)
            self._cur.append(self._rec(N.lib().sea_mlp_fc1_ln_gelu, [arr, len(fields), 1e-5, self.code], "mlp.fc1_ln_gelu" + tag, arr))   # (with or without the norm prologue: one name for the profiles)
        else:
            self._gemm([dict(A=n_e[i], W=P.act(f"{pre}mlp.{i}.layers.0.weight"), bias=P.f32_vec(f"{pre}mlp.{i}.layers.0.bias"), Cact=hbuf[i])
                        for i in fields], "mlp.fc1" + tag)
            self._norm([dict(X=hbuf[i], gamma=P.f32_vec(f"{pre}mlp.{i}.layers.1.weight"), beta=P.f32_vec(f"{pre}mlp.{i}.layers.1.bias"), Yact=hg[i])
                        for i in fields], S, "mlp.ln_gelu" + tag, x_is_act=True, gelu=True)
        # fc2 + residual, proj and — after the last layer — the model's final norm in one launch where the kernel is instantiated (the same shapes as the
        # fc1 kernel): a workgroup owns 32 complete rows through both Linear layers.  Measured (plain replay, same box): cfg2 0.2373 -> 0.2358 ms (the launch
        # 30.8 us against 19.8 + 7.1 + 5.6 with two boundaries less: a 32-row workgroup per CU streams W2 at a third of the rate three co-resident 64 x 64
        # tiles do), B = 8 1.108 -> 1.18 ms (200 us against 76 + 21 + 31) — short launches only.  SEA_PLAN=mlp2=0 / 1 forces.
        w2 = _switches.plan("mlp2", "auto")
        if (type(self) is Plan and (w2 == "1" or (w2 == "auto" and 1024 <= self.M <= 4096)) and ops.mlp_fc1_supported(self.dt, E, S) and Eo == E and len(fields) <= N.MAX_MLP_GROUPS):
            arr = (N.SeaMlp2Group * len(fields))()
            for g_, i in zip(arr, fields):
                nrm = norm_params(f"ln.{i}.", E) if final_norm else {}
                ops.fill_mlp2_group(g_, hg[i], P.act(f"{pre}mlp.{i}.layers.3.weight"), P.f32_vec(f"{pre}mlp.{i}.layers.3.bias"), xr[i],
                                    P.act(f"{pre}proj.{i}.weight"), P.f32_vec(f"{pre}proj.{i}.bias"), Y32=xr[i], ldy32=(FE if final_norm else None), **nrm)
                if final_norm:
                    self._out_patches.append((g_, "Y32", i * E * 4))
            self._cur.append(self._rec(N.lib().sea_mlp_fc2_proj_norm, [arr, len(fields), 1e-5, self.code], ("mlp.fc2_proj_norm" if final_norm else "mlp.fc2_proj") + tag, arr))
            return final_norm
        self._gemm([dict(A=hg[i], W=P.act(f"{pre}mlp.{i}.layers.3.weight"), bias=P.f32_vec(f"{pre}mlp.{i}.layers.3.bias"), R=xr[i], Cact=xm[i])
                    for i in fields], "mlp.fc2" + tag)
        # the last layer's proj + the model's final norm in one launch (sea_gemm_rownorm: a tile spans the whole output row): at B = 8 two launches of 23 + 31 us
        # (the norm re-reads the rows the proj has just written: 100 MB) -> one.  SEA_PLAN=norm=0 / projnorm=0 keep the two launches.
        if final_norm and getattr(self, "_fuse_norm", False) and type(self) is Plan and Eo <= 256 and Eo % 16 == 0 and Eo == E and _switches.plan("projnorm", "1") != "0":
            self._gemm_norm([dict(A=xm[i], W=P.act(f"{pre}proj.{i}.weight"), bias=P.f32_vec(f"{pre}proj.{i}.bias"), Y32=xo[i], ldy32=FE, Y_is_out=i * Eo * 4,
                                  **norm_params(f"ln.{i}.", Eo)) for i in fields], "proj_norm" + tag)
            return final_norm
        self._gemm([dict(A=xm[i], W=P.act(f"{pre}proj.{i}.weight"), bias=P.f32_vec(f"{pre}proj.{i}.bias"), C32=xo[i]) for i in fields], "proj" + tag)
        if final_norm:
            self._norm([dict(X=xo[i], Y32=xo[i], ldy32=FE, Y_is_out=i * Eo * 4, **norm_params(f"ln.{i}.", Eo)) for i in fields], Eo, "final.norm" + tag)
        return final_norm

    def _ib_params(self, pre: str) -> dict:
        P = self.eng.params
        return dict(w1=P.f32_vec(pre + "ib.layers.0.weight"), b1=P.f32_vec(pre + "ib.layers.0.bias"), lnw=P.f32_vec(pre + "ib.layers.1.weight"),
                    lnb=P.f32_vec(pre + "ib.layers.1.bias"), w2=P.f32(pre + "ib.layers.3.weight"), b2=P.f32_vec(pre + "ib.layers.3.bias"),
                    h=self.eng.model.ib_hidden)

    def _fill_ib(self, ib, pre: str) -> None:
        """The layer parameters of SeaIbParams for the block's ib_scale_mode (models/temporal.py:103-109)."""
        P, mode = self.eng.params, self.eng.ib_mode
        ib.mode, ib.M, ib.E = mode, self.M, self.ib_dim
        if mode == 0:
            q = self._ib_params(pre)
            ib.w1, ib.b1, ib.lnw, ib.lnb, ib.w2, ib.b2 = (q[k].data_ptr() for k in ("w1", "b1", "lnw", "lnb", "w2", "b2"))
            ib.h = q["h"]
        elif mode == 1:   # nn.Linear(1, E): weight [E, 1], bias [E]
            ib.w1, ib.b1, ib.h = P.f32_vec(pre + "ib.weight").data_ptr(), P.f32_vec(pre + "ib.bias").data_ptr(), 1
        else:             # GaussianFourierProjection: W [1, E/2]
            ib.w1, ib.h = P.f32(pre + "ib.W").data_ptr(), 1

    def _ib(self, pre: str, xr: List[torch.Tensor], drop=None) -> None:
        P = self.eng.params
        ib = N.SeaIbParams()
        if drop is not None:
            ib.drop.thr, ib.drop.stream = drop
            self._drop_structs.append(ib)
        for i, x in enumerate(xr):
            ib.X[i] = x.data_ptr()
        ib.n_fields, ib.ldx = len(xr), xr[0].stride(0)
        self._fill_ib(ib, pre)
        self._c_patches.append((ib, "c"))
        self._cur.append(self._rec(N.lib().sea_ib_add, [C.byref(ib)], "ib_add", ib))

    def _ib_rows(self, pre: str, drop=None):
        """The info-bottleneck rows ib(cond) [M, E] as a buffer of their own (fp32, and in the activation dtype): a copy of zeros, the usual ib add on top."""
        L, M, E, f32 = N.lib(), self.M, self.E, torch.float32
        zeros, ib32 = self._buf(M, E, dtype=f32, zero=True), self._buf(M, E, dtype=f32)
        self._cur.append(self._rec(L.sea_convert_f32_to_act, [zeros.data_ptr(), E, ib32.data_ptr(), E, M, E, N.SEA_F32], "ib.rows.zero", zeros))
        self._ib(pre, [ib32], drop=drop)
        if self.dt == f32:
            return ib32, ib32
        ibact = self._buf(M, E)
        self._cur.append(self._rec(L.sea_convert_f32_to_act, [ib32.data_ptr(), E, ibact.data_ptr(), E, M, E, self.code], "ib.rows.act"))
        return ib32, ibact

    def _ib_rows_multi(self, pre: str, n: int, drop) -> List[torch.Tensor]:
        """n sets of info-bottleneck rows ib(cond) [M, E] in the activation dtype, each with its OWN dropout mask (streams drop[1] .. drop[1] + n - 1 of one
        sea_ib_add over n zeroed buffers): what the reference's n evaluations of self.ib in train() produce (models/temporal.py:110-118)."""
        L, M, E, f32 = N.lib(), self.M, self.E, torch.float32
        zeros = self._buf(M, E, dtype=f32, zero=True)
        rows32 = [self._buf(M, E, dtype=f32) for _ in range(n)]
        for t in rows32:
            self._cur.append(self._rec(L.sea_convert_f32_to_act, [zeros.data_ptr(), E, t.data_ptr(), E, M, E, N.SEA_F32], "ib.rows.zero", zeros))
        self._ib(pre, rows32, drop=drop)
        if self.dt == f32:
            return rows32
        out = [self._buf(M, E) for _ in range(n)]
        for a, b in zip(rows32, out):
            self._cur.append(self._rec(L.sea_convert_f32_to_act, [a.data_ptr(), E, b.data_ptr(), E, M, E, self.code], "ib.rows.act"))
        return out

    def _act_copy(self, x32: torch.Tensor, name: str, keep: bool = False) -> torch.Tensor:
        """Activation-dtype copy of an fp32 [M, E] (possibly strided) matrix — the operand of a GEMM that reads the residual stream directly.  `keep`: a
        real copy in fp32 too (training: the rows are updated in place afterwards and the weight gradient needs them as they were)."""
        if self.dt == torch.float32 and not keep:
            return x32
        y = self._buf(self.M, x32.shape[1])
        self._cur.append(self._rec(N.lib().sea_convert_f32_to_act, [x32.data_ptr(), x32.stride(0), y.data_ptr(), y.stride(0), self.M, x32.shape[1], self.code], name))
        return y

    def _ib_attn(self, pre: str, xr: List[torch.Tensor]) -> None:
        """ib_addition_mode 'attention' (models/temporal.py:117-118; MultiHeadCrossAttention, models/base_blocks.py:205-243): x_i += proj_i(softmax(q_i(x_i)
        k_i(ib)^T / sqrt(hd)) v_i(ib)) — no mask, no rotary embedding: the usual QKV / attention launches with a zero-angle rotation table and a visibility
        window (src_len) as long as the sequence."""
        eng, P, B, H, T, E, M, cap = self.eng, self.eng.params, self.B, self.H, self.T, self.E, self.M, self.cap
        F, hd = len(xr), E // H
        _, ibact = self._ib_rows(pre)
        xq = [self._act_copy(xr[i], "ib.attn.x_act") for i in range(F)]
        Q = [self._buf(B, H, T, hd) for _ in range(F)]
        K = [self._buf(B, H, cap, hd, zero=True) for _ in range(F)]
        Vt = [self._buf(B, H, hd, cap, zero=True) for _ in range(F)]
        att = [self._buf(M, E) for _ in range(F)]
        qg = []
        for i in range(F):
            ca = f"{pre}cross_attn_ib.{i}."
            qg.append(dict(A=xq[i], W=P.act(ca + "q.weight"), bias=P.f32_vec(ca + "q.bias"), col0=0, Q=Q[i]))
            qg.append(dict(A=ibact, W=P.act(ca + "k.weight", 2 * E), bias=P.f32_vec(ca + "k.bias", 2 * E), col0=E, K=K[i], Vt=Vt[i]))
        self._qkv(qg, eng.rope_identity(hd), hd, "ib.attn.qkv")
        self._attn([dict(Q=Q[i], K=K[i], Vt=Vt[i], O=att[i]) for i in range(F)], hd, E, "ib.attn.attention", src_len=self.cap)
        self._gemm([dict(A=att[i], W=P.act(f"{pre}cross_attn_ib.{i}.projection.weight"), R=xr[i], C32=xr[i]) for i in range(F)], "ib.attn.proj")

    # ------------------------------------------------------------------ binding and replay
    def bind(self, x: torch.Tensor, ib: torch.Tensor, out: torch.Tensor) -> None:
        """Point the plan at the caller's tensors.  The plan knows them by raw address only, so it HOLDS them until the next bind: a launch list that is
        still executing (the host runs several steps ahead of the device) must never see its output block returned to the allocator — and possibly
        unmapped, e.g. by the cache flush a graph capture starts with — because the caller dropped the tensor."""
        self._bound_tensors = (x, ib, out)
        self.bind_ptrs(x.data_ptr(), ib.data_ptr(), out.data_ptr())
        if not self._audited or ptrcheck.always():
            self.audit()

    def _known_ranges(self, owners=()) -> "ptrcheck.Ranges":
        eng, R = self.eng, ptrcheck.Ranges()
        for t in self._keep:
            R.add_tensor(t, "plan workspace")
        P = eng.params
        for t, label in ((P.flat32, "flat32"), (P.flat_act, "flat_act"), (P.flat_actT, "flat_actT"), (eng.grads, "grads"), (eng.rope_self, "rope_self"),
                         (eng.rope_cross, "rope_cross"), (getattr(self, "_ws_colsum", None), "column-sum workspace"), (self._zero_ib, "zero rows")):
            R.add_tensor(t, label)
        for t in eng._eyes.values():
            R.add_tensor(t, "identity / rotation table")
        if self._cond_src is not None:
            for t in self._cond_src._keep:
                R.add_tensor(t, "hoisted condition buffers")
        for t, label in zip(getattr(self, "_bound_tensors", ()) or (), ("x", "condition", "out")):
            R.add_tensor(t, "bound " + label)
        R.add_tensor(getattr(self, "_bound_dout", None), "bound dout")
        for t in (eng._loss_ws or ()):
            R.add_tensor(t, "loss workspace")
        for t in owners:
            R.add_tensor(t, "caller buffer")
        return R

    def audit(self, owners=()) -> int:
        """Every device pointer of every launch record lies inside a buffer this plan knows, operands with their full extent (sea_amd/ptrcheck.py).
        Runs at the first bind of a plan and at every bind under SEA_CHECK_PTRS=1."""
        esz = 4 if self.dt == torch.float32 else 2
        n = ptrcheck.check_records(self._all_records(), self._known_ranges(owners), esz, f"{type(self).__name__} B={self.B} T={self.T} {self.mode}")
        self._audited = True
        return n

    def _all_records(self):
        return self.records

    def bind_ptrs(self, xp: int, cp: int, op: int) -> None:
        """Point the plan at the caller's buffers: x [M, F, E] fp32, condition [M] fp32, out [M, F, E] fp32 (row m = b*T + t)."""
        key = (xp, cp, op)
        if key == self._bound:
            return
        for tgt, field, off in self._x_patches:
            if isinstance(tgt, list):
                tgt[field] = xp + off
            else:
                setattr(tgt, field, xp + off)
        for tgt, field, off in self._out_patches:
            setattr(tgt, field, op + off)
        for tgt, field in self._c_patches:
            if isinstance(tgt, list):
                tgt[field] = cp
            else:
                setattr(tgt, field, cp)
        self._bound = key
        self._relink()

    def set_dropout_seed(self, seed: int) -> None:
        for st in self._drop_structs:
            st.drop.seed = seed & 0xFFFFFFFF

    def set_position(self, pos0: int) -> None:
        """Step mode: the T rows of this call sit at absolute positions pos0 .. pos0 + T - 1 of the K/V caches."""
        assert pos0 + self.T <= self.cap
        self.pos0 = pos0
        for s in self._pos_structs:
            if isinstance(s, N.SeaQkvCommon):
                s.pos0 = pos0
            else:
                s.q_pos0, s.Tk = pos0, pos0 + self.T

    def step_patch_table(self, x_base: int, x_stride: int, c_base: int, c_stride: int, out_base: int, out_stride: int):
        """The per-step edits of a KV-cache rollout (set_position + bind_ptrs) as a SeaStepPatch array for sea_run_list_steps: step s reads its rows at
        x_base + s * x_stride, its condition values at c_base + s * c_stride, writes its rows at out_base + s * out_stride and sits at position s of the
        caches.  None when the plan has no native launch list."""
        if self._clist is None:
            return None
        arr, _, relink = self._clist
        rows = []   # (address, kind, base, stride)

        def field_addr(tgt, field):
            if isinstance(tgt, list):   # a pointer argument kept in a record's Python args list: its native copy is a field of the SeaLaunchRec
                for i, rf, args, k in relink:
                    if args is tgt and k == field:
                        return C.addressof(arr[i]) + getattr(N.SeaLaunchRec, rf).offset
                raise KeyError("patched argument without a native copy")
            return C.addressof(tgt) + getattr(type(tgt), field).offset

        for st in self._pos_structs:
            if isinstance(st, N.SeaQkvCommon):
                rows.append((field_addr(st, "pos0"), 0, 0, 1))
            else:
                rows.append((field_addr(st, "q_pos0"), 0, 0, 1))
                rows.append((field_addr(st, "Tk"), 0, self.T, 1))
        for tgt, field, off in self._x_patches:
            rows.append((field_addr(tgt, field), 1, x_base + off, x_stride))
        for tgt, field, off in self._out_patches:
            rows.append((field_addr(tgt, field), 1, out_base + off, out_stride))
        for tgt, field in self._c_patches:
            rows.append((field_addr(tgt, field), 1, c_base, c_stride))
        for st, field, base, step in self._hoisted:
            rows.append((field_addr(st, field), 1, base, step))
        tab = (N.SeaStepPatch * max(len(rows), 1))()
        for t, (a_, k_, b_, s_) in zip(tab, rows):
            t.addr, t.kind, t.base, t.stride = a_, k_, b_, s_
        return tab, len(rows)

    def run_steps(self, n_steps: int, x_base: int, x_stride: int, c_base: int, c_stride: int, out_base: int, out_stride: int) -> bool:
        """n_steps consecutive KV-cache steps from position 0 in ONE native call (sea_run_list_steps); False when the plan has no native list."""
        pt = self.step_patch_table(x_base, x_stride, c_base, c_stride, out_base, out_stride)
        if pt is None:
            return False
        assert n_steps <= self.cap
        rc = N.lib().sea_run_list_steps(self._clist[0], self._clist[1], pt[0], pt[1], 0, n_steps, N.stream_ptr())
        self._bound = (None, None, None)   # the structs now hold the last step's pointers / position
        self.pos0 = n_steps - 1 if n_steps > 0 else self.pos0
        if rc != 0:
            N.check(rc, "KV-cache rollout (sea_run_list_steps)")
        return True

    def _compile_list(self) -> None:
        """Lower self.records to one SeaLaunchRec array (include/sea_hip.h): a sequential replay is then ONE native call instead of a
        Python/ctypes round trip per launch.  Pointers into the argument structs are stable (the structs are patched in place)."""
        L = N.lib()
        recs = [r for r in self.records if r.fn is not None]
        arr = (N.SeaLaunchRec * max(len(recs), 1))()
        relink = []   # records whose pointer arguments live in the Python args list (patched by bind): copied again at bind time
        addr = lambda o: C.addressof(o)
        for i, r in enumerate(recs):
            c, a = arr[i], r.args
            if r.fn is L.sea_gemm_grouped:
                c.op, c.p0, c.n, c.dtype = N.OP_GEMM, addr(a[0]), a[1], a[2]
            elif r.fn is L.sea_qkv_rope_grouped:
                c.op, c.p0, c.n, c.p1, c.dtype = N.OP_QKV, addr(a[0]), a[1], addr(r.keep[1]), a[3]
            elif r.fn is L.sea_gemm_fewrows:
                c.op, c.p0, c.n, c.i0, c.i1, c.f0, c.dtype = N.OP_GEMM_FEW, addr(a[0]), a[2], a[3], a[4], a[5], a[6]
                c.p1 = addr(a[1]) if a[1] is not None else None
            elif r.fn is L.sea_qkv_rope_fewrows:
                c.op, c.p0, c.n, c.p1, c.f0, c.dtype = N.OP_QKV_FEW, addr(a[0]), a[2], addr(r.keep[2]), a[4], a[5]
                c.l0 = addr(a[1]) if a[1] is not None else 0
            elif r.fn is L.sea_attention_fwd:
                c.op, c.p0, c.dtype = N.OP_ATTN, addr(r.keep), a[1]
            elif r.fn is L.sea_mlp_fc1_ln_gelu:
                c.op, c.p0, c.n, c.f0, c.dtype = N.OP_MLP1, addr(a[0]), a[1], a[2], a[3]
            elif r.fn is L.sea_mlp_fc2_proj_norm:
                c.op, c.p0, c.n, c.f0, c.dtype = N.OP_MLP2, addr(a[0]), a[1], a[2], a[3]
            elif r.fn is L.sea_adaln_qkv:
                c.op, c.p0, c.n, c.p1, c.f0, c.dtype = N.OP_AQKV, addr(a[0]), a[1], addr(r.keep[1]), a[10], a[11]
                c.l0, c.i0 = (addr(a[3]) if a[3] is not None else 0), a[4]
                c.l1, c.i1 = (addr(a[5]) if a[5] is not None else 0), a[6]
                c.l2, c.i2 = (a[7] or 0), a[8]
                if a[8]:
                    relink.append((i, "l2", a, 7))   # the condition pointer of the row riders, patched at every bind
                c.l3 = addr(r.keep[4]) if r.keep[4] is not None else 0
            elif r.fn is L.sea_splitk_finish:
                c.op, c.p0, c.n, c.dtype = N.OP_SPLITK, addr(a[0]), a[1], a[2]
            elif r.fn is L.sea_mlp_block:
                c.op, c.p0, c.p1, c.n, c.f0, c.dtype = N.OP_MLPB, addr(a[0]), addr(a[1]), a[2], a[3], a[4]
            elif r.fn is L.sea_exchange_tail:
                c.op, c.p0, c.n, c.f0, c.dtype = N.OP_XTAIL, addr(r.keep), a[1], a[2], a[3]
            elif r.fn is L.sea_gemm_adaln:
                c.op, c.p0, c.n, c.f0, c.dtype = N.OP_ADALN, addr(a[0]), a[1], a[2], a[3]
            elif r.fn is L.sea_row_chain:
                c.op, c.p0, c.n, c.p1, c.f0, c.dtype = N.OP_CHAIN, addr(a[0]), a[1], addr(r.keep[1]), a[3], a[4]
            elif r.fn is L.sea_row_chain_riders:
                c.op, c.p0, c.n, c.p1, c.f0, c.dtype = N.OP_CHAIN, addr(a[0]), a[1], addr(r.keep[1]), a[8], a[9]
                c.l0, c.i0, c.i1, c.i2 = addr(a[3]), a[4], a[5], a[6]
                c.l1 = addr(r.keep[3]) if r.keep[3] is not None else 0
            elif r.fn is L.sea_gemm_rownorm:
                c.op, c.p0, c.n, c.f0, c.dtype = N.OP_GEMM_NORM, addr(a[0]), a[1], a[2], a[3]
            elif r.fn is L.sea_rownorm:
                c.op, c.p0, c.n, c.i0, c.i1, c.i2, c.i3, c.f0, c.dtype = N.OP_NORM, addr(a[0]), a[1], a[2], a[3], a[4], a[5], a[6], a[7]
            elif r.fn is L.sea_silu_outer or r.fn is L.sea_silu_outer_ib:
                c.op, c.p0, c.n, c.i0, c.dtype = N.OP_SILU, (addr(a[0]) if a[0] is not None else None), a[1], a[3], a[4]
                if r.fn is L.sea_silu_outer_ib and a[5] is not None:
                    c.l0, c.l1 = addr(a[5]), a[6]
                relink.append((i, "p1", a, 2))
            elif r.fn is L.sea_ib_add:
                c.op, c.p0 = N.OP_IB, addr(r.keep)
            elif r.fn is L.sea_convert_f32_to_act:
                c.op, c.l0, c.p1, c.l1, c.l2, c.l3, c.dtype = N.OP_CONVERT, a[1], a[2], a[3], a[4], a[5], a[6]
                relink.append((i, "p0", a, 0))
            else:
                return  # a record the list runner does not know (training plans): keep the per-launch replay
        self._clist = (arr, len(recs), relink)
        self._relink()

    def _relink(self) -> None:
        if self._clist is not None:
            arr, _, relink = self._clist
            for i, field, args, k in relink:
                setattr(arr[i], field, args[k] if (args[k] is not None or field[0] == "p") else 0)   # (l0..l3 are integers: an unbound pointer is 0)

    def run(self, concurrent: bool = False) -> None:
        """Replay the launch list.  Sequentially on the current stream (record order is a valid order), or — `concurrent` — with every
        lane on its own stream, forked from / joined to the current stream by events; inside a graph capture the lanes become
        parallel branches of the graph."""
        main = torch.cuda.current_stream()
        stream = main.cuda_stream
        if not concurrent and self._clist is not None:
            rc = N.lib().sea_run_list(self._clist[0], self._clist[1], stream)
            if rc != 0:
                N.check(rc, "plan replay (sea_run_list)")
            return
        if not concurrent:
            for r in self.records:
                if r.fn is None:
                    continue
                rc = r.fn(*r.args, stream)
                if rc != 0:
                    N.check(rc, r.name)
            return
        open_lanes = set()
        for r in self.records:
            if r.fn is None:
                st = self._lane_streams.get(r.lane)
                if st is None:
                    st = self._lane_streams[r.lane] = torch.cuda.Stream(device=self.eng.device)
                if r.name == "fork":
                    st.wait_stream(main)
                    open_lanes.add(r.lane)
                else:
                    main.wait_stream(st)
                    open_lanes.discard(r.lane)
                continue
            rc = r.fn(*r.args, stream if r.lane == 0 else self._lane_streams[r.lane].cuda_stream)
            if rc != 0:
                N.check(rc, r.name)
        for lane in open_lanes:  # every lane rejoins (a graph capture requires it)
            main.wait_stream(self._lane_streams[lane])

    def time_records(self, iters: int = 10) -> List[Tuple[str, float]]:
        """Average device time of every launch of the plan, in milliseconds, from HIP events recorded on the launch stream
        around each launch (diagnostics / bench roofline)."""
        stream = N.stream_ptr()
        recs = [r for r in self.records if r.fn is not None]
        n = len(recs)
        tot = [0.0] * n
        for _ in range(iters):
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
            evs[0].record()
            for k, r in enumerate(recs):
                rc = r.fn(*r.args, stream)
                if rc != 0:
                    N.check(rc, r.name)
                evs[k + 1].record()
            torch.cuda.synchronize()
            for k in range(n):
                tot[k] += evs[k].elapsed_time(evs[k + 1])
        return [(r.name, t / iters) for r, t in zip(recs, tot)]


def _fill_gemm(g, A, W, bias=None, R=None, C32=None, Cact=None, n_seg=1, a_seg_stride=0, act=0, bias_scale=1.0, ldr=None,
               R_is_x=None, Z=None, ldc32=None, drop=None, silu=None, M=None) -> None:
    if drop is not None:  # (thr, stream, mode); the seed is patched every step
        g.drop.thr, g.drop.stream, g.drop.mode = drop
    if silu is not None:  # generated A operand: (w1 [K] f32, b1 [K] f32); the condition pointer is patched at bind time; rows = M
        g.A, g.W, g.lda = None, W.data_ptr(), 0
        g.silu_w1, g.silu_b1 = silu[0].data_ptr(), silu[1].data_ptr()
        g.Z, g.ldz, g.bias, g.R, g.C32, g.Cact = None, 0, N.ptr(bias), None, N.ptr(C32), N.ptr(Cact)
        g.a_seg_stride, g.ldw, g.ldr = 0, W.stride(0), 0
        g.ldc32 = C32.stride(0) if C32 is not None else 0
        g.ldcact = Cact.stride(0) if Cact is not None else 0
        g.M, g.N, g.K, g.n_seg, g.act, g.bias_scale = M, W.shape[0], W.shape[1], 1, 0, bias_scale
        return
    g.A, g.W = A.data_ptr(), W.data_ptr()
    g.Z, g.ldz = N.ptr(Z), (Z.stride(0) if Z is not None else 0)
    g.bias, g.R, g.C32, g.Cact = N.ptr(bias), N.ptr(R), N.ptr(C32), N.ptr(Cact)
    g.a_seg_stride = a_seg_stride
    g.lda, g.ldw = A.stride(-2), W.stride(0)
    g.ldr = (ldr if ldr is not None else R.stride(0)) if R is not None else 0
    g.ldc32 = (ldc32 if ldc32 is not None else C32.stride(0)) if C32 is not None else 0
    g.ldcact = Cact.stride(0) if Cact is not None else 0
    g.M, g.N, g.K = A.shape[-2], W.shape[0], W.shape[1]
    g.n_seg, g.act, g.bias_scale = n_seg, act, bias_scale


def blk_pe(eng: "TemporalEngine", layer: int) -> torch.Tensor:
    """The block's sinusoidal table [max_len, D] (PositionalEncoding buffer `pe`, models/base_blocks.py:355-368) on the device."""
    return eng.model.blocks[layer].pos_encoder.pe[0].to(device=eng.device, dtype=torch.float32)


class TemporalEngine:
    """Owns the flat parameter buffers of one TemporalModel on one GPU and the plans built over them."""

    def __init__(self, model: torch.nn.Module, device: torch.device, act_dtype: torch.dtype):
        N.lib()
        if device.type != "cuda":
            raise RuntimeError("sea_amd: TemporalModel runs only on an MI355X (no CPU fallback)")
        m = model
        self.ib_mode = {"mlp": 0, "linear": 1, "fourier": 2}.get(m.ib_scale_mode.lower(), -1)
        if (m.exchange_mode not in ("sea", "addition", "simple", "pool") or self.ib_mode < 0 or m.ib_addition_mode.lower() not in ("add", "none", "attention", "concat")
                or (self.ib_mode == 0 and m.ib_mlp_layers != 1) or m.ib_num != 1 or (self.ib_mode != 0 and m.embed_dim % 8 and m.ib_addition_mode.lower() != "concat")):
            raise NotImplementedError(
                "sea_amd native path covers exchange_mode in {'sea', 'addition', 'simple', 'pool'}, ib_scale_mode in {'mlp', 'linear', 'fourier'}, ib_addition_mode in "
                "{'add', 'attention', 'concat', 'none'}, ib_mlp_layers=1, ib_num=1; got "
                f"{m.exchange_mode}/{m.ib_scale_mode}/{m.ib_addition_mode}/{m.ib_mlp_layers}/{m.ib_num}")
        if m.ib_addition_mode.lower() == "concat" and m.add_info_after_cross:
            raise NotImplementedError("sea_amd: ib_addition_mode='concat' needs add_info_after_cross=False (with the info-bottleneck step after the exchange the "
                                      "reference's own forward fails: its attention modules are built for rows that are already widened, models/temporal.py:48,74-76,126-139)")
        E, H, D = m.internal_embed_dim, m.n_heads, m.down_dim
        for hd, what in ((E // H, "self"),) + (((D // H, "cross"),) if m.exchange_mode in ("sea", "pool") else ()):
            if hd not in (8, 16, 32, 64, 128, 256) or hd * H != (E if what == "self" else D):
                raise NotImplementedError(f"sea_amd: unsupported {what}-attention head dim {hd} (supported: 8, 16, 32, 64, 128, 256)")
        if m.src_len < 0:
            raise NotImplementedError("sea_amd: src_len must be >= 0")
        if self.ib_mode == 0 and m.ib_hidden > 64:
            raise NotImplementedError("sea_amd: info-bottleneck hidden width (scale_ratio) must be <= 64")
        self.model, self.device, self.act_dtype = model, device, act_dtype
        self.params = FlatParams(model, device, act_dtype)
        blk = model.blocks[0]
        self.rope_self = torch.view_as_real(blk.attn["self"][0].freqs_cis.to(device)).contiguous()
        self.rope_cross = None
        if m.exchange_mode == "sea":
            self.rope_cross = torch.view_as_real(blk.cross_attn[0][0].freqs_cis.to(device)).contiguous()
        elif m.exchange_mode == "pool":
            self.rope_cross = torch.view_as_real(blk.cross_attn[0].freqs_cis.to(device)).contiguous()
        self._eyes: Dict[int, torch.Tensor] = {}
        self._plans: Dict[Tuple, Plan] = {}
        self._graphs: Dict[Tuple, Tuple] = {}
        self._train_plans: Dict[Tuple, object] = {}
        self._kv_fast: Dict[int, object] = {}          # B -> kv_engine.KvFast
        self.grads: Optional[torch.Tensor] = None      # flat fp32 gradient buffer, same layout as params.flat32
        self.grads_dirty = False                       # True once a backward has accumulated into it since the last zero
        self._drop_step = 0                            # dropout streams are re-keyed every training forward
        self._dp_overlap: Optional[Tuple[int, bool]] = None   # (world size, slices under the backward?) — see dp_overlap()
        self._loss_ws: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None

    def eye(self, n: int) -> torch.Tensor:
        """Identity in the activation dtype: W of the GEMM that sums A segments and applies GELU ('addition' exchange)."""
        e = self._eyes.get(n)
        if e is None:
            e = self._eyes[n] = torch.eye(n, device=self.device, dtype=self.act_dtype)
        return e

    def rope_identity(self, hd: int) -> torch.Tensor:
        """A rotation table of zero angles for the un-rotated attention of ib_addition_mode 'attention' (models/base_blocks.py:222-243)."""
        t = self._eyes.get(("rope", hd))
        if t is None:
            t = torch.zeros(_round_up(self.model.max_len, 8), hd // 2, 2, device=self.device, dtype=torch.float32)
            t[..., 0] = 1.0
            self._eyes[("rope", hd)] = t
        return t

    def plan(self, B: int, T: int, mode: str = "full", cond=None) -> Plan:
        if mode == "step" and self.model.exchange_mode == "pool":
            raise NotImplementedError("sea_amd: the KV-cache rollout does not cover exchange_mode='pool' (its sinusoidal positions are relative to the window); "
                                      "use the recompute rollout")
        if mode == "step" and self.model.ib_addition_mode.lower() == "attention":
            raise NotImplementedError("sea_amd: the KV-cache rollout does not cover ib_addition_mode='attention' (every row attends to the info-bottleneck rows "
                                      "of ALL positions of the window, later ones included: rows already produced change as the window grows); use the recompute rollout")
        key = (B, T, mode) if cond is None else (B, T, mode, id(cond))
        p = self._plans.get(key)
        if p is None:
            if T > self.model.max_len:
                raise ValueError(f"sequence length {T} exceeds max_len {self.model.max_len}")
            if cond is not None:   # one hoisted step plan at a time per batch size: its predecessor's condition buffers are released with it
                for k in [k for k in self._plans if len(k) == 4 and k[:3] == (B, T, mode)]:
                    del self._plans[k]
            p = Plan(self, B, T, mode, cond=cond)
            self._plans[key] = p
        return p

    # ------------------------------------------------------------------ training
    def ensure_grads(self) -> None:
        if self.grads is None:
            self.grads = torch.zeros(self.params.n_total, device=self.device, dtype=torch.float32)

    def grad_view(self, name: str) -> torch.Tensor:
        o, shp = self.params.offsets[name]
        n = 1
        for s in shp:
            n *= s
        return self.grads[o:o + n].view(shp)

    def grad_vec(self, name: str, n: Optional[int] = None) -> torch.Tensor:
        o, shp = self.params.offsets[name]
        return self.grads[o:o + (shp[0] if n is None else n)]

    def grad_mat(self, name: str, rows: Optional[int] = None) -> torch.Tensor:
        o, shp = self.params.offsets[name]
        r = shp[0] if rows is None else rows
        return self.grads[o:o + r * shp[1]].view(r, shp[1])

    def zero_grads(self) -> None:
        if self.grads is not None:
            self.grads.zero_()
        self.grads_dirty = False

    def train_plan(self, B: int, T: int):
        from .train_engine import TrainPlan

        m = self.model
        thr = int(round(256 * m.dropout_p)) if (m.training and m.dropout_p > 0) else 0
        if thr > 255:
            raise ValueError("dropout probability too close to 1")
        dp = self.dp_overlap()
        key = (B, T, thr, dp)
        p = self._train_plans.get(key)
        if p is None:
            if T > m.max_len:
                raise ValueError(f"sequence length {T} exceeds max_len {m.max_len}")
            p = TrainPlan(self, B, T, drop_thr=thr, dp=dp)
            self._train_plans[key] = p
        return p

    def dp_overlap(self) -> bool:
        """Reduce the gradient buffer in slices under the backward (parallel.OverlappedGradientReduce)?  Decided ONCE per engine and process group, and
        AGREED between the ranks (a MIN all-reduce of each rank's SEA_DP_OVERLAP switch): the number of collectives a step issues follows from it, so ranks
        that disagreed, or a switch flipped between two steps, would deadlock.  False without a process group / at world size 1; a group initialised later
        is seen at the next step (the training plans are keyed on the answer)."""
        import torch.distributed as dist

        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        if world == 1:
            from .parallel import rehearse

            if not (rehearse() and dist.is_available() and dist.is_initialized()):   # (SEA_DP_REHEARSE=1: one rank issues the collectives of a larger world)
                return False
        if self._dp_overlap is None or self._dp_overlap[0] != world:
            want = torch.tensor([0 if os.environ.get("SEA_DP_OVERLAP", "1") == "0" else 1], device=self.device, dtype=torch.int32)
            dist.all_reduce(want, op=dist.ReduceOp.MIN)
            self._dp_overlap = (world, bool(int(want.item())))
        return self._dp_overlap[1]

    def forward_train(self, x: torch.Tensor, ib: torch.Tensor):
        """Forward that keeps the activations the backward needs.  Returns (out, plan)."""
        B, T, F, E = x.shape
        x, ib = x.contiguous(), ib.contiguous()
        out = torch.empty_like(x)
        self.params.sync()
        self.params.sync_transposed()
        p = self.train_plan(B, T)
        p.bind(x, ib, out)
        p.held = (x, ib)  # the backward list reads the inputs again: keep them alive until the next forward
        p.generation = getattr(p, "generation", 0) + 1   # identifies the activation set now in the plan's workspace (autograd.py checks it)
        if p.drop_thr > 0:
            self._drop_step += 1
            from .parallel import rank as _dp_rank   # data parallel: every rank draws its own masks (the shards are different trajectories of one batch)
            p.set_dropout_seed((torch.initial_seed() * 0x9E3779B1 + self._drop_step * 0x85EBCA77 + _dp_rank() * 0xC2B2AE35) & 0xFFFFFFFF)
        p.run()
        return out, p

    def backward(self, plan, dout: torch.Tensor, on_bucket=None) -> None:
        """Accumulate d loss / d parameters into self.grads from dout = d loss / d out ([B,T,F,E] fp32, contiguous).  `on_bucket(lo, hi)`: called
        as soon as self.grads[lo:hi] is final (TrainPlan.grad_buckets), while later launches are still being issued."""
        assert dout.is_contiguous() and dout.dtype == torch.float32
        plan.bind_dout(dout.data_ptr())
        plan._bound_dout = dout          # held until the next backward: the launch list knows it by address only
        if ptrcheck.always():
            plan.audit(owners=(dout,))
        plan.set_grads_fresh(not self.grads_dirty)   # a backward into zeros (the usual step) lets the big weight-gradient launches store instead of add
        plan.run_backward(on_bucket)
        self.grads_dirty = True

    def mse_loss_and_grad(self, out: torch.Tensor, target: torch.Tensor, grad_scale: float = 1.0):
        """loss = mean((out - target)^2) and dout = grad_scale * d loss / d out in one kernel pass."""
        if self._loss_ws is None or self._loss_ws[0].numel() != out.numel():
            self._loss_ws = (torch.empty_like(out), torch.zeros(1, device=self.device), torch.empty(1024, device=self.device))
        dout, loss, partial = self._loss_ws
        target = target.contiguous()
        N.check(N.lib().sea_mse_fwd_bwd(out.data_ptr(), target.data_ptr(), dout.data_ptr(), loss.data_ptr(), partial.data_ptr(), 1024,
                                        out.numel(), grad_scale, N.stream_ptr()), "sea_mse_fwd_bwd")
        return loss, dout

    def train_step(self, x: torch.Tensor, target: torch.Tensor, ib: torch.Tensor, optimizer, allreduce: bool = True, loss_weight: float = 1.0) -> torch.Tensor:
        """One fused train step (train/train_temporal.py:254-258): zero grads, forward, MSE + its gradient, backward, the gradient
        all-reduce when torch.distributed is initialised (the MLP slice as soon as it is final, the rest after the backward: sea_amd/parallel.py) (`allreduce=False`: a rank-local step, e.g. to time the step without the collective),
        AdamW.  Returns the local loss as a device scalar (no host sync).  `loss_weight` scales this rank's loss gradient (an uneven data-parallel
        shard: rows * world / global rows, train/train_temporal.py)."""
        from .parallel import OverlappedGradientReduce

        optimizer.zero_grad(set_to_none=False)
        out, plan = self.forward_train(x, ib)
        loss, dout = self.mse_loss_and_grad(out, target, grad_scale=float(loss_weight))
        red = OverlappedGradientReduce(self.grads, self.params.n_live, overlap=plan.dp) if allreduce else None
        self.backward(plan, dout, red.on_bucket if (red is not None and red.active) else None)
        scale = red.finish() if red is not None else 1.0
        self.last_allreduce_calls = red.calls if red is not None else 0
        if hasattr(optimizer, "mark_reduced"):
            optimizer.mark_reduced(scale)       # FlatAdamW.step() would otherwise all-reduce the buffer itself
        else:
            optimizer.grad_scale = scale
        optimizer.step()
        return loss

    def forward(self, x: torch.Tensor, ib: torch.Tensor) -> torch.Tensor:
        """TemporalModel.forward: x [B,T,F,E] fp32, ib [B,T,1] fp32 -> [B,T,F,E] fp32."""
        B, T, F, E = x.shape
        x = x.contiguous()
        ib = ib.contiguous()
        out = torch.empty_like(x)
        self.params.sync()
        p = self.plan(B, T, "full")
        p.bind(x, ib, out)
        p.run()
        return out

    def forward_graphed(self, x: torch.Tensor, ib: torch.Tensor) -> torch.Tensor:
        """Same forward replayed as ONE captured HIP graph (launch-bound regime: ~30 short kernels per layer).  The graph is
        keyed on the input tensors' addresses: write new data INTO x / ib, the returned tensor is reused between calls."""
        B, T, F, E = x.shape
        assert x.is_contiguous() and ib.is_contiguous()
        self.params.sync()
        key = (B, T, x.data_ptr(), ib.data_ptr())
        hit = self._graphs.get(key)
        if hit is None:
            out = torch.empty_like(x)
            p = Plan(self, B, T, "full")  # a private plan: its workspace addresses are baked into the graph
            p.bind(x, ib, out)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                p.run()  # warm-up outside capture (one-time function attributes, lazy module loading)
            torch.cuda.current_stream().wait_stream(side)
            # A capture begins with a flush of the caching allocator (freed blocks are UNMAPPED): everything issued so far — plain replays of other plans
            # that may still write tensors the caller has since dropped — has to be complete before that, or those launches fault (seen in round 2 as a
            # "Memory access fault" when bench.py captured a graph behind 200 plain forwards).  Once per (shape, inputs): not on the replay path.
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                p.run(concurrent=_switches.plan("graph_lanes", "1") != "0")
            hit = (graph, out, p, x, ib)
            self._graphs[key] = hit
        hit[0].replay()
        return hit[1]

    def rollout_kv(self, x0: torch.Tensor, ib: torch.Tensor, n_steps: int) -> torch.Tensor:
        """Exact KV-cache rollout (the loop of utils/train_utils.py:202-209 without recomputing the prefix): step s feeds the row
        at position s, appends its K/V to the per-layer caches and attends over positions <= s.  x0 [B,1,F,E], ib [B,>=n_steps,1]
        -> [B, n_steps, F, E].  The trajectory is kept time-major [n_steps+1, B, F, E] so that every step reads and writes
        contiguous [B, F, E] slabs with no copies."""
        B, one, F, E = x0.shape
        assert one == 1 and ib.shape[0] == B and ib.shape[1] >= n_steps
        if n_steps > self.model.max_len:
            raise ValueError(f"rollout of {n_steps} steps exceeds max_len {self.model.max_len}")
        if self.model.src_len > 0:
            # the reference masks with tril(diagonal=src_len) (models/base_blocks.py:173, 265): in its recompute loop the rows already produced re-attend
            # to the src_len rows appended after them, so their K/V and everything downstream change from step to step — a cache is not exact
            raise NotImplementedError("sea_amd: the KV-cache rollout is exact only for src_len == 0; use the recompute rollout (rollout(..., mode='recompute'))")
        from . import kv_engine
        if kv_engine.supported(self, B):
            # small models: seven launches per layer and step, condition-only work batched over all steps up front (sea_kv_rollout)
            kf = self._kv_fast.get(B)
            if kf is None:
                kf = self._kv_fast[B] = kv_engine.KvFast(self, B)
            return kf.rollout(x0, ib, n_steps)
        self.params.sync()
        traj = torch.empty(n_steps + 1, B, F, E, device=self.device, dtype=torch.float32)
        traj[0].copy_(x0[:, 0])
        cond = ib[:, :n_steps, 0].t().contiguous()  # [n_steps, B]
        # What depends on the condition only (the AdaLN modulations of every module, the info-bottleneck term) is evaluated for ALL steps by one batched
        # pass before the loop — the full-context plan's own silu / grouped-GEMM / ib launches on n_steps * B rows — instead of once per step on B rows:
        # at the shipped cylinder width (embed_dim 1024) that is two launches and 31 % of the weight bytes of every step.  SEA_KV=hoist=0: per step.
        cp = None
        m = self.model
        if _switches.kv("hoist", "1") != "0" and (m.LN_type.lower() == "adaln" or (m.ib_addition_mode.lower() == "add" and m.add_info_after_cross)) \
                and m.ib_addition_mode.lower() in ("add", "none"):
            cp = kv_engine.cond_plan_for(self, n_steps * B)
            for t in cp.ibufs:
                t.zero_()
            cp.bind_ptrs(0, cond.data_ptr(), 0)
            if not cp._audited or ptrcheck.always():
                cp.audit(owners=(cond,))
            cp.run()
        p = self.plan(B, 1, "step", cond=cp)
        slab = B * F * E * 4
        base, cbase = traj.data_ptr(), cond.data_ptr()
        if not p._audited or ptrcheck.always():   # the step plan is bound by raw address: audit it once against the buffers it will walk
            p.set_position(0)
            p.bind_ptrs(base, cbase, base + slab)
            p.set_hoisted_step(0)
            p.audit(owners=(traj, cond))
        # the step loop in native code (the plan's launch list + a table of the per-step edits); SEA_KV=loop=python: one Python round trip per step
        if _switches.kv("loop", "native") == "python" or not p.run_steps(n_steps, base, slab, cbase, B * 4, base + slab, slab):
            for s in range(n_steps):
                p.set_position(s)
                p.bind_ptrs(base + s * slab, cbase + s * B * 4, base + (s + 1) * slab)
                p.set_hoisted_step(s)
                p.run()
        return traj[1:].permute(1, 0, 2, 3).contiguous()
