"""Host-side audit of the device pointers a launch plan hands to libsea_hip.so.

The C ABI (include/sea_hip.h) takes raw addresses: a plan fills its argument structs once from `tensor.data_ptr()` and patches the caller's
buffers in at bind time (engine.Plan.bind_ptrs).  Nothing on the device checks them — a stale or mis-patched address is a page fault at best, a
queue abort (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION: an address outside the global aperture) or silent corruption at worst.  This module
walks every argument struct of a plan and requires that

  * every pointer field is NULL or lies inside an allocation the plan knows (its workspace, the engine's flat parameter / gradient buffers and
    tables, the tensors bound as x / condition / out), and
  * for the operands whose extent follows from the struct itself (rows, leading dimension, columns), the LAST byte the launch will touch
    lies inside the same allocation.

It runs once per plan at its first bind (a few milliseconds of Python) and at every bind under SEA_CHECK_PTRS=1.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Iterable, Iterator, List, Optional, Tuple

import torch

from . import _native as N

_HOST_STRUCTS = (N.SeaLaunchRec, N.SeaStepPatch)   # their pointers are HOST addresses (argument structs, patch targets)


def always() -> bool:
    return os.environ.get("SEA_CHECK_PTRS", "0") == "1"


class Ranges:
    """Sorted list of [lo, hi) device allocations with labels."""

    def __init__(self) -> None:
        self._r: List[Tuple[int, int, str]] = []

    def add_tensor(self, t: Optional[torch.Tensor], label: str) -> None:
        if t is None or not isinstance(t, torch.Tensor) or not t.is_cuda or t.numel() == 0:
            return
        st = t.untyped_storage()
        self._r.append((st.data_ptr(), st.data_ptr() + st.nbytes(), label))

    def add_range(self, lo: int, nbytes: int, label: str) -> None:
        if lo and nbytes > 0:
            self._r.append((lo, lo + nbytes, label))

    def find(self, p: int) -> Optional[Tuple[int, int, str]]:
        for lo, hi, label in self._r:
            if lo <= p < hi:
                return lo, hi, label
        return None


def iter_pointers(st: C.Structure, path: str = "") -> Iterator[Tuple[str, int]]:
    """(field path, address) of every non-NULL void* held DIRECTLY by a ctypes struct (plain fields and arrays of void*; nested structs are
    visited on their own by _walk_structs)."""
    for name, typ in st._fields_:
        if typ is C.c_void_p:
            v = getattr(st, name)
            if v:
                yield f"{path}.{name}", int(v)
        elif isinstance(typ, type) and issubclass(typ, C.Array) and typ._type_ is C.c_void_p:
            arr = getattr(st, name)
            for i in range(len(arr)):
                if arr[i]:
                    yield f"{path}.{name}[{i}]", int(arr[i])


def _extents(st, esz: int) -> Iterable[Tuple[str, int, int]]:
    """(field, address, bytes the launch touches from that address) for the operands whose extent the struct itself states."""
    f32 = 4
    if isinstance(st, N.SeaGemmGroup):
        M, Nn, K = st.M, st.N, st.K
        if st.A:
            yield "A", st.A, (((M - 1) * st.lda + K) + (st.n_seg - 1) * st.a_seg_stride) * esz
        if st.W:
            yield "W", st.W, ((Nn - 1) * st.ldw + K) * esz
        if st.bias:
            yield "bias", st.bias, Nn * f32
        if st.R:
            yield "R", st.R, ((M - 1) * st.ldr + Nn) * f32
        if st.C32:
            yield "C32", st.C32, ((M - 1) * st.ldc32 + Nn) * f32
        if st.Cact:
            yield "Cact", st.Cact, ((M - 1) * st.ldcact + Nn) * esz
        if st.Z:
            yield "Z", st.Z, ((M - 1) * st.ldz + Nn) * esz
        if st.silu_c:
            yield "silu_c", st.silu_c, M * f32
    elif isinstance(st, N.SeaQkvGroup):
        yield "A", st.A, ((st.M - 1) * st.lda + st.K) * esz
        yield "W", st.W, ((st.N - 1) * st.ldw + st.K) * esz
        yield "bias", st.bias, st.N * f32
    elif isinstance(st, N.SeaGemmNormGroup):
        if st.A:
            yield "A", st.A, (((st.M - 1) * st.lda + st.K) + (st.n_seg - 1) * st.a_seg_stride) * esz
        if st.W:
            yield "W", st.W, ((st.N - 1) * st.ldw + st.K) * esz
        if st.Yact:
            yield "Yact", st.Yact, ((st.M - 1) * st.ldyact + st.N) * esz
        if st.Y32:
            yield "Y32", st.Y32, ((st.M - 1) * st.ldy32 + st.N) * f32
        if st.mod:
            yield "mod", st.mod, ((st.M - 1) * st.ldmod + 2 * st.N) * esz
    elif isinstance(st, N.SeaAdalnGroup):
        yield "A", st.A, ((st.M - 1) * st.lda + st.K) * esz
        yield "W", st.W, ((2 * st.d - 1) * st.ldw + st.K) * esz
        if st.bias:
            yield "bias", st.bias, 2 * st.d * f32
        if st.X:
            yield "X", st.X, ((st.M - 1) * st.ldx + st.d) * f32
            yield "gamma", st.gamma, st.d * f32
        if st.Yact:
            yield "Yact", st.Yact, ((st.M - 1) * st.ldyact + (st.d if st.X else 2 * st.d)) * esz
        if st.Y32:
            yield "Y32", st.Y32, ((st.M - 1) * st.ldy32 + st.d) * f32
    elif isinstance(st, N.SeaSplitkGroup):
        yield "P", st.P, ((st.S - 1) * st.p_stride + (st.M - 1) * st.ldp + st.N) * f32
        if st.bias:
            yield "bias", st.bias, st.N * f32
        if st.R:
            yield "R", st.R, ((st.M - 1) * st.ldr + st.N) * f32
        if st.C32:
            yield "C32", st.C32, ((st.M - 1) * st.ldc32 + st.N) * f32
        if st.Cact:
            yield "Cact", st.Cact, ((st.M - 1) * st.ldcact + st.N) * esz
    elif isinstance(st, N.SeaAdalnQkv):
        E = st.E
        yield "X", st.X, ((st.M - 1) * st.ldx + E) * f32
        yield "cond", st.cond, st.M * f32
        yield "w1", st.w1, 2 * E * f32
        yield "b1", st.b1, 2 * E * f32
        yield "W2c", st.W2c, ((2 * E - 1) * st.ldw2c + 2 * E) * esz
        if st.b2c:
            yield "b2c", st.b2c, 2 * E * f32
        yield "gamma", st.gamma, E * f32
        yield "Wqkv", st.Wqkv, ((3 * E - 1) * st.ldw + E) * esz
        if st.bqkv:
            yield "bqkv", st.bqkv, 3 * E * f32
        if st.W3:
            yield "w13", st.w13, st.N3 * f32
            yield "b13", st.b13, st.N3 * f32
            yield "W3", st.W3, ((st.N3 - 1) * st.ldw3 + st.N3) * esz
            yield "mod3", st.mod3, ((st.M - 1) * st.ldmod3 + st.N3) * esz
    elif isinstance(st, N.SeaRowChain):
        K2 = st.D if st.n_seg > 0 else st.E
        for s_ in range(st.n_seg):
            yield f"att[{s_}]", st.att[s_], ((st.M - 1) * st.ldatt + st.D) * esz
            yield f"Wp[{s_}]", st.Wp[s_], ((st.D - 1) * st.ldwp + st.D) * esz
        if st.a2:
            yield "a2", st.a2, ((st.M - 1) * st.lda2 + st.E) * esz
        yield "W2", st.W2, ((st.E - 1) * st.ldw2 + K2) * esz
        if st.b2:
            yield "b2", st.b2, st.E * f32
        yield "Xin", st.Xin, ((st.M - 1) * st.ldxin + st.E) * f32
        yield "X", st.X, ((st.M - 1) * st.ldx + st.E) * f32
        if st.Xact:
            yield "Xact", st.Xact, ((st.M - 1) * st.ldxact + st.E) * esz
    elif isinstance(st, N.SeaMlpGroup):
        if st.A:
            yield "A", st.A, ((st.M - 1) * st.lda + st.E) * esz
        if st.X32:
            yield "X32", st.X32, ((st.M - 1) * st.ldx32 + st.E) * f32
        if st.addend:
            yield "addend", st.addend, ((st.M - 1) * st.ldadd + st.E) * f32
        if st.Xout:
            yield "Xout", st.Xout, ((st.M - 1) * st.ldxout + st.E) * f32
        if st.mod:
            yield "mod", st.mod, ((st.M - 1) * st.ldmod + 2 * st.E) * esz
        yield "W1", st.W1, ((st.S - 1) * st.ldw + st.E) * esz
        yield "Hg", st.Hg, ((st.M - 1) * st.ldh + st.S) * esz
        for k in ("b1", "lnw", "lnb"):
            yield k, getattr(st, k), st.S * f32
    elif isinstance(st, N.SeaMlp2Group):
        yield "Hg", st.Hg, ((st.M - 1) * st.ldh + st.S) * esz
        yield "W2", st.W2, ((st.E - 1) * st.ldw2 + st.S) * esz
        yield "Wproj", st.Wproj, ((st.E - 1) * st.ldwp + st.E) * esz
        yield "R", st.R, ((st.M - 1) * st.ldr + st.E) * f32
        if st.Y32:
            yield "Y32", st.Y32, ((st.M - 1) * st.ldy32 + st.E) * f32
        if st.Yact:
            yield "Yact", st.Yact, ((st.M - 1) * st.ldyact + st.E) * esz
        if st.mod:
            yield "mod", st.mod, ((st.M - 1) * st.ldmod + 2 * st.E) * esz


def _walk_structs(obj) -> Iterator[C.Structure]:
    if isinstance(obj, _HOST_STRUCTS):
        return
    if isinstance(obj, C.Structure):
        yield obj
        for name, _ in obj._fields_:
            v = getattr(obj, name)
            if isinstance(v, (C.Structure, C.Array)):
                yield from _walk_structs(v)
    elif isinstance(obj, C.Array):
        for i in range(len(obj)):
            e = obj[i]
            if isinstance(e, (C.Structure, C.Array)):
                yield from _walk_structs(e)
    elif isinstance(obj, (tuple, list)):
        for e in obj:
            yield from _walk_structs(e)


def check_records(records, ranges: Ranges, esz: int, what: str) -> int:
    """Audit every launch record; raises RuntimeError naming the record, field and address of the first violation.  Returns the number of pointers seen."""
    n = 0
    for r in records:
        if r.fn is None:
            continue
        # the argument structs of a record: arrays in its argument list, or — where the list holds C.byref(struct) — the struct kept beside it
        roots = [a for a in r.args if isinstance(a, (C.Structure, C.Array))]
        if r.keep is not None:
            roots.append(r.keep)
        for root in roots:
            for st in _walk_structs(root):
                for path, p in iter_pointers(st, type(st).__name__):
                    n += 1
                    if ranges.find(p) is None:
                        raise RuntimeError(f"sea_amd pointer audit ({what}): launch '{r.name}': {path} = {p:#x} lies in no buffer this plan knows "
                                           "(stale address, unpatched field or a tensor that was freed)")
                for field, p, nbytes in _extents(st, esz):
                    if not p:
                        continue
                    hit = ranges.find(p)
                    if hit is not None and nbytes > 0 and p + nbytes > hit[1]:
                        raise RuntimeError(f"sea_amd pointer audit ({what}): launch '{r.name}': {type(st).__name__}.{field} = {p:#x} + {nbytes} bytes runs "
                                           f"{p + nbytes - hit[1]} bytes past the end of its buffer ({hit[2]})")
        # raw integer addresses kept in the Python argument list (sea_convert_f32_to_act, the silu launch's condition pointer)
        for a in r.args:
            if isinstance(a, int) and a >= (1 << 32) and ranges.find(a) is None and r.fn is not None and r.name not in ("",):
                if _looks_like_device_pointer(a):
                    raise RuntimeError(f"sea_amd pointer audit ({what}): launch '{r.name}': raw argument {a:#x} lies in no buffer this plan knows")
    return n


def _looks_like_device_pointer(a: int) -> bool:
    # device allocations on this platform sit above 4 GiB and are at least 4-byte aligned; sizes / strides passed as plain ints are far smaller
    return a >= (1 << 40) and a % 4 == 0
