"""sea_amd/ptrcheck.py on the CPU: the walker over the ABI's argument structs (include/sea_hip.h) finds every pointer field — plain, arrays of
pointers, nested structs, structs passed by reference beside the argument list — and the extent check prices an operand from the struct's own
sizes.  (On the GPU every plan runs this audit at its first bind: tests/test_model_gpu.py::test_pointer_audit_*.)"""
import ctypes as C

import pytest

from sea_amd import _native as N
from sea_amd import ptrcheck

BASE = 1 << 41


class Rec:
    def __init__(self, args, keep=None, name="launch"):
        self.fn, self.args, self.keep, self.name = object(), args, keep, name


def _ranges():
    r = ptrcheck.Ranges()
    r.add_range(BASE, 4096, "buffer")
    return r


def test_known_pointers_pass_and_are_counted():
    g = (N.SeaGemmGroup * 1)()
    g[0].A, g[0].W, g[0].M, g[0].N, g[0].K, g[0].lda, g[0].ldw, g[0].n_seg = BASE, BASE + 1024, 4, 4, 8, 8, 8, 1
    assert ptrcheck.check_records([Rec([g, 1, 1], g)], _ranges(), 2, "t") >= 2


def test_operand_running_past_its_buffer_is_refused():
    g = (N.SeaGemmGroup * 1)()
    g[0].A, g[0].W, g[0].M, g[0].N, g[0].K, g[0].lda, g[0].ldw, g[0].n_seg = BASE, BASE + 1024, 400, 4, 8, 8, 8, 1
    with pytest.raises(RuntimeError, match=r"SeaGemmGroup\.A .* past the end"):
        ptrcheck.check_records([Rec([g, 1, 1], g)], _ranges(), 2, "t")


@pytest.mark.parametrize("make,field", [
    (lambda: _attn(), r"SeaAttnProblem\.K"),                      # a struct nested in an array inside a struct passed by reference
    (lambda: _ib(), r"SeaIbParams\.X\[2\]"),                      # an array of pointers
    (lambda: _xtail(), r"SeaGemmNormGroup\.W"),                   # a nested struct
    (lambda: _qkv(), r"SeaQkvCommon\.rope"),                      # a struct that only the record's `keep` holds
])
def test_unknown_pointer_is_refused_wherever_it_sits(make, field):
    with pytest.raises(RuntimeError, match=field + r" = 0x[0-9a-f]+ lies in no buffer"):
        ptrcheck.check_records([make()], _ranges(), 2, "t")


def _attn():
    P = N.SeaAttnParams()
    P.p[0].Q, P.p[1].K = BASE + 8, 5 << 42
    return Rec([C.byref(P), 1], P)


def _ib():
    ib = N.SeaIbParams()
    ib.X[0], ib.X[2] = BASE, 5 << 42
    return Rec([C.byref(ib)], ib)


def _xtail():
    x = (N.SeaExchangeTail * 1)()
    x[0].X, x[0].down.W = BASE, 5 << 42
    return Rec([x, 1, 1e-5, 1], x)


def _qkv():
    common = N.SeaQkvCommon()
    common.rope = 5 << 42
    q = (N.SeaQkvGroup * 1)()
    return Rec([q, 1, C.byref(common), 1], (q, common))


def test_host_side_structs_are_skipped_and_null_is_fine():
    rec = (N.SeaLaunchRec * 2)()
    rec[0].p0 = 0x1234   # a HOST address: never audited
    g = (N.SeaNormGroup * 1)()   # all-NULL
    assert ptrcheck.check_records([Rec([rec, 2]), Rec([g, 1, 8, 8, 0, 0, 1e-5, 1], g)], _ranges(), 2, "t") == 0
    assert not ptrcheck.always()
