"""CPU-side checks of the C-ABI library: it builds, loads, exports every symbol include/sea_hip.h declares, the ctypes
struct layouts match the C ones, and argument validation fails loudly without touching a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from sea_amd import build, _native

    build.build(verbose=False)
    return _native.lib()


def declared_functions():
    text = open(os.path.join(ROOT, "include", "sea_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sea_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    names = declared_functions()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/sea_hip.h but not exported by libsea_hip.so"


def test_binding_lists_every_declared_symbol():
    from sea_amd import _native

    assert set(declared_functions()) == set(_native.EXPORTED_SYMBOLS)


def test_struct_layouts_match(lib):
    from sea_amd import _native as N

    out = (C.c_int * 32)()
    n = lib.sea_struct_sizes(out, 32)
    mine = [C.sizeof(t) for t in N.ABI_STRUCTS]
    assert n == len(mine)
    assert list(out[:n]) == mine


def test_validation_without_gpu(lib):
    from sea_amd import _native as N

    g = (N.SeaGemmGroup * 1)()
    assert lib.sea_gemm_grouped(g, 1, 0, None) == -1  # null operands
    assert b"null operand" in lib.sea_last_error()
    assert lib.sea_gemm_grouped(g, 17, 0, None) == -1
    P = N.SeaAttnParams()
    assert lib.sea_attention_fwd(C.byref(P), 1, None) == -1
    assert lib.sea_abi_version() == 1


def test_cpu_tensors_are_refused():
    import torch
    from sea_amd import ops

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.convert(torch.zeros(4, 4), torch.zeros(4, 4))
