import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    path = os.path.join(GOLDEN, name + ".npz")
    if not os.path.exists(path):
        pytest.skip(f"golden fixture {name}.npz not present")
    return np.load(path, allow_pickle=False)


def cfg_from_meta(meta):
    from oracle.sea_oracle import OracleConfig

    m = [int(v) for v in meta]
    return OracleConfig(num_layers=m[0], embed_dim=m[1], n_heads=m[2], max_len=m[3], scale_ratio=m[4], src_len=m[5],
                        num_variables=m[6], down_proj=m[7], add_info_after_cross=bool(m[8]),
                        LN_type="adaln" if m[9] else "ln",
                        exchange_mode=("sea", "addition", "simple", "pool")[m[10]] if len(m) > 10 else "sea",
                        ib_addition_mode=("add", "none", "attention", "concat")[m[11]] if len(m) > 11 else "add",
                        ib_scale_mode=("mlp", "linear", "fourier")[m[12]] if len(m) > 12 else "mlp")


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def grad_err(a, b, tiny=1e-7):
    """rel_l2 for gradients, except where the reference's gradient is numerically ZERO (norm < tiny: e.g. the key bias of an un-masked, un-rotated
    attention — a constant added to every key leaves the softmax unchanged — where both sides hold rounding noise): there the ABSOLUTE difference is
    returned on a 1e-3 scale, so the callers' 1e-4-style bounds ask for |a - b| < 1e-7."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    nb = np.linalg.norm(b)
    if nb < tiny:
        return float(np.linalg.norm(a - b) / 1e-3)
    return float(np.linalg.norm(a - b) / nb)


def grad_sub_stride(numel, cap=384):
    """Stride of the committed sub-sample of a flattened cfg3-shaped gradient (tests/golden/make_fixtures.py::sub_stride)."""
    return max(1, numel // cap) | 1
