"""Spatial encoder on the MI355X: sea_amd.models.encoder_decoder.PointwiseEncode / SpatialModel against the reference's
golden vectors and the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import sea_oracle as O
from oracle.recipe import decode_params, encoder_params
from tests.conftest import load_golden, rel_l2

pytestmark = pytest.mark.gpu


def _groups(sizes):
    groups, k = [], 0
    for sz in sizes:
        groups.append(list(range(k, k + int(sz))))
        k += int(sz)
    return groups


def _build(groups, n_inp, hidden, layers, D, H, dtype):
    from sea_amd.models.encoder_decoder import PointwiseEncode

    m = PointwiseEncode(groups, n_inp, hidden, layers, D, H, 64, 0, dropout=0.0)
    p = encoder_params(groups, n_inp, hidden, layers, D)
    assert [k for k, _ in m.named_parameters()] == list(p.keys())   # the reference's state_dict schema (parameters)
    with torch.no_grad():
        for k, prm in m.named_parameters():
            prm.copy_(p[k])
    return m.set_compute_dtype(dtype).to("cuda:0").eval(), p


@pytest.mark.parametrize("name", ["encode_cyl_small", "encode_three_groups"])
@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-4), ("bf16", 3e-2)])
def test_encode_matches_reference_golden(name, dtype, tol):
    g = load_golden(name)
    n_inp, hidden, layers, D, H, P, B = (int(v) for v in g["dims"])
    groups = _groups(g["groups"])
    m, _ = _build(groups, n_inp, hidden, layers, D, H, dtype)
    assert torch.equal(m.spatial_pos_encoder.pe[0, :P].cpu(), torch.from_numpy(g["pe"]))
    with torch.no_grad():
        z = m(torch.from_numpy(g["x"]).cuda())
    assert z.shape == g["z"].shape
    assert rel_l2(z.cpu().numpy(), g["z"]) < tol


def test_encode_shipped_cylinder_dims_against_oracle_and_chunking():
    """configs/cylinder_flow.py spatial dims: 64 patches, groups [[0,1],[2]], embed 16 (width 32, 8 heads -> head dim 4, padded to 8 on the device),
    hidden 480, 12 layers, a cell size that is not a multiple of 8 (132).  Chunked passes must equal the single pass bit for bit."""
    groups, n_inp, hidden, layers, D, H, P = [[0, 1], [2]], 132, 480, 12, 16, 8, 64
    m, p = _build(groups, n_inp, hidden, layers, D, H, "fp32")
    rng = np.random.Generator(np.random.PCG64(9))
    x = torch.from_numpy(rng.standard_normal((5, P, 3, n_inp)).astype(np.float32))
    ref = O.encode(x, p, groups, H, layers)
    with torch.no_grad():
        z = m(x.cuda())
        m.CHUNK = 2
        z2 = m(x.cuda())
    assert rel_l2(z.cpu().numpy(), ref.numpy()) < 1e-4
    assert torch.equal(z, z2)
    m.set_compute_dtype("bf16")
    with torch.no_grad():
        z3 = m(x.cuda())
    assert rel_l2(z3.cpu().numpy(), ref.numpy()) < 3e-2


def test_spatial_model_encode_decode_chain():
    """SpatialModel.forward = generate_padding_mask -> encode -> decode, against the oracle's encode and decode on the same parameters."""
    from sea_amd.models.encoder_decoder import SpatialModel

    groups, n_inp, hidden, layers, D, H, P = [[0, 1], [2]], 24, 96, 3, 16, 8, 16
    sm = SpatialModel(groups, n_inp, hidden, layers, D, H, 64, 0, dropout=0.0)
    pe, pd = encoder_params(groups, n_inp, hidden, layers, D), decode_params(groups, n_inp, hidden, D)
    with torch.no_grad():
        for k, prm in sm.encode.named_parameters():
            prm.copy_(pe[k])
        for k, prm in sm.decode.named_parameters():
            prm.copy_(pd[k])
    sm = sm.to("cuda:0").eval()
    rng = np.random.Generator(np.random.PCG64(10))
    x = torch.from_numpy(rng.standard_normal((4, P, 3, n_inp)).astype(np.float32))
    x[0, 0, 0, :5] = -9999.0
    xr = x.clone()
    xr[xr == -9999.0] = 0.0
    ref = O.decode(O.encode(xr, pe, groups, H, layers), pd, groups)
    with torch.no_grad():
        out = sm(x.cuda())
    assert out.shape == (4, P, 3, n_inp)
    assert rel_l2(out.cpu().numpy(), ref.numpy()) < 1e-4


def test_encode_rejects_cpu_and_bad_dims():
    from sea_amd.models.encoder_decoder import PointwiseEncode, SpatialModel

    with pytest.raises(NotImplementedError):
        PointwiseEncode([[0, 2], [1]], 12, 48, 1, 16, 8, 64, 0)   # group not a run of consecutive fields
    with pytest.raises(NotImplementedError):
        SpatialModel([[0]], 12, 48, 1, 16, 4, 64, 0, variational=True)
    m = PointwiseEncode([[0, 1], [2]], 12, 48, 1, 16, 8, 64, 0)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 4, 3, 12))  # CPU tensor: no fallback
