"""Spatial decoder on the MI355X: sea_amd.models.encoder_decoder.Decode and the decode leg of the rollout
evaluation against the reference's golden vectors and the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import sea_oracle as O
from oracle.recipe import decode_params
from tests.conftest import load_golden, rel_l2

pytestmark = pytest.mark.gpu


def _groups(sizes):
    groups, k = [], 0
    for sz in sizes:
        groups.append(list(range(k, k + sz)))
        k += sz
    return groups


def _build(groups, n_inp, hidden, D, dtype):
    from sea_amd.models.encoder_decoder import Decode

    m = Decode(groups, n_inp, hidden, D)
    p = decode_params(groups, n_inp, hidden, D)
    assert [k for k, _ in m.named_parameters()] == list(p.keys())  # the reference's state_dict schema
    with torch.no_grad():
        for k, prm in m.named_parameters():
            prm.copy_(p[k])
    return m.set_compute_dtype(dtype).to("cuda:0").eval(), p


@pytest.mark.parametrize("name", ["decode_cyl_small", "decode_three_groups"])
@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-4), ("bf16", 2e-2)])
def test_decode_matches_reference_golden(name, dtype, tol):
    from sea_amd.utils.train_utils import decode_rollout, inverse_transform_processed_data

    g = load_golden(name)
    n_inp, hidden, D, P, tr, T = (int(v) for v in g["dims"])
    groups = _groups([int(v) for v in g["groups"]])
    m, _ = _build(groups, n_inp, hidden, D, dtype)
    roll = torch.from_numpy(g["roll"]).to("cuda:0")
    z = inverse_transform_processed_data(roll, tr, T, P, len(groups))
    assert torch.equal(z.cpu(), torch.from_numpy(g["z"]))
    with torch.no_grad():
        out = decode_rollout(m, roll, P)
    assert out.shape == g["out"].shape
    assert rel_l2(out.cpu().numpy(), g["out"]) < tol


def test_decode_cylinder_dims_against_oracle():
    """The shipped cylinder dims (64 patches, spatial embed 16, hidden 480, groups [[0,1],[2]]), ragged row count, weights updated in
    place between calls (the act-dtype shadow must follow)."""
    groups, n_inp, hidden, D, P = [[0, 1], [2]], 132, 480, 16, 64
    m, p = _build(groups, n_inp, hidden, D, "fp32")
    rng = np.random.Generator(np.random.PCG64(5))
    z = torch.from_numpy(rng.standard_normal((7, P, 2, D)).astype(np.float32))
    with torch.no_grad():
        out = m(z.cuda())
    assert rel_l2(out.cpu().numpy(), O.decode(z, p, groups).numpy()) < 1e-5
    with torch.no_grad():
        m.decoders[1].layer2.weight.mul_(0.5)
        p["decoders.1.layer2.weight"] = p["decoders.1.layer2.weight"] * 0.5
        out2 = m(z.cuda())
    assert rel_l2(out2.cpu().numpy(), O.decode(z, p, groups).numpy()) < 1e-5
    m.set_compute_dtype("bf16")
    with torch.no_grad():
        out3 = m(z.cuda())
    assert rel_l2(out3.cpu().numpy(), O.decode(z, p, groups).numpy()) < 2e-2


def test_decode_rejects_cpu_and_bad_dims():
    from sea_amd.models.encoder_decoder import Decode

    with pytest.raises(NotImplementedError):
        Decode([[0]], 12, 60, 16)  # MLP_hidden not a multiple of 8
    # any padded cell size works (it is data-dependent in the reference): 10 columns per field, padded to 12 inside, sliced on the way out
    groups = [[0, 1], [2]]
    m10 = Decode(groups, 10, 64, 16).to("cuda:0").eval()
    z = torch.randn(2, 5, 2, 16)
    from oracle import sea_oracle as O
    p = {k: v.detach().cpu() for k, v in m10.state_dict().items()}
    with torch.no_grad():
        out = m10(z.cuda())
    assert out.shape == (2, 5, 3, 10)
    assert rel_l2(out.cpu().numpy(), O.decode(z, p, groups).numpy()) < 1e-4
    m = Decode([[0]], 12, 64, 16)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 4, 1, 16))  # CPU tensor: no fallback
