"""Un-patchify + inverse scaling on the MI355X: sea_unpatchify through sea_amd.utils.data_processors against the
reference's golden vectors (DataPartitioner2D / MinMaxScaler of the reference) and the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import sea_oracle as O
from tests.conftest import load_golden, rel_l2

pytestmark = pytest.mark.gpu


def _groups(sizes):
    groups, k = [], 0
    for sz in sizes:
        groups.append(list(range(k, k + sz)))
        k += sz
    return groups


@pytest.mark.parametrize("name", ["unpatch_5x5", "unpatch_3x4"])
def test_partition_and_unpatch_match_reference_golden(name):
    from sea_amd.utils.data_processors import DataPartitioner2D, MeshUnpatcher, MinMaxScaler

    g = load_golden(name)
    m, n = (int(v) for v in g["mn"])
    groups = _groups([int(v) for v in g["groups"]])
    xy = torch.from_numpy(g["xy"])
    part = DataPartitioner2D(xy[0], xy[1], m=m, n=n, pad_id=-1, pad_field_value=0, device="cuda:0")
    # the index map (cells row-major, points ascending, -1 padding) is the reference's, bit for bit
    assert torch.equal(part.padded_index_map.cpu().long(), torch.from_numpy(g["index_map"]))
    fields = torch.from_numpy(g["fields"]).cuda()
    stacked = part.create_partitions([fields[:, :, i] for i in range(fields.shape[2])])
    assert torch.equal(stacked.cpu(), torch.from_numpy(g["stacked"]))
    # plain scatter: exact round trip; both argument layouts
    plain = MeshUnpatcher(part, groups)
    assert torch.equal(plain.inverse_scale_and_unpatch(stacked).cpu(), torch.from_numpy(g["fields"]))
    assert torch.equal(MeshUnpatcher(part, groups, gather=False).inverse_scale_and_unpatch(stacked).cpu(), torch.from_numpy(g["fields"]))   # scatter form
    assert torch.equal(plain.inverse_scale_and_unpatch(stacked.permute(0, 1, 3, 2).contiguous(), layout="BPFC").cpu(), torch.from_numpy(g["fields"]))
    # with the inverse MinMax transform
    scalers = []
    for r0, r1, lo, hi in g["scaler_params"]:
        sc = MinMaxScaler(feature_range=(r0, r1))
        sc.min_val, sc.max_val = torch.tensor(lo), torch.tensor(hi)
        scalers.append(sc)
    out = MeshUnpatcher(part, groups, scalers).inverse_scale_and_unpatch(stacked)
    assert rel_l2(out.cpu().numpy(), g["unscaled"]) < 1e-6
    # forward direction (sea_patchify): plain = the torch-indexing partition bit for bit; scaled against the reference; both layouts; row padding
    assert torch.equal(plain.patchify_and_scale(fields).cpu(), torch.from_numpy(g["stacked"]))
    mu = MeshUnpatcher(part, groups, scalers)
    fwd = mu.patchify_and_scale(fields)
    assert rel_l2(fwd.cpu().numpy(), g["scaled_stacked"]) < 1e-6
    C = stacked.shape[2]
    enc_in = mu.patchify_and_scale(fields, layout="BPFC", c_out=C + 3)                      # the encoder's [T, P, F, n_inp] input, cells padded to n_inp
    assert torch.equal(enc_in[..., :C].permute(0, 1, 3, 2), fwd) and float(enc_in[..., C:].abs().max()) == 0.0
    pad = torch.from_numpy(g["index_map"] < 0)
    assert float(fwd.cpu()[:, pad].abs().max()) == 0.0 if pad.any() else True               # padded slots hold pad_field_value, unscaled
    back = mu.inverse_scale_and_unpatch(fwd)                                                # scale -> partition -> un-partition -> un-scale = identity
    assert rel_l2(back.cpu().numpy(), g["fields"]) < 1e-6


def test_decode_then_unpatch_chain_against_oracle():
    """rollout output -> Decode -> un-patchify, all on the device, vs the oracle chain (cylinder-like: 64 patches, ragged cells)."""
    from oracle.recipe import decode_params
    from sea_amd.models.encoder_decoder import Decode
    from sea_amd.utils.data_processors import DataPartitioner2D, MeshUnpatcher, MinMaxScaler
    from sea_amd.utils.train_utils import decode_rollout

    rng = np.random.Generator(np.random.PCG64(3))
    npts, m, n, D, hidden, tr, T = 3000, 9, 9, 16, 96, 1, 4
    groups = [[0, 1], [2]]
    xy = torch.from_numpy(rng.random((2, npts)).astype(np.float32))
    part = DataPartitioner2D(xy[0], xy[1], m=m, n=n, device="cuda:0")
    P, C = part.padded_index_map.shape
    n_inp = (C + 3) // 4 * 4          # decoder output width per field: the padded cell size rounded to the kernel's 4-column granule
    dec = Decode(groups, n_inp, hidden, D)
    p = decode_params(groups, n_inp, hidden, D)
    with torch.no_grad():
        for k, prm in dec.named_parameters():
            prm.copy_(p[k])
    dec = dec.to("cuda:0").eval()
    roll = torch.from_numpy(rng.standard_normal((tr, T, len(groups), P * D)).astype(np.float32))
    scalers = []
    for lo, hi in ((-1.0, 3.0), (0.5, 2.0)):
        sc = MinMaxScaler()
        sc.min_val, sc.max_val = torch.tensor(lo), torch.tensor(hi)
        scalers.append(sc)
    with torch.no_grad():
        decoded = decode_rollout(dec, roll.cuda(), P)                                   # [T, P, 3, n_inp]
        out = MeshUnpatcher(part, groups, scalers).inverse_scale_and_unpatch(decoded[..., :C], layout="BPFC")
    ref_dec = O.decode(O.rollout_to_patches(roll, P), p, groups)[..., :C].permute(0, 1, 3, 2)
    ref = O.unpatchify(ref_dec, part.padded_index_map.cpu().long(), npts, groups, [(-1.0, 1.0, -1.0, 3.0), (-1.0, 1.0, 0.5, 2.0)])
    assert out.shape == (tr * T, npts, 3)
    assert rel_l2(out.cpu().numpy(), ref.numpy()) < 1e-5
    # the same fields with the decoder run only over the columns the un-patchify reads (patches sorted by cell size, a few buckets): bit for bit
    from sea_amd.utils.train_utils import inverse_transform_processed_data

    mu = MeshUnpatcher(part, groups, scalers)
    with torch.no_grad():
        z = inverse_transform_processed_data(roll.cuda(), tr, T, P, len(groups))
        fused = mu.decode_and_unpatch(dec, z)
    assert torch.equal(fused, out)
    order, buckets, _, _ = mu._prefix_plan(5)
    counts = (part.padded_index_map != part.pad_id).sum(dim=1)[order].cpu().tolist()
    assert buckets[0][0] == 0 and buckets[-1][1] == P and all(a[1] == b[0] for a, b in zip(buckets, buckets[1:]))
    assert all(max(counts[lo:hi]) == n for lo, hi, n in buckets) and len(buckets) <= 5
