"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE on CPU.

Runs only in the build container (it imports /root/reference, which never travels to the GPU
box); the .npz files it writes are committed.  Fixtures are data only: inputs, expected outputs,
and the names of parameters without gradients.  Weights are not stored — they are regenerated
from parameter names by oracle/recipe.py on both sides.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_fixtures.py [--only NAME] [--big]
"""
import argparse
import os
import sys

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import numpy as np
import torch

from oracle.recipe import decode_schema, encoder_schema, param_schema, recipe_inputs, recipe_tensor
from oracle.sea_oracle import OracleConfig

from models.temporal import TemporalModel  # reference
from models import base_blocks as ref_bb  # reference
from utils import train_utils as ref_tu  # reference
from models.encoder_decoder import Decode as RefDecode  # reference
from models.encoder_decoder import PointwiseEncode as RefEncode  # reference
from utils import data_processors as ref_dp  # reference

torch.set_num_threads(8)


def build_reference(cfg: OracleConfig, dropout=0.0):
    m = TemporalModel(cfg.num_layers, cfg.embed_dim, cfg.n_heads, cfg.max_len, cfg.scale_ratio, cfg.src_len,
                      cfg.num_variables, cfg.down_proj, dropout, cfg.exchange_mode, "learnable", cfg.ib_scale_mode, cfg.ib_addition_mode, 1, 1,
                      cfg.add_info_after_cross, cfg.LN_type)
    schema = param_schema(cfg)
    named = dict(m.named_parameters())
    assert list(named.keys()) == list(schema.keys()), (
        "schema/key-order mismatch", [k for k in named if k not in schema], [k for k in schema if k not in named])
    with torch.no_grad():
        for k, prm in named.items():
            shp, kind = schema[k]
            assert tuple(prm.shape) == tuple(shp), (k, prm.shape, shp)
            prm.copy_(torch.from_numpy(recipe_tensor(k, shp, kind)))
    return m


def cfg_meta(cfg: OracleConfig):
    return np.array([cfg.num_layers, cfg.embed_dim, cfg.n_heads, cfg.max_len, cfg.scale_ratio, cfg.src_len,
                     cfg.num_variables, cfg.down_proj, int(cfg.add_info_after_cross),
                     1 if cfg.LN_type == "adaln" else 0, ("sea", "addition", "simple", "pool").index(cfg.exchange_mode),
                     ("add", "none", "attention", "concat").index(cfg.ib_addition_mode), ("mlp", "linear", "fourier").index(cfg.ib_scale_mode)], dtype=np.int64)


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez(path, **arrs)
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def n(t):
    return t.detach().cpu().numpy().copy()  # copy: parameters are updated in place later


def model_case(name, cfg, B, T, train=False, lr=1e-3, seed=1234):
    print(name)
    m = build_reference(cfg)
    x, tgt, ib = recipe_inputs(B, T, cfg, seed)
    m.eval()
    with torch.no_grad():
        out = m(x, ib)
    arrs = dict(cfg=cfg_meta(cfg), x=n(x), tgt=n(tgt), ib=n(ib), out=n(out))
    if train:
        m.train()
        loss_fn = torch.nn.MSELoss()
        opt = ref_tu.initialize_optimizer(m, {"learning_rate": lr})
        losses = []
        for step in range(1, 4):
            opt.zero_grad()
            o = m(x, ib)
            loss = loss_fn(o, tgt)
            loss.backward()
            if step == 1:
                dead = [k for k, p in m.named_parameters() if p.grad is None]
                arrs["dead_keys"] = np.array(dead)
                arrs["out_train"] = n(o)
                for k, p in m.named_parameters():
                    if p.grad is not None:
                        arrs["grad:" + k] = n(p.grad)
            opt.step()
            losses.append(loss.item())
            if train == "grads":   # gradients, loss and the set of gradient-less parameters only (the ablation variants: small fixtures)
                break
            if step in (1, 3):
                for k, p in m.named_parameters():
                    if p.grad is not None:
                        arrs[f"param{step}:" + k] = n(p)
        arrs["losses"] = np.array(losses, dtype=np.float64)
        arrs["lr"] = np.array(lr)
        n_state = sum(1 for p in m.parameters() if p in opt.state)
        arrs["n_opt_state"] = np.array(n_state)
    save(name, **arrs)


def rollout_case(name, cfg, B, n_steps, seed=99):
    print(name)
    m = build_reference(cfg)
    m.eval()
    x, tgt, ib = recipe_inputs(B, n_steps, cfg, seed)
    # the loop of utils/train_utils.py:202-209, driven here so that the trajectory can be captured
    with torch.no_grad():
        a = x[:, 0:1]
        for i in range(n_steps):
            o = m(a, ib[:, : i + 1])
            a = torch.cat((a, o[:, -1:]), dim=1)
        pred = a[:, 1:]
        rel = ref_tu.relativeMSE(pred, tgt).mean()
    # the reference's own loop (first sample only) for its scalars
    loader = [(x, tgt, None, ib)]
    v_loss, v_rel = ref_tu.autoregressive_validation(m, loader, torch.nn.MSELoss(), torch.device("cpu"))
    with torch.no_grad():
        chk = torch.nn.functional.mse_loss(pred[0:1], tgt[0:1]).item()
    assert abs(chk - v_loss) <= 1e-6 * max(1.0, abs(chk)), (chk, v_loss)
    save(name, cfg=cfg_meta(cfg), x0=n(x[:, 0:1]), tgt=n(tgt), ib=n(ib), pred=n(pred),
         rel_mse=np.array(rel.item()), val_loss=np.array(v_loss), val_rel_mse_time=np.array(v_rel))


def module_cases():
    print("modules")
    rng = np.random.Generator(np.random.PCG64(7))
    arrs = {}

    def rnd(*shape):
        return torch.from_numpy(rng.standard_normal(shape).astype(np.float32))

    def kinds(mod, prefix):
        # store the (key, kind) list so the test can rebuild identical weights
        out = []
        for k, p in mod.named_parameters():
            if k.endswith("weight") and p.dim() == 2:
                kind = "lin_w"
            elif k.endswith("bias") and p.dim() == 1 and _is_linear_bias(mod, k):
                kind = "lin_b"
            elif k.endswith("weight"):
                kind = "norm_w"
            else:
                kind = "norm_b"
            out.append((prefix + k, kind, tuple(p.shape)))
        return out

    def _is_linear_bias(mod, k):
        sub = mod.get_submodule(k.rsplit(".", 1)[0]) if "." in k else mod
        return isinstance(sub, torch.nn.Linear)

    def load(mod, prefix):
        ks = kinds(mod, prefix)
        with torch.no_grad():
            for (key, kind, shp), (_, p) in zip(ks, mod.named_parameters()):
                p.copy_(torch.from_numpy(recipe_tensor(key, shp, kind)))
        arrs["kinds:" + prefix] = np.array([f"{k}|{kind}|{','.join(map(str, shp))}" for k, kind, shp in ks])

    for d in (64, 128):
        mod = ref_bb.AdaLN(d, 1)
        load(mod, f"adaln{d}.")
        x, c = rnd(2, 5, d) * 1.7 + 0.3, torch.from_numpy(rng.random((2, 5, 1)).astype(np.float32))
        arrs[f"adaln{d}.x"], arrs[f"adaln{d}.c"], arrs[f"adaln{d}.y"] = n(x), n(c), n(mod(x, c))
    mod = ref_bb.LayerNorm(64)
    load(mod, "ln64.")
    x = rnd(2, 5, 64) * 2.0 - 0.5
    arrs["ln64.x"], arrs["ln64.y"] = n(x), n(mod(x, None))
    for T in (1, 7, 16):
        mod = ref_bb.MaskedMultiHeadAttention(4, 64, 32, 0, 0.0)
        load(mod, "self64.")
        x = rnd(2, T, 64)
        arrs[f"self64.T{T}.x"], arrs[f"self64.T{T}.y"] = n(x), n(mod(x))
    mod = ref_bb.MaskedMultiHeadAttention(4, 64, 32, 3, 0.0)
    load(mod, "self64s3.")
    x = rnd(1, 11, 64)
    arrs["self64s3.x"], arrs["self64s3.y"] = n(x), n(mod(x))
    mod = ref_bb.MaskedMultiHeadCrossAttention(4, 32, 32, 0, 0.0)
    load(mod, "cross32.")
    x1, x2 = rnd(2, 9, 32), rnd(2, 9, 32)
    arrs["cross32.x1"], arrs["cross32.x2"], arrs["cross32.y"] = n(x1), n(x2), n(mod(x1, x2))
    mod = ref_bb.MLP(64, 0.0, 8)
    load(mod, "mlp64.")
    x = rnd(2, 5, 64)
    arrs["mlp64.x"], arrs["mlp64.y"] = n(x), n(mod(x))
    mod = ref_bb.MLP(1, 0.0, 8, 64, 1)
    load(mod, "ibmlp.")
    c = torch.from_numpy(rng.random((2, 5, 1)).astype(np.float32))
    arrs["ibmlp.c"], arrs["ibmlp.y"] = n(c), n(mod(c))
    fc = ref_bb.precompute_freqs_cis(16, 32)
    arrs["rope16.cos"], arrs["rope16.sin"] = n(fc.real), n(fc.imag)
    q, k = rnd(2, 6, 4, 16), rnd(2, 6, 4, 16)
    qo, ko = ref_bb.apply_rotary_emb(q, k, fc[:6])
    arrs["rope16.q"], arrs["rope16.k"], arrs["rope16.qo"], arrs["rope16.ko"] = n(q), n(k), n(qo), n(ko)
    p, t = rnd(2, 4, 3, 8), rnd(2, 4, 3, 8)
    arrs["relmse.p"], arrs["relmse.t"], arrs["relmse.y"] = n(p), n(t), n(ref_tu.relativeMSE(p, t))
    with torch.no_grad():
        save("modules", **{k: (v if isinstance(v, np.ndarray) else np.asarray(v)) for k, v in arrs.items()})


def exchange_cases():
    """One SEA block's exchange step alone (pins the Gauss-Seidel order, models/temporal.py:187-192)."""
    print("exchange")
    arrs = {}
    for F, ln in ((2, "adaln"), (3, "adaln"), (3, "ln")):
        cfg = OracleConfig(1, 64, 4, 16, 8, 0, F, 2, True, ln)
        m = build_reference(cfg)
        m.eval()
        x, _, ib = recipe_inputs(2, 10, cfg, seed=5 + F)
        xs = [x[:, :, i, :].clone() for i in range(F)]
        with torch.no_grad():
            ys = m.blocks[0]._apply_exchange(list(xs), ib)
            blk = m.blocks[0](*[x[:, :, i, :] for i in range(F)], x_add=ib)
        tag = f"F{F}{ln}"
        arrs[tag + ".cfg"] = cfg_meta(cfg)
        arrs[tag + ".x"], arrs[tag + ".ib"] = n(x), n(ib)
        arrs[tag + ".y"] = np.stack([n(y) for y in ys], axis=2)
        arrs[tag + ".block"] = np.stack([n(y) for y in blk], axis=2)
    save("exchange", **arrs)


def decode_cases():
    """Decode (models/encoder_decoder.py:126-146) fed by inverse_transform_processed_data (utils/train_utils.py:339-362) on a
    rollout-shaped tensor: cylinder-like groups [[0,1],[2]] with small dims, and a 3-group variant with ragged group sizes."""
    for name, groups, n_inp, hidden, D, P, tr, T in (("decode_cyl_small", [[0, 1], [2]], 24, 96, 16, 9, 2, 5),
                                                     ("decode_three_groups", [[0], [1, 2, 3], [4, 5]], 12, 64, 8, 4, 1, 7)):
        dec = RefDecode(groups, n_inp, hidden, D, dropout=0.0).eval()
        sch = decode_schema(groups, n_inp, hidden, D)
        named = dict(dec.named_parameters())
        assert list(named.keys()) == list(sch.keys()), (list(named.keys()), list(sch.keys()))
        with torch.no_grad():
            for k, prm in named.items():
                shp, kind = sch[k]
                assert tuple(prm.shape) == tuple(shp), (k, prm.shape, shp)
                prm.copy_(torch.from_numpy(recipe_tensor(k, shp, kind)))
        rng = np.random.Generator(np.random.PCG64(77))
        roll = torch.from_numpy(rng.standard_normal((tr, T, len(groups), P * D)).astype(np.float32))
        with torch.no_grad():
            z = ref_tu.inverse_transform_processed_data(roll, tr, T, P, len(groups))
            out = dec(z)
        save(name, groups=np.array([len(g) for g in groups]), dims=np.array([n_inp, hidden, D, P, tr, T]), roll=n(roll), z=n(z), out=n(out))


def encode_cases():
    """PointwiseEncode (models/encoder_decoder.py:75-123) in eval mode on padded patch tensors [B, P, F, C]: a cylinder-like case whose
    attention head dim is 4 (width 2 x 16, 8 heads — the shipped spatial dims) and a three-group case with head dim 8."""
    for name, groups, n_inp, hidden, layers, D, H, P, B in (("encode_cyl_small", [[0, 1], [2]], 12, 48, 2, 16, 8, 9, 3),
                                                           ("encode_three_groups", [[0], [1, 2, 3], [4, 5]], 8, 64, 3, 16, 6, 10, 2)):
        print(name)
        enc = RefEncode(groups, n_inp, hidden, layers, D, H, 64, 0, dropout=0.0).eval()
        sch = encoder_schema(groups, n_inp, hidden, layers, D)
        named = dict(enc.named_parameters())
        assert list(named.keys()) == list(sch.keys()), (list(named.keys()), list(sch.keys()))
        with torch.no_grad():
            for k, prm in named.items():
                shp, kind = sch[k]
                assert tuple(prm.shape) == tuple(shp), (k, prm.shape, shp)
                prm.copy_(torch.from_numpy(recipe_tensor(k, shp, kind)))
        rng = np.random.Generator(np.random.PCG64(zlib_seed(name)))
        Fn = sum(len(g) for g in groups)
        x = torch.from_numpy(rng.standard_normal((B, P, Fn, n_inp)).astype(np.float32))
        x[:, :, :, -2:] = 0.0     # padded slots of a cell (pad_field_value = 0, configs/cylinder_flow.py:23)
        with torch.no_grad():
            z = enc(x)
        save(name, dims=np.array([n_inp, hidden, layers, D, H, P, B], dtype=np.int64), groups=np.array([len(g) for g in groups], dtype=np.int64),
             x=n(x), z=n(z), pe=n(enc.spatial_pos_encoder.pe[0, :P]))


def zlib_seed(name):
    import zlib
    return zlib.crc32(name.encode())


def unpatch_cases():
    """DataPartitioner2D.create_partitions / inverse_partition and MinMaxScaler.inverse_transform of the reference on a random 2-D point
    cloud (ragged cells, one empty-ish corner), as chained by MeshProcessor.inverse_scale_and_unpatch (utils/data_processors.py:553-573)."""
    import contextlib, io
    rng = np.random.Generator(np.random.PCG64(31))
    for name, npts, m, nn, T, F, groups in (("unpatch_5x5", 700, 6, 6, 3, 3, [[0, 1], [2]]), ("unpatch_3x4", 90, 4, 5, 2, 2, [[0], [1]])):
        xy = rng.random((2, npts)).astype(np.float32)
        xy[0, : npts // 4] *= 0.3   # uneven density -> ragged cells
        fields = rng.standard_normal((T, npts, F)).astype(np.float32)
        part = ref_dp.DataPartitioner2D(torch.from_numpy(xy[0]), torch.from_numpy(xy[1]), m=m, n=nn, pad_id=-1, pad_field_value=0)
        parts, imap = part.create_partitions([torch.from_numpy(fields[:, :, i]) for i in range(F)])
        stacked = torch.stack([p_[1] for p_ in parts], dim=1)          # [T, P, C, F]
        index_map = torch.stack(imap, dim=0)                           # [P, C]
        coords, recon = part.inverse_partition([(p_[0], p_[1]) for p_ in parts], time_dim=T)
        assert torch.equal(recon, torch.from_numpy(fields))
        # scaled variant: fields are "scaled" values; inverse transform per group with fixed min / max
        scal, params = [], []
        for gi, g in enumerate(groups):
            sc = ref_dp.MinMaxScaler(feature_range=(-1, 1), name=f"g{gi}", save_dir=HERE)
            sc.min_val, sc.max_val = torch.tensor(-2.5 - gi), torch.tensor(4.0 + 0.5 * gi)
            scal.append(sc)
            params.append([-1.0, 1.0, float(sc.min_val), float(sc.max_val)])
        unscaled = torch.zeros_like(recon)
        with contextlib.redirect_stdout(io.StringIO()):
            for sc, g in zip(scal, groups):
                unscaled[..., g] = sc.inverse_transform(recon[..., g])
        # forward direction (MeshProcessor._scale_fields then create_partitions, utils/data_processors.py:528-536, 511-523): scale, THEN pad with 0
        fwd = torch.zeros(T, npts, F)
        with contextlib.redirect_stdout(io.StringIO()):
            for sc, g in zip(scal, groups):
                fwd[..., g] = sc.transform(torch.from_numpy(fields)[..., g])
        parts_s, _ = part.create_partitions([fwd[:, :, i].to(torch.float32) for i in range(F)])
        scaled_stacked = torch.stack([p_[1] for p_ in parts_s], dim=1)   # [T, P, C, F]
        save(name, xy=xy, mn=np.array([m, nn]), fields=fields, stacked=n(stacked), index_map=index_map.numpy().astype(np.int64),
             unscaled=n(unscaled), groups=np.array([len(g) for g in groups]), scaler_params=np.array(params, dtype=np.float64),
             scaled_stacked=n(scaled_stacked))


def dataset_cases():
    """TemporalDataset (utils/data_processors.py:388-452): window indexing over two trajectories of different lengths, with and without overlap."""
    print("temporal_dataset")
    rng = np.random.Generator(np.random.PCG64(41))
    lens = (23, 17)
    data = [torch.from_numpy(rng.standard_normal((L_, 2, 6)).astype(np.float32)) for L_ in lens]
    orig = [torch.from_numpy(rng.standard_normal((L_, 5, 3)).astype(np.float32)) for L_ in lens]
    ib = [torch.from_numpy(rng.random((L_, 1)).astype(np.float32)) for L_ in lens]
    arrs = {f"data{i}": n(d) for i, d in enumerate(data)}
    arrs.update({f"orig{i}": n(d) for i, d in enumerate(orig)})
    arrs.update({f"ib{i}": n(d) for i, d in enumerate(ib)})
    for tag, src_len, overlap in (("a", 5, 0), ("b", 6, 2)):
        ds = ref_dp.TemporalDataset(data, orig, ib, src_len=src_len, overlap=overlap)
        arrs[f"{tag}.len"] = np.array([len(ds), src_len, overlap])
        for idx in range(len(ds)):
            src, tgt, tgt_o, fib = ds[idx]
            arrs[f"{tag}.{idx}.src"], arrs[f"{tag}.{idx}.tgt"], arrs[f"{tag}.{idx}.tgto"], arrs[f"{tag}.{idx}.ib"] = n(src), n(tgt), n(tgt_o), n(fib)
    save("temporal_dataset", **arrs)


def big_cases():
    # cfg2 shape (BASELINE.json configs[1]): E=256, H=8, F=3, T=2024, B=1
    print("cfg2_shape")
    cfg = OracleConfig(1, 256, 8, 2024, 8, 0, 3, 2, True, "adaln")
    m = build_reference(cfg)
    m.eval()
    x, _, ib = recipe_inputs(1, 2024, cfg, seed=1234)
    with torch.no_grad():
        out = m(x, ib)
    save("cfg2_shape", cfg=cfg_meta(cfg), seed=np.array(1234), out_sub=n(out[:, ::97, :, ::13]),
         out_l2=n(out.pow(2).sum(dim=(0, 1, 3)).sqrt()), out_mean=n(out.mean(dim=(0, 1, 3))),
         out_last=n(out[:, -1]))
    del m
    # cfg5-like (multiphase structure): LN_type='ln', F=2, 100-step rollout at E=256
    rollout_case("rollout100_ln_f2_e256", OracleConfig(1, 256, 8, 128, 8, 0, 2, 2, True, "ln"), 1, 100, seed=77)
    # cfg1 (BASELINE.json configs[0]): shipped cylinder dims E=1024, F=2, 1 trajectory, 8-step rollout
    print("cfg1_cylinder_rollout8")
    cfg = OracleConfig(1, 1024, 8, 2024, 8, 0, 2, 2, True, "adaln")
    m = build_reference(cfg)
    m.eval()
    x, tgt, ib = recipe_inputs(1, 8, cfg, seed=4321)
    with torch.no_grad():
        a = x[:, 0:1]
        for i in range(8):
            o = m(a, ib[:, : i + 1])
            a = torch.cat((a, o[:, -1:]), dim=1)
        pred = a[:, 1:]
    save("cfg1_cylinder_rollout8", cfg=cfg_meta(cfg), seed=np.array(4321), pred_sub=n(pred[:, :, :, ::37]),
         pred_l2=n(pred.pow(2).sum(dim=(0, 3)).sqrt()))


def sub_stride(numel, cap=384):
    """Stride of the committed sub-sample of a flattened gradient: at most `cap` elements, never a multiple of a power-of-two row length
    (odd), so the samples walk across rows and columns."""
    st = max(1, numel // cap)
    return st | 1


def cfg3_train_case():
    """cfg3-shaped training step of the reference (BASELINE.json configs[2] at one trajectory: E=256, H=8, F=3, T=2024, B=1 — every multi-tile
    path of the backward is exercised at this size): train/train_temporal.py:254-258 once, fp32.  Stored: loss, eval- and train-mode output
    sub-samples, and for every parameter with a gradient its L2 norm, its sum and a strided sub-sample (sub_stride)."""
    print("cfg3_train")
    cfg = OracleConfig(1, 256, 8, 2024, 8, 0, 3, 2, True, "adaln")
    m = build_reference(cfg)
    x, tgt, ib = recipe_inputs(1, 2024, cfg, seed=2024)
    m.train()
    loss_fn = torch.nn.MSELoss()
    opt = ref_tu.initialize_optimizer(m, {"learning_rate": 1e-4})
    opt.zero_grad()
    o = m(x, ib)
    loss = loss_fn(o, tgt)
    loss.backward()
    arrs = dict(cfg=cfg_meta(cfg), seed=np.array(2024), loss=np.array(loss.item(), dtype=np.float64), out_sub=n(o[:, ::97, :, ::13]),
                dead_keys=np.array([k for k, p in m.named_parameters() if p.grad is None]))
    keys = [k for k, p in m.named_parameters() if p.grad is not None]
    arrs["grad_keys"] = np.array(keys)
    arrs["grad_l2"] = np.array([float(m.get_parameter(k).grad.double().pow(2).sum().sqrt()) for k in keys], dtype=np.float64)
    arrs["grad_sum"] = np.array([float(m.get_parameter(k).grad.double().sum()) for k in keys], dtype=np.float64)
    for k in keys:
        g = m.get_parameter(k).grad.reshape(-1)
        arrs["gsub:" + k] = n(g[:: sub_stride(g.numel())])
    opt.step()
    # parameters after the AdamW step, same sub-sampling (lr 1e-4: the update is +-lr per element at step 1, so this pins signs)
    for k in keys:
        q = m.get_parameter(k).detach().reshape(-1)
        arrs["p1sub:" + k] = n(q[:: sub_stride(q.numel())])
    save("cfg3_train", **arrs)


def eval_case():
    """The `temporal test` evaluation of the reference on a synthetic mesh (main.py:101-123 -> utils/train_utils.py:186-312), end to end with the
    reference's OWN classes: MeshProcessor.patchify_and_scale -> ProcessData (frozen SpatialModel, recipe weights, loaded from a checkpoint file
    as the reference does) -> transform_processed_data -> TemporalDataset / DataLoader -> full_autoregressive_evaluation (rollout, decode,
    un-patchify, per-field relative MSE) and autoregressive_validation.  Stored: the raw mesh + fields + conditions (inputs), the dims, and the
    reference's metrics and intermediate tensors."""
    import contextlib, io, tempfile
    import matplotlib
    matplotlib.use("Agg")
    from models.encoder_decoder import SpatialModel as RefSpatial
    from torch.utils.data import DataLoader

    print("eval_synth")
    tmp = tempfile.mkdtemp(prefix="sea_eval_")
    groups, D, hidden, layers, Hs = [[0, 1], [2]], 8, 32, 2, 2
    m_ = n_ = 5
    P = (m_ - 1) * (n_ - 1)
    tr, T, npts, F = 2, 7, 420, 3
    rng = np.random.Generator(np.random.PCG64(2025))
    xy = rng.random((2, npts)).astype(np.float32)
    xy[0, : npts // 3] = 0.2 + 0.15 * xy[0, : npts // 3]          # a denser band: ragged cells
    tt = np.arange(T + 1, dtype=np.float32)[None, :, None]
    ph = rng.random((tr, 1, 1)).astype(np.float32) * 6.28
    fields = np.stack([np.sin(3 * xy[0][None, None, :] + 0.3 * tt + ph), np.cos(2 * xy[1][None, None, :] - 0.2 * tt + ph) * xy[0][None, None, :],
                       0.5 * np.sin(4 * xy[0][None, None, :] * xy[1][None, None, :] + 0.1 * tt) + 0.1 * ph], axis=-1).astype(np.float32)   # [tr, T+1, N, F]
    cond = (0.2 + 0.6 * rng.random((tr, 1, 1)).astype(np.float32)) * np.ones((tr, T + 1, 1), dtype=np.float32)
    E = P * D
    config = dict(device="cpu", save_dir=tmp, dimension="2D", field_groups=groups, scale_feature_range=None, csv_scale_name="scaler", m=m_, n=n_, k=None,
                  pad_id=-1, pad_field_value=0, MLP_hidden_spatial=hidden, num_layers_spatial=layers, embed_dim_spatial=D, n_heads_spatial=Hs,
                  block_size_spatial=64, dropout_spatial=0.0, variational_spatial=False, src_len_spatial=0,
                  encoder_decoder_path=os.path.join(tmp, "encoder_decoder.pt"), spatial_batch_size=1000, perform_initial_test=False,
                  random_seed=42, SEA_isolate=True, SEA_mixed=False, case_name="synth", run_name="fixture")
    coords = torch.from_numpy(xy)
    data = torch.from_numpy(fields)
    with contextlib.redirect_stdout(io.StringIO()):
        mp_ = ref_dp.MeshProcessor(config, coords)
        stacked_coords, scaled = mp_.patchify_and_scale(data.reshape(tr * (T + 1), npts, F), train_indices=np.arange(1))
    n_inp = scaled.shape[2]
    config["n_inp"] = n_inp
    _, imap = mp_.partitioner.create_partitions([data[0, :1, :, i].to(torch.float32) for i in range(F)])
    index_map = torch.stack(imap, dim=0).numpy().astype(np.int64)                   # [P, C], pad_id = -1
    scaled_p = scaled.permute(0, 1, 3, 2)                        # SEA_isolate (train/train_temporal.py:152-153)
    # the frozen spatial autoencoder: recipe weights under the reference's own state_dict keys, saved the way the reference expects to load them
    sp = RefSpatial(groups, n_inp, hidden, layers, D, Hs, 64, 0, variational=False, dropout=0.0)
    sch = OrderedDictMerge(encoder_schema(groups, n_inp, hidden, layers, D, pre="encode."), decode_schema(groups, n_inp, hidden, D, pre="decode.decoders."))
    named = dict(sp.named_parameters())
    assert sorted(named.keys()) == sorted(sch.keys()), (sorted(set(named) ^ set(sch)))
    with torch.no_grad():
        for k, prm in named.items():
            shp, kind = sch[k]
            prm.copy_(torch.from_numpy(recipe_tensor(k, shp, kind)))
    torch.save(sp.state_dict(), config["encoder_decoder_path"])
    with contextlib.redirect_stdout(io.StringIO()):
        proc = ref_dp.ProcessData(n_inp, config)
        z = proc.initialize_and_process_data(scaled_p.clone())                     # [tr*(T+1), P, G, D]
    enc = ref_tu.transform_processed_data(z, tr, T + 1, P, len(groups))       # [tr, T+1, G, P*D]
    ds = ref_dp.TemporalDataset([enc[i] for i in range(tr)], [data[i] for i in range(tr)], [torch.from_numpy(cond[i]) for i in range(tr)], src_len=T, overlap=0)
    loader = DataLoader(ds, batch_size=tr, shuffle=False)
    cfg = OracleConfig(1, E, 4, 32, 8, 0, len(groups), 2, True, "adaln")
    model = build_reference(cfg)
    with contextlib.redirect_stdout(io.StringIO()):
        res = ref_tu.full_autoregressive_evaluation(model, loader, torch.nn.MSELoss(), torch.device("cpu"), proc, mp_, config, 0, plot_traj=False)
        v_loss, v_rel = ref_tu.autoregressive_validation(model, loader, torch.nn.MSELoss(), torch.device("cpu"))
    import csv
    with open(os.path.join(tmp, "rollout_error_synth_fixture.csv")) as f:
        rows = [r for r in csv.reader(f)][1:]
    per_step = np.array([[float(v) for v in r[1:]] for r in rows], dtype=np.float64)     # [T, F] decoded rel-MSE of the LAST batch (there is one)
    save("eval_synth", xy=xy, fields=fields, cond=cond, dims=np.array([m_, n_, D, hidden, layers, Hs, n_inp, tr, T], dtype=np.int64),
         groups=np.array([len(g) for g in groups]), cfg=cfg_meta(cfg), scaled=n(scaled), index_map=index_map, z=n(z), enc=n(enc),
         encoded_rel_mse=np.array(res["encoded_rel_mse"]), decoded_rel_mse=np.array(res["decoded_rel_mse"]), decoded_per_step=per_step,
         val_loss=np.array(v_loss), val_rel_mse_time=np.array(v_rel))
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)


def OrderedDictMerge(a, b):
    out = dict(a)
    out.update(b)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--big", action="store_true", help="also the cfg1/cfg2-shaped cases (slow, ~GBs of RAM)")
    args = ap.parse_args()

    cases = {
        "model_tiny_adaln_f3": lambda: model_case("model_tiny_adaln_f3", OracleConfig(1, 32, 2, 24, 8, 0, 3, 2, True, "adaln"), 2, 16, train=True),
        "model_tiny_ln_f2": lambda: model_case("model_tiny_ln_f2", OracleConfig(2, 32, 2, 24, 8, 0, 2, 2, True, "ln"), 2, 12, train=True),
        "model_tiny_adaln_f2_pre": lambda: model_case("model_tiny_adaln_f2_pre", OracleConfig(1, 32, 2, 24, 8, 0, 2, 2, False, "adaln"), 2, 8, train=True),
        "model_small_srclen2": lambda: model_case("model_small_srclen2", OracleConfig(1, 64, 4, 16, 8, 2, 2, 2, True, "adaln"), 2, 9),
        "rollout8_adaln_f3": lambda: rollout_case("rollout8_adaln_f3", OracleConfig(1, 64, 4, 16, 8, 0, 3, 2, True, "adaln"), 2, 8),
        "rollout100_ln_f2": lambda: rollout_case("rollout100_ln_f2", OracleConfig(1, 64, 4, 128, 8, 0, 2, 2, True, "ln"), 1, 100),
        "modules": module_cases,
        "exchange": exchange_cases,
        "decode": decode_cases,
        "encode": encode_cases,
        "unpatch": unpatch_cases,
        "dataset": dataset_cases,
    }
    # ablation variants of the exchange / info-bottleneck: forward only
    cases["model_addition_adaln_f3"] = lambda: model_case("model_addition_adaln_f3", OracleConfig(2, 64, 4, 80, 8, 0, 3, 2, True, "adaln", "addition"), 2, 33)
    cases["model_addition_ln_f2_pre"] = lambda: model_case("model_addition_ln_f2_pre", OracleConfig(1, 64, 4, 80, 8, 0, 2, 2, False, "ln", "addition"), 2, 17)
    cases["model_simple_adaln_f3"] = lambda: model_case("model_simple_adaln_f3", OracleConfig(2, 64, 4, 80, 8, 0, 3, 2, True, "adaln", "simple"), 2, 33)
    cases["model_pool_adaln_f3"] = lambda: model_case("model_pool_adaln_f3", OracleConfig(2, 64, 4, 80, 8, 0, 3, 2, True, "adaln", "pool"), 2, 27)
    cases["model_pool_ln_f2"] = lambda: model_case("model_pool_ln_f2", OracleConfig(1, 64, 4, 80, 8, 1, 2, 2, True, "ln", "pool"), 2, 18)
    cases["model_sea_fourier_adaln_f3"] = lambda: model_case("model_sea_fourier_adaln_f3", OracleConfig(2, 64, 4, 80, 8, 0, 3, 2, True, "adaln", "sea", "add", "fourier"), 2, 21)
    cases["model_sea_linear_ln_f2_pre"] = lambda: model_case("model_sea_linear_ln_f2_pre", OracleConfig(1, 64, 4, 80, 8, 0, 2, 2, False, "ln", "sea", "add", "linear"), 2, 19)
    cases["model_sea_noib_adaln_f2"] = lambda: model_case("model_sea_noib_adaln_f2", OracleConfig(1, 64, 4, 80, 8, 0, 2, 2, True, "adaln", "sea", "none"), 2, 20)
    # ... and their training step (gradients of every parameter + the set of gradient-less ones), small dims
    cases["train_addition_adaln_f3"] = lambda: model_case("train_addition_adaln_f3", OracleConfig(2, 32, 2, 24, 8, 0, 3, 2, True, "adaln", "addition"), 2, 12, train="grads")
    cases["train_simple_ln_f2"] = lambda: model_case("train_simple_ln_f2", OracleConfig(1, 32, 2, 24, 8, 0, 2, 2, True, "ln", "simple"), 2, 12, train="grads")
    cases["train_sea_noib_adaln_f2"] = lambda: model_case("train_sea_noib_adaln_f2", OracleConfig(1, 32, 2, 24, 8, 0, 2, 2, True, "adaln", "sea", "none"), 2, 12, train="grads")
    cases["train_sea_linear_ln_f2_pre"] = lambda: model_case("train_sea_linear_ln_f2_pre", OracleConfig(1, 32, 2, 24, 8, 0, 2, 2, False, "ln", "sea", "add", "linear"), 2, 12, train="grads")
    cases["train_addition_fourier_adaln_f3"] = lambda: model_case("train_addition_fourier_adaln_f3", OracleConfig(1, 32, 2, 24, 8, 0, 3, 2, True, "adaln", "addition", "add", "fourier"), 2, 12, train="grads")
    cases["train_pool_adaln_f3"] = lambda: model_case("train_pool_adaln_f3", OracleConfig(2, 32, 2, 24, 8, 0, 3, 2, True, "adaln", "pool"), 2, 12, train="grads")
    cases["train_pool_ln_f1"] = lambda: model_case("train_pool_ln_f1", OracleConfig(1, 32, 2, 24, 8, 0, 1, 2, False, "ln", "pool"), 2, 12, train="grads")
    # ib_addition_mode 'attention' (models/temporal.py:49-53,117-118): forward at two shapes (after / before the exchange), the train step of each
    cases["model_ibattn_adaln_f3"] = lambda: model_case("model_ibattn_adaln_f3", OracleConfig(2, 64, 4, 80, 8, 0, 3, 2, True, "adaln", "sea", "attention"), 2, 27)
    cases["model_ibattn_ln_f2_pre"] = lambda: model_case("model_ibattn_ln_f2_pre", OracleConfig(1, 64, 4, 80, 8, 0, 2, 2, False, "ln", "addition", "attention", "linear"), 2, 70)
    cases["train_ibattn_adaln_f3"] = lambda: model_case("train_ibattn_adaln_f3", OracleConfig(2, 32, 2, 24, 8, 0, 3, 2, True, "adaln", "sea", "attention"), 2, 12, train="grads")
    cases["train_ibattn_ln_f2_pre"] = lambda: model_case("train_ibattn_ln_f2_pre", OracleConfig(1, 32, 2, 24, 8, 0, 2, 2, False, "ln", "simple", "attention", "fourier"), 2, 12, train="grads")
    # ib_addition_mode 'concat' (models/temporal.py:48,115-116: blocks widened by 64 info-bottleneck columns; the reference only runs it with the
    # info-bottleneck step BEFORE the block, add_info_after_cross=False): embed_dim 64 -> rows of 128 (head dims 32 / 16)
    cases["model_ibconcat_adaln_f3"] = lambda: model_case("model_ibconcat_adaln_f3", OracleConfig(2, 64, 4, 80, 8, 0, 3, 2, False, "adaln", "sea", "concat"), 2, 27)
    cases["model_ibconcat_ln_f2"] = lambda: model_case("model_ibconcat_ln_f2", OracleConfig(1, 64, 4, 80, 8, 0, 2, 2, False, "ln", "addition", "concat", "linear"), 2, 70)
    # (training: small hidden widths — scale_ratio 4 — keep the per-parameter gradients of these 128-wide models under 2 MB per file)
    cases["train_ibconcat_ln_f2"] = lambda: model_case("train_ibconcat_ln_f2", OracleConfig(1, 64, 4, 24, 4, 0, 2, 2, False, "ln", "sea", "concat"), 2, 12, train="grads")
    cases["train_ibconcat_adaln_f1"] = lambda: model_case("train_ibconcat_adaln_f1", OracleConfig(1, 64, 4, 24, 4, 0, 1, 2, False, "adaln", "addition", "concat", "fourier"), 2, 12, train="grads")
    for T in (1, 7, 16, 65):
        cases[f"model_small_adaln_f3_T{T}"] = (lambda T=T: model_case(
            f"model_small_adaln_f3_T{T}", OracleConfig(1, 64, 4, 80, 8, 0, 3, 2, True, "adaln"), 2, T))
    for name, fn in cases.items():
        if args.only is None or args.only == name:
            fn()
    if args.big and args.only is None or args.only == "big":
        big_cases()
    if args.big and args.only is None or args.only == "cfg3_train":
        cfg3_train_case()
    if args.only is None or args.only == "eval_synth":
        eval_case()


if __name__ == "__main__":
    main()
