"""Pin the CPU oracle (oracle/sea_oracle.py) against golden vectors produced by the reference itself
(tests/golden/make_fixtures.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import sea_oracle as O
from oracle.recipe import encoder_params, param_schema, recipe_params, recipe_tensor
from tests.conftest import cfg_from_meta, grad_err, load_golden, rel_l2

TOL = 2e-6  # fp32 accumulation-order noise between two CPU restatements


def T(a):
    return torch.from_numpy(np.asarray(a))


def module_params(g, prefix):
    out = {}
    for row in g["kinds:" + prefix]:
        key, kind, shp = str(row).split("|")
        shape = tuple(int(s) for s in shp.split(",")) if shp else ()
        out[key] = T(recipe_tensor(key, shape, kind))
    return out


@pytest.mark.parametrize("d", [64, 128])
def test_adaln(d):
    g = load_golden("modules")
    p = module_params(g, f"adaln{d}.")
    y = O.adaln(T(g[f"adaln{d}.x"]), T(g[f"adaln{d}.c"]), p, f"adaln{d}.")
    assert rel_l2(y, g[f"adaln{d}.y"]) < TOL


def test_layernorm():
    g = load_golden("modules")
    p = module_params(g, "ln64.")
    y = O.layer_norm(T(g["ln64.x"]), p["ln64.weight"], None)
    assert rel_l2(y, g["ln64.y"]) < TOL


@pytest.mark.parametrize("T_", [1, 7, 16])
def test_self_attention(T_):
    g = load_golden("modules")
    p = module_params(g, "self64.")
    x = T(g[f"self64.T{T_}.x"])
    y = O.masked_attention(x, x, p, "self64.", 4, 0)
    assert rel_l2(y, g[f"self64.T{T_}.y"]) < TOL


def test_self_attention_src_len():
    g = load_golden("modules")
    p = module_params(g, "self64s3.")
    x = T(g["self64s3.x"])
    y = O.masked_attention(x, x, p, "self64s3.", 4, 3)
    assert rel_l2(y, g["self64s3.y"]) < TOL


def test_cross_attention():
    g = load_golden("modules")
    p = module_params(g, "cross32.")
    y = O.masked_attention(T(g["cross32.x1"]), T(g["cross32.x2"]), p, "cross32.", 4, 0)
    assert rel_l2(y, g["cross32.y"]) < TOL


def test_mlp_and_ib_mlp():
    g = load_golden("modules")
    y = O.mlp(T(g["mlp64.x"]), module_params(g, "mlp64."), "mlp64.")
    assert rel_l2(y, g["mlp64.y"]) < TOL
    y = O.mlp(T(g["ibmlp.c"]), module_params(g, "ibmlp."), "ibmlp.")
    assert rel_l2(y, g["ibmlp.y"]) < TOL


def test_rope():
    g = load_golden("modules")
    cos, sin = O.rope_tables(16, 32)
    # torch.polar and torch.cos/sin may differ in the last ulp
    assert np.allclose(cos.numpy(), g["rope16.cos"], atol=2e-7, rtol=0)
    assert np.allclose(sin.numpy(), g["rope16.sin"], atol=2e-7, rtol=0)
    qo = O.rope(T(g["rope16.q"]), cos[:6], sin[:6])
    ko = O.rope(T(g["rope16.k"]), cos[:6], sin[:6])
    assert rel_l2(qo, g["rope16.qo"]) < 1e-7 and rel_l2(ko, g["rope16.ko"]) < 1e-7


def test_relative_mse():
    g = load_golden("modules")
    y = O.relative_mse(T(g["relmse.p"]), T(g["relmse.t"]))
    assert rel_l2(y, g["relmse.y"]) < 1e-6


@pytest.mark.parametrize("tag", ["F2adaln", "F3adaln", "F3ln"])
def test_exchange_is_gauss_seidel(tag):
    g = load_golden("exchange")
    cfg = cfg_from_meta(g[tag + ".cfg"])
    p = recipe_params(cfg)
    x, ib = T(g[tag + ".x"]), T(g[tag + ".ib"])
    xs = [x[:, :, i, :] for i in range(cfg.num_variables)]
    ys = torch.stack(O.sea_exchange(xs, ib, p, "blocks.0.", cfg), dim=2)
    assert rel_l2(ys, g[tag + ".y"]) < TOL
    blk = torch.stack(O.block_forward(xs, ib, p, "blocks.0.", cfg), dim=2)
    assert rel_l2(blk, g[tag + ".block"]) < TOL


MODEL_CASES = ["model_tiny_adaln_f3", "model_tiny_ln_f2", "model_tiny_adaln_f2_pre", "model_small_srclen2",
               "model_small_adaln_f3_T1", "model_small_adaln_f3_T7", "model_small_adaln_f3_T16",
               "model_small_adaln_f3_T65",
               # ablation variants: exchange_mode 'addition' / 'simple', ib_addition_mode 'none'
               "model_addition_adaln_f3", "model_addition_ln_f2_pre", "model_simple_adaln_f3", "model_sea_noib_adaln_f2",
               # ib_scale_mode 'fourier' (the constructor's default) and 'linear'
               "model_sea_fourier_adaln_f3", "model_sea_linear_ln_f2_pre",
               # exchange_mode 'pool' (field -> pool-token cross-attention, src_len = 1 in the second case)
               "model_pool_adaln_f3", "model_pool_ln_f2",
               # ib_addition_mode 'attention' (un-masked cross-attention from the field rows to the info-bottleneck rows), after and before the exchange
               "model_ibattn_adaln_f3", "model_ibattn_ln_f2_pre",
               # ib_addition_mode 'concat' (rows widened by 64 info-bottleneck columns inside every block)
               "model_ibconcat_adaln_f3", "model_ibconcat_ln_f2"]


@pytest.mark.parametrize("name", MODEL_CASES)
def test_model_forward(name):
    g = load_golden(name)
    cfg = cfg_from_meta(g["cfg"])
    p = recipe_params(cfg)
    out = O.model_forward(T(g["x"]), T(g["ib"]), p, cfg)
    assert rel_l2(out, g["out"]) < TOL


@pytest.mark.parametrize("name", ["model_tiny_adaln_f3", "model_tiny_ln_f2", "model_tiny_adaln_f2_pre"])
def test_model_grads_and_adamw(name):
    g = load_golden(name)
    cfg = cfg_from_meta(g["cfg"])
    p = recipe_params(cfg)
    x, tgt, ib = T(g["x"]), T(g["tgt"]), T(g["ib"])
    out, loss, grads = O.loss_and_grads(x, ib, tgt, p, cfg)
    assert rel_l2(out, g["out_train"]) < TOL
    assert abs(float(loss) - float(g["losses"][0])) < 1e-6 * abs(float(g["losses"][0]))
    # the set of parameters without gradient is exactly the reference's
    dead = set(str(k) for k in g["dead_keys"])
    assert set(p.keys()) - set(O.live_param_keys(p, cfg)) == dead
    assert set(grads.keys()) == set(p.keys()) - dead
    assert int(g["n_opt_state"]) == len(grads)
    worst = max(rel_l2(grads[k], g["grad:" + k]) for k in grads)
    assert worst < 2e-5, worst
    # 3 AdamW steps
    lr = float(g["lr"])
    p3, losses = O.train_steps(x, ib, tgt, p, cfg, 3, lr)
    assert np.allclose(losses, g["losses"], rtol=2e-5)
    p1, _ = O.train_steps(x, ib, tgt, p, cfg, 1, lr)
    for k in grads:
        assert rel_l2(p1[k], g["param1:" + k]) < 1e-5, k
        assert rel_l2(p3[k], g["param3:" + k]) < 1e-4, k
    for k in dead:
        assert torch.equal(p3[k], p[k])


@pytest.mark.parametrize("name", ["train_addition_adaln_f3", "train_simple_ln_f2", "train_sea_noib_adaln_f2", "train_sea_linear_ln_f2_pre",
                                  "train_addition_fourier_adaln_f3", "train_pool_adaln_f3", "train_pool_ln_f1", "train_ibattn_adaln_f3", "train_ibattn_ln_f2_pre", "train_ibconcat_ln_f2", "train_ibconcat_adaln_f1"])
def test_variant_model_grads(name):
    """The ablation variants' train step (reference models/temporal.py:197-312, 103-116): output, loss, every gradient and the set of gradient-less
    parameters of the oracle against what the reference produced."""
    g = load_golden(name)
    cfg = cfg_from_meta(g["cfg"])
    p = recipe_params(cfg)
    out, loss, grads = O.loss_and_grads(T(g["x"]), T(g["ib"]), T(g["tgt"]), p, cfg)
    assert rel_l2(out, g["out_train"]) < TOL
    assert abs(float(loss) - float(g["losses"][0])) < 1e-6 * abs(float(g["losses"][0]))
    dead = set(str(k) for k in g["dead_keys"])
    assert set(p.keys()) - set(O.live_param_keys(p, cfg)) == dead
    assert set(grads.keys()) == set(p.keys()) - dead
    worst = max(grad_err(grads[k], g["grad:" + k]) for k in grads)
    assert worst < 2e-5, worst


@pytest.mark.parametrize("name", ["rollout8_adaln_f3", "rollout100_ln_f2", "rollout100_ln_f2_e256"])
def test_rollout(name):
    g = load_golden(name)
    cfg = cfg_from_meta(g["cfg"])
    p = recipe_params(cfg)
    tgt = T(g["tgt"])
    pred = O.rollout(T(g["x0"]), T(g["ib"]), tgt.shape[1], p, cfg)
    assert rel_l2(pred, g["pred"]) < 1e-4  # autoregressive: per-step 1e-6 noise compounds
    rel = O.relative_mse(pred, tgt).mean()
    assert abs(float(rel) - float(g["rel_mse"])) < 1e-4 * float(g["rel_mse"])
    v_loss = O.mse_loss(pred[0:1], tgt[0:1])
    assert abs(float(v_loss) - float(g["val_loss"])) < 1e-4 * float(g["val_loss"])
    v_rel = O.relative_mse(pred[0:1], tgt[0:1], dim=3).mean()
    assert abs(float(v_rel) - float(g["val_rel_mse_time"])) < 1e-4 * float(g["val_rel_mse_time"])


def test_cfg2_shape_forward():
    """BASELINE.json configs[1] shape (E=256, H=8, F=3, T=2024, B=1) against the reference's strided sample."""
    g = load_golden("cfg2_shape")
    cfg = cfg_from_meta(g["cfg"])
    from oracle.recipe import recipe_inputs

    p = recipe_params(cfg)
    x, _, ib = recipe_inputs(1, 2024, cfg, seed=int(g["seed"]))
    with torch.no_grad():
        out = O.model_forward(x, ib, p, cfg)
    assert rel_l2(out[:, ::97, :, ::13], g["out_sub"]) < 5e-6
    assert rel_l2(out[:, -1], g["out_last"]) < 5e-6
    assert np.allclose(out.pow(2).sum(dim=(0, 1, 3)).sqrt().numpy(), g["out_l2"], rtol=1e-5)


def test_cfg3_train_step_matches_reference():
    """BASELINE.json configs[2] shape at one trajectory (E=256, H=8, F=3, T=2024): loss, the set of gradient-less parameters, and every live
    parameter's gradient (L2 norm, sum, strided sub-sample) of the oracle's autograd against the reference's (train/train_temporal.py:254-258)."""
    from oracle.recipe import recipe_inputs
    from tests.conftest import grad_sub_stride

    g = load_golden("cfg3_train")
    cfg = cfg_from_meta(g["cfg"])
    p = recipe_params(cfg)
    x, tgt, ib = recipe_inputs(1, 2024, cfg, seed=int(g["seed"]))
    out, loss, grads = O.loss_and_grads(x, ib, tgt, p, cfg)
    assert abs(float(loss) - float(g["loss"])) < 1e-6 * float(g["loss"])
    assert rel_l2(out[:, ::97, :, ::13], g["out_sub"]) < 5e-6
    keys = [str(k) for k in g["grad_keys"]]
    assert sorted(grads.keys()) == sorted(keys)
    assert sorted(k for k in p if k not in grads) == sorted(str(k) for k in g["dead_keys"])
    for k, l2 in zip(keys, g["grad_l2"]):
        gr = grads[k].reshape(-1)
        assert abs(float(gr.double().norm()) - l2) < 2e-5 * l2, k
        assert rel_l2(gr[:: grad_sub_stride(gr.numel())], g["gsub:" + k]) < 5e-5, k


def test_temporal_test_evaluation_chain_matches_reference():
    """The whole `temporal test` chain of the reference on a synthetic mesh (tests/golden/eval_synth.npz: its own MeshProcessor, ProcessData,
    TemporalDataset and full_autoregressive_evaluation) restated with the oracle's functions: patchify -> encode -> rollout -> decode -> un-patchify ->
    per-step, per-field relative MSE."""
    from oracle.recipe import decode_schema, encoder_schema

    g = load_golden("eval_synth")
    m_, n_, D, hidden, layers, Hs, n_inp, tr, T = (int(v) for v in g["dims"])
    groups, f0 = [], 0
    for n_f in g["groups"]:
        groups.append(list(range(f0, f0 + int(n_f))))
        f0 += int(n_f)
    P = (m_ - 1) * (n_ - 1)
    fields, imap = T_(g["fields"]), T_(g["index_map"])
    npts, F = fields.shape[2], fields.shape[3]
    scaled = O.patchify(fields.reshape(tr * (T + 1), npts, F), imap)                 # [B, P, C, F]
    assert torch.equal(scaled, T_(g["scaled"]))
    ep = {k: T_(recipe_tensor("encode." + k, shp, kind)) for k, (shp, kind) in encoder_schema(groups, n_inp, hidden, layers, D).items()}
    dp = {k: T_(recipe_tensor("decode." + k, shp, kind)) for k, (shp, kind) in decode_schema(groups, n_inp, hidden, D).items()}
    z = O.encode(scaled.permute(0, 1, 3, 2), ep, groups, Hs, layers)
    assert rel_l2(z, g["z"]) < 1e-5
    enc = z.reshape(tr, T + 1, P, len(groups), D).permute(0, 1, 3, 2, 4).reshape(tr, T + 1, len(groups), P * D)
    cfg = cfg_from_meta(g["cfg"])
    pred = O.rollout(enc[:, :1], T_(g["cond"])[:, :T], T, recipe_params(cfg), cfg)
    enc_rel = O.relative_mse(pred, enc[:, 1:]).mean()
    assert abs(float(enc_rel) - float(g["encoded_rel_mse"])) < 1e-4 * float(g["encoded_rel_mse"])
    dec = O.decode(O.rollout_to_patches(pred, P), dp, groups)                         # [tr*T, P, F, C]
    rec = O.unpatchify(dec.permute(0, 1, 3, 2), imap, npts).reshape(tr, T, npts, F)
    per_step = O.relative_mse(rec, fields[:, 1:], dim=2).mean(dim=0)
    assert rel_l2(per_step, g["decoded_per_step"]) < 1e-4
    assert abs(float(per_step.mean()) - float(g["decoded_rel_mse"])) < 1e-4 * float(g["decoded_rel_mse"])


def T_(a):
    return torch.from_numpy(np.asarray(a))


def test_schema_counts():
    cfg = O.OracleConfig(1, 256, 8, 2024, 8, 0, 3, 2, True, "adaln")
    s = param_schema(cfg)
    n = sum(int(np.prod(shp)) for shp, _ in s.values())
    assert len(s) == 224 and n == 8_381_472
    live = O.live_param_keys({k: None for k in s}, cfg)
    n_dead = sum(int(np.prod(s[k][0])) for k in s if k not in live)
    assert n_dead == 1_057_408


@pytest.mark.parametrize("name", ["decode_cyl_small", "decode_three_groups"])
def test_decode_matches_reference(name):
    """Decode + inverse_transform_processed_data against the reference's own modules."""
    from oracle.recipe import decode_params

    g = load_golden(name)
    sizes = [int(v) for v in g["groups"]]
    n_inp, hidden, D, P, tr, T = (int(v) for v in g["dims"])
    groups, k = [], 0
    for sz in sizes:
        groups.append(list(range(k, k + sz)))
        k += sz
    p = decode_params(groups, n_inp, hidden, D)
    z = O.rollout_to_patches(torch.from_numpy(g["roll"]), P)
    assert torch.equal(z, torch.from_numpy(g["z"]))
    out = O.decode(z, p, groups)
    assert out.shape == g["out"].shape
    assert rel_l2(out.numpy(), g["out"]) < 1e-6


@pytest.mark.parametrize("name", ["unpatch_5x5", "unpatch_3x4"])
def test_unpatchify_matches_reference(name):
    """inverse_partition + MinMaxScaler.inverse_transform against the reference's own classes."""
    g = load_golden(name)
    sizes = [int(v) for v in g["groups"]]
    groups, k = [], 0
    for sz in sizes:
        groups.append(list(range(k, k + sz)))
        k += sz
    stacked, imap = torch.from_numpy(g["stacked"]), torch.from_numpy(g["index_map"])
    n_points = g["fields"].shape[1]
    assert torch.equal(O.unpatchify(stacked, imap, n_points), torch.from_numpy(g["fields"]))   # pure scatter: exact
    out = O.unpatchify(stacked, imap, n_points, groups, [tuple(r) for r in g["scaler_params"]])
    assert rel_l2(out.numpy(), g["unscaled"]) < 1e-6


@pytest.mark.parametrize("name", ["encode_cyl_small", "encode_three_groups"])
def test_encode_matches_reference(name):
    """Spatial encoder: oracle.encode against PointwiseEncode run by the reference; the sinusoidal position table too."""
    g = load_golden(name)
    n_inp, hidden, layers, D, H, P, B = (int(v) for v in g["dims"])
    groups, k = [], 0
    for sz in g["groups"]:
        groups.append(list(range(k, k + int(sz))))
        k += int(sz)
    p = encoder_params(groups, n_inp, hidden, layers, D)
    z = O.encode(T(g["x"]), p, groups, H, layers)
    assert z.shape == g["z"].shape
    assert rel_l2(z, g["z"]) < TOL
    assert np.array_equal(O.sinusoidal_pe(P, len(groups) * D).numpy(), g["pe"])


@pytest.mark.parametrize("name", ["unpatch_5x5", "unpatch_3x4"])
def test_patchify_matches_reference(name):
    """Forward direction of the mesh partition: scale, then partition and pad."""
    g = load_golden(name)
    groups, k = [], 0
    for sz in g["groups"]:
        groups.append(list(range(k, k + int(sz))))
        k += int(sz)
    imap = T(g["index_map"])
    assert torch.equal(O.patchify(T(g["fields"]), imap), T(g["stacked"]))
    out = O.patchify(T(g["fields"]), imap, groups, [tuple(r) for r in g["scaler_params"]])
    assert rel_l2(out, g["scaled_stacked"]) < 1e-6
    pad = (imap < 0)[None, :, :, None].expand_as(out)
    assert float(out[pad].abs().max()) == 0.0 if pad.any() else True


def test_temporal_dataset_windows_match_reference():
    from sea_amd.utils.data_processors import TemporalDataset

    g = load_golden("temporal_dataset")
    data, orig, ib = [T(g[f"data{i}"]) for i in range(2)], [T(g[f"orig{i}"]) for i in range(2)], [T(g[f"ib{i}"]) for i in range(2)]
    for tag in ("a", "b"):
        n, src_len, overlap = (int(v) for v in g[f"{tag}.len"])
        ds = TemporalDataset(data, orig, ib, src_len=src_len, overlap=overlap)
        assert len(ds) == n
        for idx in range(n):
            src, tgt, tgt_o, fib = ds[idx]
            seg, a, b = O.dataset_window(data, idx, src_len, overlap)
            assert torch.equal(src, data[seg][a:b]) and src.data_ptr() == data[seg][a:b].data_ptr()   # a view, not a copy
            assert torch.equal(src, T(g[f"{tag}.{idx}.src"])) and torch.equal(tgt, T(g[f"{tag}.{idx}.tgt"]))
            assert torch.equal(tgt_o, T(g[f"{tag}.{idx}.tgto"])) and torch.equal(fib, T(g[f"{tag}.{idx}.ib"]))
        with pytest.raises(IndexError):
            ds[n]
        b4 = ds.batch([0, 1])
        assert b4[0].shape[0] == 2 and torch.equal(b4[1][1], T(g[f"{tag}.1.tgt"]))

