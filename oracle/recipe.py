"""Deterministic parameter recipe shared by the fixture generator and the parity tests.

TEST INFRASTRUCTURE ONLY (see oracle/sea_oracle.py header).

The golden fixtures under tests/golden/ store inputs and expected outputs but not the
model weights: weights are regenerated from the parameter NAME with this recipe, so the
reference (in the build container), the oracle and the HIP path (on the GPU box) all see
bit-identical parameters without shipping megabytes of weights.

``param_schema`` restates the reference's parameter set (names and shapes of
``TemporalModel(...).named_parameters()``, models/temporal.py:326-403 with the blocks of
:21-192 and the layers of models/base_blocks.py); tests/golden/make_fixtures.py asserts it equals
the reference's own, key for key.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np
import torch

from .sea_oracle import OracleConfig

Schema = "OrderedDict[str, Tuple[Tuple[int, ...], str]]"


def _adaln(s, pre: str, d: int):
    s[pre + "weight"] = ((d,), "norm_w")
    s[pre + "bias"] = ((d,), "norm_b")
    s[pre + "cond_mlp.0.weight"] = ((2 * d, 1), "lin_w")
    s[pre + "cond_mlp.0.bias"] = ((2 * d,), "lin_b")
    s[pre + "cond_mlp.2.weight"] = ((2 * d, 2 * d), "lin_w")
    s[pre + "cond_mlp.2.bias"] = ((2 * d,), "lin_b")


def _norm(s, pre: str, d: int, ln_type: str):
    if ln_type.lower() == "adaln":
        _adaln(s, pre, d)
    else:
        s[pre + "weight"] = ((d,), "norm_w")  # custom LayerNorm(bias=False), base_blocks.py:82-85


def _linear(s, pre: str, out_f: int, in_f: int, bias: bool = True):
    s[pre + "weight"] = ((out_f, in_f), "lin_w")
    if bias:
        s[pre + "bias"] = ((out_f,), "lin_b")


def _attention(s, pre: str, d: int):
    for n in ("k", "q", "v"):
        _linear(s, f"{pre}{n}.", d, d)
    _linear(s, pre + "projection.", d, d, bias=False)


def param_schema(cfg: OracleConfig) -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    Eo, D, F = cfg.embed_dim, cfg.down_dim, cfg.num_variables
    E = cfg.internal_embed_dim                         # the block's row width ('concat': embed_dim + 64, models/temporal.py:48)
    Eib = 64 if cfg.ib_addition_mode == "concat" else Eo   # the info-bottleneck layer's output width (models/temporal.py:100-101)
    S = max(1, int(E * cfg.scale_ratio))
    sr = max(1, int(1 * cfg.scale_ratio))  # ib MLP hidden: dim_in = ib_num = 1
    s: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()
    for layer in range(cfg.num_layers):
        b = f"blocks.{layer}."
        if cfg.exchange_mode == "pool":     # SEAPoolBlockTemporal (models/temporal.py:197-241): a module's own parameters precede its sub-modules'
            s[b + "pool_token"] = ((1, 1, D), "randn")
        if cfg.ib_addition_mode == "attention":   # cross_attn_ib is registered before the info-bottleneck layer (models/temporal.py:49-57)
            for i in range(F):
                _attention(s, f"{b}cross_attn_ib.{i}.", E)
        if cfg.ib_scale_mode == "fourier":    # GaussianFourierProjection(1, E/2): a fixed random W (models/base_blocks.py:143-148)
            s[b + "ib.W"] = ((1, Eib // 2), "randn")
        elif cfg.ib_scale_mode == "linear":   # nn.Linear(1, E)
            _linear(s, b + "ib.", Eib, 1)
        else:
            _linear(s, b + "ib.residual_projection.", Eib, 1)
            _linear(s, b + "ib.layers.0.", sr, 1)
            s[b + "ib.layers.1.weight"] = ((sr,), "norm_w")
            s[b + "ib.layers.1.bias"] = ((sr,), "norm_b")
            _linear(s, b + "ib.layers.3.", Eib, sr)
        for i in range(F):
            for n in range(3):
                _norm(s, f"{b}ln.exp.{i}.{n}.", E, cfg.LN_type)
        _norm(s, b + "ln.cross.", D, cfg.LN_type)
        for i in range(F):
            _attention(s, f"{b}attn.self.{i}.", E)
        for i in range(F):
            _linear(s, f"{b}mlp.{i}.layers.0.", S, E)
            s[f"{b}mlp.{i}.layers.1.weight"] = ((S,), "norm_w")
            s[f"{b}mlp.{i}.layers.1.bias"] = ((S,), "norm_b")
            _linear(s, f"{b}mlp.{i}.layers.3.", E, S)
        for i in range(F):
            _linear(s, f"{b}proj.{i}.", Eo, E)
        if cfg.exchange_mode == "simple":   # SimpleBlockTemporal adds no parameters (models/temporal.py:304-306)
            continue
        for i in range(F):
            _linear(s, f"{b}cross_down.{i}.", D, E)
        for i in range(F):
            _linear(s, f"{b}cross_up.{i}.", E, D)
        if cfg.exchange_mode == "sea":      # AddBlockTemporal has no cross-attention modules (models/temporal.py:279-289)
            for i in range(F):
                for j in range(F):
                    _attention(s, f"{b}cross_attn.{i}.{j}.", D)
        elif cfg.exchange_mode == "pool":   # one cross-attention per field (field -> pool)
            for i in range(F):
                _attention(s, f"{b}cross_attn.{i}.", D)
        for i in range(F):
            _norm(s, f"{b}ln_cross.{i}.", D, cfg.LN_type)
        if cfg.exchange_mode == "pool":
            _norm(s, b + "ln_pool.", D, cfg.LN_type)
            _linear(s, b + "pool_update.0.", 2 * D, D * F)
            _linear(s, b + "pool_update.2.", D, 2 * D)
    for i in range(F):
        _norm(s, f"ln.{i}.", Eo, cfg.LN_type)
    return s


def recipe_tensor(key: str, shape: Tuple[int, ...], kind: str) -> np.ndarray:
    """Values depend only on (key, shape, kind).  Scales are chosen so that activations stay O(1)
    and every term (biases, norm gains, modulation) is visibly exercised, unlike the reference's
    N(0, 0.02) init under which softmax is near-uniform."""
    rng = np.random.Generator(np.random.PCG64(zlib.crc32(key.encode("utf-8"))))
    if kind == "lin_w":
        fan_in = shape[1]
        return (rng.standard_normal(shape) * (0.8 / np.sqrt(fan_in))).astype(np.float32)
    if kind == "lin_b":
        return (rng.standard_normal(shape) * 0.1).astype(np.float32)
    if kind == "norm_w":
        return (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
    if kind == "norm_b":
        return (0.1 * rng.standard_normal(shape)).astype(np.float32)
    if kind == "randn":
        return rng.standard_normal(shape).astype(np.float32)
    raise ValueError(kind)


def recipe_params(cfg: OracleConfig, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(recipe_tensor(k, shp, kind)).to(dtype) for k, (shp, kind) in param_schema(cfg).items()}


def recipe_inputs(B: int, T: int, cfg: OracleConfig, seed: int = 1234) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Synthetic encoded fields x ~ N(0,1) [B,T,F,E], target ~ N(0,1), condition ib ~ U(0,1) [B,T,1]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    x = rng.standard_normal((B, T, cfg.num_variables, cfg.embed_dim)).astype(np.float32)
    tgt = rng.standard_normal((B, T, cfg.num_variables, cfg.embed_dim)).astype(np.float32)
    ib = rng.random((B, T, 1)).astype(np.float32)
    return torch.from_numpy(x), torch.from_numpy(tgt), torch.from_numpy(ib)


def decode_schema(field_groups, n_inp: int, mlp_hidden: int, embed_dim: int, pre: str = "decoders."):
    """Parameter names / shapes / kinds of the reference's Decode module (models/encoder_decoder.py:126-136), in named_parameters order."""
    sch = {}
    for i, group in enumerate(field_groups):
        sch[f"{pre}{i}.layer1.weight"] = ((mlp_hidden, embed_dim), "lin_w")
        sch[f"{pre}{i}.layer2.weight"] = ((n_inp * len(group), mlp_hidden), "lin_w")
        sch[f"{pre}{i}.layer2.bias"] = ((n_inp * len(group),), "lin_b")
    return sch


def decode_params(field_groups, n_inp: int, mlp_hidden: int, embed_dim: int, dtype=torch.float32):
    return {k: torch.from_numpy(recipe_tensor(k, shp, kind)).to(dtype) for k, (shp, kind) in decode_schema(field_groups, n_inp, mlp_hidden, embed_dim).items()}


def encoder_schema(field_groups, n_inp: int, mlp_hidden: int, num_layers: int, embed_dim: int, pre: str = ""):
    """Parameter names / shapes / kinds of the reference's PointwiseEncode (models/encoder_decoder.py:75-101), in named_parameters order:
    blocks (EncoderBlock: two weight-only LayerNorms, MultiHeadAttention k/q/v + bias-free projection, MLP x4), the final nn.LayerNorm,
    then the per-group downScaleMLPs (created after the blocks)."""
    W = len(field_groups) * embed_dim
    sch = OrderedDict()
    for l in range(num_layers):
        b = f"{pre}blocks.{l}."
        sch[b + "ln_exp1_1.weight"] = ((W,), "norm_w")
        sch[b + "ln_exp1_2.weight"] = ((W,), "norm_w")
        _attention(sch, b + "attn_1.", W)
        _linear(sch, b + "mlp_1.layers.0.", 4 * W, W)
        sch[b + "mlp_1.layers.1.weight"] = ((4 * W,), "norm_w")
        sch[b + "mlp_1.layers.1.bias"] = ((4 * W,), "norm_b")
        _linear(sch, b + "mlp_1.layers.3.", W, 4 * W)
    sch[pre + "ln.weight"] = ((W,), "norm_w")
    sch[pre + "ln.bias"] = ((W,), "norm_b")
    for i, group in enumerate(field_groups):
        sch[f"{pre}encoders.{i}.layer1.weight"] = ((mlp_hidden, n_inp * len(group)), "lin_w")
        sch[f"{pre}encoders.{i}.layer2.weight"] = ((embed_dim, mlp_hidden), "lin_w")
        sch[f"{pre}encoders.{i}.layer2.bias"] = ((embed_dim,), "lin_b")
    return sch


def encoder_params(field_groups, n_inp: int, mlp_hidden: int, num_layers: int, embed_dim: int, dtype=torch.float32):
    return {k: torch.from_numpy(recipe_tensor(k, shp, kind)).to(dtype)
            for k, (shp, kind) in encoder_schema(field_groups, n_inp, mlp_hidden, num_layers, embed_dim).items()}

