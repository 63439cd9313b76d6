"""Host logic of the model mirror, on CPU: constructor surface, checkpoint schema, error behaviour, flat-buffer layout."""
import numpy as np
import pytest
import torch

from oracle.recipe import param_schema
from oracle.sea_oracle import OracleConfig
from sea_amd.models.temporal import SEABlockTemporal, TemporalModel, create_block_temporal


def make(cfg: OracleConfig, **kw):
    return TemporalModel(cfg.num_layers, cfg.embed_dim, cfg.n_heads, cfg.max_len, cfg.scale_ratio, cfg.src_len, cfg.num_variables,
                         cfg.down_proj, 0.0, cfg.exchange_mode, "learnable", cfg.ib_scale_mode, cfg.ib_addition_mode, 1, 1, cfg.add_info_after_cross, cfg.LN_type, **kw)


@pytest.mark.parametrize("cfg", [OracleConfig(1, 64, 4, 24, 8, 0, 3, 2, True, "adaln"), OracleConfig(2, 64, 4, 24, 8, 0, 2, 2, True, "ln"),
                                 OracleConfig(2, 64, 4, 24, 8, 0, 3, 2, True, "adaln", "addition"), OracleConfig(1, 64, 4, 24, 8, 0, 2, 2, False, "ln", "simple"),
                                 OracleConfig(1, 64, 4, 24, 8, 0, 2, 2, True, "adaln", "sea", "none"),
                                 OracleConfig(2, 64, 4, 24, 8, 0, 3, 2, True, "adaln", "sea", "add", "fourier"), OracleConfig(1, 64, 4, 24, 8, 0, 2, 2, True, "ln", "sea", "add", "linear"),
                                 OracleConfig(2, 64, 4, 24, 8, 0, 3, 2, True, "adaln", "pool"), OracleConfig(1, 64, 4, 24, 8, 0, 2, 2, True, "ln", "pool"),
                                 OracleConfig(2, 64, 4, 24, 8, 0, 3, 2, True, "adaln", "sea", "attention"),
                                 OracleConfig(1, 64, 4, 24, 8, 0, 2, 2, False, "ln", "simple", "attention", "fourier"),
                                 OracleConfig(2, 64, 4, 24, 8, 0, 3, 2, False, "adaln", "sea", "concat"),
                                 OracleConfig(1, 64, 4, 24, 8, 0, 2, 2, False, "ln", "addition", "concat", "linear")])
def test_parameter_schema_matches_reference(cfg):
    """Names, order and shapes equal the reference's named_parameters() (oracle/recipe.param_schema is asserted equal to the
    reference's own by tests/golden/make_fixtures.py)."""
    m = make(cfg)
    schema = param_schema(cfg)
    named = list(m.named_parameters())
    assert [k for k, _ in named] == list(schema.keys())
    for k, p in named:
        assert tuple(p.shape) == tuple(schema[k][0]), k


def test_state_dict_buffers_and_roundtrip():
    cfg = OracleConfig(1, 64, 4, 24, 8, 2, 2, 2, True, "adaln")
    m = make(cfg)
    sd = m.state_dict()
    # buffers of the reference's checkpoint schema
    t = sd["blocks.0.attn.self.0.tril"]
    assert t.shape == (1, 1, 24, 24) and torch.equal(t[0, 0], torch.tril(torch.ones(24, 24), diagonal=2))
    assert sd["blocks.0.cross_attn.1.0.tril"].shape == (1, 1, 24, 24)
    assert sd["blocks.0.attn.self.1.freqs_cis"].dtype == torch.complex64 and sd["blocks.0.attn.self.1.freqs_cis"].shape == (24, 8)
    assert sd["blocks.0.cross_attn.0.1.freqs_cis"].shape == (24, 4)
    assert sd["blocks.0.pos_encoder.pe"].shape == (1, 5000, 32)
    n_buffers = sum(1 for k in sd if k.endswith(("tril", "freqs_cis", "pos_encoder.pe")))
    assert len(sd) == len(param_schema(cfg)) + n_buffers
    m2 = make(cfg)
    missing, unexpected = m2.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    for (k, a), (_, b) in zip(m.named_parameters(), m2.named_parameters()):
        assert torch.equal(a, b), k


def test_init_matches_reference_recipe():
    torch.manual_seed(0)
    m = make(OracleConfig(1, 64, 4, 24, 8, 0, 2, 2, True, "adaln"))
    sd = dict(m.named_parameters())
    assert abs(float(sd["blocks.0.attn.self.0.q.weight"].std()) - 0.02) < 2e-3
    assert float(sd["blocks.0.attn.self.0.q.bias"].abs().max()) == 0
    assert torch.equal(sd["ln.0.weight"], torch.ones(64)) and torch.equal(sd["ln.0.bias"], torch.zeros(64))
    assert torch.equal(sd["blocks.0.mlp.0.layers.1.weight"], torch.ones(512))


def test_error_behaviour():
    with pytest.raises(ValueError):
        TemporalModel(1, 64, 4, 24, 8, 0, 2, exchange_mode="bogus")
    with pytest.raises(ValueError):
        TemporalModel(1, 64, 4, 24, 8, 0, 2, exchange_mode="sea", ib_scale_mode="mlp", pos_encoding_mode="bogus")
    with pytest.raises(ValueError):
        TemporalModel(1, 64, 4, 24, 8, 0, 2, exchange_mode="sea", ib_scale_mode="mlp", LN_type="bogus")
    with pytest.raises(ValueError):
        TemporalModel(1, 64, 4, 24, 8, 0, 2, exchange_mode="sea", ib_scale_mode="mlp", ib_addition_mode="bogus")
    wide = TemporalModel(1, 64, 4, 24, 8, 0, 2, exchange_mode="sea", ib_scale_mode="mlp", ib_addition_mode="concat")   # rows of 64 + 64 inside the block
    assert wide.blocks[0].internal_embed_dim == 128 and wide.blocks[0].proj[0].weight.shape == (64, 128) and wide.blocks[0].ib.layers[-1].out_features == 64
    with pytest.raises(ValueError):
        create_block_temporal("nope")
    m = make(OracleConfig(1, 64, 4, 24, 8, 0, 2, 2, True, "adaln"))
    with pytest.raises(AssertionError):
        m(torch.zeros(1, 4, 3, 64), torch.zeros(1, 4, 1))
    with pytest.raises(RuntimeError, match="no CPU"):
        m(torch.zeros(1, 4, 2, 64), torch.zeros(1, 4, 1))
    blk = m.blocks[0]
    assert isinstance(blk, SEABlockTemporal)
    with pytest.raises(AssertionError):
        blk(torch.zeros(1, 4, 64), x_add=torch.zeros(1, 4, 1))


def test_flat_layout_offsets():
    """Flat-buffer layout rules the engine relies on: live parameters first, q|k|v adjacent in that order, 8-element alignment."""
    from sea_amd.engine import dead_prefixes, _round_up, _QKV, _QKV_RANK  # noqa: F401

    cfg = OracleConfig(1, 64, 4, 24, 8, 0, 3, 2, True, "adaln")
    m = make(cfg)
    dead = dead_prefixes(m)
    names = [k for k, _ in m.named_parameters()]
    n_dead = sum(any(k.startswith(d) for d in dead) for k in names)
    from oracle.sea_oracle import live_param_keys

    assert len(names) - n_dead == len(live_param_keys({k: None for k in names}, cfg))
