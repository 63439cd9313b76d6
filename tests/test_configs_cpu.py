"""Config mirrors (reference configs/cylinder_flow.py:73-162, configs/multiphase_flow.py:73-162): key / value equality with the imported
reference dicts in the build container; the shipped temporal-model values pinned literally everywhere else."""
import importlib.util
import os

import pytest

from sea_amd.configs import get_config
from sea_amd.configs import cylinder_flow, multiphase_flow

REF = "/root/reference/configs"
MODEL_KEYS = ('num_layers', 'embed_dim', 'n_heads', 'block_size', 'scale_ratio', 'src_len', 'num_fields', 'down_proj', 'dropout', 'exchange_mode',
              'pos_encoding_mode', 'ib_scale_mode', 'ib_addition_mode', 'ib_mlp_layers', 'ib_num', 'add_info_after_cross', 'LN_type')


def test_shipped_temporal_values():
    cyl, mp = cylinder_flow.get_config_temporal(), multiphase_flow.get_config_temporal()
    assert [cyl[k] for k in MODEL_KEYS] == [1, 1024, 8, 2024, 8, 0, 2, 2, 0.1, 'sea', 'learnable', 'mlp', 'add', 1, 1, True, 'adaln']
    assert [mp[k] for k in MODEL_KEYS] == [1, 2048, 8, 2024, 8, 0, 2, 2, 0.0, 'sea', 'learnable', 'mlp', 'add', 1, 1, True, 'ln']
    assert (cyl['batch_size'], cyl['dataset_src_len'], cyl['learning_rate']) == (2, 399, 1e-4)
    assert (mp['batch_size'], mp['dataset_src_len'], mp['learning_rate']) == (4, 199, 8e-5)
    for c in (cyl, mp):
        assert c['dtype'] == 'bf16' and c['world_size'] == 1 and c['rollout_mode'] in ('kv', 'recompute')
        assert not any('KEY' in k.upper() for k in c), "no credential keys in the build's configs"
    assert get_config('cylinder_flow', 'temporal') == cyl and get_config('multiphase_flow', 'spatial') == multiphase_flow.get_config_spatial()


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists only in the build container")
@pytest.mark.parametrize("case", ["cylinder_flow", "multiphase_flow"])
@pytest.mark.parametrize("kind", ["temporal", "spatial"])
def test_keys_and_values_equal_reference(case, kind):
    import sys
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location("ref_cfg_" + case, f"{REF}/{case}.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ref = getattr(mod, "get_config_" + kind)()
    mine = get_config(case, kind)
    for k, v in ref.items():
        if 'KEY' in k.upper() or k == 'use_wandb':   # credentials are never mirrored; wandb is not part of this build
            continue
        assert k in mine, k
        assert mine[k] == v, (k, mine[k], v)
    assert set(mine) - set(ref) <= {'dtype', 'world_size', 'rollout_mode'}
