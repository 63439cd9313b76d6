"""Shared builder of the two case configurations (reference configs/cylinder_flow.py:2-162, configs/multiphase_flow.py:2-162).

The reference's configs are flat dicts; `get_config_temporal()` inherits the data / mesh / spatial-model keys from
`get_config_spatial()` and adds the temporal model, logging, dataset and optimiser keys.  Here both cases come from one table:
what the two cases share, then the per-case differences, so that the keys `get_model` (sea_amd/train/train_temporal.py) and
`train` consume have the reference's names and values.  Keys this build adds: `dtype` (compute dtype of the HIP kernels),
`world_size` (data-parallel ranks; 1 process per GPU) and `rollout_mode` ('kv' exact incremental decode | 'recompute').
No credential of any kind lives here: wandb is not part of this build.
"""
from __future__ import annotations

from typing import Any, Dict

import torch

# data / mesh keys (the same in both cases: reference configs/*:5-24)
_DATA = dict(save_dir='./checkpoints', field_data_path='./data/CF/all_data/field_data.npy', input_path='./data/CF/all_data/input_data.npy',
             coordinates_path='./data/CF/all_data/coordinates.npy', random_seed=42, dimension='2D', field_groups=[[0, 1], [2]],
             scale_feature_range=None, csv_scale_name='scaler', m=9, n=9, k=None, pad_id=-1, pad_field_value=0)

# spatial autoencoder (frozen at temporal time): what differs is the MLP width and the embedding (reference configs/*:26-32)
_SPATIAL = dict(num_layers=12, n_heads=8, block_size=2024, src_len=0, dropout=0.0, variational=False)
_SPATIAL_CASE = {"cylinder_flow": dict(MLP_hidden=480, embed_dim=16), "multiphase_flow": dict(MLP_hidden=624, embed_dim=32)}

# temporal model + loop keys shared by the cases (reference configs/*:112-162)
_TEMPORAL = dict(num_layers=1, n_heads=8, block_size=2024, scale_ratio=8, src_len=0, down_proj=2, exchange_mode='sea', pos_encoding_mode='learnable',
                 ib_scale_mode='mlp', ib_addition_mode='add', ib_mlp_layers=1, ib_num=1, add_info_after_cross=True,
                 test_mesh_structure=False, perform_initial_test=True, validation_interval=10, full_eval_interval=100, final_save=False,
                 dataset_overlap=0, dataset_time_shifting_flag=False, variational=False, KL_weight_min=0, KL_weight_max=0, epoch_num=3000,
                 run_name='run1', project_name='SEA_Temporal')
_TEMPORAL_CASE = {
    "cylinder_flow": dict(embed_dim=1024, dropout=0.1, LN_type='adaln', batch_size=2, dataset_src_len=399, learning_rate=1e-4, use_wandb=False),
    "multiphase_flow": dict(embed_dim=2048, dropout=0.0, LN_type='ln', batch_size=4, dataset_src_len=199, learning_rate=8e-5, use_wandb=False),
}


def _device() -> str:
    return 'cuda' if torch.cuda.is_available() else 'cpu'


def spatial_config(case: str) -> Dict[str, Any]:
    c: Dict[str, Any] = dict(device=_device(), **_DATA, train_fraction=0.8, val_fraction=0.1)
    c.update(_SPATIAL)
    c.update(_SPATIAL_CASE[case])
    c.update(test_mesh_structure=False, perform_initial_test=True, validation_interval=10, final_save=False, batch_size=128, learning_rate=1e-4,
             KL_weight_min=0, KL_weight_max=0, epoch_num=5000, use_wandb=False, run_name='run1', case_name=case, project_name='SEA_Encoder_Decoder',
             spatial_batch_size=1000, SEA_isolate=True, SEA_mixed=False)
    for k in ('embed_dim', 'n_heads', 'block_size', 'dropout', 'MLP_hidden', 'num_layers', 'src_len', 'variational'):
        c[k + '_spatial'] = c[k]
    return c


def temporal_config(case: str) -> Dict[str, Any]:
    sp = spatial_config(case)
    c: Dict[str, Any] = {k: sp[k] for k in ('device', *_DATA.keys())}
    c.update(train_fraction=0.6, val_fraction=0.2)
    for k in ('MLP_hidden', 'num_layers', 'embed_dim', 'n_heads', 'block_size', 'dropout', 'variational', 'src_len'):
        c[k + '_spatial'] = sp[k]
    c['encoder_decoder_path'] = f"{sp['save_dir']}/encoder_decoder_{sp['case_name']}_{sp['run_name']}.pt"
    c['spatial_batch_size'] = sp['batch_size']
    c.update(_TEMPORAL)
    c.update(_TEMPORAL_CASE[case])
    c['num_fields'] = len(sp['field_groups'])
    # the reference's multiphase temporal config keeps case_name 'cylinder_flow' (configs/multiphase_flow.py:156): it names the checkpoint
    # file `temporal_{case_name}_{run_name}.pt`, so the value is mirrored, not corrected
    c.update(case_name='cylinder_flow', SEA_isolate=sp['SEA_isolate'], SEA_mixed=sp['SEA_mixed'])
    # this build's own keys
    c.update(dtype='bf16', world_size=1, rollout_mode='kv')
    return c
