"""Host-side mirror of the hot-path half of the reference's train/train_temporal.py: get_model (:190-223) and the train loop
(:232-348).  Dataset preparation (process_data / get_datasets, :13-189) needs the mesh data and the frozen spatial autoencoder and
is out of scope: `train` takes ready DataLoaders of (data, target, original, ib) batches — what the reference's
TemporalDataset yields (utils/data_processors.py:444-452) — through config['loaders'] = (train, val, test)."""
from __future__ import annotations

import time
from typing import Any, Dict, Tuple

import torch

from ..models.temporal import TemporalModel
from ..utils.train_utils import SeaMSELoss, full_autoregressive_evaluation, initialize_optimizer


def get_model(config: Dict[str, Any], device: torch.device) -> Tuple[TemporalModel, torch.nn.Module, torch.optim.Optimizer]:
    """Same 17 positional config keys as the reference (:191-207); returns (model, loss_fn, optimizer)."""
    model = TemporalModel(config['num_layers'], config['embed_dim'], config['n_heads'], config['block_size'], config['scale_ratio'],
                          config['src_len'], config['num_fields'], config['down_proj'], config['dropout'], config['exchange_mode'],
                          config['pos_encoding_mode'], config['ib_scale_mode'], config['ib_addition_mode'], config['ib_mlp_layers'],
                          config['ib_num'], config['add_info_after_cross'], config['LN_type'])
    if config.get('load_pretrained', False):
        model.load_state_dict(torch.load(config['pretrained_model_path'], map_location='cpu'))
        print(f"Loaded pre-trained model from {config['pretrained_model_path']}")
    if 'dtype' in config:
        model.set_compute_dtype(config['dtype'])
    model = model.to(device)
    optimizer = initialize_optimizer(model, config)
    if config.get('variational', False):
        raise NotImplementedError("sea_amd: the variational loss belongs to the spatial autoencoder path (out of scope)")
    return model, SeaMSELoss(), optimizer


def train(config: Dict[str, Any], error_tracker):
    """The reference's epoch loop (:247-339): teacher-forced next-step training, periodic validation, best-model checkpoints."""
    if 'loaders' not in config:
        raise RuntimeError("sea_amd.train: pass config['loaders'] = (trainLoader, validationLoader, testLoader); building them from "
                           "raw mesh data needs the spatial autoencoder pipeline, which this build does not cover")
    trainLoader, validationLoader, _ = config['loaders']
    device = torch.device(config['device'])
    model, loss_fn, optimizer = get_model(config, device)
    scheduler = None
    if isinstance(optimizer, tuple):
        optimizer, scheduler = optimizer
    start = time.time()
    best_val = float('inf')
    best_rollout = float('inf')
    processor, mesh_processor = config.get('processor'), config.get('mesh_processor')   # the reference builds both in get_datasets (:225-230)
    error_tracker.log_model(model, loss_fn, optimizer)
    full_eval_interval = config.get('full_eval_interval', 50)
    for epoch in range(1, config['epoch_num'] + 1):
        model.train()
        loss_sum = torch.zeros((), device=device)  # accumulated on the device: one host sync per epoch, not per step
        for data, target, _, ib in trainLoader:
            data, target, ib = data.to(device), target.to(device), ib.to(device)
            optimizer.zero_grad()
            outputs = model(data, ib)
            loss = loss_fn(outputs, target)
            loss.backward()
            optimizer.step()
            loss_sum += loss.detach()
        if scheduler is not None:
            scheduler.step()
        train_loss = loss_sum.item() / max(len(trainLoader), 1)
        error_tracker.record_error("train", epoch, {"Loss": train_loss})
        if epoch % config.get('validation_interval', 1) == 0 or epoch == config['epoch_num']:
            model.eval()
            val_sum, n_val = torch.zeros((), device=device), 0
            with torch.no_grad():
                for v_data, v_target, _, v_ib in validationLoader:
                    v_out = model(v_data.to(device), v_ib.to(device))
                    val_sum += loss_fn(v_out, v_target.to(device))
                    n_val += 1
            val_loss = val_sum.item() / max(n_val, 1)
            val_metrics = {"Loss": val_loss}
            if epoch % full_eval_interval == 0:
                res = full_autoregressive_evaluation(model, validationLoader, loss_fn, device, processor, mesh_processor, config, epoch, plot_traj=False)
                val_metrics["Full_Encoded_Rel_MSE"] = res['encoded_rel_mse']
                if processor is not None and mesh_processor is not None:
                    val_metrics["Full_Decoded_Rel_MSE"] = res['decoded_rel_mse']
                # the reference keeps a second checkpoint on the best rollout error (:305-318), the decoded one when it can be computed
                score = res['decoded_rel_mse'] if res['decoded_rel_mse'] == res['decoded_rel_mse'] else res['encoded_rel_mse']
                if score < best_rollout and config.get('save_dir'):
                    best_rollout = score
                    path = f"{config['save_dir']}/temporal_Checkpoint_{config.get('case_name', 'case')}_{config.get('run_name', 'run')}.pt"
                    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, path)
                    print("--- New Best Rollout Checkpoint Saved ---")
            error_tracker.record_error("val", epoch, val_metrics)
            print(f"\nEpoch: {epoch}/{config['epoch_num']}  Train Loss: {train_loss:.8f}  " + "  ".join(f"{k}: {v:.8f}" for k, v in val_metrics.items()))
            if val_loss < best_val:
                best_val = val_loss
                if config.get('save_dir'):
                    path = f"{config['save_dir']}/temporal_{config.get('case_name', 'case')}_{config.get('run_name', 'run')}.pt"
                    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, path)
                    print("--- New Best Model Saved ---")
    print(f"Total training time: {time.time() - start:.2f} seconds")
    error_tracker.finish()
    return model
