"""Host-side mirror of the hot-path half of the reference's train/train_temporal.py: get_model (:190-223) and the train loop
(:232-348).  Dataset preparation (process_data / get_datasets, :13-189) needs the mesh data and the frozen spatial autoencoder and
is out of scope: `train` takes ready DataLoaders of (data, target, original, ib) batches — what the reference's
TemporalDataset yields (utils/data_processors.py:444-452) — through config['loaders'] = (train, val, test).

Data parallel (BASELINE.json north_star: replicate the model, split the trajectory batch, ONE gradient all-reduce per step; the reference itself
is single-process): `train` run by N processes — one per GPU, torch.distributed initialised by the launcher, or by `train` itself from the usual
RANK / WORLD_SIZE / MASTER_* environment when config['world_size'] > 1 — starts every rank from rank 0's parameters, gives rank r the
trajectories [r B/N, (r+1) B/N) of every training batch (a loader that already shards — a DistributedSampler — is left alone), and reduces
the flat gradient buffer once per step (inside the fused step, or in FlatAdamW.step() on the autograd path).  Metrics are averaged over the
ranks; rank 0 writes the checkpoints."""
from __future__ import annotations

import time
from typing import Any, Dict, Tuple

import torch

from .. import parallel
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


def _init_data_parallel(config: Dict[str, Any], device: torch.device) -> Tuple[int, int]:
    """(rank, world) of this process.  config['world_size'] > 1 without a process group: initialise one from the environment a launcher
    (torch.distributed.run, bench.py) sets — backend "nccl" (= RCCL) on the GPU; config['dist_backend'] overrides (tests use gloo)."""
    import torch.distributed as dist

    want = int(config.get('world_size', 1) or 1)
    if want > 1 and dist.is_available() and not dist.is_initialized():
        import os

        if "RANK" not in os.environ or "WORLD_SIZE" not in os.environ:
            raise RuntimeError(f"sea_amd.train: config['world_size'] = {want} needs one process per GPU started by a launcher that sets RANK / WORLD_SIZE / "
                               "MASTER_ADDR / MASTER_PORT (python -m torch.distributed.run --nproc-per-node N ...), or an initialised torch.distributed group")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(config.get('dist_backend', 'nccl' if device.type == 'cuda' else 'gloo'))
    world, rank = parallel.world_size(), parallel.rank()
    if want > 1 and world != want:
        raise RuntimeError(f"sea_amd.train: config['world_size'] = {want} but the process group has {world} ranks")
    return rank, world


def _loader_shards_itself(loader) -> bool:
    from torch.utils.data.distributed import DistributedSampler

    return isinstance(getattr(loader, 'sampler', None), DistributedSampler) or isinstance(getattr(getattr(loader, 'batch_sampler', None), 'sampler', None), DistributedSampler)


def _mean_over_ranks(value: torch.Tensor, world: int) -> torch.Tensor:
    if world > 1:
        import torch.distributed as dist

        value = value.clone()
        dist.all_reduce(value, op=dist.ReduceOp.SUM)
        value /= world
    return value


def train(config: Dict[str, Any], error_tracker):
    """The reference's epoch loop (:247-339): teacher-forced next-step training, periodic validation, best-model checkpoints.  One step is the
    reference's `zero_grad; model(data, ib); loss_fn; backward; step` (:252-258) — as ONE fused launch sequence (engine.train_step: no ATen kernel,
    the gradient all-reduce in slices under the backward) when the loss is the MSE the reference uses, or literally through autograd
    (config['fused_step'] = False): both are data-parallel (module docstring)."""
    if 'loaders' not in config:
        raise RuntimeError("sea_amd.train: pass config['loaders'] = (trainLoader, validationLoader, testLoader); building them from "
                           "raw mesh data needs the spatial autoencoder pipeline, which this build does not cover")
    trainLoader, validationLoader, _ = config['loaders']
    device = torch.device(config['device'])
    rank, world = _init_data_parallel(config, device)
    model, loss_fn, optimizer = get_model(config, device)
    scheduler = None
    if isinstance(optimizer, tuple):
        optimizer, scheduler = optimizer
    if world > 1:
        eng = model.engine(device)
        parallel.broadcast_parameters(eng.params.flat32)   # every rank starts from rank 0's parameters (the flat buffer IS the parameters)
        eng.params.sync(force=True)
        eng.params.sync_transposed(force=True)
    shard = world > 1 and not _loader_shards_itself(trainLoader)
    fused = bool(config.get('fused_step', True)) and isinstance(loss_fn, SeaMSELoss)
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
            if shard:   # rank r's trajectories of the global batch (parallel.shard_batch raises when the batch does not divide)
                data, target, ib = (parallel.shard_batch(t, rank, world) for t in (data, target, ib))
            data, target, ib = data.to(device), target.to(device), ib.to(device)
            if fused:
                loss = model.engine(device).train_step(data.float(), target.float(), ib.float(), optimizer)
                loss_sum += loss.reshape(())
                continue
            optimizer.zero_grad()
            outputs = model(data, ib)
            loss = loss_fn(outputs, target)
            loss.backward()
            optimizer.step()
            loss_sum += loss.detach()
        if scheduler is not None:
            scheduler.step()
        train_loss = _mean_over_ranks(loss_sum, world).item() / max(len(trainLoader), 1)
        error_tracker.record_error("train", epoch, {"Loss": train_loss})
        if epoch % config.get('validation_interval', 1) == 0 or epoch == config['epoch_num']:
            model.eval()
            val_sum, n_val = torch.zeros((), device=device), 0
            with torch.no_grad():
                for v_data, v_target, _, v_ib in validationLoader:   # every rank validates on the whole loader: identical models give identical metrics
                    v_out = model(v_data.to(device), v_ib.to(device))
                    val_sum += loss_fn(v_out, v_target.to(device))
                    n_val += 1
            val_loss = val_sum.item() / max(n_val, 1)
            val_metrics = {"Loss": val_loss}
            if epoch % full_eval_interval == 0:
                res = full_autoregressive_evaluation(model, validationLoader, loss_fn, device, processor, mesh_processor, config, epoch, plot_traj=False)
                if res is not None:   # None: an empty validation loader
                    val_metrics["Full_Encoded_Rel_MSE"] = res['encoded_rel_mse']
                    if processor is not None and mesh_processor is not None:
                        val_metrics["Full_Decoded_Rel_MSE"] = res['decoded_rel_mse']
                    # the reference keeps a second checkpoint on the best rollout error (:305-318), the decoded one when it can be computed
                    score = res['decoded_rel_mse'] if res['decoded_rel_mse'] == res['decoded_rel_mse'] else res['encoded_rel_mse']
                    if score < best_rollout and config.get('save_dir'):
                        best_rollout = score
                        if rank == 0:
                            path = f"{config['save_dir']}/temporal_Checkpoint_{config.get('case_name', 'case')}_{config.get('run_name', 'run')}.pt"
                            torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, path)
                            print("--- New Best Rollout Checkpoint Saved ---")
            error_tracker.record_error("val", epoch, val_metrics)
            if rank == 0:
                print(f"\nEpoch: {epoch}/{config['epoch_num']}  Train Loss: {train_loss:.8f}  " + "  ".join(f"{k}: {v:.8f}" for k, v in val_metrics.items()))
            if val_loss < best_val:
                best_val = val_loss
                if config.get('save_dir') and rank == 0:
                    path = f"{config['save_dir']}/temporal_{config.get('case_name', 'case')}_{config.get('run_name', 'run')}.pt"
                    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, path)
                    print("--- New Best Model Saved ---")
    if rank == 0:
        print(f"Total training time: {time.time() - start:.2f} seconds")
    error_tracker.finish()
    return model
