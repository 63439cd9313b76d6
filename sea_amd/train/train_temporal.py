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


def _rank_device(device: torch.device, want: int) -> torch.device:
    """The GPU of THIS rank.  The reference configs say config['device'] = 'cuda' (configs/cylinder_flow.py:5) — device 0 in every process; with one process
    per GPU rank r must run on the launcher's LOCAL_RANK (torch.distributed.run sets it; without it: RANK modulo the visible devices, which is 0 for
    every rank of a one-GPU box).  The device is made CURRENT before anything is allocated: the native launches go to torch's current stream of the
    current device (_native.stream_ptr), and RCCL refuses two ranks on one device."""
    import os

    if device.type != 'cuda' or want <= 1:
        return device
    n = torch.cuda.device_count()
    local = os.environ.get("LOCAL_RANK")
    if local is None:
        local = int(os.environ.get("RANK", "0")) % max(n, 1)
    local = int(local)
    if n and local >= n:
        raise RuntimeError(f"sea_amd.train: LOCAL_RANK = {local} but only {n} GPU(s) are visible to this process")
    torch.cuda.set_device(local)
    return torch.device('cuda', local)


def _init_data_parallel(config: Dict[str, Any], device: torch.device) -> Tuple[int, int, torch.device]:
    """(rank, world, device) of this process.  config['world_size'] > 1: the rank is bound to its own GPU first (_rank_device), then — without a process
    group — one is initialised from the environment a launcher (torch.distributed.run, bench.py) sets: backend "nccl" (= RCCL) on the GPU, handed the
    rank's device (`device_id`: RCCL communicators are then created eagerly on that device); config['dist_backend'] overrides (tests use gloo)."""
    import torch.distributed as dist

    want = int(config.get('world_size', 1) or 1)
    if want > 1 and dist.is_available():
        device = _rank_device(device, want)
    if want > 1 and dist.is_available() and not dist.is_initialized():
        import os

        if "RANK" not in os.environ or "WORLD_SIZE" not in os.environ:
            raise RuntimeError(f"sea_amd.train: config['world_size'] = {want} needs one process per GPU started by a launcher that sets RANK / WORLD_SIZE / "
                               "MASTER_ADDR / MASTER_PORT (python -m torch.distributed.run --nproc-per-node N ...), or an initialised torch.distributed group")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = config.get('dist_backend', 'nccl' if device.type == 'cuda' else 'gloo')
        if backend == 'nccl' and device.type == 'cuda':
            dist.init_process_group(backend, device_id=device)
        else:
            dist.init_process_group(backend)
    world, rank = parallel.world_size(), parallel.rank()
    if want > 1 and world != want:
        raise RuntimeError(f"sea_amd.train: config['world_size'] = {want} but the process group has {world} ranks")
    return rank, world, device


class _RankZeroTracker:
    """The caller's error tracker on rank 0, a no-op elsewhere: one logger run and one set of records per JOB (every rank holds the same averaged numbers)."""

    def __init__(self, tracker, rank: int):
        self._t, self._on = tracker, rank == 0

    def __getattr__(self, name):
        attr = getattr(self._t, name)
        if self._on or not callable(attr):
            return attr
        return lambda *a, **k: None


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
    rank, world, device = _init_data_parallel(config, device)
    if world > 1:
        error_tracker = _RankZeroTracker(error_tracker, rank)   # one logger run / one set of records per job
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
    skipped_small = False
    for epoch in range(1, config['epoch_num'] + 1):
        model.train()
        loss_sum = torch.zeros((), device=device)  # accumulated on the device: one host sync per epoch, not per step
        n_batches = 0
        for data, target, _, ib in trainLoader:
            weight = 1.0
            if shard:
                # rank r's trajectories of the global batch.  A batch that does not divide by the world size — the reference's loader has no drop_last
                # (train/train_temporal.py:84), so the last batch of an epoch is usually ragged — is split unevenly and the rank's mean loss weighted by its
                # share, so that SUM over ranks / world is the mean over the global batch; a batch with fewer trajectories than ranks is skipped by
                # EVERY rank (a rank without rows would have no forward to run the fused step around), once with a warning
                Bg = data.shape[0]
                if Bg < world:
                    if not skipped_small:
                        skipped_small = True
                        if rank == 0:
                            print(f"sea_amd.train: a batch of {Bg} trajectories cannot be split over {world} ranks: such batches are skipped (use drop_last or a batch size >= world_size)")
                    continue
                lo, hi = parallel.shard_bounds(Bg, rank, world)
                weight = (hi - lo) * world / Bg
                data, target, ib = (t[lo:hi] for t in (data, target, ib))
            n_batches += 1
            data, target, ib = data.to(device), target.to(device), ib.to(device)
            if fused:
                loss = model.engine(device).train_step(data.float(), target.float(), ib.float(), optimizer, loss_weight=weight)
                loss_sum += loss.reshape(()) * weight
                continue
            optimizer.zero_grad()
            outputs = model(data, ib)
            loss = loss_fn(outputs, target)
            (loss * weight if weight != 1.0 else loss).backward()
            optimizer.step()
            loss_sum += loss.detach() * weight
        if scheduler is not None:
            scheduler.step()
        train_loss = _mean_over_ranks(loss_sum, world).item() / max(n_batches, 1)
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
                # every rank evaluates (identical models: identical numbers, and the checkpoint decision below must agree); rank 0 alone writes the CSV
                res = full_autoregressive_evaluation(model, validationLoader, loss_fn, device, processor, mesh_processor,
                                                     (config if rank == 0 else {**config, 'save_dir': None}), epoch, plot_traj=False)
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
