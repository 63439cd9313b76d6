"""Data-parallel plumbing on CPU with world_size 2 (gloo): batch sharding, the single flat-gradient all-reduce with the 1/world mean,
parameter broadcast and the bitwise-sync check.  The math being reduced is a stand-in (the HIP path needs a GPU); what is under test is
that N ranks with B/N trajectories each end up with the gradient of the global batch and identical parameters."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sea_amd.parallel import allreduce_flat_gradients, broadcast_parameters, parameters_in_sync, shard_batch

        torch.manual_seed(0)
        n_total, n_live = 1000, 840
        gx = torch.randn(8, 16)                      # global batch of 8 "trajectories"
        w = torch.randn(n_total) + rank              # ranks start out of sync on purpose
        assert not parameters_in_sync(w)
        broadcast_parameters(w, src=0)
        assert parameters_in_sync(w)
        xs = shard_batch(gx, rank, world)
        assert xs.shape[0] == 8 // world and torch.equal(xs, gx[rank * 4:(rank + 1) * 4])
        # stand-in loss: mean over the local batch of (sum_j x_bj) * (w . v); its flat gradient is linear in the local batch mean
        v = torch.linspace(-1, 1, n_total)
        local_grad = xs.sum(1).mean() * v
        flat = local_grad.clone()
        flat[n_live:] = 123.0                        # the dead tail must not be touched by the collective
        scale = allreduce_flat_gradients(flat, n_live)
        assert scale == 1.0 / world
        global_grad = gx.sum(1).mean() * v
        assert torch.allclose(flat[:n_live] * scale, global_grad[:n_live], atol=1e-6)
        assert torch.all(flat[n_live:] == 123.0)
        w[:n_live] -= 0.1 * flat[:n_live] * scale     # every rank applies the same update
        assert parameters_in_sync(w)
        # the same reduction in slices: two buckets announced early (out of order, as a backward would), the three remainders by finish()
        from sea_amd.parallel import OverlappedGradientReduce

        flat2 = local_grad.clone()
        flat2[n_live:] = 123.0
        red = OverlappedGradientReduce(flat2, n_live)
        assert red.active
        red.on_bucket(400, 650)
        red.on_bucket(100, 180)
        assert red.finish() == 1.0 / world and red.calls == 5
        assert torch.equal(flat2, flat)              # every element reduced exactly once, the dead tail untouched
        ret[rank] = "ok"
    except Exception as e:  # pragma: no cover
        ret[rank] = repr(e)
    finally:
        dist.destroy_process_group()


def test_flat_allreduce_world2_gloo():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}


def test_single_process_is_identity():
    from sea_amd.parallel import allreduce_flat_gradients, parameters_in_sync, shard_batch

    g = torch.arange(10.0)
    assert allreduce_flat_gradients(g, 8) == 1.0 and torch.equal(g, torch.arange(10.0))
    assert parameters_in_sync(g)
    with pytest.raises(ValueError):
        shard_batch(torch.zeros(7, 2), 0, 2)


def test_bench_parent_launches_ranks_and_fails_cleanly_without_gpus():
    """`python bench.py --gpus 2` as a plain command: the parent starts the rank processes itself (no torch.distributed.run needed) and, in this
    GPU-less container, reports their failure with a non-zero exit code instead of hanging or raising a traceback of its own."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--mode", "train", "--steps", "1", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: this checks the no-GPU failure path only")
    assert r.returncode == 1
    assert "rank exit codes" in r.stderr and "needs an MI355X" in r.stderr
    assert "{" not in r.stdout


def test_train_entry_data_parallel_plumbing(monkeypatch):
    """sea_amd.train.train_temporal: config['world_size'] > 1 without a launcher's environment is an error (never a silent single-process run); a loader
    with a DistributedSampler is recognised (train() then leaves its batches alone), a plain list / DataLoader is sharded by rank."""
    from torch.utils.data import DataLoader, TensorDataset
    from torch.utils.data.distributed import DistributedSampler

    from sea_amd.train import train_temporal as tt

    for k in ("RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(RuntimeError, match="world_size"):
        tt._init_data_parallel({"world_size": 2}, torch.device("cpu"))
    assert tt._init_data_parallel({"world_size": 1}, torch.device("cpu")) == (0, 1, torch.device("cpu"))
    assert tt._init_data_parallel({}, torch.device("cpu")) == (0, 1, torch.device("cpu"))
    ds = TensorDataset(torch.arange(8.0))
    assert not tt._loader_shards_itself(DataLoader(ds, batch_size=4))
    assert not tt._loader_shards_itself([(1, 2, 3, 4)])
    assert tt._loader_shards_itself(DataLoader(ds, batch_size=4, sampler=DistributedSampler(ds, num_replicas=2, rank=0)))


def test_train_entry_binds_each_rank_to_its_own_gpu(monkeypatch):
    """Row j1: with one process per GPU, rank r of train() runs on the launcher's LOCAL_RANK — config['device'] = 'cuda' (the reference configs) is device 0
    in EVERY process otherwise, and RCCL refuses two ranks on one device.  torch.cuda is mocked (no GPU here): the device is made current before any
    allocation and handed back for everything train() creates; without LOCAL_RANK the rank is folded onto the visible devices."""
    from sea_amd.train import train_temporal as tt

    calls = []
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: calls.append(d))
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setenv("RANK", "5")
    monkeypatch.setenv("LOCAL_RANK", "5")
    assert tt._rank_device(torch.device("cuda"), 8) == torch.device("cuda", 5) and calls == [5]
    monkeypatch.delenv("LOCAL_RANK")
    monkeypatch.setenv("RANK", "11")   # second node of a two-node job without LOCAL_RANK: 11 % 8
    assert tt._rank_device(torch.device("cuda"), 16) == torch.device("cuda", 3) and calls == [5, 3]
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)   # the one-GPU test box: both gloo ranks share cuda:0
    monkeypatch.setenv("RANK", "1")
    assert tt._rank_device(torch.device("cuda"), 2) == torch.device("cuda", 0) and calls[-1] == 0
    monkeypatch.setenv("LOCAL_RANK", "3")
    with pytest.raises(RuntimeError, match="LOCAL_RANK"):
        tt._rank_device(torch.device("cuda"), 8)
    n = len(calls)
    assert tt._rank_device(torch.device("cuda"), 1) == torch.device("cuda") and len(calls) == n   # single process: untouched
    assert tt._rank_device(torch.device("cpu"), 2) == torch.device("cpu") and len(calls) == n


def test_uneven_shards_cover_a_ragged_batch():
    """parallel.shard_bounds: any global batch size is split into contiguous, disjoint, covering shards whose sizes differ by at most one, and the
    loss weights (rows * world / B) average to 1 — the SUM all-reduce times 1 / world is then the mean over the global batch."""
    from sea_amd import parallel

    for world in (1, 2, 3, 8):
        for B in (1, 2, 3, 5, 8, 13, 64):
            spans = [parallel.shard_bounds(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
            assert abs(sum(n * world / B for n in sizes) / world - 1.0) < 1e-12


def _train_plumbing_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        from sea_amd.train import train_temporal as tt

        got = tt._init_data_parallel({"world_size": world, "dist_backend": "gloo"}, torch.device("cpu"))   # starts the group from the environment
        assert got == (rank, world, torch.device("cpu")) and dist.is_initialized()
        v = tt._mean_over_ranks(torch.tensor(float(rank + 1)), world)
        assert float(v) == 1.5
        try:
            tt._init_data_parallel({"world_size": 4}, torch.device("cpu"))
            ret[rank] = "no error for a world-size mismatch"
        except RuntimeError:
            ret[rank] = "ok"
    except Exception as e:  # pragma: no cover
        ret[rank] = repr(e)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_train_entry_starts_its_group_from_the_launcher_environment():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_train_plumbing_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}
