"""Data-parallel train step on the device (SURVEY.md §8(e) "Verification"): engine.train_step at world size N against the single-process step on
the global batch — gradients after the all-reduce equal the global-batch gradients (<= 1e-6, fp32), parameters stay bit-identical across ranks.
The ranks of this test share the box's one GPU and reduce through gloo (two RCCL ranks cannot share a device); on a multi-GPU node the same body
runs one rank per GPU over RCCL when SEA_TEST_DP_BACKEND=nccl.  The step reduces the gradient buffer in slices (sea_amd/parallel.py:
OverlappedGradientReduce; the flat layout is ordered by the phase of the backward that completes a gradient, engine.grad_phase): the equality with the
single-process gradients covers every slice boundary."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

CFGS = {"cfg3_like": (1, 64, 4, 48, 8, 0, 3, 2, True, "adaln"),                 # one layer, AdaLN, info-bottleneck step behind the exchange (the benchmark's structure)
        "two_layers_ln_pre": (2, 64, 4, 48, 8, 0, 2, 2, False, "ln")}             # two layers, plain LayerNorm, info-bottleneck step in front of the block
CFG = CFGS["cfg3_like"]
B_GLOBAL, T, STEPS, LR = 4, 40, 3, 1e-3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(dtype="fp32", cfg="cfg3_like"):
    from oracle import sea_oracle as O
    from tests.test_model_gpu import build

    return build(O.OracleConfig(*CFGS[cfg]), dtype).train()


def _data(cfg="cfg3_like"):
    from oracle import sea_oracle as O
    from oracle.recipe import recipe_inputs

    return recipe_inputs(B_GLOBAL, T, O.OracleConfig(*CFGS[cfg]), seed=31)


def _run_steps(x, tgt, ib, world, rank, cfg="cfg3_like"):
    from sea_amd.parallel import parameters_in_sync, shard_batch
    from sea_amd.utils.train_utils import initialize_optimizer

    m = _model(cfg=cfg)
    eng = m.engine()
    opt = initialize_optimizer(m, {"learning_rate": LR})
    xs, ts, cs = (shard_batch(t, rank, world).cuda().contiguous() for t in (x, tgt, ib))
    grads1 = None
    for step in range(STEPS):
        eng.train_step(xs, ts, cs, opt)
        if step == 0:
            grads1 = (eng.grads[:eng.params.n_live] * opt.grad_scale).cpu()   # the mean gradient AdamW consumed
        assert parameters_in_sync(eng.params.flat32)
        if world > 1:
            # three collectives, every element once: the MLP + proj slice and the exchange / norm / condition-MLP slice go early (each as soon as the backward
            # has finished it, under the launches that follow), the self-attention slice after the backward — most of the buffer is on its way before then
            bk = eng.train_plan(xs.shape[0], xs.shape[1]).grad_buckets()
            assert eng.last_allreduce_calls == len(bk) + 1, (eng.last_allreduce_calls, bk)
            assert bk[0][1] == 0 and all(a[2] == b[1] and a[0] < b[0] for a, b in zip(bk, bk[1:]))     # contiguous slices, in the order the backward finishes them
            if cfg == "cfg3_like":
                assert len(bk) == 2 and bk[1][2] / eng.params.n_live > 0.7, bk
            else:
                assert len(bk) == 5, bk    # two layers: (MLP, middle, self-attention) of the last layer, (MLP, middle) of the first
    return grads1, eng.params.flat32[:eng.params.n_live].cpu()


def _worker(rank, world, port, backend, ret, cfg):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank if backend == "nccl" else 0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        x, tgt, ib = _data(cfg)
        g, p = _run_steps(x, tgt, ib, world, rank, cfg)
        ret[rank] = (g.numpy(), p.numpy())
    except Exception as e:  # pragma: no cover
        ret[rank] = repr(e)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,cfg", [(1, "cfg3_like"), (2, "cfg3_like"), (2, "two_layers_ln_pre")])
def test_train_step_world_n_equals_single_process_global_batch(world, cfg):
    backend = os.environ.get("SEA_TEST_DP_BACKEND", "gloo")
    if backend == "nccl" and torch.cuda.device_count() < world:
        pytest.skip("needs one GPU per rank")
    x, tgt, ib = _data(cfg)
    g_ref, p_ref = _run_steps(x, tgt, ib, 1, 0, cfg)  # single process, the global batch
    if world == 1:
        g2, p2 = _run_steps(x, tgt, ib, 1, 0, cfg)    # the body at N = 1: run-to-run agreement (fp32 atomics order)
        assert np.linalg.norm(g2.numpy() - g_ref.numpy()) <= 1e-6 * np.linalg.norm(g_ref.numpy())
        return
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), backend, ret, cfg), nprocs=world, join=True)
    got = dict(ret)
    assert all(not isinstance(v, str) for v in got.values()), got
    for r in range(world):
        g, p = got[r]
        assert np.linalg.norm(g - g_ref.numpy()) <= 1e-6 * np.linalg.norm(g_ref.numpy()), r
        # three AdamW steps at lr 1e-3: parameters within the update's own rounding of the single-process run
        assert np.abs(p - p_ref.numpy()).max() <= 2e-5, r
    assert np.array_equal(got[0][1], got[1][1])       # bit-identical across ranks


def _rccl_worker(rank, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["SEA_DP_REHEARSE"] = "1"
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        x, tgt, ib = _data()
        from sea_amd.utils.train_utils import initialize_optimizer

        m = _model()
        eng = m.engine()
        opt = initialize_optimizer(m, {"learning_rate": LR})
        xs, ts, cs = (t.cuda().contiguous() for t in (x, tgt, ib))
        calls = []
        for _ in range(STEPS):
            eng.train_step(xs, ts, cs, opt)
            calls.append(int(eng.last_allreduce_calls))
        torch.cuda.synchronize()
        ret[0] = (eng.params.flat32[:eng.params.n_live].cpu().numpy(), calls, dist.get_backend())
    except Exception as e:  # pragma: no cover
        import traceback

        ret[0] = repr(e) + "\n" + traceback.format_exc()
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_rccl_collectives_rehearsed_on_one_rank():
    """The RCCL path on the one GPU a test box has (SEA_DP_REHEARSE=1, sea_amd/parallel.py): a process group of ONE `nccl` rank issues the three gradient
    collectives of a data-parallel step — two asynchronous slices under the backward, the rest behind it — on the engine's flat buffer.  A one-rank SUM is
    the identity, so three AdamW steps must land on the single-process parameters (1e-6: the fp32 atomics of the weight gradients are unordered)."""
    x, tgt, ib = _data()
    _, p_ref = _run_steps(x, tgt, ib, 1, 0)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rccl_worker, args=(_free_port(), ret), nprocs=1, join=True)
    got = ret[0]
    assert not isinstance(got, str), got
    p, calls, backend = got
    assert backend == "nccl"
    assert calls == [3] * STEPS, calls
    assert np.abs(p - p_ref.numpy()).max() <= 2e-5


# ---------------------------------------------------------------------------------------------------------------- the named entry: train(config, tracker)
def _train_config(save_dir, fused, world):
    from sea_amd.configs import get_config

    config = get_config("cylinder_flow", "temporal")
    config.update(device="cuda", save_dir=save_dir, embed_dim=64, n_heads=4, block_size=32, num_fields=3, dropout=0.0, dtype="fp32", learning_rate=1e-3,
                  epoch_num=3, validation_interval=3, full_eval_interval=100, run_name=f"dp{world}", rollout_mode="kv", fused_step=fused,
                  world_size=world, dist_backend=os.environ.get("SEA_TEST_DP_BACKEND", "gloo"))
    torch.manual_seed(3)
    base = torch.randn(8, 13, 3, 64).cumsum(dim=1) * 0.1
    ib = torch.rand(8, 13, 1)
    batch = lambda sl: (base[sl, :-1], base[sl, 1:], base[sl, 1:], ib[sl, :-1])   # noqa: E731  (host tensors: train() shards, then moves to the device)
    # global batches of 4, 2 and — ragged at world 2, like the last batch of a reference epoch (no drop_last) — 3 trajectories
    config["loaders"] = ([batch(slice(0, 4)), batch(slice(4, 6)), batch(slice(3, 6))], [batch(slice(6, 8))], [batch(slice(6, 8))])
    return config


def _run_train(save_dir, fused, world):
    from sea_amd.train.train_temporal import train
    from sea_amd.utils.train_utils import NoOpErrorTracker

    class Tracker(NoOpErrorTracker):
        def __init__(self):
            self.rows = []

        def record_error(self, phase, epoch, metrics):
            self.rows.append((phase, epoch, dict(metrics)))

    tr = Tracker()
    cfg = _train_config(save_dir, fused, world)
    # get_model draws the initial weights: rank 0 (and the single process) from seed 11, rank 1 from seed 12 — train() must broadcast rank 0's
    torch.manual_seed(11 + int(os.environ.get("RANK", "0")) if world > 1 else 11)
    model = train(cfg, tr)
    eng = model.engine()
    losses = [m["Loss"] for ph, _, m in tr.rows if ph == "train"]
    return eng.params.flat32[:eng.params.n_live].cpu().numpy(), np.asarray(losses)


def _train_worker(rank, world, port, save_dir, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))   # what a launcher sets: train() starts the group itself
    if os.environ.get("SEA_TEST_DP_BACKEND", "gloo") == "nccl":
        os.environ["LOCAL_RANK"] = str(rank)   # what torch.distributed.run sets; train() itself binds the rank to that GPU (no set_device here)
    try:
        # both step forms in ONE pair of processes (a fresh process costs the box a minute of imports): the first train() starts the process group from the
        # environment, the second finds it initialised
        ret[rank] = {fused: _run_train(save_dir, fused, world) for fused in (True, False)}
    except Exception as e:  # pragma: no cover
        ret[rank] = repr(e)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_train_entry_is_data_parallel(tmp_path):
    """train(config, tracker) (reference loop train/train_temporal.py:252-258) run by 2 ranks — each started like a launcher would (RANK / WORLD_SIZE /
    MASTER_* in the environment, config['world_size'] = 2) — equals the single-process run on the global batches: same epoch losses, parameters equal
    to the rounding of 9 AdamW steps (rel-L2 <= 1e-5, every element within 9e-5 = a few percent of one lr-sized update), bit-identical across the ranks.
    The third batch of every epoch has 3 trajectories: uneven shards (2 + 1) with the loss weights of parallel.shard_bounds.
    Covers the parameter broadcast (rank 1 draws different initial weights, see _run_train), the per-rank shard of every loader batch, and the gradient
    all-reduce of both step forms: the fused step (slices under the backward) and `loss.backward(); optimizer.step()` (one collective inside
    FlatAdamW.step)."""
    backend = os.environ.get("SEA_TEST_DP_BACKEND", "gloo")
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("needs one GPU per rank")
    ref = {fused: _run_train(str(tmp_path), fused, 1) for fused in (True, False)}
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_train_worker, args=(2, _free_port(), str(tmp_path), ret), nprocs=2, join=True)
    got = dict(ret)
    assert all(not isinstance(v, str) for v in got.values()), got
    for fused in (True, False):
        p_ref, l_ref = ref[fused]
        for r in range(2):
            p, losses = got[r][fused]
            if r == 0:
                assert np.allclose(losses, l_ref, rtol=1e-5, atol=0), (fused, r, losses, l_ref)
            else:
                assert losses.size == 0   # one set of tracker records per job: rank 0's
            assert np.linalg.norm(p - p_ref) <= 1e-5 * np.linalg.norm(p_ref) and np.abs(p - p_ref).max() <= 9e-5, (fused, r)
        assert np.array_equal(got[0][fused][0], got[1][fused][0])
    # the two step forms are the same arithmetic up to the order of the gradient reduction
    assert np.linalg.norm(ref[True][0] - ref[False][0]) <= 1e-5 * np.linalg.norm(ref[True][0])
    assert os.path.exists(os.path.join(str(tmp_path), "temporal_cylinder_flow_dp2.pt"))   # rank 0 wrote the best-validation checkpoint
