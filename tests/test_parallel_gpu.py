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

CFG = (1, 64, 4, 48, 8, 0, 3, 2, True, "adaln")
B_GLOBAL, T, STEPS, LR = 4, 40, 3, 1e-3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(dtype="fp32"):
    from oracle import sea_oracle as O
    from tests.test_model_gpu import build

    return build(O.OracleConfig(*CFG), dtype).train()


def _data():
    from oracle import sea_oracle as O
    from oracle.recipe import recipe_inputs

    return recipe_inputs(B_GLOBAL, T, O.OracleConfig(*CFG), seed=31)


def _run_steps(x, tgt, ib, world, rank):
    from sea_amd.parallel import parameters_in_sync, shard_batch
    from sea_amd.utils.train_utils import initialize_optimizer

    m = _model()
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
            assert eng.last_allreduce_calls == 3, eng.last_allreduce_calls
            bk = eng.train_plan(xs.shape[0], xs.shape[1]).grad_buckets()
            assert len(bk) == 2 and bk[0][1] == 0 and bk[0][2] == bk[1][1] and bk[0][0] < bk[1][0]
            assert bk[1][2] / eng.params.n_live > 0.7, bk
    return grads1, eng.params.flat32[:eng.params.n_live].cpu()


def _worker(rank, world, port, backend, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank if backend == "nccl" else 0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        x, tgt, ib = _data()
        g, p = _run_steps(x, tgt, ib, world, rank)
        ret[rank] = (g.numpy(), p.numpy())
    except Exception as e:  # pragma: no cover
        ret[rank] = repr(e)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2])
def test_train_step_world_n_equals_single_process_global_batch(world):
    backend = os.environ.get("SEA_TEST_DP_BACKEND", "gloo")
    if backend == "nccl" and torch.cuda.device_count() < world:
        pytest.skip("needs one GPU per rank")
    x, tgt, ib = _data()
    g_ref, p_ref = _run_steps(x, tgt, ib, 1, 0)       # single process, the global batch
    if world == 1:
        g2, p2 = _run_steps(x, tgt, ib, 1, 0)         # the body at N = 1: run-to-run agreement (fp32 atomics order)
        assert np.linalg.norm(g2.numpy() - g_ref.numpy()) <= 1e-6 * np.linalg.norm(g_ref.numpy())
        return
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), backend, ret), nprocs=world, join=True)
    got = dict(ret)
    assert all(not isinstance(v, str) for v in got.values()), got
    for r in range(world):
        g, p = got[r]
        assert np.linalg.norm(g - g_ref.numpy()) <= 1e-6 * np.linalg.norm(g_ref.numpy()), r
        # three AdamW steps at lr 1e-3: parameters within the update's own rounding of the single-process run
        assert np.abs(p - p_ref.numpy()).max() <= 2e-5, r
    assert np.array_equal(got[0][1], got[1][1])       # bit-identical across ranks
