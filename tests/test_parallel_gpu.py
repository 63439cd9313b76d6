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
