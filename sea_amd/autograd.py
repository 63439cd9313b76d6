"""torch.autograd bridge: keeps `loss.backward()` of the reference train loop (train/train_temporal.py:255-257) working while the
forward and the backward both run as pre-built HIP launch lists (sea_amd/train_engine.py)."""
from __future__ import annotations

import torch

from . import _native as N


class _TemporalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, x, ib, model, eng):
        out, plan = eng.forward_train(x, ib)
        ctx.plan, ctx.eng, ctx.model = plan, eng, model
        ctx.generation = plan.generation   # the saved activations live in the plan's workspace: a later forward of the same shape replaces them
        return out

    @staticmethod
    def backward(ctx, dout):
        eng, model = ctx.eng, ctx.model
        if ctx.plan.generation != ctx.generation:
            raise RuntimeError("sea_amd: backward() of a TemporalModel forward whose saved activations were overwritten by a later grad-enabled forward of the "
                               "same shape (one activation set per (batch, length) is kept): run backward() before the next forward, or run the other "
                               "forward under torch.no_grad()")
        live = model._live_params()
        # torch semantics: gradients accumulate until zero_grad().  A step that starts from p.grad is None starts from zero.
        if eng.grads_dirty and live and live[0].grad is None:
            eng.zero_grads()
        eng.backward(ctx.plan, dout.contiguous().float())
        for name, p in zip(eng.params.live_names, live):
            if p.grad is None:
                p.grad = eng.grad_view(name)
        return None, None, None, None, None


def temporal_forward_with_grad(model, eng, x, ib):
    return _TemporalFn.apply(model._grad_anchor(), x.float(), ib.float(), model, eng)


class MSELossFn(torch.autograd.Function):
    """mean((output - target)^2) with forward and backward in ONE kernel pass (sea_mse_fwd_bwd)."""

    @staticmethod
    def forward(ctx, output, target):
        N.require_gpu(output, "output")
        out = output.contiguous().float()
        tgt = target.contiguous().float()
        dout = torch.empty_like(out)
        loss = torch.empty(1, device=out.device)
        partial = torch.empty(1024, device=out.device)
        N.check(N.lib().sea_mse_fwd_bwd(out.data_ptr(), tgt.data_ptr(), dout.data_ptr(), loss.data_ptr(), partial.data_ptr(), 1024,
                                        out.numel(), 1.0, N.stream_ptr()), "sea_mse_fwd_bwd")
        ctx.save_for_backward(dout)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dout,) = ctx.saved_tensors
        return dout * g, None
