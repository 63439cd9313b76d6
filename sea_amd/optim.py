"""AdamW over the model's flat parameter buffer: one kernel launch per step (sea_adamw_flat), which also refreshes the
activation-dtype weight shadow.  Semantics follow torch.optim.AdamW as the reference configures it
(utils/train_utils.py:33-34): decoupled weight decay, bias-corrected moments; parameters that never receive a gradient are not
touched and get no state (they sit beyond the live prefix of the flat buffer).

Data parallel (one process per GPU, torch.distributed initialised): `step()` SUM-all-reduces the live prefix of the flat gradient buffer
itself when nothing has reduced it since the last `zero_grad()` — so the reference's loop `loss.backward(); optimizer.step()`
(train/train_temporal.py:256-257) is correct on N ranks as written — and folds the 1/world mean into the kernel's grad_scale.  The fused step
(engine.train_step) reduces in slices under the backward and tells the optimizer so (`mark_reduced`)."""
from __future__ import annotations

from typing import Optional

import torch

from . import _native as N


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        params = list(model.parameters())
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.model = model
        self._m: Optional[torch.Tensor] = None
        self._v: Optional[torch.Tensor] = None
        self._step = 0
        self._eng = None
        self.grad_scale = 1.0   # data parallel: 1 / world_size after the SUM all-reduce
        self._reduced = False   # True once this step's gradients have been all-reduced (by engine.train_step or by step() itself)
        self.allreduce_calls = 0   # collectives issued by step() over the optimizer's lifetime (the bench line reports it)
        self.process_group = None

    def _buffers(self):
        eng = self.model.engine()
        if self._eng is not eng:  # first use, or the model moved / changed dtype: (re)allocate the moments
            n = eng.params.n_live
            self._m = torch.zeros(n, device=eng.device, dtype=torch.float32)
            self._v = torch.zeros(n, device=eng.device, dtype=torch.float32)
            self._eng = eng
            eng.ensure_grads()
        return eng

    def zero_grad(self, set_to_none: bool = True):
        eng = self._buffers()
        eng.zero_grads()
        self._reduced = False
        if set_to_none:
            for p in self.model._live_params():
                p.grad = None

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        eng = self._buffers()
        if not eng.grads_dirty:
            return loss  # no backward since the last zero_grad: like torch, parameters without gradients are skipped
        if not self._reduced:
            # data parallel: ONE all-reduce of the live prefix (a no-op returning 1.0 without a process group or at world size 1)
            from .parallel import allreduce_flat_gradients, world_size

            if world_size(self.process_group) > 1:
                self.grad_scale = allreduce_flat_gradients(eng.grads, eng.params.n_live, self.process_group)
                self.allreduce_calls += 1
            else:
                self.grad_scale = 1.0
            self._reduced = True
        g = self.param_groups[0]
        self._step += 1
        P = eng.params
        n = P.n_live
        shadow = P.flat_act if P.act_dtype != torch.float32 else None
        N.check(N.lib().sea_adamw_flat(P.flat32.data_ptr(), eng.grads.data_ptr(), self._m.data_ptr(), self._v.data_ptr(),
                                       None if shadow is None else shadow.data_ptr(), N.dtype_code(P.act_dtype), n, float(g["lr"]),
                                       float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), self._step,
                                       float(self.grad_scale), N.stream_ptr()), "sea_adamw_flat")
        # the kernel wrote through raw pointers: mark both shadows as current for the row-major one, stale for the transposed one
        P._synced_version = P.flat32._version
        P.sync_transposed(force=True)
        return loss

    def mark_reduced(self, grad_scale: float) -> None:
        """engine.train_step: the gradients of this step are already summed over the ranks; `grad_scale` turns the sum into the mean."""
        self.grad_scale, self._reduced = grad_scale, True

    def state_dict(self):
        sd = super().state_dict()
        sd["sea_flat"] = dict(step=self._step, m=None if self._m is None else self._m.cpu(), v=None if self._v is None else self._v.cpu())
        return sd

    def load_state_dict(self, state_dict):
        flat = state_dict.get("sea_flat")
        super().load_state_dict({k: v for k, v in state_dict.items() if k != "sea_flat"})
        if flat is not None and flat["m"] is not None:
            self._buffers()
            self._step = flat["step"]
            self._m.copy_(flat["m"])
            self._v.copy_(flat["v"])
