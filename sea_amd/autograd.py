"""torch.autograd bridge for TemporalModel (backward kernels land here)."""


def temporal_forward_with_grad(model, engine, x, ib):
    raise NotImplementedError("sea_amd: the backward pass is not built yet; run the forward under torch.no_grad()")
