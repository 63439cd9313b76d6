"""Spatial decoder of the rollout's consumer side: the reference's `Decode` module
(models/encoder_decoder.py:126-146) and its `upScaleMLP` (models/base_blocks.py:49-63) with the same constructor arguments,
parameter names and shapes, computed by two grouped-GEMM launches of libsea_hip.so (sea_gemm_grouped: Linear + GELU epilogue, then
Linear + bias written straight into the concatenated [B, P, n_fields, n_inp] output) — no per-group Python loop of ATen ops, no cat.

Only the decoder is here: the 12-layer spatial encoder, the mesh partitioner and the MinMax scalers are out of scope for this round.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.nn as nn

from .. import _native as N
from .. import ops


class upScaleMLP(nn.Module):  # noqa: N801  (reference name, models/base_blocks.py:49)
    def __init__(self, d_model: int, d_output: int, hidden_dim: int):
        super().__init__()
        self.d_model, self.d_output, self.hidden_dim = d_model, d_output, hidden_dim
        self.layer1 = nn.Linear(d_model, hidden_dim, bias=False)
        self.activation = nn.GELU()
        self.layer2 = nn.Linear(hidden_dim, d_output)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        raise RuntimeError("sea_amd.upScaleMLP is a parameter container; call Decode.forward (grouped native launch)")


class Decode(nn.Module):
    """Decode(field_groups, n_inp, MLP_hidden, embed_dim, dropout=0.1).forward(z [B, P, n_groups, embed_dim]) -> [B, P, n_fields, n_inp].
    `dropout` is accepted and unused, as in the reference (models/encoder_decoder.py:127-136 never applies it)."""

    def __init__(self, field_groups: Sequence[Sequence[int]], n_inp: int, MLP_hidden: int, embed_dim: int, dropout: float = 0.1):
        super().__init__()
        self.field_groups = [list(g) for g in field_groups]
        self.num_groups = len(self.field_groups)
        self.n_inp, self.MLP_hidden, self.embed_dim = n_inp, MLP_hidden, embed_dim
        self.decoders = nn.ModuleList([upScaleMLP(d_model=embed_dim, d_output=n_inp * len(g), hidden_dim=MLP_hidden) for g in self.field_groups])
        self.compute_dtype = "fp32"
        self._shadow = None  # (key, [W1 act], [W2 act])
        if embed_dim % 8 or MLP_hidden % 8 or n_inp % 4:
            raise NotImplementedError("sea_amd.Decode: embed_dim and MLP_hidden must be multiples of 8, n_inp a multiple of 4 (16-byte operand rows)")

    def set_compute_dtype(self, dtype) -> "Decode":
        name = {torch.float32: "fp32", torch.bfloat16: "bf16"}.get(dtype, dtype)
        if name not in ("fp32", "bf16"):
            raise ValueError("compute dtype must be 'fp32' or 'bf16'")
        self.compute_dtype, self._shadow = name, None
        return self

    def _weights(self, dt: torch.dtype):
        """Activation-dtype copies of the Linear weights, refreshed when a parameter was written (version counter) or moved."""
        ps = [d.layer1.weight for d in self.decoders] + [d.layer2.weight for d in self.decoders]
        key = (dt, tuple((p.data_ptr(), p._version) for p in ps))
        if self._shadow is None or self._shadow[0] != key:
            with torch.no_grad():
                conv = lambda p: p.detach().contiguous() if dt == torch.float32 else p.detach().to(dt).contiguous()
                self._shadow = (key, [conv(d.layer1.weight) for d in self.decoders], [conv(d.layer2.weight) for d in self.decoders])
        return self._shadow[1], self._shadow[2]

    def forward(self, z: torch.Tensor) -> torch.Tensor:
        N.require_gpu(z, "Decode input")
        if z.requires_grad or (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()) and False):
            raise NotImplementedError("sea_amd.Decode: inference only")
        B, P, G, D = z.shape
        assert G == self.num_groups and D == self.embed_dim, (z.shape, self.num_groups, self.embed_dim)
        dt = torch.float32 if self.compute_dtype == "fp32" else torch.bfloat16
        M = B * P
        zf = z.detach().to(torch.float32).contiguous().view(M, G * D)
        za = zf if dt == torch.float32 else torch.empty(M, G * D, device=z.device, dtype=dt)
        if dt != torch.float32:
            ops.convert(zf, za)
        W1, W2 = self._weights(dt)
        n_fields = sum(len(g) for g in self.field_groups)
        out = torch.empty(M, n_fields * self.n_inp, device=z.device, dtype=torch.float32)
        hid: List[torch.Tensor] = [torch.empty(M, self.MLP_hidden, device=z.device, dtype=dt) for _ in range(G)]
        ops.gemm_grouped([dict(A=za[:, g * D:(g + 1) * D], W=W1[g], Cact=hid[g], act=1) for g in range(G)], dt)
        groups, off = [], 0
        for g, grp in enumerate(self.field_groups):
            w = len(grp) * self.n_inp
            groups.append(dict(A=hid[g], W=W2[g], bias=self.decoders[g].layer2.bias.detach(), C32=out[:, off:off + w]))
            off += w
        ops.gemm_grouped(groups, dt)
        return out.view(B, P, n_fields, self.n_inp)
