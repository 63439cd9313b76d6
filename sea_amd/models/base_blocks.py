"""Host-side mirror of the reference's models/base_blocks.py for the temporal path.

Same class names, constructor signatures, parameter/buffer names and shapes (so reference checkpoints load unchanged);
the forward()s launch the HIP kernels of libsea_hip.so through sea_amd.ops instead of eager ATen ops.  These module-level
forwards are the un-fused building blocks (one kernel group per call); the fused whole-model path is
TemporalModel.forward -> sea_amd.engine.

Only what the temporal-rollout path uses is mirrored: MLP (:9-47), LayerNorm (:80-88), MaskedMultiHeadAttention
(:155-203), MaskedMultiHeadCrossAttention (:246-295), precompute_freqs_cis (:300-305), AdaLN (:330-350) and
PositionalEncoding (:355-372, a buffer that the path never reads).  The spatial-autoencoder classes are out of scope.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn

from .. import ops

_COMPUTE_DTYPE = {"dtype": torch.float32}


def set_module_compute_dtype(dtype: torch.dtype) -> None:
    """Activation/weight operand dtype of the stand-alone module forwards (float32 = exact-f32 MFMA, bfloat16)."""
    assert dtype in (torch.float32, torch.bfloat16)
    _COMPUTE_DTYPE["dtype"] = dtype


def _act_dtype() -> torch.dtype:
    return _COMPUTE_DTYPE["dtype"]


def _as_act(t: torch.Tensor) -> torch.Tensor:
    """fp32 [.., n] -> activation-dtype 2-D copy made by the conversion kernel."""
    t2 = t.reshape(-1, t.shape[-1])
    if _act_dtype() == torch.float32:
        return t2.contiguous()
    out = torch.empty(t2.shape, device=t.device, dtype=_act_dtype())
    ops.convert(t2.contiguous(), out)
    return out


_DROP_CALLS = [0]


def _dropout_key(module: nn.Module, p: float):
    """(seed, stream, thr) of this call's counter-based dropout masks, or None in eval mode / for p = 0: keep iff a hash byte of (seed, stream, row, column)
    >= thr = round(256 p), kept values scaled by 256 / (256 - thr) (include/sea_hip.h, SeaDropout) — where the reference applies nn.Dropout.  Its torch RNG
    stream cannot be reproduced; every call draws fresh masks from torch.initial_seed() and a call counter."""
    if not module.training or p <= 0.0:
        return None
    thr = int(round(256 * p))
    if thr <= 0:
        return None
    if thr > 255:
        raise ValueError("dropout probability too close to 1")
    _DROP_CALLS[0] += 1
    return (torch.initial_seed() * 0x9E3779B1 + _DROP_CALLS[0] * 0x85EBCA77) & 0xFFFFFFFF, (id(module) >> 4) & 0xFFFF, thr


def precompute_freqs_cis(dim: int, end: int, theta: float = 10000.0) -> torch.Tensor:
    """RoPE table e^{i t f_k}, f_k = theta^(-2k/dim), as complex64 [end, dim/2] (reference :300-305).  Frequencies and
    angles are formed in fp32 exactly as the reference does so that the table (a state_dict buffer) is identical."""
    k = torch.arange(0, dim, 2)[: dim // 2].float()
    freqs = 1.0 / (theta ** (k / dim))
    angles = torch.outer(torch.arange(end, dtype=torch.float32), freqs)
    return torch.polar(torch.ones_like(angles), angles)


class LayerNorm(nn.Module):
    """Bias-optional LayerNorm that ignores its condition argument (reference :80-88)."""

    def __init__(self, ndim, bias=False):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(ndim))
        self.bias = nn.Parameter(torch.zeros(ndim)) if bias else None

    def forward(self, input, cond=None):
        x = input.reshape(-1, input.shape[-1]).contiguous()
        y = torch.empty_like(x)
        ops.rownorm([dict(X=x, gamma=self.weight, beta=self.bias, Y32=y)], x.shape[0], x.shape[1], False, False, 1e-5, _act_dtype())
        return y.view(input.shape)


class AdaLN(nn.Module):
    """Adaptive LayerNorm: (x - mean)/sqrt(var + 1e-5) * (weight + 1 + w(cond)) + (bias + b(cond)) with
    [w, b] = Linear(SiLU(Linear(cond))) (reference :330-350).  cond_dim must be 1 on the native path."""

    def __init__(self, embed_dim, cond_dim):
        super().__init__()
        self.embed_dim = embed_dim
        self.cond_dim = cond_dim
        self.weight = nn.Parameter(torch.ones(embed_dim))
        self.bias = nn.Parameter(torch.zeros(embed_dim))
        self.cond_mlp = nn.Sequential(nn.Linear(cond_dim, 2 * embed_dim), nn.SiLU(), nn.Linear(2 * embed_dim, 2 * embed_dim))

    def forward(self, x, cond):
        if self.cond_dim != 1:
            raise NotImplementedError("sea_amd: AdaLN native path supports a scalar condition (cond_dim=1)")
        dt = _act_dtype()
        d = self.embed_dim
        x2 = x.reshape(-1, d).contiguous()
        M = x2.shape[0]
        c = cond.reshape(-1).contiguous()
        hid = torch.empty(M, 2 * d, device=x.device, dtype=dt)
        ops.silu_outer([dict(w1=self.cond_mlp[0].weight.reshape(-1), b1=self.cond_mlp[0].bias, Hid=hid)], c, M, dt)
        mod = torch.empty(M, 2 * d, device=x.device, dtype=dt)
        ops.gemm_grouped([dict(A=hid, W=_as_act(self.cond_mlp[2].weight), bias=self.cond_mlp[2].bias, Cact=mod)], dt)
        y = torch.empty_like(x2)
        ops.rownorm([dict(X=x2, mod=mod, gamma=self.weight, beta=self.bias, Y32=y)], M, d, False, False, 1e-5, dt)
        return y.view(x.shape)


class MLP(nn.Module):
    """Linear -> nn.LayerNorm(hidden) -> GELU(erf) -> Linear (reference :9-47, num_layers None/1 only)."""

    def __init__(self, dim_in, dropout, scale_ratio=4, dim_out=None, num_layers=None):
        super().__init__()
        if dim_out is None:
            dim_out = dim_in
        self.residual_projection = None
        if dim_in != dim_out:
            self.residual_projection = nn.Linear(dim_in, dim_out)  # created, never used (as in the reference)
        if not (num_layers is None or num_layers == 1):
            raise NotImplementedError("sea_amd: MLP with num_layers > 1 is outside the temporal hot path")
        hidden = max(1, int(dim_in * scale_ratio))
        self.layers = nn.ModuleList([nn.Linear(dim_in, hidden), nn.LayerNorm(hidden), nn.GELU(), nn.Linear(hidden, dim_out)])
        self.dropout = nn.Dropout(dropout)
        self._p = dropout

    def forward(self, x, residual=None):
        """`residual` (fp32, same shape as the output) is an extension over the reference signature: when given, the second
        GEMM's epilogue returns residual + MLP(x)."""
        dk = _dropout_key(self, self._p)
        dt = _act_dtype()
        fc1, ln, _, fc2 = self.layers
        lead = x.shape[:-1]
        if fc1.in_features == 1:
            # the information-bottleneck MLP on a scalar condition: one fused kernel (sea_ib_add) into a zero buffer
            c = x.reshape(-1).contiguous()
            if residual is None:
                y = torch.zeros(c.numel(), fc2.out_features, device=x.device, dtype=torch.float32)
            else:
                y = residual.reshape(c.numel(), fc2.out_features).clone()  # the kernel accumulates in place
            ops.ib_add([y], c, fc1.weight.reshape(-1), fc1.bias, ln.weight, ln.bias, fc2.weight, fc2.bias, drop=dk)
            return y.view(*lead, fc2.out_features)
        a = _as_act(x)
        M = a.shape[0]
        h = torch.empty(M, fc1.out_features, device=x.device, dtype=dt)
        ops.gemm_grouped([dict(A=a, W=_as_act(fc1.weight), bias=fc1.bias, Cact=h)], dt)
        hg = torch.empty_like(h)
        ops.rownorm([dict(X=h, gamma=ln.weight, beta=ln.bias, Yact=hg)], M, h.shape[1], dt != torch.float32, True, ln.eps, dt)
        y = torch.empty(M, fc2.out_features, device=x.device, dtype=torch.float32)
        R = None if residual is None else residual.reshape(M, fc2.out_features).contiguous()
        # nn.Dropout on the MLP output (reference :47), before the residual: mode 1 of the GEMM epilogue
        ops.gemm_grouped([dict(A=hg, W=_as_act(fc2.weight), bias=fc2.bias, R=R, C32=y, drop=None if dk is None else (dk[0], dk[1], dk[2], 1))], dt)
        return y.view(*lead, fc2.out_features)


class _MaskedAttentionBase(nn.Module):
    """Parameters and buffers shared by the two masked attention modules: k, q, v Linear(+bias), projection without bias,
    `freqs_cis` (complex64 RoPE table) and `tril` (the dense causal mask).

    The dense [1,1,max_len,max_len] mask (16 MB at max_len=2024, x12 modules) is part of the reference's checkpoint schema but
    is never needed by the flash kernel, so it is not kept resident: it is synthesised when a state_dict is written and
    ignored when one is loaded."""

    def __init__(self, n_heads, embed_dim, max_len, src_len, dropout):
        super().__init__()
        self.n_heads = n_heads
        self.max_len = max_len
        self.src_len = src_len
        self.head_dim = embed_dim // n_heads
        self.dropout = nn.Dropout(dropout)
        self._p = dropout
        self.embed_dim = embed_dim
        self.k = nn.Linear(embed_dim, self.head_dim * n_heads)
        self.q = nn.Linear(embed_dim, self.head_dim * n_heads)
        self.v = nn.Linear(embed_dim, self.head_dim * n_heads)
        self.projection = nn.Linear(embed_dim, embed_dim, bias=False)
        self.register_buffer("freqs_cis", precompute_freqs_cis(self.head_dim, max_len))
        self._register_state_dict_hook(_MaskedAttentionBase._emit_tril)
        self._register_load_state_dict_pre_hook(_MaskedAttentionBase._drop_tril)

    @staticmethod
    def _emit_tril(module, state_dict, prefix, local_metadata):
        n = module.max_len
        state_dict[prefix + "tril"] = torch.tril(torch.ones(n, n), diagonal=module.src_len).view(1, 1, n, n)
        return state_dict

    @staticmethod
    def _drop_tril(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        state_dict.pop(prefix + "tril", None)

    def _attend(self, x_q: torch.Tensor, x_kv: torch.Tensor, residual: Optional[torch.Tensor] = None, gelu_out: bool = False) -> torch.Tensor:
        dk = _dropout_key(self, self._p)   # nn.Dropout on the attention probabilities (reference :194, :286)
        dt = _act_dtype()
        B, T, Cdim = x_q.shape
        H, hd = self.n_heads, self.head_dim
        if T > self.max_len:
            raise ValueError(f"sequence length {T} exceeds max_len {self.max_len}")
        cap = (T + 7) // 8 * 8
        dev = x_q.device
        aq = _as_act(x_q)
        akv = aq if x_kv is x_q else _as_act(x_kv)
        Wq, Wkv = _as_act(self.q.weight), _as_act(torch.cat((self.k.weight, self.v.weight), dim=0))
        bkv = torch.cat((self.k.bias, self.v.bias))
        Q = torch.empty(B, H, T, hd, device=dev, dtype=dt)
        K = torch.zeros(B, H, cap, hd, device=dev, dtype=dt)
        Vt = torch.zeros(B, H, hd, cap, device=dev, dtype=dt)
        rope = torch.view_as_real(self.freqs_cis).contiguous()
        ops.qkv_rope_grouped([dict(A=aq, W=Wq, bias=self.q.bias, col0=0, Q=Q), dict(A=akv, W=Wkv, bias=bkv, col0=Cdim, K=K, Vt=Vt)],
                             rope, H, hd, T, 0, cap, ops.q_scale(hd), dt)
        O = torch.empty(B, T, Cdim, device=dev, dtype=dt)
        ops.attention_fwd([dict(Q=Q, K=K, Vt=Vt, O=O)], B, H, hd, T, T, cap, 0, self.src_len, dt, drop=dk)
        y = torch.empty(B * T, Cdim, device=dev, dtype=torch.float32)
        R = None if residual is None else residual.reshape(B * T, Cdim).contiguous()
        ops.gemm_grouped([dict(A=O.view(B * T, Cdim), W=_as_act(self.projection.weight), R=R, C32=y, act=int(gelu_out))], dt)
        return y.view(B, T, Cdim)


class MaskedMultiHeadAttention(_MaskedAttentionBase):
    """Causal self-attention with RoPE (reference :155-203)."""

    def forward(self, x):
        return self._attend(x, x)


class MaskedMultiHeadCrossAttention(_MaskedAttentionBase):
    """Causal cross-attention with RoPE: q from x_1, k and v from x_2, same length and positions (reference :246-295)."""

    def forward(self, x_1, x_2):
        return self._attend(x_1, x_2)


class MultiHeadCrossAttention(nn.Module):
    """Un-masked cross-attention without rotary embedding: q from x_1, k and v from x_2 (reference models/base_blocks.py:205-243; the block's
    ib_addition_mode 'attention' attends from the field rows to the info-bottleneck rows of every position with it, models/temporal.py:49-53,117-118).
    Same parameters as the masked modules (k, q, v with bias, bias-free projection) and the same dense `tril` buffer in the checkpoint schema
    (never used by the forward, synthesised on save / dropped on load like theirs)."""

    def __init__(self, n_heads, embed_dim, max_len, src_len, dropout):
        super().__init__()
        self.n_heads, self.max_len, self.src_len = n_heads, max_len, src_len
        self.head_dim = embed_dim // n_heads
        self.dropout = nn.Dropout(dropout)
        self._p = dropout
        self.embed_dim = embed_dim
        self.k = nn.Linear(embed_dim, self.head_dim * n_heads)
        self.q = nn.Linear(embed_dim, self.head_dim * n_heads)
        self.v = nn.Linear(embed_dim, self.head_dim * n_heads)
        self.projection = nn.Linear(embed_dim, embed_dim, bias=False)
        self._register_state_dict_hook(_MaskedAttentionBase._emit_tril)
        self._register_load_state_dict_pre_hook(_MaskedAttentionBase._drop_tril)

    def forward(self, x_1, x_2):
        if self.training and self._p > 0:
            raise NotImplementedError("sea_amd.MultiHeadCrossAttention: dropout > 0 in train() is not covered")
        dt = _act_dtype()
        B, T, Cdim = x_1.shape
        Ts = x_2.shape[1]
        H, hd = self.n_heads, self.head_dim
        cap = (Ts + 7) // 8 * 8
        dev = x_1.device
        Wkv = _as_act(torch.cat((self.k.weight, self.v.weight), dim=0))
        bkv = torch.cat((self.k.bias, self.v.bias))
        Q = torch.empty(B, H, T, hd, device=dev, dtype=dt)
        K = torch.zeros(B, H, cap, hd, device=dev, dtype=dt)
        Vt = torch.zeros(B, H, hd, cap, device=dev, dtype=dt)
        ident = identity_rope(hd, max(T, Ts), dev)
        ops.qkv_rope_grouped([dict(A=_as_act(x_1), W=_as_act(self.q.weight), bias=self.q.bias, col0=0, Q=Q)], ident, H, hd, T, 0, cap, ops.q_scale(hd), dt)
        ops.qkv_rope_grouped([dict(A=_as_act(x_2), W=Wkv, bias=bkv, col0=Cdim, K=K, Vt=Vt)], ident, H, hd, Ts, 0, cap, ops.q_scale(hd), dt)
        O = torch.empty(B, T, Cdim, device=dev, dtype=dt)
        ops.attention_fwd([dict(Q=Q, K=K, Vt=Vt, O=O)], B, H, hd, T, Ts, cap, 0, Ts, dt)   # src_len >= Ts: every key is visible to every query
        y = torch.empty(B * T, Cdim, device=dev, dtype=torch.float32)
        ops.gemm_grouped([dict(A=O.view(B * T, Cdim), W=_as_act(self.projection.weight), C32=y)], dt)
        return y.view(B, T, Cdim)


def identity_rope(hd: int, length: int, device) -> torch.Tensor:
    """A rotation table of zero angles ([length, hd/2, (cos, sin)] = (1, 0)): the QKV kernel's RoPE epilogue as a no-op for the un-rotated attention modules."""
    t = torch.zeros(length, hd // 2, 2, device=device, dtype=torch.float32)
    t[..., 0] = 1.0
    return t


class PositionalEncoding(nn.Module):
    """Sinusoidal table kept only because its buffer `pe` [1, 5000, d] is in the checkpoint schema (reference :355-372);
    with RoPE in the attention modules the temporal path never adds it."""

    def __init__(self, d_model, dropout=0.1, max_len=5000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        pe = torch.zeros(max_len, d_model)
        pos = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(pos * div)
        pe[:, 1::2] = torch.cos(pos * div[: d_model // 2])
        self.register_buffer("pe", pe.unsqueeze(0))


class GaussianFourierProjection(nn.Module):
    """Random Fourier features of the condition (reference models/base_blocks.py:143-151): a fixed (requires_grad=False) W [input_dim, half_dim];
    forward(x) = [sin(2 pi x W), cos(2 pi x W)].  Parameter container for the whole-model path (sea_ib_add mode 2)."""

    def __init__(self, input_dim, half_dim=256, scale=1.0):
        super().__init__()
        self.input_dim, self.half_dim = input_dim, half_dim
        self.W = nn.Parameter(torch.randn(input_dim, half_dim) * scale, requires_grad=False)

    def forward(self, x):
        raise RuntimeError("sea_amd.GaussianFourierProjection is a parameter container; call TemporalModel.forward")


# ---------------------------------------------------------------------------------------------- spatial-encoder layers
class downScaleMLP(nn.Module):  # noqa: N801  (reference name, models/base_blocks.py:65-78)
    """Linear(d_input, hidden, bias=False) -> GELU -> Linear(hidden, d_model).  Parameter container: PointwiseEncode.forward runs all
    groups as grouped native launches."""

    def __init__(self, d_input, d_model, hidden_dim):
        super().__init__()
        self.d_model, self.d_input = d_model, d_input
        self.layer1 = nn.Linear(d_input, hidden_dim, bias=False)
        self.activation = nn.GELU()
        self.layer2 = nn.Linear(hidden_dim, d_model)

    def forward(self, x):
        raise RuntimeError("sea_amd.downScaleMLP is a parameter container; call PointwiseEncode.forward (grouped native launch)")


class MultiHeadAttention(nn.Module):
    """Un-masked multi-head self-attention of the spatial encoder (reference models/base_blocks.py:91-121): k, q, v Linear with bias (in
    that registration order), bias-free projection, no rotary embedding.  Parameter container for PointwiseEncode.forward."""

    def __init__(self, n_heads, embed_dim, dropout):
        super().__init__()
        self.n_heads, self.head_dim, self.embed_dim = n_heads, embed_dim // n_heads, embed_dim
        self.dropout = nn.Dropout(dropout)
        self.k = nn.Linear(embed_dim, self.head_dim * n_heads)
        self.q = nn.Linear(embed_dim, self.head_dim * n_heads)
        self.v = nn.Linear(embed_dim, self.head_dim * n_heads)
        self.projection = nn.Linear(embed_dim, embed_dim, bias=False)

    def forward(self, x):
        raise RuntimeError("sea_amd.MultiHeadAttention is a parameter container; call PointwiseEncode.forward")


class EncoderBlock(nn.Module):
    """x += MHA(LN(x)); x += MLP_x4(LN(x)) with weight-only LayerNorms (reference models/base_blocks.py:123-139).  Parameter container."""

    def __init__(self, n_heads, max_len, embed_dim, src_len, dropout):
        super().__init__()
        self.ln_exp1_1 = LayerNorm(embed_dim, bias=False)
        self.ln_exp1_2 = LayerNorm(embed_dim, bias=False)
        self.attn_1 = MultiHeadAttention(n_heads, embed_dim, dropout)
        self.mlp_1 = MLP(embed_dim, dropout)

    def forward(self, x):
        raise RuntimeError("sea_amd.EncoderBlock is a parameter container; call PointwiseEncode.forward")

