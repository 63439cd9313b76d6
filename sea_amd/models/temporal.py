"""Host-side mirror of the reference's models/temporal.py: the State-Exchange Attention temporal transformer.

Class names, constructor signatures, attribute names, parameter/buffer schema and error behaviour follow the reference
(BaseBlockTemporal :21-148, SEABlockTemporal :152-192, create_block_temporal :314-324, TemporalModel :326-416) so that this
module is a drop-in for the temporal-rollout path.  Compute does not go through module-by-module eager ops:
TemporalModel.forward hands the whole forward to sea_amd.engine (≈30 fused HIP launches per layer over flat HBM buffers).

Scope: exchange_mode='sea' with ib_scale_mode='mlp', ib_addition_mode='add' — what both shipped configs select
(configs/cylinder_flow.py:112-128, configs/multiphase_flow.py:112-128) — forward, rollout and training; the ablation variants
exchange_mode 'addition' / 'simple' / 'pool', ib_scale_mode 'fourier' (the constructor's default) / 'linear' and ib_addition_mode 'none' / 'attention' through the
same plans, forward, rollout and training, and ib_addition_mode 'concat' (blocks widened by 64 info-bottleneck columns; like the reference
it only runs with add_info_after_cross=False).  Invalid names raise ValueError as in the reference.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.nn as nn

from .base_blocks import (AdaLN, GaussianFourierProjection, LayerNorm, MLP, MaskedMultiHeadAttention, MaskedMultiHeadCrossAttention, MultiHeadCrossAttention,
                          PositionalEncoding)

_EXCHANGE_MODES = {"sea", "simple", "addition", "pool"}
_IB_SCALE_MODES = {"fourier", "linear", "mlp"}
_IB_ADD_MODES = {"add", "concat", "attention", "none"}


def _make_norm(ln_type: str, dim: int, ib_num: int) -> nn.Module:
    t = ln_type.lower()
    if t == "adaln":
        return AdaLN(dim, ib_num)
    if t == "ln":
        return LayerNorm(dim, bias=False)
    raise ValueError(f"Invalid LN_type: {ln_type}. Must be one of {'adaln', 'ln'}.")


class BaseBlockTemporal(nn.Module):
    """One temporal block: per-field causal self-attention, state exchange (subclass), information-bottleneck add, per-field
    MLP and output projection (reference :21-148)."""

    def __init__(self, n_heads, max_len, embed_dim, src_len, scale_ratio, num_variables, down_proj=2, ib_scale_mode='fourier',
                 ib_addition_mode='add', ib_mlp_layers=1, ib_num=1, dropout=0, add_info_after_cross=False, LN_type='adaln'):
        super().__init__()
        self.num_variables = num_variables
        self.ib_dim_concat = 64
        self.original_embed_dim = embed_dim
        self.ib_mlp_layers = ib_mlp_layers
        self.ib_addition_mode = self._validate_ib_addition_mode(ib_addition_mode)
        self.ib_num = ib_num
        self.add_info_after_cross = add_info_after_cross
        # 'concat' (reference :48): the block works on rows widened by ib_dim_concat info-bottleneck columns; proj maps them back to embed_dim
        self.internal_embed_dim = embed_dim + self.ib_dim_concat if self.ib_addition_mode == "concat" else embed_dim
        if self.ib_addition_mode == "attention":   # reference :49-53 (registered before the info-bottleneck layer: the checkpoint's key order)
            self.cross_attn_ib = nn.ModuleList([MultiHeadCrossAttention(n_heads, self.internal_embed_dim, max_len, src_len, dropout) for _ in range(num_variables)])
        self.ib_scale_mode = self._validate_ib_mode(ib_scale_mode)
        self.ib_dim = self.ib_dim_concat if self.ib_addition_mode == "concat" else embed_dim   # reference :100-101
        if self.ib_scale_mode == "fourier":      # reference :103-109
            self.ib = GaussianFourierProjection(self.ib_num, int(self.ib_dim // 2))
        elif self.ib_scale_mode == "linear":
            self.ib = nn.Linear(self.ib_num, self.ib_dim)
        else:
            self.ib = MLP(self.ib_num, dropout, scale_ratio, self.ib_dim, self.ib_mlp_layers)
        self.down_ratio = down_proj
        self.down_dim = self.internal_embed_dim // self.down_ratio
        F, E = num_variables, self.internal_embed_dim
        _make_norm(LN_type, 1, 1)  # validates LN_type before any allocation
        self.ln = nn.ModuleDict({
            'exp': nn.ModuleList([nn.ModuleList([_make_norm(LN_type, E, ib_num) for _ in range(3)]) for _ in range(F)]),
            'cross': _make_norm(LN_type, self.down_dim, ib_num),
        })
        self.attn = nn.ModuleDict({
            'self': nn.ModuleList([MaskedMultiHeadAttention(n_heads, E, max_len, src_len, dropout) for _ in range(F)])
        })
        self.mlp = nn.ModuleList([MLP(E, dropout, scale_ratio) for _ in range(F)])
        self.act = nn.GELU()
        self.pos_encoder = PositionalEncoding(self.down_dim, dropout)
        self.proj = nn.ModuleList([nn.Linear(E, self.original_embed_dim) for _ in range(F)])

    @staticmethod
    def _validate_ib_mode(mode):
        mode = mode.lower()
        if mode not in _IB_SCALE_MODES:
            raise ValueError(f"Invalid ib_scale_mode '{mode}'. Must be one of {_IB_SCALE_MODES}.")
        return mode

    @staticmethod
    def _validate_ib_addition_mode(mode):
        mode = mode.lower()
        if mode not in _IB_ADD_MODES:
            raise ValueError(f"Invalid ib_addition_mode '{mode}'. Must be one of {_IB_ADD_MODES}.")
        return mode

    def _add_info(self, x, add_info, var_idx):
        if self.ib_addition_mode == "none":   # reference :113-114
            return x
        if self.ib_scale_mode != "mlp":
            raise NotImplementedError("sea_amd: the stand-alone block forward covers ib_scale_mode='mlp'; call TemporalModel.forward")
        if self.ib_addition_mode == "attention":   # reference :117-118
            return x + self.cross_attn_ib[var_idx](x, self.ib(add_info))
        if self.ib_addition_mode == "concat":      # reference :115-116
            return torch.cat([x, self.ib(add_info)], dim=-1)
        return self.ib(add_info, residual=x)

    def _apply_exchange(self, x_vars, x_add):
        raise NotImplementedError

    def forward(self, *x_vars, x_add):
        """Un-fused block forward composed from the module kernels (the fused path is TemporalModel.forward)."""
        assert len(x_vars) == self.num_variables, f"Expected {self.num_variables} input variables, but got {len(x_vars)}"
        xs = [x.contiguous() for x in x_vars]
        if not self.add_info_after_cross:
            xs = [self._add_info(x, x_add, i) for i, x in enumerate(xs)]
        for i in range(self.num_variables):
            n = self.ln['exp'][i][0](xs[i], x_add)
            xs[i] = self.attn['self'][i]._attend(n, n, residual=xs[i])
        xs = self._apply_exchange(xs, x_add)
        if self.add_info_after_cross:
            xs = [self._add_info(x, x_add, i) for i, x in enumerate(xs)]
        for i in range(self.num_variables):
            xs[i] = self.mlp[i](self.ln['exp'][i][2](xs[i], x_add), residual=xs[i])
            xs[i] = _linear(self.proj[i], xs[i])
        return tuple(xs)


def _linear(lin: nn.Linear, x: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.Linear (+ optional fp32 residual) through the grouped-GEMM kernel (stand-alone module path)."""
    from .. import ops
    from .base_blocks import _act_dtype, _as_act

    dt = _act_dtype()
    a = _as_act(x)
    y = torch.empty(a.shape[0], lin.out_features, device=x.device, dtype=torch.float32)
    R = None if residual is None else residual.reshape(a.shape[0], lin.out_features).contiguous()
    ops.gemm_grouped([dict(A=a, W=_as_act(lin.weight), bias=lin.bias, R=R, C32=y)], dt)
    return y.view(*x.shape[:-1], lin.out_features)


class SEABlockTemporal(BaseBlockTemporal):
    """State-Exchange Attention block: every field attends every other field through a down-projected, normalised,
    masked cross-attention, sequentially over fields with in-place update (reference :152-192; Gauss-Seidel order,
    SURVEY.md §0 item 4)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.n_heads = kwargs['n_heads']
        self.max_len = kwargs['max_len']
        self.src_len = kwargs['src_len']
        self.dropout = kwargs['dropout']
        self.LN_type = kwargs['LN_type']
        F, E, D = self.num_variables, self.internal_embed_dim, self.down_dim
        self.cross_down = nn.ModuleList([nn.Linear(E, D) for _ in range(F)])
        self.cross_up = nn.ModuleList([nn.Linear(D, E) for _ in range(F)])
        self.cross_attn = nn.ModuleList([
            nn.ModuleList([MaskedMultiHeadCrossAttention(self.n_heads, D, self.max_len, self.src_len, self.dropout) for _ in range(F)])
            for _ in range(F)])
        self.ln_cross = nn.ModuleList([_make_norm(self.LN_type, D, self.ib_num) for _ in range(F)])

    def _apply_cross_attention(self, x_i, x_j, i, j, x_add):
        n_i = self.ln_cross[i](_linear(self.cross_down[i], x_i), x_add)
        n_j = self.ln_cross[j](_linear(self.cross_down[j], x_j), x_add)
        g = self.cross_attn[i][j]._attend(n_i, n_j, gelu_out=True)  # GELU fused into the projection epilogue
        return g

    def _apply_exchange(self, x_vars, x_add):
        for i in range(len(x_vars)):
            x_i = x_vars[i]
            acc = x_i
            for j in range(len(x_vars)):
                if j != i:
                    g = self._apply_cross_attention(x_i, x_vars[j], i, j, x_add)
                    acc = _linear(self.cross_up[i], g, residual=acc)  # running sum carried as the residual operand
            x_vars[i] = acc
        return x_vars


class AddBlockTemporal(BaseBlockTemporal):
    """Ablation block (reference :279-301): x_i += cross_up_i(GELU(sum_j ln_cross_j(cross_down_j(x_j)))) with every field read at its
    pre-exchange value.  Parameter container for the whole-model path (sea_amd.engine runs it as three grouped launches); the stand-alone
    block forward is not provided."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.LN_type = kwargs['LN_type']
        F, E, D = self.num_variables, self.internal_embed_dim, self.down_dim
        self.cross_down = nn.ModuleList([nn.Linear(E, D) for _ in range(F)])
        self.cross_up = nn.ModuleList([nn.Linear(D, E) for _ in range(F)])
        self.ln_cross = nn.ModuleList([_make_norm(self.LN_type, D, self.ib_num) for _ in range(F)])

    def _apply_exchange(self, x_vars, x_add):
        raise NotImplementedError("sea_amd.AddBlockTemporal: call TemporalModel.forward (whole-model native path)")


class SEAPoolBlockTemporal(BaseBlockTemporal):
    """Ablation block with an information pool (reference :197-277, pool_update_method='mlp' — the only one create_block_temporal selects): every
    field attends a pool token sequence computed from all (pre-exchange) fields.  Parameter container for the whole-model path; the pool token
    and ln_pool are created, as in the reference, and never reach the output (the code overwrites the token it prepares)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.n_heads, self.max_len, self.src_len = kwargs['n_heads'], kwargs['max_len'], kwargs['src_len']
        self.dropout, self.LN_type = kwargs['dropout'], kwargs['LN_type']
        self.pool_update_method = kwargs.get('pool_update_method', 'mlp')
        if self.pool_update_method != 'mlp':
            raise NotImplementedError("sea_amd.SEAPoolBlockTemporal: pool_update_method 'mlp' only")
        F, E, D = self.num_variables, self.internal_embed_dim, self.down_dim
        self.pool_token = nn.Parameter(torch.randn(1, 1, D))
        self.cross_down = nn.ModuleList([nn.Linear(E, D) for _ in range(F)])
        self.cross_up = nn.ModuleList([nn.Linear(D, E) for _ in range(F)])
        self.cross_attn = nn.ModuleList([MaskedMultiHeadCrossAttention(self.n_heads, D, self.max_len, self.src_len, self.dropout) for _ in range(F)])
        self.ln_cross = nn.ModuleList([_make_norm(self.LN_type, D, self.ib_num) for _ in range(F)])
        self.ln_pool = _make_norm(self.LN_type, D, self.ib_num)
        self.pool_update = nn.Sequential(nn.Linear(D * F, D * 2), nn.GELU(), nn.Linear(D * 2, D))

    def _apply_exchange(self, x_vars, x_add):
        raise NotImplementedError("sea_amd.SEAPoolBlockTemporal: call TemporalModel.forward (whole-model native path)")


class SimpleBlockTemporal(BaseBlockTemporal):
    """Ablation block without exchange (reference :304-306)."""

    def _apply_exchange(self, x_vars, x_add):
        return x_vars


def create_block_temporal(exchange_mode, *args, **kwargs):
    if exchange_mode == 'sea':
        return SEABlockTemporal(*args, **kwargs)
    if exchange_mode == 'addition':
        return AddBlockTemporal(*args, **kwargs)
    if exchange_mode == 'simple':
        return SimpleBlockTemporal(*args, **kwargs)
    if exchange_mode == 'pool':
        return SEAPoolBlockTemporal(*args, **kwargs)
    raise ValueError(f"Invalid exchange_mode: {exchange_mode}")


class TemporalModel(nn.Module):
    """Causal transformer over time steps with one token stream per field group (reference :326-416).

    forward(x, x_additional_info): x [batch, time, field, cell] fp32, x_additional_info [batch, time, ib_num=1] -> same shape
    as x.  The tensors must live on the GPU: the forward runs through sea_amd.engine on libsea_hip.so and raises on CPU.

    compute dtype: `model.compute_dtype` ('fp32' default = exact-f32 MFMA, numerically the reference; 'bf16' = bf16 MFMA
    operands with fp32 accumulation, statistics and residual stream).  Set it with set_compute_dtype() or the
    SEA_AMD_DTYPE environment variable before the first forward.
    """

    def __init__(self, num_layers, embed_dim, n_heads, max_len, scale_ratio, src_len, num_variables, down_proj=2, dropout=0.0,
                 exchange_mode='sea', pos_encoding_mode='learnable', ib_scale_mode='fourier', ib_addition_mode='add',
                 ib_mlp_layers=1, ib_num=1, add_info_after_cross=True, LN_type='adaln'):
        super().__init__()
        self.num_variables = num_variables
        self.exchange_mode = self._validate_exchange_mode(exchange_mode)
        self.pos_encoding_mode = self._validate_pos_encoding_mode(pos_encoding_mode)
        self.ib_scale_mode = ib_scale_mode
        self.ib_addition_mode = ib_addition_mode
        self.ib_num = ib_num
        self.LN_type = LN_type
        # sizes the engine needs
        self.num_layers, self.embed_dim, self.n_heads, self.max_len = num_layers, embed_dim, n_heads, max_len
        self.src_len, self.down_proj, self.dropout_p = src_len, down_proj, dropout
        self.ib_mlp_layers, self.add_info_after_cross = ib_mlp_layers, add_info_after_cross
        # block-internal width: 'concat' widens the rows of every block by 64 info-bottleneck columns (models/temporal.py:40,48); down_dim / the MLP
        # hidden width follow it (:58-59, base_blocks.py:13)
        self.internal_embed_dim = embed_dim + 64 if str(ib_addition_mode).lower() == "concat" else embed_dim
        self.down_dim = self.internal_embed_dim // down_proj
        self.mlp_hidden = max(1, int(self.internal_embed_dim * scale_ratio))
        self.ib_hidden = max(1, int(1 * scale_ratio))  # blocks always see ib_num=1
        self.blocks = nn.ModuleList([
            create_block_temporal(
                self.exchange_mode, n_heads=n_heads, max_len=max_len, embed_dim=embed_dim, src_len=src_len, down_proj=down_proj,
                scale_ratio=scale_ratio, dropout=dropout, ib_scale_mode=self.ib_scale_mode, ib_addition_mode=self.ib_addition_mode,
                ib_mlp_layers=ib_mlp_layers, num_variables=num_variables, add_info_after_cross=add_info_after_cross,
                LN_type=self.LN_type)
            for _ in range(num_layers)])
        self.ln = nn.ModuleList([_make_norm(self.LN_type, embed_dim, self.ib_num) for _ in range(num_variables)])
        self.apply(self._init_weights)
        self.compute_dtype = os.environ.get("SEA_AMD_DTYPE", "fp32").lower()
        self._engine = None

    @staticmethod
    def _validate_exchange_mode(mode):
        mode = mode.lower()
        if mode not in _EXCHANGE_MODES:
            raise ValueError(f"Invalid exchange_mode '{mode}'. Must be one of {_EXCHANGE_MODES}.")
        return mode

    @staticmethod
    def _validate_pos_encoding_mode(mode):
        if mode not in {'learnable', 'fixed'}:
            raise ValueError(f"Invalid pos_encoding_mode '{mode}'. Must be one of {{'learnable', 'fixed'}}.")
        return mode

    def _init_weights(self, module):
        # every nn.Linear ~ N(0, 0.02) with zero bias; nn.LayerNorm / AdaLN gain 1, bias 0 (reference :392-399)
        if isinstance(module, nn.Linear):
            torch.nn.init.normal_(module.weight, mean=0.0, std=0.02)
            if module.bias is not None:
                torch.nn.init.zeros_(module.bias)
        elif isinstance(module, (nn.LayerNorm, AdaLN)):
            nn.init.constant_(module.bias, 0)
            nn.init.constant_(module.weight, 1.0)

    # ------------------------------------------------------------------ native engine plumbing
    def set_compute_dtype(self, dtype) -> "TemporalModel":
        name = {torch.float32: "fp32", torch.bfloat16: "bf16"}.get(dtype, dtype)
        if name not in ("fp32", "bf16"):
            raise ValueError("compute dtype must be 'fp32' or 'bf16'")
        if name != self.compute_dtype:
            self.compute_dtype = name
            self._engine = None
        return self

    def _apply(self, fn, *args, **kwargs):
        # .to()/.cuda()/.cpu() re-create parameter storage: drop the engine (and its flat buffers); it is rebuilt lazily
        self._engine = None
        return super()._apply(fn, *args, **kwargs)

    def _grad_anchor(self) -> torch.Tensor:
        """A leaf that requires grad, so that autograd calls the hand-written backward of the whole-model Function."""
        a = getattr(self, "_anchor", None)
        if a is None or a.device != next(self.parameters()).device:
            a = torch.zeros(1, device=next(self.parameters()).device, requires_grad=True)
            object.__setattr__(self, "_anchor", a)
        return a

    def _live_params(self):
        eng = self._engine
        cache = getattr(self, "_live_cache", None)
        if cache is None or cache[0] is not eng:
            named = dict(self.named_parameters())
            cache = (eng, [named[n] for n in eng.params.live_names])
            object.__setattr__(self, "_live_cache", cache)
        return cache[1]

    def engine(self, device: Optional[torch.device] = None):
        from ..engine import TemporalEngine

        if self._engine is None:
            dev = device if device is not None else next(self.parameters()).device
            if dev.type != "cuda":
                raise RuntimeError(
                    "sea_amd.TemporalModel: parameters are on the CPU. This implementation has no CPU path: move the model and its "
                    "inputs to the MI355X (model.to('cuda')). The CPU oracle used by the tests lives under oracle/.")
            act = torch.bfloat16 if self.compute_dtype == "bf16" else torch.float32
            self._engine = TemporalEngine(self, dev, act)
        return self._engine

    def forward(self, x, x_additional_info):
        # x shape: [batch, time, field, cell]
        assert x.shape[2] == self.num_variables, f"Expected {self.num_variables} variables, but got {x.shape[2]}"
        if not x.is_cuda:
            raise RuntimeError("sea_amd.TemporalModel.forward: input is on the CPU; this path has no CPU fallback")
        eng = self.engine(x.device)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from ..autograd import temporal_forward_with_grad

            return temporal_forward_with_grad(self, eng, x, x_additional_info)
        if self.training and self.dropout_p > 0.0:
            return eng.forward_train(x.float(), x_additional_info.float())[0]  # dropout is active in train(): counter-based masks
        return eng.forward(x.float(), x_additional_info.float())
