"""CPU oracle for SEA's temporal-rollout hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``sea_amd/`` may import this file; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do,
and there only as the checker / the timed CPU baseline — never as the product.

This is a from-spec, functional restatement (plain tensor algebra on the CPU, no
``nn.Module``) of what the reference computes on the path named by
BASELINE.json:north_star.  Every function cites the reference lines it follows
(paths are relative to the reference repository root).  Parameters are addressed
by the reference's ``state_dict`` key names, so a reference-trained checkpoint
can be fed to it unchanged.

Pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md §4), so the oracle is pinned against outputs of the reference itself,
generated in the build container by ``tests/golden/make_fixtures.py`` and
committed as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every
function here against them.

The arithmetic is floating point; ``dtype`` may be ``torch.float32`` (what the
reference computes in) or ``torch.float64`` (a tighter truth for tolerance
studies).  Gradients come from autograd over these functions.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

Params = Dict[str, torch.Tensor]


@dataclass(frozen=True)
class OracleConfig:
    """Constructor arguments of the reference ``TemporalModel`` that shape the
    computation (models/temporal.py:327-344).  Restated: exchange_mode in {'sea' (both
    shipped configs), 'addition', 'simple'}, ib_scale_mode in {'mlp', 'linear', 'fourier'},
    ib_addition_mode in {'add', 'attention', 'concat', 'none'}, ib_mlp_layers=1, ib_num=1."""

    num_layers: int
    embed_dim: int
    n_heads: int
    max_len: int
    scale_ratio: int
    src_len: int
    num_variables: int
    down_proj: int = 2
    add_info_after_cross: bool = True
    LN_type: str = "adaln"
    exchange_mode: str = "sea"
    ib_addition_mode: str = "add"
    ib_scale_mode: str = "mlp"

    @property
    def internal_embed_dim(self) -> int:  # models/temporal.py:40,48: 'concat' widens the block's rows by ib_dim_concat = 64 columns
        return self.embed_dim + 64 if self.ib_addition_mode == "concat" else self.embed_dim

    @property
    def down_dim(self) -> int:  # models/temporal.py:58-59
        return self.internal_embed_dim // self.down_proj


# --------------------------------------------------------------------------- primitives
def linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.Linear: y = x W^T + b with W stored [out, in]."""
    y = x @ w.t()
    return y if b is None else y + b


def gelu_erf(x: torch.Tensor) -> torch.Tensor:
    """nn.GELU() default = exact erf form (models/base_blocks.py:24,74)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def silu(x: torch.Tensor) -> torch.Tensor:
    return x / (1.0 + torch.exp(-x))


def layer_norm(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], eps: float = 1e-5) -> torch.Tensor:
    """F.layer_norm over the last dim, biased variance (models/base_blocks.py:87-88 and the
    nn.LayerNorm inside MLP, :23)."""
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    y = (x - mean) / torch.sqrt(var + eps) * w
    return y if b is None else y + b


def adaln(x: torch.Tensor, cond: torch.Tensor, p: Params, pre: str) -> torch.Tensor:
    """AdaLN.forward (models/base_blocks.py:343-350)."""
    h = silu(linear(cond, p[pre + "cond_mlp.0.weight"], p[pre + "cond_mlp.0.bias"]))
    c = linear(h, p[pre + "cond_mlp.2.weight"], p[pre + "cond_mlp.2.bias"])
    d = x.shape[-1]
    w, b = c[..., :d], c[..., d:]
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    xn = (x - mean) / torch.sqrt(var + 1e-5)
    return xn * (p[pre + "weight"] + (w + 1.0)) + (p[pre + "bias"] + b)


def norm(x: torch.Tensor, cond: torch.Tensor, p: Params, pre: str, ln_type: str) -> torch.Tensor:
    """Dispatch on LN_type (models/temporal.py:61-72): 'adaln' -> AdaLN, 'ln' -> the custom
    bias-free LayerNorm that ignores cond (models/base_blocks.py:80-88)."""
    if ln_type.lower() == "adaln":
        return adaln(x, cond, p, pre)
    return layer_norm(x, p[pre + "weight"], None)


def rope_tables(head_dim: int, length: int, theta: float = 10000.0, dtype=torch.float32) -> Tuple[torch.Tensor, torch.Tensor]:
    """precompute_freqs_cis (models/base_blocks.py:300-305) as separate cos/sin tables
    [length, head_dim/2]; frequencies and angles are formed in fp32 as the reference does."""
    freqs = 1.0 / (theta ** (torch.arange(0, head_dim, 2)[: head_dim // 2].float() / head_dim))
    ang = torch.outer(torch.arange(length, dtype=torch.float32), freqs)
    return torch.cos(ang).to(dtype), torch.sin(ang).to(dtype)


def rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """apply_rotary_emb (models/base_blocks.py:314-324): consecutive (even, odd) pairs of the
    head dim are complex numbers multiplied by e^{i t f_k}.  x: [B, T, H, hd]."""
    xe, xo = x[..., 0::2], x[..., 1::2]
    c = cos[None, :, None, :]
    s = sin[None, :, None, :]
    out = torch.stack((xe * c - xo * s, xe * s + xo * c), dim=-1)
    return out.flatten(-2)


def masked_attention(x_q: torch.Tensor, x_kv: torch.Tensor, p: Params, pre: str, n_heads: int, src_len: int,
                     pos0: int = 0) -> torch.Tensor:
    """MaskedMultiHeadAttention.forward (x_q is x_kv; models/base_blocks.py:175-203) and
    MaskedMultiHeadCrossAttention.forward (:267-295): q from x_q, k/v from x_kv, the same RoPE
    positions for q and k, causal mask tril(diagonal=src_len), softmax, projection without bias."""
    B, T, C = x_q.shape
    hd = C // n_heads
    q = linear(x_q, p[pre + "q.weight"], p[pre + "q.bias"]).view(B, T, n_heads, hd)
    k = linear(x_kv, p[pre + "k.weight"], p[pre + "k.bias"]).view(B, T, n_heads, hd)
    v = linear(x_kv, p[pre + "v.weight"], p[pre + "v.bias"]).view(B, T, n_heads, hd)
    cos, sin = rope_tables(hd, pos0 + T, dtype=x_q.dtype)
    q = rope(q, cos[pos0:], sin[pos0:])
    k = rope(k, cos[pos0:], sin[pos0:])
    q, k, v = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
    att = (q @ k.transpose(-2, -1)) * hd ** -0.5
    i = torch.arange(T)
    allowed = i[None, :] <= i[:, None] + src_len  # tril(ones, diagonal=src_len) != 0
    att = att.masked_fill(~allowed, float("-inf"))
    att = att - att.max(dim=-1, keepdim=True).values
    e = torch.exp(att)
    att = e / e.sum(dim=-1, keepdim=True)
    out = (att @ v).transpose(1, 2).reshape(B, T, C)
    return linear(out, p[pre + "projection.weight"])


def mlp(x: torch.Tensor, p: Params, pre: str) -> torch.Tensor:
    """MLP with num_layers None/1: Linear -> nn.LayerNorm(hidden, affine) -> GELU(erf) -> Linear
    (models/base_blocks.py:22-26, 44-47); dropout is identity in eval / p=0."""
    h = linear(x, p[pre + "layers.0.weight"], p[pre + "layers.0.bias"])
    h = layer_norm(h, p[pre + "layers.1.weight"], p[pre + "layers.1.bias"])
    h = gelu_erf(h)
    return linear(h, p[pre + "layers.3.weight"], p[pre + "layers.3.bias"])


# --------------------------------------------------------------------------- block / model
def sea_exchange(xs: List[torch.Tensor], cond: torch.Tensor, p: Params, pre: str, cfg: OracleConfig) -> List[torch.Tensor]:
    """SEABlockTemporal._apply_exchange / _apply_cross_attention (models/temporal.py:176-192).

    Gauss-Seidel: the list is updated in place while it is iterated, so field i is exchanged
    against the ALREADY UPDATED fields j < i and the not-yet-updated fields j > i; x_i itself
    enters as its pre-exchange value."""
    F = cfg.num_variables
    xs = list(xs)
    for i in range(F):
        x_i = xs[i]
        acc = None
        for j in range(F):
            if j == i:
                continue
            d_i = linear(x_i, p[f"{pre}cross_down.{i}.weight"], p[f"{pre}cross_down.{i}.bias"])
            d_j = linear(xs[j], p[f"{pre}cross_down.{j}.weight"], p[f"{pre}cross_down.{j}.bias"])
            n_i = norm(d_i, cond, p, f"{pre}ln_cross.{i}.", cfg.LN_type)
            n_j = norm(d_j, cond, p, f"{pre}ln_cross.{j}.", cfg.LN_type)
            a = masked_attention(n_i, n_j, p, f"{pre}cross_attn.{i}.{j}.", cfg.n_heads, cfg.src_len)
            u = linear(gelu_erf(a), p[f"{pre}cross_up.{i}.weight"], p[f"{pre}cross_up.{i}.bias"])
            acc = u if acc is None else acc + u
        xs[i] = x_i if acc is None else x_i + acc
    return xs


def add_exchange(xs: List[torch.Tensor], cond: torch.Tensor, p: Params, pre: str, cfg: OracleConfig) -> List[torch.Tensor]:
    """AddBlockTemporal._apply_exchange (models/temporal.py:291-301): every field is down-projected and normalised FROM ITS
    PRE-EXCHANGE VALUE (Jacobi, unlike 'sea'); x_i += cross_up_i(GELU(normalized_i + sum_{j != i} normalized_j))."""
    F = cfg.num_variables
    nrm = [norm(linear(xs[j], p[f"{pre}cross_down.{j}.weight"], p[f"{pre}cross_down.{j}.bias"]), cond, p, f"{pre}ln_cross.{j}.", cfg.LN_type)
           for j in range(F)]
    out = []
    for i in range(F):
        others = None
        for j in range(F):
            if j != i:
                others = nrm[j] if others is None else others + nrm[j]
        combined = nrm[i] if others is None else nrm[i] + others
        out.append(xs[i] + linear(gelu_erf(combined), p[f"{pre}cross_up.{i}.weight"], p[f"{pre}cross_up.{i}.bias"]))
    return out


def pool_exchange(xs: List[torch.Tensor], cond: torch.Tensor, p: Params, pre: str, cfg: OracleConfig) -> List[torch.Tensor]:
    """SEAPoolBlockTemporal._apply_exchange (models/temporal.py:255-277) with pool_update_method='mlp' (the only one create_block_temporal
    can select, :314-324) in eval mode: every field is down-projected, normalised and given the sinusoidal positions of
    PositionalEncoding (models/base_blocks.py:355-372) from its pre-exchange value; the pool token the code prepares (pool_token,
    ln_pool) is overwritten by pool_update(cat(normalized)) (:270: Linear(F D -> 2 D), GELU, Linear(2 D -> D)); field i attends the
    pool (masked cross-attention, q from the field), and x_i += cross_up_i(GELU(normalized_i + attention))."""
    F = cfg.num_variables
    T_ = xs[0].shape[1]
    pe = sinusoidal_pe(T_, cfg.down_dim, xs[0].dtype)[None]
    nrm = [norm(linear(xs[j], p[f"{pre}cross_down.{j}.weight"], p[f"{pre}cross_down.{j}.bias"]), cond, p, f"{pre}ln_cross.{j}.", cfg.LN_type) + pe
           for j in range(F)]
    cat = torch.cat(nrm, dim=-1)
    pool = linear(gelu_erf(linear(cat, p[pre + "pool_update.0.weight"], p[pre + "pool_update.0.bias"])), p[pre + "pool_update.2.weight"], p[pre + "pool_update.2.bias"])
    out = []
    for i in range(F):
        a = masked_attention(nrm[i], pool, p, f"{pre}cross_attn.{i}.", cfg.n_heads, cfg.src_len)
        out.append(xs[i] + linear(gelu_erf(nrm[i] + a), p[f"{pre}cross_up.{i}.weight"], p[f"{pre}cross_up.{i}.bias"]))
    return out


def exchange(xs: List[torch.Tensor], cond: torch.Tensor, p: Params, pre: str, cfg: OracleConfig) -> List[torch.Tensor]:
    """create_block_temporal's dispatch (models/temporal.py:314-324); 'simple' = SimpleBlockTemporal (:304-306), no exchange."""
    if cfg.exchange_mode == "sea":
        return sea_exchange(xs, cond, p, pre, cfg)
    if cfg.exchange_mode == "addition":
        return add_exchange(xs, cond, p, pre, cfg)
    if cfg.exchange_mode == "simple":
        return list(xs)
    if cfg.exchange_mode == "pool":
        return pool_exchange(xs, cond, p, pre, cfg)
    raise ValueError(f"Invalid exchange_mode: {cfg.exchange_mode}")


def info_bottleneck(cond: torch.Tensor, p: Params, pre: str, mode: str = "mlp") -> torch.Tensor:
    """The ib layer of BaseBlockTemporal (models/temporal.py:103-109) as used by _add_info with
    ib_addition_mode='add' (:111-116).  'mlp': MLP(1 -> scale_ratio -> E), whose residual_projection is
    created but never used (models/base_blocks.py:15-17); 'linear': nn.Linear(1, E); 'fourier':
    GaussianFourierProjection (models/base_blocks.py:143-151): x_proj = x @ W * 2 * pi in fp32, [sin, cos]."""
    if mode == "mlp":
        return mlp(cond, p, pre + "ib.")
    if mode == "linear":
        return linear(cond, p[pre + "ib.weight"], p[pre + "ib.bias"])
    if mode == "fourier":
        x_proj = cond @ p[pre + "ib.W"] * 2 * math.pi
        return torch.cat([torch.sin(x_proj), torch.cos(x_proj)], dim=-1)
    raise ValueError(f"Invalid ib_scale_mode '{mode}'")


def plain_cross_attention(x_q: torch.Tensor, x_kv: torch.Tensor, p: Params, pre: str, n_heads: int) -> torch.Tensor:
    """MultiHeadCrossAttention.forward (models/base_blocks.py:222-243): q from x_q, k / v from x_kv, no rotary embedding, NO mask (the module's
    `tril` buffer is never read), softmax over all source rows, projection without bias."""
    B, T, C = x_q.shape
    Ts = x_kv.shape[1]
    hd = C // n_heads
    q = linear(x_q, p[pre + "q.weight"], p[pre + "q.bias"]).view(B, T, n_heads, hd).transpose(1, 2)
    k = linear(x_kv, p[pre + "k.weight"], p[pre + "k.bias"]).view(B, Ts, n_heads, hd).transpose(1, 2)
    v = linear(x_kv, p[pre + "v.weight"], p[pre + "v.bias"]).view(B, Ts, n_heads, hd).transpose(1, 2)
    att = (q @ k.transpose(-2, -1)) * hd ** -0.5
    att = att - att.max(dim=-1, keepdim=True).values
    e = torch.exp(att)
    att = e / e.sum(dim=-1, keepdim=True)
    out = (att @ v).transpose(1, 2).reshape(B, T, C)
    return linear(out, p[pre + "projection.weight"])


def add_info(xs: List[torch.Tensor], cond: torch.Tensor, p: Params, pre: str, cfg: OracleConfig) -> List[torch.Tensor]:
    """BaseBlockTemporal._add_info over the fields (models/temporal.py:111-120): 'none' returns x, 'add' adds the info-bottleneck rows, 'attention' adds
    cross_attn_ib[i](x_i, ib rows) — every row attends to the info-bottleneck rows of ALL positions of its trajectory."""
    if cfg.ib_addition_mode == "none":
        return xs
    ib = info_bottleneck(cond, p, pre, cfg.ib_scale_mode)
    if cfg.ib_addition_mode == "add":
        return [x + ib for x in xs]
    if cfg.ib_addition_mode == "attention":
        return [x + plain_cross_attention(x, ib, p, f"{pre}cross_attn_ib.{i}.", cfg.n_heads) for i, x in enumerate(xs)]
    if cfg.ib_addition_mode == "concat":   # :115-116 (the layer's width is ib_dim_concat, :100-101)
        return [torch.cat([x, ib], dim=-1) for x in xs]
    raise ValueError(f"ib_addition_mode {cfg.ib_addition_mode!r} is not restated")


def block_forward(xs: Sequence[torch.Tensor], cond: torch.Tensor, p: Params, pre: str, cfg: OracleConfig) -> List[torch.Tensor]:
    """BaseBlockTemporal.forward (models/temporal.py:126-148)."""
    F = cfg.num_variables
    assert len(xs) == F
    xs = list(xs)
    if not cfg.add_info_after_cross:
        xs = add_info(xs, cond, p, pre, cfg)
    for i in range(F):
        n = norm(xs[i], cond, p, f"{pre}ln.exp.{i}.0.", cfg.LN_type)
        xs[i] = xs[i] + masked_attention(n, n, p, f"{pre}attn.self.{i}.", cfg.n_heads, cfg.src_len)
    xs = exchange(xs, cond, p, pre, cfg)
    if cfg.add_info_after_cross:
        xs = add_info(xs, cond, p, pre, cfg)
    for i in range(F):
        n = norm(xs[i], cond, p, f"{pre}ln.exp.{i}.2.", cfg.LN_type)
        xs[i] = xs[i] + mlp(n, p, f"{pre}mlp.{i}.")
        xs[i] = linear(xs[i], p[f"{pre}proj.{i}.weight"], p[f"{pre}proj.{i}.bias"])
    return xs


def model_forward(x: torch.Tensor, cond: torch.Tensor, p: Params, cfg: OracleConfig) -> torch.Tensor:
    """TemporalModel.forward (models/temporal.py:405-416).  x: [B, T, F, E], cond: [B, T, 1]."""
    assert x.shape[2] == cfg.num_variables
    xs = [x[:, :, i, :] for i in range(cfg.num_variables)]
    for layer in range(cfg.num_layers):
        xs = block_forward(xs, cond, p, f"blocks.{layer}.", cfg)
    xs = [norm(xs[i], cond, p, f"ln.{i}.", cfg.LN_type) for i in range(cfg.num_variables)]
    return torch.stack(xs, dim=2)


# --------------------------------------------------------------------------- loss / metrics / loops
def mse_loss(out: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """nn.MSELoss() (train/train_temporal.py:221,256)."""
    return ((out - target) ** 2).mean()


def relative_mse(pred: torch.Tensor, truth: torch.Tensor, dim: int = -1) -> torch.Tensor:
    """relativeMSE (utils/train_utils.py:112-116)."""
    return ((pred - truth) ** 2).sum(dim=dim) / ((truth ** 2).sum(dim=dim) + 1e-8)


def rollout(x0: torch.Tensor, cond: torch.Tensor, n_steps: int, p: Params, cfg: OracleConfig) -> torch.Tensor:
    """The autoregressive loop of full_autoregressive_evaluation / autoregressive_validation
    (utils/train_utils.py:202-209, :170-177): start from step 0, run the FULL forward on the
    growing prefix, append the last predicted step.  x0: [B, 1, F, E]; cond: [B, >=n_steps, 1].
    Returns the n_steps predictions [B, n_steps, F, E]."""
    a = x0
    with torch.no_grad():
        for i in range(n_steps):
            out = model_forward(a, cond[:, : i + 1], p, cfg)
            a = torch.cat((a, out[:, -1:]), dim=1)
    return a[:, 1:]


def live_param_keys(p: Params, cfg: OracleConfig) -> List[str]:
    """Parameters that receive a gradient on this path: everything except
    ln.exp.{i}.1, ln.cross, the diagonal cross_attn.{i}.{i}, ib.residual_projection (and, per variant, the unused info-bottleneck layer / the pool token)."""
    dead_marks = []
    for layer in range(cfg.num_layers):
        pre = f"blocks.{layer}."
        dead_marks.append(pre + "ln.cross.")
        dead_marks.append(pre + "ib.residual_projection.")
        if getattr(cfg, "ib_addition_mode", "add") == "none":      # the layer is evaluated but its output dropped (models/temporal.py:112-114)
            dead_marks.append(pre + "ib.")
        elif getattr(cfg, "ib_scale_mode", "mlp") == "fourier":    # fixed random features (models/base_blocks.py:147: requires_grad=False)
            dead_marks.append(pre + "ib.W")
        for i in range(cfg.num_variables):
            dead_marks.append(f"{pre}ln.exp.{i}.1.")
            dead_marks.append(f"{pre}cross_attn.{i}.{i}.")
        if getattr(cfg, "exchange_mode", "sea") == "pool":         # the 'mlp' pool update ignores its token argument (models/temporal.py:244-249)
            dead_marks += [pre + "pool_token", pre + "ln_pool."]
    return [k for k in p if not any(k.startswith(d) for d in dead_marks)]


def loss_and_grads(x: torch.Tensor, cond: torch.Tensor, target: torch.Tensor, p: Params, cfg: OracleConfig
                   ) -> Tuple[torch.Tensor, torch.Tensor, Dict[str, torch.Tensor]]:
    """One forward + MSE + backward (train/train_temporal.py:255-257).  Returns (out, loss, grads of
    live parameters)."""
    keys = live_param_keys(p, cfg)
    q = dict(p)
    for k in keys:
        q[k] = p[k].detach().clone().requires_grad_(True)
    out = model_forward(x, cond, q, cfg)
    loss = mse_loss(out, target)
    gs = torch.autograd.grad(loss, [q[k] for k in keys], allow_unused=True)
    grads = {k: g for k, g in zip(keys, gs) if g is not None}
    return out.detach(), loss.detach(), grads


def adamw_update(param: torch.Tensor, grad: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float,
                 beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, weight_decay: float = 0.0
                 ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """torch.optim.AdamW as configured by initialize_optimizer (utils/train_utils.py:33-34):
    decoupled weight decay, bias-corrected moments.  ``step`` counts from 1."""
    param = param * (1.0 - lr * weight_decay)
    m = beta1 * m + (1.0 - beta1) * grad
    v = beta2 * v + (1.0 - beta2) * grad * grad
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = torch.sqrt(v) / math.sqrt(bc2) + eps
    param = param - (lr / bc1) * m / denom
    return param, m, v


def train_steps(x: torch.Tensor, cond: torch.Tensor, target: torch.Tensor, p: Params, cfg: OracleConfig, n_steps: int,
                lr: float, weight_decay: float = 0.0) -> Tuple[Params, List[float]]:
    """n_steps iterations of the reference train step (train/train_temporal.py:254-258) on one batch."""
    p = {k: v.clone() for k, v in p.items()}
    ms: Dict[str, torch.Tensor] = {}
    vs: Dict[str, torch.Tensor] = {}
    losses = []
    for step in range(1, n_steps + 1):
        _, loss, grads = loss_and_grads(x, cond, target, p, cfg)
        losses.append(float(loss))
        for k, g in grads.items():
            if k not in ms:
                ms[k] = torch.zeros_like(p[k])
                vs[k] = torch.zeros_like(p[k])
            p[k], ms[k], vs[k] = adamw_update(p[k], g, ms[k], vs[k], step, lr, weight_decay=weight_decay)
    return p, losses


# ---------------------------------------------------------------------------------------------------- spatial decoder
def rollout_to_patches(rollout_out: torch.Tensor, n_patches: int) -> torch.Tensor:
    """utils/train_utils.py:339-362 (inverse_transform_processed_data): [tr, T, G, P*D] -> [tr*T, P, G, D]."""
    tr, T, G, PD = rollout_out.shape
    D = PD // n_patches
    return rollout_out.reshape(tr, T, G, n_patches, D).permute(0, 1, 3, 2, 4).reshape(tr * T, n_patches, G, D)


def decode(z: torch.Tensor, p: Params, field_groups: Sequence[Sequence[int]], pre: str = "decoders.") -> torch.Tensor:
    """models/encoder_decoder.py:126-146 (Decode.forward) with upScaleMLP of models/base_blocks.py:49-63:
    per field group g, y = W2 gelu(W1 z[:, :, g, :]) + b2 (layer1 has no bias), reshaped to [B, P, |group|, n_inp]; groups are
    concatenated along the field axis -> [B, P, n_fields, n_inp]."""
    B, P_, _, _ = z.shape
    outs = []
    for i, group in enumerate(field_groups):
        h = gelu_erf(linear(z[:, :, i:i + 1, :], p[f"{pre}{i}.layer1.weight"]))
        y = linear(h, p[f"{pre}{i}.layer2.weight"], p[f"{pre}{i}.layer2.bias"])
        outs.append(y.reshape(B, P_, len(group), -1))
    return torch.cat(outs, dim=2)


def unpatchify(scaled_fields: torch.Tensor, index_map: torch.Tensor, n_points: int, field_groups: Sequence[Sequence[int]] = (),
               scalers: Sequence[Tuple[float, float, float, float]] = ()) -> torch.Tensor:
    """utils/data_processors.py:93-111 (DataPartitioner2D.inverse_partition) + :553-573 (inverse_scale_and_unpatch) + :258-273
    (MinMaxScaler.inverse_transform).  scaled_fields [T, P, C, F]; index_map [P, C] with -1 padding; scalers: one
    (range_lo, range_hi, min_val, max_val) per field group."""
    T_, P_, C_, F_ = scaled_fields.shape
    out = torch.empty(T_, n_points, F_, dtype=scaled_fields.dtype)
    for p_ in range(P_):
        idx = index_map[p_]
        valid = idx >= 0
        out[:, idx[valid].long(), :] = scaled_fields[:, p_, valid, :]
    if scalers:
        res = torch.zeros_like(out)
        for group, (r0, r1, lo, hi) in zip(field_groups, scalers):
            std = (out[..., list(group)] - r0) / (r1 - r0)
            res[..., list(group)] = std * (hi - lo) + lo
        out = res
    return out


# ---------------------------------------------------------------------------------------------------- spatial encoder
def sinusoidal_pe(length: int, d_model: int, dtype=torch.float32) -> torch.Tensor:
    """PositionalEncoding's buffer (models/base_blocks.py:355-368), formed in fp32 as the reference does: even columns sin, odd cos."""
    pe = torch.zeros(length, d_model)
    position = torch.arange(0, length, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term[: d_model // 2])
    return pe.to(dtype)


def full_attention(x: torch.Tensor, p: Params, pre: str, n_heads: int) -> torch.Tensor:
    """MultiHeadAttention.forward (models/base_blocks.py:105-121): no mask, no rotary embedding, projection without bias."""
    B, T, C = x.shape
    hd = C // n_heads
    q = linear(x, p[pre + "q.weight"], p[pre + "q.bias"]).view(B, T, n_heads, hd).transpose(1, 2)
    k = linear(x, p[pre + "k.weight"], p[pre + "k.bias"]).view(B, T, n_heads, hd).transpose(1, 2)
    v = linear(x, p[pre + "v.weight"], p[pre + "v.bias"]).view(B, T, n_heads, hd).transpose(1, 2)
    att = (q @ k.transpose(-2, -1)) * hd ** -0.5
    att = att - att.max(dim=-1, keepdim=True).values
    e = torch.exp(att)
    att = e / e.sum(dim=-1, keepdim=True)
    out = (att @ v).transpose(1, 2).reshape(B, T, C)
    return linear(out, p[pre + "projection.weight"])


def encode(x: torch.Tensor, p: Params, field_groups: Sequence[Sequence[int]], n_heads: int, num_layers: int, pre: str = "") -> torch.Tensor:
    """PointwiseEncode.forward (models/encoder_decoder.py:103-123) in eval mode: per-group downScaleMLP (models/base_blocks.py:65-78; layer1
    without bias, GELU, layer2), groups concatenated along the feature axis, sinusoidal positions over the P patch tokens, num_layers
    EncoderBlocks (models/base_blocks.py:123-139: x += MHA(LN(x)); x += MLP_{x4}(LN(x)), both LayerNorms weight-only), nn.LayerNorm.
    x: [B, P, F, C] -> [B, P, n_groups, embed_dim]."""
    B, P, F, C = x.shape
    zs = []
    for i, group in enumerate(field_groups):
        xg = x[:, :, list(group), :].reshape(B, P, -1)
        h = gelu_erf(linear(xg, p[f"{pre}encoders.{i}.layer1.weight"]))
        zs.append(linear(h, p[f"{pre}encoders.{i}.layer2.weight"], p[f"{pre}encoders.{i}.layer2.bias"]))
    z = torch.cat(zs, dim=-1)
    W = z.shape[-1]
    z = z + sinusoidal_pe(P, W, z.dtype)[None]
    for l in range(num_layers):
        b = f"{pre}blocks.{l}."
        z = z + full_attention(layer_norm(z, p[b + "ln_exp1_1.weight"], None), p, b + "attn_1.", n_heads)
        z = z + mlp(layer_norm(z, p[b + "ln_exp1_2.weight"], None), p, b + "mlp_1.")
    z = layer_norm(z, p[pre + "ln.weight"], p[pre + "ln.bias"])
    return z.reshape(B, P, len(field_groups), -1)


# ---------------------------------------------------------------------------------------------------- patchify / dataset windows
def patchify(fields: torch.Tensor, index_map: torch.Tensor, field_groups: Sequence[Sequence[int]] = (), scaler_params: Sequence[Sequence[float]] = (),
             pad_value: float = 0.0) -> torch.Tensor:
    """MeshProcessor._scale_fields (utils/data_processors.py:528-536; MinMaxScaler.transform :241-247) followed by DataPartitioner2D.create_partitions /
    pad_partitions (:60-92) and the stack of :519-522.  fields [T, N, F]; index_map [P, C] with -1 padding; scaler_params per group
    (range_lo, range_hi, min, max).  Returns [T, P, C, F]; padded slots hold pad_value (not scaled)."""
    x = fields.clone()
    for g, (r0, r1, lo, hi) in zip(field_groups, scaler_params):
        std = (fields[..., list(g)] - lo) / (hi - lo)
        x[..., list(g)] = std * (r1 - r0) + r0
    idx = index_map.long()
    out = x[:, idx.clamp_min(0).reshape(-1), :].reshape(x.shape[0], idx.shape[0], idx.shape[1], x.shape[2])
    return torch.where((idx >= 0)[None, :, :, None], out, torch.full_like(out, pad_value))


def dataset_window(data_list: Sequence[torch.Tensor], idx: int, src_len: int, overlap: int) -> Tuple[int, int, int]:
    """TemporalDataset.__getitem__ without time shifting (utils/data_processors.py:420-452): (segment, first step, one-past-last step) of sample idx;
    the target window is the same one step later."""
    step = src_len - overlap
    cum = 0
    for s, d in enumerate(data_list):
        n = d.shape[0] // step
        if idx < cum + n:
            a = (idx - cum) * step
            return s, a, a + src_len
        cum += n
    raise IndexError("Index out of range")

