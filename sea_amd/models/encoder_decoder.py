"""Spatial decoder of the rollout's consumer side: the reference's `Decode` module
(models/encoder_decoder.py:126-146) and its `upScaleMLP` (models/base_blocks.py:49-63) with the same constructor arguments,
parameter names and shapes, computed by two grouped-GEMM launches of libsea_hip.so (sea_gemm_grouped: Linear + GELU epilogue, then
Linear + bias written straight into the concatenated [B, P, n_fields, n_inp] output) — no per-group Python loop of ATen ops, no cat.

`PointwiseEncode` (models/encoder_decoder.py:75-123) is the inference forward of the 12-layer spatial encoder: the same
constructor, sub-module and parameter names, run as a fixed sequence of libsea_hip.so launches per chunk of snapshots — grouped down-scale
MLPs (GELU epilogue; the sinusoidal patch positions ride as the residual operand of the second Linear), and per EncoderBlock:
weight-only LayerNorm, fused q/k/v projection written in the attention layouts, un-masked flash attention (the causal kernel with the whole
row visible), projection + residual, LayerNorm, Linear, LayerNorm + GELU, Linear + residual.  `SpatialModel` ties encoder and decoder
together as the reference does.  Training of the spatial model is out of scope.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.nn as nn

from .. import _native as N
from .. import ops
from .base_blocks import EncoderBlock, PositionalEncoding, downScaleMLP


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
        if embed_dim % 8 or MLP_hidden % 8:
            raise NotImplementedError("sea_amd.Decode: embed_dim and MLP_hidden must be multiples of 8 (16-byte operand rows)")
        # a field's output columns are padded to whole 128-byte lines inside (n_inp is the padded cell size: data-dependent in the reference): every row and
        # every field of the fp32 output then starts on a cache line — rows at odd multiples of 16 B cost the second layer 12 % (1.48 -> 1.29 ms for the
        # 2.5 GB of a 2024-snapshot rollout: partial lines at both ends of every 512-byte tile row)
        self._n_inp_p = (n_inp + 31) // 32 * 32

    def set_compute_dtype(self, dtype) -> "Decode":
        name = {torch.float32: "fp32", torch.bfloat16: "bf16"}.get(dtype, dtype)
        if name not in ("fp32", "bf16"):
            raise ValueError("compute dtype must be 'fp32' or 'bf16'")
        self.compute_dtype, self._shadow = name, None
        return self

    def _weights(self, dt: torch.dtype):
        """Activation-dtype copies of the Linear weights, refreshed when a parameter was written (version counter) or moved."""
        ps = [d.layer1.weight for d in self.decoders] + [d.layer2.weight for d in self.decoders] + [d.layer2.bias for d in self.decoders]
        key = (dt, tuple((p.data_ptr(), p._version) for p in ps))
        if self._shadow is None or self._shadow[0] != key:
            with torch.no_grad():
                conv = lambda p: p.detach().contiguous() if dt == torch.float32 else p.detach().to(dt).contiguous()
                C, Cp = self.n_inp, self._n_inp_p

                def pad_out(t, grp):   # layer2 rows / bias entries of field f at [f C, (f + 1) C) -> [f Cp, f Cp + C); the pad rows are zero
                    if Cp == C:
                        return t.detach()
                    out = torch.zeros((len(grp), Cp) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
                    out[:, :C] = t.detach().view((len(grp), C) + tuple(t.shape[1:]))
                    return out.view((len(grp) * Cp,) + tuple(t.shape[1:]))

                self._shadow = (key, [conv(d.layer1.weight) for d in self.decoders], [conv(pad_out(d.layer2.weight, g)) for d, g in zip(self.decoders, self.field_groups)],
                                [pad_out(d.layer2.bias, g).float().contiguous() for d, g in zip(self.decoders, self.field_groups)])
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
        Cp = self._n_inp_p
        out = torch.empty(M, n_fields * Cp, device=z.device, dtype=torch.float32)
        hid: List[torch.Tensor] = [torch.empty(M, self.MLP_hidden, device=z.device, dtype=dt) for _ in range(G)]
        ops.gemm_grouped([dict(A=za[:, g * D:(g + 1) * D], W=W1[g], Cact=hid[g], act=1) for g in range(G)], dt)
        groups, off = [], 0
        for g, grp in enumerate(self.field_groups):
            w = len(grp) * Cp
            groups.append(dict(A=hid[g], W=W2[g], bias=self._shadow[3][g], C32=out[:, off:off + w]))
            off += w
        ops.gemm_grouped(groups, dt)
        out = out.view(B, P, n_fields, Cp)
        return out if Cp == self.n_inp else out[..., :self.n_inp]   # a strided view: sea_unpatchify reads it in place


    def forward_prefix(self, z: torch.Tensor, buckets) -> torch.Tensor:
        """The decoder for a consumer that only reads the first cells of a patch (MeshUnpatcher.decode_and_unpatch: a patch holds as many mesh points as its
        cell has, the rest of its n_inp columns is padding nobody reads — 71 % of the columns on the bench's wake-refined mesh).  z [B, P, n_groups,
        embed_dim] with the patches in the CALLER's order; buckets: [(p_lo, p_hi, n_cols)] covering 0 .. P — for the patches p_lo .. p_hi-1 only the first
        n_cols columns of every field are computed.  Rows are patch-major inside (a bucket is one contiguous row range: one GEMM group per (bucket, field)
        with the first n_cols rows of the field's second-layer weights).  Returns [B, P, n_fields, n_inp] as a strided view; columns >= n_cols of a
        bucket's patches are UNDEFINED."""
        N.require_gpu(z, "Decode input")
        B, P, G, D = z.shape
        assert G == self.num_groups and D == self.embed_dim, (z.shape, self.num_groups, self.embed_dim)
        dt = torch.float32 if self.compute_dtype == "fp32" else torch.bfloat16
        M = B * P
        zf = z.detach().to(torch.float32).permute(1, 0, 2, 3).contiguous().view(M, G * D)      # patch-major rows
        za = zf if dt == torch.float32 else torch.empty(M, G * D, device=z.device, dtype=dt)
        if dt != torch.float32:
            ops.convert(zf, za)
        W1, W2 = self._weights(dt)
        bias = self._shadow[3]
        n_fields = sum(len(g) for g in self.field_groups)
        Cp = self._n_inp_p
        out = torch.empty(M, n_fields * Cp, device=z.device, dtype=torch.float32)
        hid: List[torch.Tensor] = [torch.empty(M, self.MLP_hidden, device=z.device, dtype=dt) for _ in range(G)]
        ops.gemm_grouped([dict(A=za[:, g * D:(g + 1) * D], W=W1[g], Cact=hid[g], act=1) for g in range(G)], dt)
        groups = []
        for p_lo, p_hi, n_cols in buckets:
            nn = min(_round_up(max(int(n_cols), 1), 32), Cp)
            r0, r1 = p_lo * B, p_hi * B
            f = 0
            for g, grp in enumerate(self.field_groups):
                for j in range(len(grp)):
                    groups.append(dict(A=hid[g][r0:r1], W=W2[g][j * Cp:j * Cp + nn], bias=bias[g][j * Cp:j * Cp + nn], C32=out[r0:r1, f * Cp:f * Cp + nn]))
                    f += 1
        for s0 in range(0, len(groups), N.MAX_GROUPS):
            ops.gemm_grouped(groups[s0:s0 + N.MAX_GROUPS], dt)
        return out.view(P, B, n_fields, Cp).permute(1, 0, 2, 3)[..., :self.n_inp]


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class PointwiseEncode(nn.Module):
    """PointwiseEncode(field_groups, n_inp, MLP_hidden, num_layers, embed_dim, n_heads, max_len, src_len, dropout=0.1)
    .forward(x [B, P, n_fields, n_inp]) -> z [B, P, n_groups, embed_dim]      (inference; reference models/encoder_decoder.py:75-123)

    Device layout: rows m = snapshot * P + patch; fp32 residual stream [M, W], W = n_groups * embed_dim; matrix operands in the compute dtype.
    Attention heads narrower than 8 (the shipped cylinder model: W = 32, 8 heads) are zero-padded to 8 inside the packed q/k/v and
    projection weights — exact, the pad lanes contribute 0 to every score and every output."""

    CHUNK = 2048   # snapshots per pass (workspace ~ M * 4W act elements)

    def __init__(self, field_groups: Sequence[Sequence[int]], n_inp: int, MLP_hidden: int, num_layers: int, embed_dim: int, n_heads: int,
                 max_len: int, src_len: int, dropout: float = 0.1):
        super().__init__()
        self.field_groups = [list(g) for g in field_groups]
        self.num_groups = len(self.field_groups)
        self.n_inp, self.MLP_hidden, self.embed_dim, self.n_heads = n_inp, MLP_hidden, embed_dim, n_heads
        W = self.num_groups * embed_dim
        self.spatial_pos_encoder = PositionalEncoding(W, dropout)
        self.blocks = nn.ModuleList([EncoderBlock(n_heads=n_heads, max_len=max_len, embed_dim=W, src_len=src_len, dropout=dropout) for _ in range(num_layers)])
        self.ln = nn.LayerNorm(W)
        self.apply(self._init_weights)   # before the down-scale MLPs exist, as in the reference (:90-95): they keep torch's default init
        self.encoders = nn.ModuleList([downScaleMLP(d_input=n_inp * len(g), d_model=embed_dim, hidden_dim=MLP_hidden) for g in self.field_groups])
        self.compute_dtype = "fp32"
        self._pack = None
        for g in self.field_groups:
            if g != list(range(g[0], g[0] + len(g))):
                raise NotImplementedError("sea_amd.PointwiseEncode: a field group must be a run of consecutive field indices (column slice of the [M, F*C] input)")
        if W % n_heads or (W // n_heads) % 4 or W % 8 or MLP_hidden % 8 or embed_dim % 4:
            raise NotImplementedError("sea_amd.PointwiseEncode: need n_groups*embed_dim a multiple of 8 and of n_heads, head dim and embed_dim multiples of 4, "
                                      "MLP_hidden a multiple of 8")
        self._n_inp_p = (n_inp + 3) // 4 * 4   # a field's cell columns are padded to 16 bytes inside (zero weight columns: exact)
        if (W // n_heads) > 128:
            raise NotImplementedError("sea_amd.PointwiseEncode: head dim above 128")

    @staticmethod
    def _init_weights(module):
        if isinstance(module, nn.Linear):
            torch.nn.init.normal_(module.weight, mean=0.0, std=0.02)
            if module.bias is not None:
                torch.nn.init.zeros_(module.bias)
        elif isinstance(module, nn.LayerNorm):
            nn.init.constant_(module.bias, 0)
            nn.init.constant_(module.weight, 1.0)

    def set_compute_dtype(self, dtype) -> "PointwiseEncode":
        name = {torch.float32: "fp32", torch.bfloat16: "bf16"}.get(dtype, dtype)
        if name not in ("fp32", "bf16"):
            raise ValueError("compute dtype must be 'fp32' or 'bf16'")
        self.compute_dtype, self._pack = name, None
        return self

    # ------------------------------------------------------------------ packed weights (refreshed when a parameter was written or moved)
    def _packed(self, dt: torch.dtype, dev: torch.device):
        ps = list(self.parameters())
        key = (dt, dev, tuple((p.data_ptr(), p._version) for p in ps))
        if self._pack is not None and self._pack["key"] == key:
            return self._pack
        W, H = self.num_groups * self.embed_dim, self.n_heads
        hd = W // H
        hdp = min(v for v in (8, 16, 32, 64, 128) if v >= hd)   # the attention kernel's head dims; narrower heads are zero-padded (exact)
        Wp = H * hdp
        f32 = torch.float32
        with torch.no_grad():
            conv = lambda t: t.detach().to(device=dev, dtype=dt).contiguous()  # noqa: E731
            enc = []
            C, Cp = self.n_inp, self._n_inp_p
            for g, m in zip(self.field_groups, self.encoders):
                K = len(g) * Cp
                Kp = _round_up(K, 8)
                w1 = torch.zeros(self.MLP_hidden, Kp, device=dev, dtype=dt)
                w1[:, :K].view(self.MLP_hidden, len(g), Cp)[:, :, :C] = m.layer1.weight.detach().to(dt).view(self.MLP_hidden, len(g), C)
                enc.append(dict(K=K, Kp=Kp, col0=g[0] * Cp, W1=w1, W2=conv(m.layer2.weight), b2=m.layer2.bias.detach().to(dev, f32).contiguous()))

            def pad_rows(w):   # [H*hd, ...] -> [H*hdp, ...], head h at rows h*hdp .. h*hdp + hd - 1
                out = torch.zeros((H, hdp) + tuple(w.shape[1:]), device=dev, dtype=w.dtype)
                out[:, :hd] = w.detach().to(dev).view((H, hd) + tuple(w.shape[1:]))
                return out.view((Wp,) + tuple(w.shape[1:]))

            blocks = []
            for b in self.blocks:
                a = b.attn_1
                wqkv = torch.cat([pad_rows(a.q.weight), pad_rows(a.k.weight), pad_rows(a.v.weight)], 0).to(dt).contiguous()
                bqkv = torch.cat([pad_rows(a.q.bias), pad_rows(a.k.bias), pad_rows(a.v.bias)], 0).to(f32).contiguous()
                wo = pad_rows(a.projection.weight.detach().t().contiguous()).t().to(dt).contiguous()      # [W, Wp]: zero columns at the pad lanes
                fc1, ln, _, fc2 = b.mlp_1.layers
                blocks.append(dict(g1=b.ln_exp1_1.weight.detach().to(dev, f32), g2=b.ln_exp1_2.weight.detach().to(dev, f32), wqkv=wqkv, bqkv=bqkv, wo=wo,
                                   w1=conv(fc1.weight), b1=fc1.bias.detach().to(dev, f32), lnw=ln.weight.detach().to(dev, f32), lnb=ln.bias.detach().to(dev, f32),
                                   w2=conv(fc2.weight), b2=fc2.bias.detach().to(dev, f32)))
            fin_w, fin_b = self.ln.weight.detach().to(dev, f32).contiguous(), self.ln.bias.detach().to(dev, f32).contiguous()
        self._pack = dict(key=key, enc=enc, blocks=blocks, hd=hd, hdp=hdp, Wp=Wp, ws={}, fin_w=fin_w, fin_b=fin_b)
        return self._pack

    def _workspace(self, pk, Bc: int, P: int, dt: torch.dtype, dev: torch.device):
        ws = pk["ws"].get((Bc, P))
        if ws is not None:
            return ws
        W, H, hdp, Wp, S = self.num_groups * self.embed_dim, self.n_heads, pk["hdp"], pk["Wp"], 4 * self.num_groups * self.embed_dim
        M, cap = Bc * P, _round_up(P, 8)
        e = lambda *shape, dtype=dt: torch.empty(*shape, device=dev, dtype=dtype)  # noqa: E731
        z = lambda *shape, dtype=dt: torch.zeros(*shape, device=dev, dtype=dtype)  # noqa: E731
        rope = torch.zeros(cap, hdp // 2, 2, device=dev, dtype=torch.float32)
        rope[..., 0] = 1.0   # (cos, sin) = (1, 0): the projection kernel's rotation is the identity — this attention has no positional rotation
        ws = dict(A=[z(M, g["Kp"]) for g in pk["enc"]], hid=[e(M, self.MLP_hidden) for _ in pk["enc"]], zr=e(M, W, dtype=torch.float32), n=e(M, W),
                  Q=e(Bc, H, P, hdp), K=z(Bc, H, cap, hdp), Vt=z(Bc, H, hdp, cap), att=e(Bc, P, Wp), h=e(M, S), hg=e(M, S), rope=rope,
                  pe=self.spatial_pos_encoder.pe[0, :P].to(dev, torch.float32).repeat(Bc, 1).contiguous(), cap=cap)
        pk["ws"] = {(Bc, P): ws}   # one shape kept: a new chunk shape replaces it
        return ws

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        N.require_gpu(x, "PointwiseEncode input")
        if torch.is_grad_enabled() and x.requires_grad:
            raise NotImplementedError("sea_amd.PointwiseEncode: inference only")
        B, P, F, C = x.shape
        assert F == sum(len(g) for g in self.field_groups) and C == self.n_inp, (x.shape, self.field_groups, self.n_inp)
        if P > self.spatial_pos_encoder.pe.shape[1]:
            raise ValueError(f"{P} patches exceed the positional table ({self.spatial_pos_encoder.pe.shape[1]})")
        dt = torch.float32 if self.compute_dtype == "fp32" else torch.bfloat16
        dev = x.device
        pk = self._packed(dt, dev)
        W, G, H, hdp = self.num_groups * self.embed_dim, self.num_groups, self.n_heads, pk["hdp"]
        xf = x.detach().to(torch.float32)
        if self._n_inp_p != C:   # pad each field's cell columns to 16 bytes (one copy; sea_patchify can write this layout directly: c_out)
            xf = torch.nn.functional.pad(xf, (0, self._n_inp_p - C))
        xf = xf.contiguous().view(B * P, F * self._n_inp_p)
        out = torch.empty(B * P, W, device=dev, dtype=torch.float32)
        x_is_act = dt != torch.float32
        for b0 in range(0, B, self.CHUNK):
            Bc = min(self.CHUNK, B - b0)
            M = Bc * P
            ws = self._workspace(pk, Bc, P, dt, dev)
            rows = slice(b0 * P, b0 * P + M)
            # down-scale MLPs of all groups: (compute-dtype copy of the group's columns) -> Linear + GELU -> Linear + bias + positions
            for g, A in zip(pk["enc"], ws["A"]):
                ops.convert(xf[rows, g["col0"]:g["col0"] + g["K"]], A[:, :g["K"]])
            ops.gemm_grouped([dict(A=A, W=g["W1"], Cact=hid, act=1) for g, A, hid in zip(pk["enc"], ws["A"], ws["hid"])], dt)
            E = self.embed_dim
            ops.gemm_grouped([dict(A=hid, W=g["W2"], bias=g["b2"], R=ws["pe"][:, i * E:(i + 1) * E], C32=ws["zr"][:, i * E:(i + 1) * E])
                              for i, (g, hid) in enumerate(zip(pk["enc"], ws["hid"]))], dt)
            zr = ws["zr"]
            for blk in pk["blocks"]:
                ops.rownorm([dict(X=zr, gamma=blk["g1"], Yact=ws["n"])], M, W, False, False, 1e-5, dt)
                ops.qkv_rope_grouped([dict(A=ws["n"], W=blk["wqkv"], bias=blk["bqkv"], col0=0, Q=ws["Q"], K=ws["K"], Vt=ws["Vt"])], ws["rope"], H, hdp, P, 0,
                                     ws["cap"], ops.q_scale(pk["hd"]), dt)
                ops.attention_fwd([dict(Q=ws["Q"], K=ws["K"], Vt=ws["Vt"], O=ws["att"])], Bc, H, hdp, P, P, ws["cap"], 0, P, dt)   # src_len = P: every key visible
                ops.gemm_grouped([dict(A=ws["att"].view(M, -1), W=blk["wo"], R=zr, C32=zr)], dt)
                ops.rownorm([dict(X=zr, gamma=blk["g2"], Yact=ws["n"])], M, W, False, False, 1e-5, dt)
                ops.gemm_grouped([dict(A=ws["n"], W=blk["w1"], bias=blk["b1"], Cact=ws["h"])], dt)
                ops.rownorm([dict(X=ws["h"], gamma=blk["lnw"], beta=blk["lnb"], Yact=ws["hg"])], M, ws["h"].shape[1], x_is_act, True, 1e-5, dt)
                ops.gemm_grouped([dict(A=ws["hg"], W=blk["w2"], bias=blk["b2"], R=zr, C32=zr)], dt)
            ops.rownorm([dict(X=zr, gamma=pk["fin_w"], beta=pk["fin_b"], Y32=out[rows])], M, W, False, False, self.ln.eps, dt)
        return out.view(B, P, G, self.embed_dim)


class SpatialModel(nn.Module):
    """SpatialModel(field_groups, n_inp, MLP_hidden, num_layers, embed_dim, n_heads, max_len, src_len, dropout=0.1, variational=False): encoder +
    decoder under the reference's attribute names `encode` / `decode` (models/encoder_decoder.py:148-176), inference only; the variational encoder
    (`Encode`, sampling) is not provided."""

    def __init__(self, field_groups, n_inp, MLP_hidden, num_layers, embed_dim, n_heads, max_len, src_len, dropout=0.1, variational=False):
        super().__init__()
        if variational:
            raise NotImplementedError("sea_amd.SpatialModel: the variational encoder is outside the accelerated path (both shipped configs use variational=False)")
        self.variational = False
        self.encode = PointwiseEncode(field_groups, n_inp, MLP_hidden, num_layers, embed_dim, n_heads, max_len, src_len, dropout)
        self.decode = Decode(field_groups, n_inp, MLP_hidden, embed_dim, dropout)

    def set_compute_dtype(self, dtype) -> "SpatialModel":
        self.encode.set_compute_dtype(dtype)
        self.decode.set_compute_dtype(dtype)
        return self

    def generate_padding_mask(self, x, pad_idx=-9999):
        """In place, as the reference (:171-174): entries equal to pad_idx become 0."""
        x[x == pad_idx] = 0.0
        return x

    def forward(self, x):
        x = self.generate_padding_mask(x)
        return self.decode(self.encode(x))

