"""Mesh-side consumers of the rollout: the reference's 2-D mesh partitioner, MinMax scaler and the un-patchify / inverse
scaling step of `MeshProcessor` (utils/data_processors.py:9-111, 225-290, 553-573) with the same names and argument meaning.

What is native: `MeshUnpatcher.inverse_scale_and_unpatch` — one launch of sea_unpatchify (scatter by the partition's index map fused with
the inverse MinMax transform and with the [B,P,F,C] -> [B,P,C,F] permute).  The index map itself is built once per mesh with a handful of
torch calls on the device (bucketize + sort), not per time step, and is plumbing; the forward patchify/scale direction is provided for
round-trip tests with torch indexing."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from .. import _native as N
from .. import ops


class MinMaxScaler:
    """MinMaxScaler(feature_range).fit / transform / inverse_transform (reference utils/data_processors.py:225-273), without the file I/O."""

    def __init__(self, feature_range=(-1, 1), name: str = "scaler"):
        self.feature_range, self.name = feature_range, name
        self.min_val = self.max_val = None

    def fit(self, data: torch.Tensor) -> None:
        self.min_val, self.max_val = torch.min(data), torch.max(data)
        if self.min_val == self.max_val:
            raise ValueError("Data has zero variance")

    def transform(self, data: torch.Tensor) -> torch.Tensor:
        if self.min_val is None:
            raise ValueError("The scaler has not been fitted yet. Call 'fit' with training data before using 'transform'.")
        std = (data - self.min_val) / (self.max_val - self.min_val)
        return std * (self.feature_range[1] - self.feature_range[0]) + self.feature_range[0]

    def forward_affine(self) -> Tuple[float, float]:
        """(scale, shift) of transform as x * scale + shift."""
        if self.min_val is None:
            raise ValueError("The scaler has not been fitted yet. Call 'fit' with training data before using 'transform'.")
        r0, r1 = float(self.feature_range[0]), float(self.feature_range[1])
        scale = (r1 - r0) / (float(self.max_val) - float(self.min_val))
        return scale, r0 - float(self.min_val) * scale

    def inverse_affine(self) -> Tuple[float, float]:
        """(scale, shift) of inverse_transform as y * scale + shift."""
        if self.min_val is None:
            raise ValueError("The scaler has not been fitted yet.")
        r0, r1 = float(self.feature_range[0]), float(self.feature_range[1])
        scale = (float(self.max_val) - float(self.min_val)) / (r1 - r0)
        return scale, float(self.min_val) - r0 * scale


class DataPartitioner2D:
    """Same constructor as the reference (utils/data_processors.py:9-19).  `padded_index_map` is one int32 tensor [P, C] (pad_id where a
    cell has fewer than C points), cell order (i, j) row-major over the (m-1) x (n-1) grid, points of a cell in ascending point index —
    exactly the reference's list of per-cell index tensors after pad_partitions."""

    def __init__(self, x_coords, y_coords, m=9, n=9, pad_id=-1, pad_field_value=0, device="cuda"):
        self.device = torch.device(device)
        self.x_coords = x_coords.to(self.device).float()
        self.y_coords = y_coords.to(self.device).float()
        self.full_coords = torch.stack((self.x_coords, self.y_coords), dim=1)
        self.m, self.n, self.pad_id, self.pad_field_value = m, n, pad_id, pad_field_value
        self.padded_index_map = self._build_index_map()

    def _build_index_map(self) -> torch.Tensor:
        x, y = self.x_coords, self.y_coords
        xb = torch.linspace(float(x.min()), float(x.max()), self.m, device=self.device)
        yb = torch.linspace(float(y.min()), float(y.max()), self.n, device=self.device)
        xi = torch.bucketize(x, xb, right=True).clamp_(1, self.m - 1)
        yi = torch.bucketize(y, yb, right=True).clamp_(1, self.n - 1)
        cell = (xi - 1) * (self.n - 1) + (yi - 1)                      # reference loop order: i outer, j inner
        P = (self.m - 1) * (self.n - 1)
        order = torch.sort(cell, stable=True).indices                  # points of a cell stay in ascending index order
        counts = torch.bincount(cell, minlength=P)
        C = int(counts.max())
        start = torch.cumsum(counts, 0) - counts
        rank = torch.arange(cell.numel(), device=self.device) - start[cell[order]]
        imap = torch.full((P, C), self.pad_id, dtype=torch.int32, device=self.device)
        imap[cell[order], rank] = order.to(torch.int32)
        # inverse: the (cell, slot) of every point, as one flat index p * C + c (the gather form of sea_unpatchify)
        self.point_slot = torch.empty(cell.numel(), dtype=torch.int32, device=self.device)
        self.point_slot[order] = (cell[order] * C + rank).to(torch.int32)
        return imap

    def create_partitions(self, vars: Sequence[torch.Tensor]) -> torch.Tensor:
        """vars: list of [T, N] tensors (one per field) -> stacked padded fields [T, P, C, F] (the reference returns the same data as a
        list of per-cell (coords, fields) tuples which its caller stacks, utils/data_processors.py:519-525)."""
        fields = torch.stack([v.to(self.device).float() for v in vars], dim=2)        # [T, N, F]
        idx = self.padded_index_map.long()
        out = fields[:, idx.clamp_min(0).view(-1), :].view(fields.shape[0], idx.shape[0], idx.shape[1], -1)
        return torch.where((idx >= 0)[None, :, :, None], out, torch.full_like(out, float(self.pad_field_value)))


class MeshUnpatcher:
    """The inverse leg of the reference's MeshProcessor (inverse_scale_and_unpatch, utils/data_processors.py:553-573): partitioner +
    one scaler per field group."""

    def __init__(self, partitioner: DataPartitioner2D, field_groups: Sequence[Sequence[int]], scalers: Sequence[MinMaxScaler] = (), gather: bool = True):
        self.partitioner, self.field_groups, self.scalers, self.gather = partitioner, [list(g) for g in field_groups], list(scalers), gather
        F = sum(len(g) for g in self.field_groups)
        scale, shift = [1.0] * F, [0.0] * F
        for g, sc in zip(self.field_groups, self.scalers):
            a, b = sc.inverse_affine()
            for f in g:
                scale[f], shift[f] = a, b
        dev = partitioner.device
        self._scale = torch.tensor(scale, device=dev, dtype=torch.float32)
        self._shift = torch.tensor(shift, device=dev, dtype=torch.float32)

    def patchify_and_scale(self, data: torch.Tensor, layout: str = "BPCF", c_out: Optional[int] = None) -> torch.Tensor:
        """The forward leg (reference MeshProcessor.patchify_and_scale, utils/data_processors.py:484-526, with fitted scalers): data [T, N, F] ->
        scaled, partitioned, padded fields [T, P, C, F] (layout "BPCF", the reference's) or [T, P, F, c_out] ("BPFC", what the encoder reads) in ONE
        launch of sea_patchify; padded slots hold pad_field_value, unscaled, as in the reference."""
        N.require_gpu(data, "data")
        if not hasattr(self, "_fscale"):
            F = sum(len(g) for g in self.field_groups)
            scale, shift = [1.0] * F, [0.0] * F
            for g, sc in zip(self.field_groups, self.scalers):
                a, b = sc.forward_affine()
                for f in g:
                    scale[f], shift[f] = a, b
            dev = self.partitioner.device
            self._fscale = torch.tensor(scale, device=dev, dtype=torch.float32)
            self._fshift = torch.tensor(shift, device=dev, dtype=torch.float32)
        return ops.patchify(data.float().contiguous(), self.partitioner.padded_index_map, self._fscale, self._fshift, layout, c_out,
                            float(self.partitioner.pad_field_value))

    def inverse_scale_and_unpatch(self, scaled_fields: torch.Tensor, layout: str = "BPCF") -> torch.Tensor:
        """scaled_fields [T, P, C, F] (reference layout) or, with layout="BPFC", the decoder's [T, P, F, C] output directly -> [T, N, F]."""
        N.require_gpu(scaled_fields, "scaled_fields")
        return ops.unpatchify(scaled_fields.float(), layout, self.partitioner.padded_index_map, self._scale, self._shift, self.partitioner.x_coords.numel(),
                              point_slot=self.partitioner.point_slot if self.gather else None)


class TemporalDataset:
    """Windows over encoded trajectories with the reference's indexing (utils/data_processors.py:388-452): sample idx of segment s covers steps
    [k*step + shift, k*step + shift + src_len), the target is the same window one step later.  Items are VIEWS of the (device-resident) trajectory
    tensors — no copies, no host round trip; `batch()` stacks a list of samples for the train step."""

    def __init__(self, data_list, data_list_original, field_ib, src_len=64, overlap=0, device='cpu', time_shifting_flag=False):
        self.device, self.data_list, self.data_list_original, self.field_ib = device, data_list, data_list_original, field_ib
        self.src_len, self.overlap, self.step, self.time_shifting_flag = src_len, overlap, src_len - overlap, time_shifting_flag
        self.segment_samples = [d.shape[0] // self.step for d in data_list]
        self.num_samples = sum(self.segment_samples)

    def __len__(self):
        return self.num_samples

    def locate(self, idx: int) -> Tuple[int, int]:
        """(segment, sample inside the segment) of a flat index; IndexError past the end, as the reference."""
        cum = 0
        for s, n in enumerate(self.segment_samples):
            if idx < cum + n:
                return s, idx - cum
            cum += n
        raise IndexError("Index out of range")

    def __getitem__(self, idx):
        import numpy as np

        seg, k = self.locate(idx)
        shift = int(np.random.randint(0, self.data_list[seg].shape[0] - self.step)) if self.time_shifting_flag else 0
        a, b = k * self.step + shift, k * self.step + shift + self.src_len
        return self.data_list[seg][a:b], self.data_list[seg][a + 1:b + 1], self.data_list_original[seg][a + 1:b + 1], self.field_ib[seg][a:b]

    def batch(self, indices: Sequence[int]):
        items = [self[i] for i in indices]
        return tuple(torch.stack([it[j] for it in items]) for j in range(4))

