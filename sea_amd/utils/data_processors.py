"""Mesh-side consumers of the rollout: the reference's 2-D mesh partitioner, MinMax scaler and the un-patchify / inverse
scaling step of `MeshProcessor` (utils/data_processors.py:9-111, 225-290, 553-573) with the same names and argument meaning.

What is native: `MeshUnpatcher.inverse_scale_and_unpatch` — one launch of sea_unpatchify (scatter by the partition's index map fused with
the inverse MinMax transform and with the [B,P,F,C] -> [B,P,C,F] permute).  The index map itself is built once per mesh with a handful of
torch calls on the device (bucketize + sort), not per time step, and is plumbing; the forward patchify/scale direction is provided for
round-trip tests with torch indexing."""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

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


    # ------------------------------------------------------------------ decoder + un-patchify without the padding
    def _prefix_plan(self, max_buckets: int):
        """Patches sorted by how many mesh points their cell holds, cut into <= max_buckets runs that minimise sum(run length x longest cell of the run) —
        the columns the decoder has to produce; the index map and the point -> slot table in that patch order."""
        key = max_buckets
        plan = getattr(self, "_pplan", None)
        if plan is None or plan[0] != key:
            part = self.partitioner
            idx = part.padded_index_map
            P, C = idx.shape
            counts = (idx != part.pad_id).sum(dim=1).cpu().tolist()       # a cell's valid slots are a prefix of its row (DataPartitioner2D pads at the end)
            order = sorted(range(P), key=lambda q: -counts[q])
            cs = [counts[q] for q in order]                                # non-increasing
            K = max(1, min(max_buckets, P))
            INF = float("inf")
            cost = [[INF] * (P + 1) for _ in range(K + 1)]                 # cost[k][i]: the first i patches in k runs
            back = [[0] * (P + 1) for _ in range(K + 1)]
            cost[0][0] = 0
            for k in range(1, K + 1):
                for i in range(1, P + 1):
                    for j in range(k - 1, i):                             # the last run is j .. i-1: its longest cell is cs[j]
                        c = cost[k - 1][j] + (i - j) * cs[j]
                        if c < cost[k][i]:
                            cost[k][i], back[k][i] = c, j
            k_best = min(range(1, K + 1), key=lambda k: cost[k][P])
            cuts, i = [], P
            for k in range(k_best, 0, -1):
                j = back[k][i]
                cuts.append((j, i, cs[j]))
                i = j
            buckets = cuts[::-1]
            order_t = torch.tensor(order, device=idx.device, dtype=torch.long)
            rank = torch.empty(P, dtype=torch.long, device=idx.device)
            rank[order_t] = torch.arange(P, device=idx.device)
            ps = part.point_slot.long()
            slot_sorted = (rank[ps // C] * C + ps % C).to(torch.int32).contiguous()
            plan = self._pplan = (key, order_t, buckets, idx[order_t].contiguous(), slot_sorted)
        return plan[1:]

    def decode_and_unpatch(self, decoder, z: torch.Tensor) -> torch.Tensor:
        """decoder(z) followed by inverse_scale_and_unpatch(..., layout="BPFC") — z [T, P, n_groups, embed_dim] -> [T, N, F] — with the decoder run only over
        the columns the un-patchify reads: a patch's first `count(cell)` columns per field (Decode.forward_prefix; the rest of a padded cell is never
        produced, so the result equals the two-call chain bit for bit where it is defined — everywhere)."""
        N.require_gpu(z, "z")
        n_fields = sum(len(g) for g in self.field_groups)
        order, buckets, idx_sorted, slot_sorted = self._prefix_plan(max(1, N.MAX_GROUPS // n_fields))
        cells = decoder.forward_prefix(z[:, order], buckets)
        C = idx_sorted.shape[1]
        return ops.unpatchify(cells[..., :C], "BPFC", idx_sorted, self._scale, self._shift, self.partitioner.x_coords.numel(), point_slot=slot_sorted)


class MeshProcessor:
    """MeshProcessor(config, coordinates) of the reference (utils/data_processors.py:454-573) on the device: `patchify_and_scale` (scale, partition into
    the (m-1) x (n-1) cells, pad) and `inverse_scale_and_unpatch` (scatter back, inverse scaling), ONE launch each (sea_patchify / sea_unpatchify).
    config keys as the reference: 'dimension' ('2D'; the 3-D partitioner is out of scope), 'field_groups', 'scale_feature_range' (None in both shipped
    configs; a (lo, hi) pair builds one MinMaxScaler per field group — the reference's own constructor call at :481 passes a dict and cannot run),
    'm', 'n', 'pad_id', 'pad_field_value'.  coordinates: [2, N]."""

    def __init__(self, config, coordinates, device="cuda"):
        self.config = config
        self.dimension = config.get('dimension', '3D')
        if 'field_groups' not in config:
            raise ValueError("'field_groups' must be specified in the config dictionary")
        if self.dimension != '2D':
            raise NotImplementedError("sea_amd.MeshProcessor: only the 2-D partitioner is built (both shipped configs are 2-D)")
        self.field_groups = [list(g) for g in config['field_groups']]
        self.scale_feature_range = config.get('scale_feature_range')
        self.coordinates = coordinates
        self.device = torch.device(device)
        self.scalers = [MinMaxScaler(tuple(self.scale_feature_range), name=f"{config.get('csv_scale_name', 'scaler')}-group{i}")
                        for i in range(len(self.field_groups))] if self.scale_feature_range is not None else []
        self.partitioner: Optional[DataPartitioner2D] = None
        self._unpatcher: Optional[MeshUnpatcher] = None

    def patchify_and_scale(self, data: torch.Tensor, train_indices=None):
        """data [T, N, F] -> (stacked_coords [1, P, C, 2], fields [T, P, C, F]); the scalers are fitted on `data` when train_indices is given
        (reference :484-526: it fits on everything it is handed, whatever the indices say)."""
        data = data.to(self.device).float()
        if self.scalers:
            if train_indices is None and any(sc.min_val is None for sc in self.scalers):
                raise ValueError("No saved scaler values found and train_indices is None.")
            if train_indices is not None:
                for sc, g in zip(self.scalers, self.field_groups):
                    sc.fit(data[:, :, g])
        self.partitioner = DataPartitioner2D(self.coordinates[0], self.coordinates[1], m=self.config['m'], n=self.config['n'], pad_id=self.config.get('pad_id', -1),
                                             pad_field_value=self.config.get('pad_field_value', 0), device=self.device)
        self._unpatcher = MeshUnpatcher(self.partitioner, self.field_groups, self.scalers)
        fields = self._unpatcher.patchify_and_scale(data, layout="BPCF")
        idx = self.partitioner.padded_index_map.long()
        coords = self.partitioner.full_coords[idx.clamp_min(0)] * (idx >= 0)[..., None]
        self.stacked_coords = coords[None]
        return self.stacked_coords, fields

    def inverse_scale_and_unpatch(self, scaled_fields: torch.Tensor, layout: str = "BPCF") -> torch.Tensor:
        """[T, P, C, F] (the reference's argument; layout="BPFC" reads the decoder's [T, P, F, C] output in place) -> [T, N, F]."""
        if self._unpatcher is None:
            raise ValueError("call patchify_and_scale first (it builds the partition)")
        return self._unpatcher.inverse_scale_and_unpatch(scaled_fields.to(self.device), layout=layout)

    def decode_and_unpatch(self, decoder, z: torch.Tensor) -> torch.Tensor:
        """decoder(z) + inverse_scale_and_unpatch(..., layout="BPFC") in one go, the decoder run only over a cell's mesh points (MeshUnpatcher.decode_and_unpatch):
        z [T, P, n_groups, embed_dim] -> [T, N, F]."""
        if self._unpatcher is None:
            raise ValueError("call patchify_and_scale first (it builds the partition)")
        return self._unpatcher.decode_and_unpatch(decoder, z.to(self.device))


class ProcessData:
    """ProcessData(n_inp, config) of the reference (utils/data_processors.py:291-373): the frozen spatial autoencoder around the temporal model —
    `initialize_and_process_data` / `process_data` encode padded patches [B, P, F, C] -> [B, P, G, D], `decode_data` decodes [B, P, G, D] ->
    [B, P, n_fields, n_inp].  The model is built and loaded ONCE (the reference rebuilds it and re-reads the checkpoint on every call) and stays on the
    device; inputs and outputs stay there too."""

    def __init__(self, n_inp, config, device=None):
        self.config, self.n_inp = config, n_inp
        self.model_path = config['encoder_decoder_path']
        self.batch_size = config.get('spatial_batch_size', 1000)
        self.device = torch.device(device if device is not None else config.get('device', 'cuda'))
        self.embed_dim = config['embed_dim_spatial']
        if config['dimension'] == '3D':
            self.P = (config['m'] - 1) * (config['n'] - 1) * (config['k'] - 1)
        else:
            self.P = (config['m'] - 1) * (config['n'] - 1)
        self.model_spatial = None

    def initialize_spatial_model(self):
        from ..models.encoder_decoder import SpatialModel

        c = self.config
        m = SpatialModel(field_groups=c['field_groups'], n_inp=self.n_inp, MLP_hidden=c['MLP_hidden_spatial'], num_layers=c['num_layers_spatial'],
                         embed_dim=c['embed_dim_spatial'], n_heads=c['n_heads_spatial'], max_len=c['block_size_spatial'], src_len=c['src_len_spatial'],
                         variational=c['variational_spatial'], dropout=c['dropout_spatial'])
        if 'dtype_spatial' in c:
            m.set_compute_dtype(c['dtype_spatial'])
        return m.to(self.device)

    def load_model(self):
        sd = self.model_path if isinstance(self.model_path, dict) else torch.load(self.model_path, map_location='cpu')
        self.model_spatial.load_state_dict({k.replace("module.", ""): v for k, v in sd.items()})
        self.model_spatial.eval()

    def _model(self):
        if self.model_spatial is None:
            self.model_spatial = self.initialize_spatial_model()
            self.load_model()
        return self.model_spatial

    def initialize_and_process_data(self, data):
        if isinstance(data, torch.Tensor):
            data = [data[i:i + self.batch_size] for i in range(0, data.shape[0], self.batch_size)]
        return self.process_data(data)

    def process_data(self, dataloader):
        m = self._model()
        out = []
        with torch.no_grad():
            for data in dataloader:
                data = m.generate_padding_mask(data.to(self.device).float().clone())
                out.append(m.encode(data))
        return torch.cat(out, dim=0)

    def decode_data(self, data):
        with torch.no_grad():
            return self._model().decode(data.to(self.device))

    def decoder(self):
        """The frozen spatial decoder module (sea_amd.Decode) — for MeshProcessor.decode_and_unpatch."""
        return self._model().decode


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

