"""Gather (by mesh point) against scatter (by cell slot) form of sea_unpatchify on the decode bench's mesh — development aid."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sea_amd import ops
from sea_amd.utils.data_processors import DataPartitioner2D
from tools.bench_ops import timeit

dev = torch.device("cuda:0")


def main():
    gen = torch.Generator().manual_seed(7)
    n_points, T = 30000, 2024
    px = torch.rand(n_points, generator=gen); px[: n_points // 3] = 0.25 + 0.1 * torch.rand(n_points // 3, generator=gen)
    py = torch.rand(n_points, generator=gen)
    part = DataPartitioner2D(px, py, m=9, n=9, device=dev)
    P, C = part.padded_index_map.shape
    Cp = (C + 31) // 32 * 32
    cells = torch.randn(T, P, 3, Cp, device=dev)[..., :C]
    scale, shift = torch.ones(3, device=dev), torch.zeros(3, device=dev)
    for name, ps in (("gather by point", part.point_slot), ("scatter by slot", None)):
        us = timeit(lambda: ops.unpatchify(cells, "BPFC", part.padded_index_map, scale, shift, n_points, point_slot=ps), iters=5, reps=3)
        print(f"{name:18s} {us:8.1f} us", flush=True)
    # mesh points numbered cell by cell (what a spatially sorted mesh looks like): the gather then reads runs of consecutive slots
    order = torch.argsort(part.point_slot)
    inv = torch.empty_like(order); inv[order] = torch.arange(n_points, device=dev)
    part2 = DataPartitioner2D(px[order.cpu()], py[order.cpu()], m=9, n=9, device=dev)
    us = timeit(lambda: ops.unpatchify(cells, "BPFC", part2.padded_index_map, scale, shift, n_points, point_slot=part2.point_slot), iters=5, reps=3)
    print(f"{'gather, mesh points sorted by cell':18s} {us:8.1f} us", flush=True)


if __name__ == "__main__":
    main()
