"""Development aid: what a plain streaming WRITE of a GEMM-output-sized buffer costs on this device (torch fill / copy kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.bench_ops import timeit
dev = torch.device("cuda:0")
for mb in (25, 75, 149, 199, 398, 796):
    n = mb * 1000 * 1000 // 2
    x = torch.empty(n, device=dev, dtype=torch.bfloat16)
    y = torch.randn(n, device=dev).to(torch.bfloat16)
    t_fill = timeit(lambda: x.zero_())
    t_copy = timeit(lambda: x.copy_(y))
    t_read = timeit(lambda: y.sum())
    print(f"{mb:4d} MB: fill {t_fill:6.1f} us = {mb / t_fill / 1e3 * 1e3:5.2f} GB/ms ({mb / t_fill:5.2f} TB/s)   copy {t_copy:6.1f} us ({mb / t_copy:5.2f} TB/s written)   sum-read {t_read:6.1f} us ({mb / t_read:5.2f} TB/s)", flush=True)
