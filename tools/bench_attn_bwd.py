"""Micro-benchmark of the attention backward pair (development aid): python tools/bench_attn_bwd.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sea_amd import ops
from tools.bench_ops import timeit, dev

def bwd(B, H, hd, T, n, dt=torch.bfloat16):
    cap = (T + 7) // 8 * 8
    E = H * hd
    probs, bprobs = [], []
    for _ in range(n):
        Q = (torch.randn(B, H, T, hd, device=dev) * hd ** -0.25).to(dt)
        K = (torch.randn(B, H, cap, hd, device=dev) * hd ** -0.25).to(dt)
        V = torch.randn(B, H, cap, hd, device=dev).to(dt)
        O = torch.empty(B, T, E, device=dev, dtype=dt)
        LSE = torch.empty(B, H, T, device=dev)
        probs.append(dict(Q=Q, K=K, Vt=V.transpose(2, 3).contiguous(), O=O, LSE=LSE))
        bprobs.append(dict(Q=Q, K=K, V=V, O=O, dO=torch.randn(B, T, E, device=dev).to(dt), LSE=LSE, delta=torch.empty(B, H, T, device=dev),
                           dQ=torch.empty(B * T, E, device=dev, dtype=dt), dK=torch.empty(B * T, E, device=dev, dtype=dt), dV=torch.empty(B * T, E, device=dev, dtype=dt)))
    rope = torch.zeros(cap, hd // 2, 2, device=dev); rope[..., 0] = 1
    ops.attention_fwd(probs, B, H, hd, T, T, cap, 0, 0, dt)
    us = timeit(lambda: ops.attention_bwd(bprobs, rope, B, H, hd, T, T, cap, 0, 0, ops.q_scale(hd), dt), iters=10)
    fl = n * 10 * B * H * (T * (T + 1) // 2) * hd
    print(f"attn bwd B={B} H={H} hd={hd} T={T} nprob={n}: {us:8.1f} us  {fl / us / 1e6:8.1f} TFLOP/s", flush=True)

if __name__ == "__main__":
    for a in [(8, 8, 32, 2024, 3), (8, 8, 16, 2024, 2), (1, 8, 32, 2024, 3), (1, 8, 16, 2024, 2)]:
        bwd(*a)
