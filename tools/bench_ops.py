"""Micro-benchmarks of single kernels (development aid): python tools/bench_ops.py attn|gemm ..."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sea_amd import ops

dev = torch.device("cuda:0")

def timeit(fn, iters=20, warm=3, reps=5):
    """GPU time per call with the calls captured in one HIP graph (no host launch cost between them)."""
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best  # us

def attn(B, H, hd, T, nprob, dtype=torch.bfloat16):
    cap = (T + 7) // 8 * 8
    probs = []
    for _ in range(nprob):
        Q = torch.randn(B, H, T, hd, device=dev).to(dtype) * hd ** -0.25
        K = torch.randn(B, H, cap, hd, device=dev).to(dtype) * hd ** -0.25
        Vt = torch.randn(B, H, hd, cap, device=dev).to(dtype)
        O = torch.empty(B, T, H * hd, device=dev, dtype=dtype)
        probs.append(dict(Q=Q, K=K, Vt=Vt, O=O))
    us = timeit(lambda: ops.attention_fwd(probs, B, H, hd, T, T, cap, 0, 0, dtype))
    fl = nprob * 4 * B * H * (T * (T + 1) // 2) * hd
    print(f"attn B={B} H={H} hd={hd} T={T} nprob={nprob}: {us:8.1f} us  {fl / us / 1e6:8.1f} TFLOP/s", flush=True)

def gemm(M, N, K, ngroups, dtype=torch.bfloat16, **kw):
    gs = []
    for _ in range(ngroups):
        A = torch.randn(M, K, device=dev).to(dtype); W = torch.randn(N, K, device=dev).to(dtype)
        gs.append(dict(A=A, W=W, Cact=torch.empty(M, N, device=dev, dtype=dtype), **kw))
    us = timeit(lambda: ops.gemm_grouped(gs, dtype))
    print(f"gemm M={M} N={N} K={K} groups={ngroups}: {us:8.1f} us  {ngroups * 2 * M * N * K / us / 1e6:8.1f} TFLOP/s", flush=True)

if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("attn", "all"):
        for (B, H, hd, T, n) in [(1, 8, 32, 2024, 3), (1, 8, 32, 2024, 1), (1, 8, 32, 1024, 3), (1, 8, 32, 512, 3), (8, 8, 32, 2024, 3), (1, 8, 16, 2024, 2), (8, 8, 16, 2024, 2), (1, 8, 32, 64, 3)]:
            attn(B, H, hd, T, n)
    if what == "biggemm":
        for (M, N, K, g) in [(129536, 3328, 512, 1), (129536, 3328, 448, 1), (16192, 2048, 256, 3), (16192, 512, 512, 9)]:
            gemm(M, N, K, g)
    if what in ("gemm", "all"):
        for (M, N, K, g) in [(2024, 2048, 256, 3), (2024, 256, 2048, 3), (2024, 256, 256, 3), (2024, 768, 256, 3), (16192, 2048, 256, 3), (16192, 256, 2048, 3), (4096, 4096, 4096, 1), (2024, 512, 512, 9), (64, 256, 256, 1)]:
            gemm(M, N, K, g)
