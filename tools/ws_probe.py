"""Development aid: in-kernel stamps of gemm_ws.hip's iteration (workgroup 0, iterations 8 .. 15) on cfg3's mlp.fc1 shape:  SEA_TUNE=gemm_ws_probe=2 python tools/ws_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sea_amd import ops
dev = torch.device("cuda:0"); bf = torch.bfloat16
M, N, K = 16192, 2048, int(sys.argv[1]) if len(sys.argv) > 1 else 256
gs = [dict(A=torch.randn(M, K, device=dev).to(bf), W=(torch.randn(N, K, device=dev) * 0.05).to(bf), Cact=torch.empty(M, N, device=dev, dtype=bf)) for _ in range(3)]
os.environ["SEA_TUNE"] = "gemm_ws_probe=1"
for _ in range(5):
    ops.gemm_grouped(gs, bf)
torch.cuda.synchronize()
os.environ["SEA_TUNE"] = "gemm_ws_probe=2"
ops.gemm_grouped(gs, bf)
torch.cuda.synchronize()
