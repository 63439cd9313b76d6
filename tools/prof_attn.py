import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.bench_ops import attn
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
attn(B, 8, 32, 2024, 3)
