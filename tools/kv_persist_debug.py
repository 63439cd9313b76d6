"""Per-step difference between the persistent and the seven-launch forms of sea_kv_rollout (debug aid)."""
import os
import subprocess
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sea_oracle as O  # noqa: E402
from oracle.recipe import recipe_inputs  # noqa: E402
from tests.test_model_gpu import build  # noqa: E402


def main():
    dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    mode = os.environ.get("SEA_KV_PERSIST", "1")
    cfg = O.OracleConfig(1, 256, 8, 160, 8, 0, 3, 2, True, "adaln")
    m = build(cfg, dtype)
    x, _, ib = recipe_inputs(1, n, cfg, seed=5)
    out = m.engine().rollout_kv(x[:, :1].cuda().contiguous(), ib.cuda().contiguous(), n)
    np.save(f"gpurun_out/kvp_{dtype}_{mode}.npy", out.cpu().numpy())


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "cmp":
        a, b = np.load(f"gpurun_out/kvp_{sys.argv[2]}_1.npy"), np.load(f"gpurun_out/kvp_{sys.argv[2]}_0.npy")
        for s in range(a.shape[1]):
            d = np.abs(a[0, s] - b[0, s])
            print(s, "max abs diff %.3e" % d.max(), "per field", ["%.2e" % d[f].max() for f in range(a.shape[2])])
    else:
        main()
