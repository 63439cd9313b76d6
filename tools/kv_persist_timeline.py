"""Timeline of one step of the persistent KV rollout (sea_kv_rollout, one launch): 100 MHz clock stamps of the hand-offs, averaged over steps.
Usage (on the MI355X box): python tools/kv_persist_timeline.py [n_steps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sea_oracle as O  # noqa: E402
from oracle.recipe import recipe_inputs  # noqa: E402
from sea_amd import _native as N  # noqa: E402
from tests.test_model_gpu import build  # noqa: E402

NAMES = {(0, 0): "self f0 h0: x in LDS", (0, 1): "self: q/k/v projected, row appended", (0, 2): "self: head output published",
         (1, 0): "oproj f0: att gathered", (1, 1): "oproj f0: x', nd_old published",
         (2, 0): "cross pair 0: nd gathered", (2, 1): "cross pair 0: published", (3, 0): "cross last pair: nd gathered", (3, 1): "cross last pair: published",
         (4, 0): "tail f0: first oc arrived", (4, 2): "tail f0: nd_new published", (4, 3): "tail f0: x'' published",
         (5, 0): "tail f1: first oc arrived", (5, 1): "tail f1: nd_new_0 arrived", (5, 2): "tail f1: nd_new published", (5, 3): "tail f1: x'' published",
         (6, 1): "tail f2: nd_new_1 arrived", (6, 3): "tail f2: x'' published",
         (8, 0): "fc f0 k0: x'' gathered", (8, 1): "fc f0: fc1 rows published", (8, 2): "fc f0: h gathered", (8, 3): "fc f0: fc2 rows published",
         (9, 0): "fc f2 k0: x'' gathered", (9, 1): "fc f2: fc1 rows published", (9, 2): "fc f2: h gathered", (9, 3): "fc f2: fc2 rows published",
         (10, 0): "proj f0: x3 gathered", (10, 1): "proj f0: x published", (11, 0): "proj f2: x3 gathered", (11, 1): "proj f2: x published"}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2024
    cfg = O.OracleConfig(1, 256, 8, 2024, 8, 0, 3, 2, True, "adaln")
    m = build(cfg, "bf16")
    x, _, ib = recipe_inputs(1, n, cfg, seed=5)
    x0, ibg = x[:, :1].cuda().contiguous(), ib.cuda().contiguous()
    m.engine().rollout_kv(x0, ibg, n)   # warm-up
    st = torch.zeros(n * 64, dtype=torch.int64, device="cuda")
    N.lib().sea_kv_debug_stamps(st.data_ptr())
    m.engine().rollout_kv(x0, ibg, n)
    torch.cuda.synchronize()
    N.lib().sea_kv_debug_stamps(None)
    t = st.cpu().numpy().reshape(n, 16, 4).astype(np.float64) * 0.01   # us
    for lo, hi in ((50, 150), (n - 150, n - 50)):
        if lo < 1 or hi > n:
            continue
        base = t[lo:hi, 10, 1]   # proj f0 publishes x of the step
        prev = t[lo - 1:hi - 1, 10, 1]
        print(f"steps {lo}..{hi}: step period {np.mean(base - prev):.2f} us; events relative to the previous step's 'proj f0: x published':")
        ev = sorted(((np.mean(t[lo:hi, r, k] - prev), NAMES[(r, k)]) for (r, k) in NAMES if t[lo:hi, r, k].min() > 0))
        for dt, nm in ev:
            print(f"  {dt:7.2f} us  {nm}")


if __name__ == "__main__":
    main()
