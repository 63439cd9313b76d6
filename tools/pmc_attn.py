"""PMC probe of the attention kernels alone (development aid).  Under rocprofv3:
   rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d <dir> -- python3 tools/pmc_attn.py [fwd|bwd]
then   python tools/pmc_attn.py fold <dir> [<dir2> ...]   prints per-kernel counter sums / per-wave figures."""
import sys, os, glob, csv, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(what):
    import torch
    from sea_amd import ops
    dev, dt = torch.device("cuda:0"), torch.bfloat16
    for (B, H, hd, T, n) in [(8, 8, 32, 2024, 3), (8, 8, 16, 2024, 2), (1, 8, 32, 2024, 3), (1, 8, 16, 2024, 2)]:
        cap = (T + 7) // 8 * 8
        probs, bprobs = [], []
        for _ in range(n):
            Q = (torch.randn(B, H, T, hd, device=dev) * hd ** -0.25).to(dt)
            K = (torch.randn(B, H, cap, hd, device=dev) * hd ** -0.25).to(dt)
            V = torch.randn(B, H, cap, hd, device=dev).to(dt)
            O = torch.empty(B, T, H * hd, device=dev, dtype=dt)
            LSE = torch.empty(B, H, T, device=dev)
            probs.append(dict(Q=Q, K=K, Vt=V.transpose(2, 3).contiguous(), O=O, LSE=LSE))
            E = H * hd
            bprobs.append(dict(Q=Q, K=K, V=V, O=O, dO=torch.randn(B, T, E, device=dev).to(dt), LSE=LSE, delta=torch.empty(B, H, T, device=dev),
                               dQ=torch.empty(B * T, E, device=dev, dtype=dt), dK=torch.empty(B * T, E, device=dev, dtype=dt), dV=torch.empty(B * T, E, device=dev, dtype=dt)))
        rope = torch.zeros(cap, hd // 2, 2, device=dev)
        rope[..., 0] = 1
        for _ in range(3):
            ops.attention_fwd(probs, B, H, hd, T, T, cap, 0, 0, dt)
            if what == "bwd":
                ops.attention_bwd(bprobs, rope, B, H, hd, T, T, cap, 0, 0, ops.q_scale(hd), dt)
        torch.cuda.synchronize()


def fold(dirs):
    rows = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            seen = set()
            for r in csv.DictReader(open(f)):
                k = (r["Kernel_Name"][:60], r.get("Grid_Size", ""))
                rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
                if (r["Dispatch_Id"], f) not in seen and r["Counter_Name"] == "SQ_WAVES":
                    seen.add((r["Dispatch_Id"], f)); cnt[k] += 1
    for k, c in sorted(rows.items()):
        if "attn" not in k[0] and "attention" not in k[0]:
            continue
        n = max(cnt[k], 1)
        print(k, "dispatches", n)
        w = c.get("SQ_WAVES", 1) / n
        for name in sorted(c):
            print(f"    {name:28s} {c[name] / n:16.0f}   per wave {c[name] / n / w:12.1f}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "fold":
        fold(sys.argv[2:])
    else:
        run(sys.argv[1] if len(sys.argv) > 1 else "fwd")
