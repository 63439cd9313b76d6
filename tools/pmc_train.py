"""PMC evidence for the cfg3 training step (fwd + bwd + AdamW, B = 8, T = 2024, bf16): HBM bytes and MFMA / VALU utilisation per KERNEL of the step.

  on the GPU box (each pass its own process; counters never share a run with a trace domain other than --kernel-trace):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE  --output-format csv -d <dir>/f -- python3 tools/train_trace.py run
    rocprofv3 --kernel-trace --pmc WRITE_SIZE  --output-format csv -d <dir>/w -- python3 tools/train_trace.py run
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d <dir>/u -- python3 tools/train_trace.py run
  then
    python tools/pmc_train.py <dir>/f <dir>/w <dir>/u <traffic.json> <utilisation.json>

Counters as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE / WRITE_SIZE count KiB; FETCH_SIZE is doubled (wide reads are tallied at half).  tools/train_trace.py
runs 8 fused steps; figures are per step (sum over a kernel's dispatches / 8) and per dispatch.  The launch names of bench.py's train leg map to kernels as listed in
RECORD_KERNELS (the attention backward of a field is its dQ and its dK/dV kernel)."""
import collections
import csv
import glob
import json
import re
import sys

STEPS = 8
RECORD_KERNELS = {   # bench.py record name -> (regex over the kernel symbol, dispatches of that kernel per step that belong to the record)
    "bwd.self.attention": (r"attn_bwd_d(q|kv)_kernelIDF16bLi32", None),
    "bwd.cross.attention (one field)": (r"attn_bwd_d(q|kv)_kernelIDF16bLi16", 3),
    "self.attention": (r"attention_fwd_kernelIDF16bLi32", None),
    "cross.attention (one field)": (r"attention_fwd_kernelIDF16bLi16", 3),
    "bwd.mlp.ln_gelu": (r"rownorm_bwd_wide_kernel", None),
}


def per_kernel(d, counters):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] in counters:
                agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
                disp[r["Kernel_Name"]].add(r["Dispatch_Id"])
    return agg, {k: len(v) for k, v in disp.items()}


def durations(d):
    tot, n = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            tot[r["Kernel_Name"]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            n[r["Kernel_Name"]] += 1
    return tot, n


def main():
    fd, wd, ud, out_t, out_u = sys.argv[1:6]
    fe, nf = per_kernel(fd, {"FETCH_SIZE"})
    wr, _ = per_kernel(wd, {"WRITE_SIZE"})
    ut, _ = per_kernel(ud, {"SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE", "SQ_WAVES"})
    dur, nd = durations(fd)
    ours = [k for k in dur if not k.startswith("void at::") and "rocclr" not in k]
    kernels = collections.OrderedDict()
    for k in sorted(ours, key=lambda k: -dur[k]):
        n = max(nd[k], 1)
        f = fe[k].get("FETCH_SIZE", 0.0) * 1024 * 2
        w = wr[k].get("WRITE_SIZE", 0.0) * 1024
        kernels[k[:110]] = {"dispatches_per_step": n / STEPS, "avg_us": round(dur[k] / n, 2), "us_per_step": round(dur[k] / STEPS, 1),
                            "hbm_bytes_per_dispatch": int((f + w) / n), "hbm_bytes_per_step": int((f + w) / STEPS),
                            "achieved_GBps": round((f + w) / n / (dur[k] / n * 1e-6) / 1e9, 1) if dur[k] else None}
    launches = {}
    for name, (pat, per_step) in RECORD_KERNELS.items():
        ks = [k for k in ours if re.search(pat, k)]
        if not ks:
            continue
        f = sum(fe[k].get("FETCH_SIZE", 0.0) for k in ks) * 1024 * 2
        w = sum(wr[k].get("WRITE_SIZE", 0.0) for k in ks) * 1024
        div = STEPS * (per_step or 1)
        launches[name] = {"kernels": [k[:90] for k in ks], "hbm_bytes": int((f + w) / div), "us": round(sum(dur[k] for k in ks) / div, 1)}
    total = sum(v["hbm_bytes_per_step"] for v in kernels.values())
    json.dump({"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/train_trace.py run (8 fused cfg3 train steps, bf16, B=8, T=2024); "
                       "bytes = FETCH_SIZE * 1024 * 2 + WRITE_SIZE * 1024 (gfx950 corrections of MI355X_MICROARCH.md); per step = sum over the kernel's dispatches / 8",
               "hbm_bytes_per_step": total, "launches": launches, "kernels": kernels}, open(out_t, "w"), indent=1)
    util = collections.OrderedDict()
    for k in sorted(ours, key=lambda k: -dur[k]):
        c = ut[k]
        simd_cycles = c.get("GRBM_GUI_ACTIVE", 0.0) / 8 * 1024
        if simd_cycles <= 0:
            continue
        util[k[:110]] = {"mfma_busy": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simd_cycles, 4), "valu_busy": round(4 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / simd_cycles, 4),
                         "mfma_insts_per_step": int(c.get("SQ_INSTS_MFMA", 0.0) / STEPS), "valu_insts_per_step": int(c.get("SQ_INSTS_VALU", 0.0) / STEPS)}
    json.dump({"note": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES over tools/train_trace.py run; "
                       "busy fractions are of SIMD-cycles while the kernel runs (mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024), valu_busy = 4 * SQ_ACTIVE_INST_VALU / same)",
               "kernels": util}, open(out_u, "w"), indent=1)
    print("hbm bytes per step: %.1f MB" % (total / 1e6))
    for k, v in list(kernels.items())[:14]:
        print(f"{v['us_per_step']:8.1f} us/step  {v['hbm_bytes_per_step'] / 1e6:8.1f} MB/step  {v['achieved_GBps']} GB/s   {k[:70]}")


if __name__ == "__main__":
    main()
