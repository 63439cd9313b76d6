"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/pmc_forward.py into HBM bytes per plan launch.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <names.json> <out.json>

Counter handling as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE and WRITE_SIZE count kilobytes (x1024); FETCH_SIZE is
doubled (wide reads are counted as half).  Dispatches are matched to plan records by order: the script ran WARM + N forwards, each
forward dispatching the plan's records in order (the first record of a forward is found by its kernel name)."""
import csv, glob, json, sys, collections

fetch_dir, write_dir, names_json, out = sys.argv[1:5]
meta = json.load(open(names_json))
names, n_fwd = meta["names"], meta["n"]


def per_dispatch(d, counter):
    rows = collections.OrderedDict()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                k = int(r["Dispatch_Id"])
                rows[k] = (r["Kernel_Name"], rows.get(k, (None, 0.0))[1] + float(r["Counter_Value"]))
    return [rows[k] for k in sorted(rows)]


def durations(d):
    out_ = {}
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            out_[int(r["Dispatch_Id"])] = (r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return [out_[k] for k in sorted(out_)]


def fold(seq):
    """keep the dispatches of the plan's kernels only, last n_fwd forwards, grouped by record index"""
    ours = [x for x in seq if not x[0].startswith("void at::") and "rocclr" not in x[0] and "convert_kernel" not in x[0]]
    per = len(names)
    ours = ours[-per * n_fwd:]
    assert len(ours) == per * n_fwd, (len(ours), per, n_fwd)
    acc = [[] for _ in names]
    for i, x in enumerate(ours):
        acc[i % per].append(x)
    return acc

fe, wr, du = fold(per_dispatch(fetch_dir, "FETCH_SIZE")), fold(per_dispatch(write_dir, "WRITE_SIZE")), fold(durations(fetch_dir))
launches = collections.OrderedDict()
for i, nme in enumerate(names):
    f = sum(v for _, v in fe[i]) / len(fe[i]) * 1024 * 2
    w = sum(v for _, v in wr[i]) / len(wr[i]) * 1024
    launches[nme] = {"kernel": fe[i][0][0][:100], "avg_us": round(sum(v for _, v in du[i]) / len(du[i]), 2), "fetch_bytes_corrected": int(f),
                     "write_bytes": int(w), "hbm_bytes": int(f + w)}
json.dump({"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/pmc_forward.py, plain-replay "
                   "forwards at cfg2 (B=1,T=2024,bf16); FETCH_SIZE x1024 x2 (gfx950 wide-read correction), WRITE_SIZE x1024; per launch",
           "launches": launches}, open(out, "w"), indent=1)
print(json.dumps({k: (v["avg_us"], v["hbm_bytes"]) for k, v in launches.items()}))
