"""Fold a rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES) over
tools/pmc_forward.py into per-launch MFMA / VALU utilisation:   python tools/pmc_util.py <dir> <names.json> <out.json>

  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)      fraction of SIMD-cycles with the matrix core busy
  valu_busy = 4 * SQ_ACTIVE_INST_VALU / (same denominator)                            SQ_ACTIVE_INST_VALU counts quad-cycles
GRBM_GUI_ACTIVE is summed over the 8 XCDs by rocprofv3, hence the division."""
import csv, glob, json, sys, collections

d, names_json, out = sys.argv[1:4]
meta = json.load(open(names_json)); names, n_fwd = meta["names"], meta["n"]
rows = collections.defaultdict(dict)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = int(r["Dispatch_Id"])
        rows[k]["name"] = r["Kernel_Name"]
        rows[k][r["Counter_Name"]] = rows[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
seq = [rows[k] for k in sorted(rows)]
seq = [x for x in seq if not x["name"].startswith("void at::") and "rocclr" not in x["name"] and "convert_kernel" not in x["name"]]
per = len(names); seq = seq[-per * n_fwd:]
assert len(seq) == per * n_fwd, (len(seq), per, n_fwd)
res = collections.OrderedDict()
for i, nme in enumerate(names):
    xs = seq[i::per]
    avg = lambda c: sum(x.get(c, 0.0) for x in xs) / len(xs)
    simd_cycles = avg("GRBM_GUI_ACTIVE") / 8 * 1024
    res[nme] = {"kernel": xs[0]["name"][:90], "mfma_busy": round(avg("SQ_VALU_MFMA_BUSY_CYCLES") / simd_cycles, 4),
                "valu_busy": round(4 * avg("SQ_ACTIVE_INST_VALU") / simd_cycles, 4), "mfma_insts": int(avg("SQ_INSTS_MFMA")),
                "valu_insts": int(avg("SQ_INSTS_VALU")), "waves": int(avg("SQ_WAVES")), "gpu_cycles_per_xcd": int(avg("GRBM_GUI_ACTIVE") / 8)}
json.dump({"note": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES over "
                   "tools/pmc_forward.py (cfg2, B=1, T=2024, bf16, plain replay); busy fractions are of SIMD-cycles while the kernel runs", "launches": res},
          open(out, "w"), indent=1)
for k, v in res.items():
    print(f"{k:22s} mfma_busy {v['mfma_busy']:.3f} valu_busy {v['valu_busy']:.3f} mfma {v['mfma_insts']:8d} valu {v['valu_insts']:9d}")
