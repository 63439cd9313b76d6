"""Fold a rocprofv3 counter_collection.csv: per kernel-name substring and dispatch, the sum of every counter over its rows.  python tools/pmc_kernel_sum.py <dir> <substring>"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        d = agg.setdefault(r["Dispatch_Id"], collections.OrderedDict())
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k, d in agg.items():
    print(k, "  ".join(f"{n}={v:.4g}" for n, v in d.items()))
