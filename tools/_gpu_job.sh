cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04m; mkdir -p $O
rm -rf $O/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python3 tools/chain_probe.py replay 50 > $O/prof.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r04m/prof/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if int(r["Calls"])>=50: print(f'{r["Name"][:64]:64s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us')
PY
rm -rf $O/prof/*kernel_trace.csv
