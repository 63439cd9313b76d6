"""Copy the judged files of a tools/measure_all.sh run into profiles/ under this round's names:
    python tools/collect_profiles.py <tag> <rNN>        gpurun_out/<tag>/... -> profiles/<rNN>_*"""
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stats_csv(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    return f[0] if f else None


def main():
    tag, rnd = sys.argv[1], sys.argv[2]
    src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
    plain = {"bench_n1.json": "bench_n1.json", "bench_rollout_graph.json": "bench_rollout_graph.json", "bench_rollout_b8.json": "bench_rollout_b8.json",
             "bench_decode.json": "bench_decode.json", "bench_encode.json": "bench_encode.json", "kv_shipped_widths.txt": "kv_shipped_widths.txt",
             "train_launch_table.txt": "train_cfg3_launch_table.txt", "train_step_trace.txt": "train_cfg3_step_trace.txt",
             "forward_cfg2_launch_breakdown.txt": "forward_cfg2_launch_breakdown.txt", "forward_cfg2_pmc_traffic.json": "forward_cfg2_pmc_traffic.json",
             "forward_cfg2_pmc_utilisation.json": "forward_cfg2_pmc_utilisation.json", "train_cfg3_pmc_traffic.json": "train_cfg3_pmc_traffic.json",
             "train_cfg3_pmc_utilisation.json": "train_cfg3_pmc_utilisation.json"}
    for a, b in plain.items():
        f = os.path.join(src, a)
        if os.path.exists(f):
            shutil.copy(f, os.path.join(dst, f"{rnd}_{b}"))
            print("copied", a)
        else:
            print("MISSING", a)
    for d, b in (("prof_fwd", "forward_cfg2_kernel_stats.csv"), ("prof_train", "train_cfg3_kernel_stats.csv"), ("prof_kv", "kv_kernel_stats.csv"),
                 ("prof_kvs", "kv_shipped_widths_kernel_stats.csv")):
        f = stats_csv(os.path.join(src, d))
        if f:
            shutil.copy(f, os.path.join(dst, f"{rnd}_{b}"))
            print("copied", d)
        else:
            print("MISSING", d)


if __name__ == "__main__":
    main()
