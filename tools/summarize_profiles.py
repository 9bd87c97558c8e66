#!/usr/bin/env python3
"""gpurun_out/<tag>/ (written by tools/collect_profiles.sh on the GPU box) -> the summaries kept under profiles/:
kernel statistics as rocprofv3 wrote them, counter collections condensed to one row per (kernel, counter) with the
mean over dispatches, pmc_traffic.json for bench.py's roofline.traffic (tools/pmc_traffic.py).

usage: python tools/summarize_profiles.py <tag>"""
import csv
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def condense(src, dst):
    acc = {}
    for r in csv.DictReader(open(src)):
        k = (r["Kernel_Name"], r["Counter_Name"])
        a = acc.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Counter_Name", "Dispatches", "Average_Counter_Value"])
        for (kn, cn), (n, s) in sorted(acc.items()):
            w.writerow([kn[:160], cn, n, s / n])


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    cp = lambda a, b: shutil.copyfile(os.path.join(src, a), os.path.join(dst, b))
    cp("kt2/kt_kernel_stats.csv", f"{tag}_kernel_stats.csv")
    cp("kt/kt_kernel_stats.csv", f"{tag}_kernel_stats_all_configs.csv")
    cp("train/kt_kernel_stats.csv", f"{tag}_training_step_kernel_stats.csv")
    cp("block/kt_kernel_stats.csv", f"{tag}_training_block_kernel_stats.csv")
    cp("mag/kt_kernel_stats.csv", f"{tag}_mag_layer_kernel_stats.csv")
    if os.path.exists(os.path.join(src, "small/kt_kernel_stats.csv")):
        cp("small/kt_kernel_stats.csv", f"{tag}_small_batch_step_kernel_stats.csv")
        cp("small.log", f"{tag}_small_batch_step.log")
    cp("bench.json", f"{tag}_bench.json")
    cp("mag_bench.json", f"{tag}_mag_bench.json")
    cp("rmag_bench.json", f"{tag}_rmag_bench.json")
    for w in ("molhiv", "cifar"):
        if os.path.exists(os.path.join(src, f"tile_{w}/kt_kernel_stats.csv")):
            cp(f"tile_{w}/kt_kernel_stats.csv", f"{tag}_tile_path_{w}_kernel_stats.csv")
            cp(f"tile_{w}.log", f"{tag}_tile_path_{w}.log")
    for name, out in (("pmc_tile_fetch", "pmc_tile_path_cifar_fetch_size"), ("pmc_tile_write", "pmc_tile_path_cifar_write_size")):
        f = os.path.join(src, name, "pmc_counter_collection.csv")
        if os.path.exists(f):
            condense(f, os.path.join(dst, f"{tag}_{out}.csv"))
    # round 4: the one-launch batch layer and the long-k GEMM
    for w in ("molhiv", "cifar", "zinc"):
        if os.path.exists(os.path.join(src, f"fused_{w}/kt_kernel_stats.csv")):
            cp(f"fused_{w}/kt_kernel_stats.csv", f"{tag}_fused_tile_{w}_kernel_stats.csv")
            cp(f"fused_{w}.log", f"{tag}_fused_tile_{w}.log")
    for name in ("pmc_fused_molhiv_fetch", "pmc_fused_molhiv_write", "pmc_fused_cifar_fetch", "pmc_fused_cifar_write", "pmc_fused_sq",
                 "pmc_fused_lds", "pmc_magk_fetch", "pmc_magk_write", "pmc_magk_sq"):
        f = os.path.join(src, name, "pmc_counter_collection.csv")
        if os.path.exists(f):
            condense(f, os.path.join(dst, f"{tag}_{name}.csv"))
    if os.path.exists(os.path.join(src, "magk/kt_kernel_stats.csv")):
        cp("magk/kt_kernel_stats.csv", f"{tag}_mag_gemm_kernel_stats.csv")
        cp("magk.log", f"{tag}_mag_gemm.log")
    for name in ("gemm_error", "gemm_time_wide"):
        if os.path.exists(os.path.join(src, name + ".log")):
            cp(name + ".log", f"{tag}_{name}.log")
    if os.path.exists(os.path.join(src, "host_overhead.log")):
        cp("host_overhead.log", f"{tag}_host_overhead.log")
    for name in ("reference_shapes", "gemm_time"):
        if os.path.exists(os.path.join(src, name + ".log")):
            cp(name + ".log", f"{tag}_{name}.log")
    for name in ("pmc_fetch", "pmc_write", "pmc_sq"):
        out = {"pmc_fetch": "pmc_fetch_size", "pmc_write": "pmc_write_size", "pmc_sq": "pmc_sq_counters"}[name]
        condense(os.path.join(src, name, "pmc_counter_collection.csv"), os.path.join(dst, f"{tag}_{out}.csv"))
    for name, out in (("pmc_train_fetch", "pmc_training_step_fetch_size"), ("pmc_train_write", "pmc_training_step_write_size")):
        f = os.path.join(src, name, "pmc_counter_collection.csv")
        if os.path.exists(f):
            condense(f, os.path.join(dst, f"{tag}_{out}.csv"))
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"),
                           os.path.join(src, "pmc_fetch", "pmc_counter_collection.csv"),
                           os.path.join(src, "pmc_write", "pmc_counter_collection.csv"),
                           os.path.join(dst, "pmc_traffic.json")])
    j = json.load(open(os.path.join(dst, "pmc_traffic.json")))
    commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"]).decode().strip()
    j["collected"] = f"{tag}, binary of commit {commit}, profiles/{tag}_pmc_fetch_size.csv + {tag}_pmc_write_size.csv"
    json.dump(j, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
