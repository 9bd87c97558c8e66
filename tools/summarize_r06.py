#!/usr/bin/env python3
"""gpurun_out/r06_final/ (tools/r06_final.sh, run on the GPU box) + the round's earlier calls -> the summaries kept under profiles/r06_*.
Kernel statistics as rocprofv3 wrote them; counter collections condensed to one row per (kernel, counter) with the mean over
dispatches; profiles/pmc_traffic.json (what bench.py's roofline.traffic reads) refreshed from this round's FETCH_SIZE / WRITE_SIZE passes."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r06_final")
DST = os.path.join(ROOT, "profiles")


def cp(a, b):
    shutil.copyfile(os.path.join(SRC, a), os.path.join(DST, b))


def counters(path):
    acc = {}
    for r in csv.DictReader(open(path)):
        a = acc.setdefault((r["Kernel_Name"], r["Counter_Name"]), [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return {k: (n, s / n) for k, (n, s) in acc.items()}


def condense(paths, dst):
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Pass", "Kernel_Name", "Counter_Name", "Dispatches", "Average_Counter_Value"])
        for tag, p in paths:
            for (kn, cn), (n, v) in sorted(counters(p).items()):
                if "egc::" in kn:
                    w.writerow([tag, kn[:170], cn, n, v])


def pmc(name):
    return glob.glob(os.path.join(SRC, name, "**", "*counter_collection.csv"), recursive=True)[0]


def traffic(fetch_csv, write_csv, match):
    f = {k[0]: v for k, v in counters(fetch_csv).items() if k[1] == "FETCH_SIZE" and match(k[0])}
    w = {k[0]: v for k, v in counters(write_csv).items() if k[1] == "WRITE_SIZE" and match(k[0])}
    out = {}
    for kn, (n, v) in f.items():
        fb, wb = 2.0 * v * 1024.0, w.get(kn, (0, 0.0))[1] * 1024.0
        out[kn] = {"kernel": kn[:140], "launches": n, "fetch_bytes_corrected": fb, "write_bytes": wb, "hbm_bytes_per_launch": fb + wb}
    return out


def main():
    cp("kt2/kt_kernel_stats.csv", "r06_kernel_stats.csv")
    cp("bench_config2_under_rocprof.json", "r06_bench_config2_under_rocprof.json")
    cp("bench_line.json", "r06_bench.json")
    cp("bench_detail.json", "r06_bench_detail.json")
    for w in ("zinc", "molhiv", "cifar"):
        cp(f"fused_{w}/kt_kernel_stats.csv", f"r06_fused_tile_{w}_kernel_stats.csv")
    for w in ("molhiv", "zinc"):
        cp(f"step_{w}/kt_kernel_stats.csv", f"r06_batch_train_step_{w}_kernel_stats.csv")
    for t in ("224", "296", "168"):
        cp(f"wide_{t}/kt_kernel_stats.csv", f"r06_wide_train_step_{t}_kernel_stats.csv")
        cp(f"wide_step_{t}.log", f"r06_wide_train_step_{t}.log")
    cp("gemm_rows_scan.log", "r06_gemm_rows_scan.log")
    cp("xt_wide_sweep.log", "r06_xt_wide_sweep.log")
    for a, b in (("batch_train_step.log", "r06_batch_train_step.log"), ("batch_shapes.log", "r06_batch_shapes.log"),
                 ("stdvar_modes.log", "r06_stdvar_modes.log"), ("eager_step_native.log", "r06_eager_step_native.log"),
                 ("eager_step_python.log", "r06_eager_step_python.log"), ("eager_step_variants.log", "r06_eager_step_variants.log"),
                 ("lds_atomic_order.log", "r06_lds_atomic_order.log")):
        txt = "".join(ln for ln in open(os.path.join(SRC, a)) if "amdgpu.ids" not in ln)
        open(os.path.join(DST, b), "w").write(txt)
    condense([("config2 FETCH_SIZE", pmc("pmc_fetch")), ("config2 WRITE_SIZE", pmc("pmc_write"))], os.path.join(DST, "r06_pmc_config2.csv"))
    condense([(f"{w} {c}", pmc(f"pmc_fused_{w}_{d}")) for w in ("molhiv", "cifar") for c, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write"))]
             + [("molhiv training step FETCH_SIZE", pmc("pmc_step_fetch")), ("molhiv training step WRITE_SIZE", pmc("pmc_step_write"))],
             os.path.join(DST, "r06_pmc_fused_tile.csv"))
    # pmc_traffic.json: config 2's aggregate + GEMM, the one-launch kernels of configs 3 / 4, the backward launch
    pj_path = os.path.join(DST, "pmc_traffic.json")
    pj = json.load(open(pj_path))
    t2 = traffic(pmc("pmc_fetch"), pmc("pmc_write"), lambda k: "agg_fast_kernel" in k or "basis_gemm_f16x2_kernel" in k)
    for kn, rec in t2.items():
        pj["kernels"]["aggregate" if "agg_fast" in kn else "gemm"] = rec
        if "agg_fast" in kn:
            pj["aggregate_kernel_hbm_bytes_per_launch"] = rec["hbm_bytes_per_launch"]
    pj["collected"] = "r06, tools/r06_final.sh: profiles/r06_pmc_config2.csv (FETCH_SIZE x 2 + WRITE_SIZE, separate --pmc passes)"
    for w, key in (("molhiv", "config3_molhiv_b2048"), ("cifar", "config4_cifar_b2048")):
        t = traffic(pmc(f"pmc_fused_{w}_fetch"), pmc(f"pmc_fused_{w}_write"), lambda k: "fused_tile_kernel" in k)
        rec = max(t.values(), key=lambda r: r["launches"])
        rec["collected"] = f"r06, profiles/r06_pmc_fused_tile.csv ({rec['launches']} launches)"
        pj["side_configs"][key] = rec
    t = traffic(pmc("pmc_step_fetch"), pmc("pmc_step_write"), lambda k: "fused_tile_kernel" in k)
    for kn, rec in t.items():
        rec["collected"] = "r06, profiles/r06_pmc_fused_tile.csv (4-block molhiv training step)"
        pj["side_configs"]["molhiv_step_backward_launch" if kn.rstrip().endswith("1>(egc::AggArgs, egc::FusedTileArgs)") else "molhiv_step_forward_launch"] = rec
    json.dump(pj, open(pj_path, "w"), indent=1)
    # the round's earlier calls
    for a, b in (("gpurun_out/r06e/magk_kt.log", None),):
        pass
    print("profiles/r06_* written")


if __name__ == "__main__":
    sys.exit(main())
