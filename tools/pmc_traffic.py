#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; collected separately, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes) of `python bench.py ...` into
profiles/pmc_traffic.json, which bench.py reads to fill roofline.traffic.

gfx950 corrections applied (MI355X_MICROARCH.md, section HBM): the counters are in KiB, and FETCH_SIZE
reports exactly half of the bytes of wide (16 B / lane) coalesced reads -> doubled.  The figure is the
per-launch average over the dispatches of the dominant kernel.

usage: tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json]
"""
import csv
import json
import sys


def per_kernel(path, counter):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        acc.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over bench.py; KiB -> bytes; "
                     "FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B)", "kernels": {}}
    for name in fetch:
        short = "aggregate" if "agg_fast_kernel" in name or "agg_rows_kernel" in name else \
            "gemm" if "basis_gemm_f16x2_kernel" in name else \
            "gemm_bases_only_of_the_fused_experiment" if "basis_gemm" in name else None
        if short is None:
            continue
        f_b = 2.0 * fetch[name] * 1024.0
        w_b = write.get(name, 0.0) * 1024.0
        out["kernels"][short] = {"kernel": name[:120], "fetch_bytes_corrected": f_b, "write_bytes": w_b,
                                 "hbm_bytes_per_launch": f_b + w_b}
    if "aggregate" in out["kernels"]:
        out["aggregate_kernel_hbm_bytes_per_launch"] = out["kernels"]["aggregate"]["hbm_bytes_per_launch"]
    dst = sys.argv[3] if len(sys.argv) > 3 else "profiles/pmc_traffic.json"
    try:       # the side configs' entries (bench.py: side_traffic) come from their own passes: kept across a refresh of the headline's
        out["side_configs"] = json.load(open(dst)).get("side_configs", {})
    except Exception:   # noqa: BLE001
        pass
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
