#!/bin/bash
# round 5, GPU call 11: A/B on one box -- the backward launch with the helpers' next-tile work skipped behind the last tile (-DEGC_FTB_LAST_TILE_SKIP)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05n; mkdir -p $O
for rep in 1 2; do
for lib in default skip; do
  for w in zinc molhiv; do
    if [ $lib = skip ]; then export EGC_HIP_LIB=$R/egc_amd/lib/var_ftb_skip.so; else unset EGC_HIP_LIB; fi
    EGC_SMALL_ONLY=$w timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_${lib}_${w}_$rep -o kt --output-format csv -- python3 $R/tools/batch_train_step_time.py > /dev/null 2>&1
  done
done
done
find $O -name "*kernel_trace.csv" -delete
cd $R
python3 - <<'PY'
import csv, glob
for w in ("zinc","molhiv"):
    for lib in ("default","skip"):
        for rep in (1,2):
            f = glob.glob(f"gpurun_out/r05n/kt_{lib}_{w}_{rep}/**/*kernel_stats.csv", recursive=True)[0]
            for r in csv.DictReader(open(f)):
                if "fused_tile_kernel" in r["Name"]:
                    print(w, lib, rep, "bwd" if r["Name"].split("(")[0].rstrip().endswith("0, 1>") else "fwd", "calls", r["Calls"], "avg %.2f us" % (float(r["AverageNs"])/1e3), "min %.2f" % (float(r["MinNs"])/1e3))
PY
