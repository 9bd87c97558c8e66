#!/bin/bash
# round 5, GPU call 10: A/B on one box -- the helpers' last-tile skip (-DEGC_FT_LAST_TILE_SKIP) against the default, forward launch, three batches
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05m; mkdir -p $O
for rep in 1 2; do
for lib in default skip; do
  for w in zinc molhiv cifar; do
    if [ $lib = skip ]; then export EGC_HIP_LIB=$R/egc_amd/lib/var_ft_skip.so; else unset EGC_HIP_LIB; fi
    EGC_TILE_ONLY=$w timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_${lib}_${w}_$rep -o kt --output-format csv -- python3 $R/tools/fused_tile_time.py > /dev/null 2>&1
  done
done
done
find $O -name "*kernel_trace.csv" -delete
cd $R
python3 - <<'PY'
import csv, glob
for w in ("zinc","molhiv","cifar"):
    for lib in ("default","skip"):
        for rep in (1,2):
            f = glob.glob(f"gpurun_out/r05m/kt_{lib}_{w}_{rep}/**/*kernel_stats.csv", recursive=True)[0]
            for r in csv.DictReader(open(f)):
                if "fused_tile_kernel" in r["Name"]:
                    print(w, lib, rep, "calls", r["Calls"], "avg %.2f us" % (float(r["AverageNs"])/1e3), "min %.2f" % (float(r["MinNs"])/1e3))
PY
