cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05l; mkdir -p $O
for w in zinc molhiv cifar; do
  EGC_TILE_ONLY=$w timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$w -o kt --output-format csv -- python3 $R/tools/fused_tile_time.py > /dev/null 2>&1
done
EGC_SMALL_ONLY=molhiv timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_step -o kt --output-format csv -- python3 $R/tools/batch_train_step_time.py > /dev/null 2>&1
find $O -name "*kernel_trace.csv" -delete
cd $R
python3 - <<'PY'
import csv, glob
for w in ("zinc","molhiv","cifar","step"):
    f = glob.glob(f"gpurun_out/r05l/kt_{w}/**/*kernel_stats.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if "fused_tile_kernel" in r["Name"]:
            print(w, r["Name"][:40], r["Name"][-60:-30], "calls", r["Calls"], "avg %.1f us" % (float(r["AverageNs"])/1e3), "min %.1f" % (float(r["MinNs"])/1e3))
PY
