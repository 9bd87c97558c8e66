cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06k; mkdir -p $O
for lib in base bwdw5 bwdw6; do
  if [ $lib = base ]; then unset EGC_HIP_LIB; else export EGC_HIP_LIB=$R/egc_amd/lib/var_$lib.so; fi
  for sh in "224,4,4,add+mean+max,1,lay" "296,8,4,symadd,1,lay"; do
    tag=$(echo $sh | cut -d, -f1)
    EGC_SMALL_ONLY=molhiv EGC_STEP_SHAPE="$sh" rocprofv3 --kernel-trace --stats -d $O/kt_${lib}_$tag -o kt --output-format csv -- python3 $R/tools/batch_train_step_time.py > $O/step_${lib}_$tag.log 2>&1
  done
done
unset EGC_HIP_LIB
find $O -name "*kernel_trace.csv" -delete
cd $R
python3 - <<'PY'
import csv, glob
for lib in ("base", "bwdw5", "bwdw6"):
    for tag in ("224", "296"):
        fs = glob.glob(f"gpurun_out/r06k/kt_{lib}_{tag}/**/*kernel_stats.csv", recursive=True)
        for r in csv.DictReader(open(fs[0])):
            if "bwd_dst_fast" in r["Name"] or "bwd_src" in r["Name"]:
                print(lib, tag, r["Name"][:50], "calls", r["Calls"], "avg %.2f us" % (float(r["AverageNs"])/1e3))
PY
