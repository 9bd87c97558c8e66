cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06l; mkdir -p $O
cd $R; timeout 900 python -m pytest tests/test_backward_gpu.py tests/test_backward_golden.py tests/test_native_ext.py tests/test_nets_golden.py -x -q 2>&1 | tail -4; cd /tmp
for sh in "224,4,4,add+mean+max,1,lay" "296,8,4,symadd,1,lay" "168,8,4,symadd,1,lay"; do
  tag=$(echo $sh | cut -d, -f1); w=molhiv; [ $tag = 168 ] && w=zinc
  EGC_SMALL_ONLY=$w EGC_STEP_SHAPE="$sh" rocprofv3 --kernel-trace --stats -d $O/kt_$tag -o kt --output-format csv -- python3 $R/tools/batch_train_step_time.py > $O/step_$tag.log 2>&1
  EGC_SMALL_ONLY=$w EGC_STEP_SHAPE="$sh" python3 $R/tools/batch_train_step_time.py 2>&1 | grep -v amdgpu | head -2
done
find $O -name "*kernel_trace.csv" -delete
cd $R
python3 - <<'PY'
import csv, glob
for tag in ("224", "296", "168"):
    fs = glob.glob(f"gpurun_out/r06l/kt_{tag}/**/*kernel_stats.csv", recursive=True)
    for r in csv.DictReader(open(fs[0])):
        if "bwd_dst_fast" in r["Name"] or "bwd_src" in r["Name"] or "agg_fast" in r["Name"]:
            print(tag, r["Name"][:70], "calls", r["Calls"], "avg %.2f us" % (float(r["AverageNs"])/1e3))
PY
