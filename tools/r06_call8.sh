cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06j; mkdir -p $O
for sh in "224,4,4,add+mean+max,1,lay" "296,8,4,symadd,1,lay" "168,8,4,symadd,1,lay"; do
  tag=$(echo $sh | cut -d, -f1)
  w=molhiv; [ $tag = 168 ] && w=zinc
  EGC_SMALL_ONLY=$w EGC_STEP_SHAPE="$sh" rocprofv3 --kernel-trace --stats -d $O/kt_step_$tag -o kt --output-format csv -- python3 $R/tools/batch_train_step_time.py > $O/step_$tag.log 2>&1
  grep -v "amdgpu.ids\|rocprofv3\|Opened" $O/step_$tag.log | tail -n 3
done
EGC_SMALL_ONLY=cifar true
find $O -name "*kernel_trace.csv" -delete
cd $R
python3 - <<'PY'
import csv, glob
for tag in ("224", "296", "168"):
    fs = glob.glob(f"gpurun_out/r06j/kt_step_{tag}/**/*kernel_stats.csv", recursive=True)
    if not fs: continue
    print("==", tag)
    for r in list(csv.DictReader(open(fs[0])))[:16]:
        print(f"  {r['Name'][:110]:110s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us {r['Percentage']}%")
PY
