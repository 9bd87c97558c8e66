# round 6, call 6: the code-shaped workload (test + which path serves it + times); where the training step of the reference's own
# 224 / H4 / B4 molhiv net goes (kernel stats)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06h; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_callers.py tests/test_fused_bwd_gpu.py tests/test_native_ext.py -x -q 2>&1 | tail -5
EGC_SHAPES_ONLY="code-shaped" python3 tools/batch_shapes_time.py 2>&1 | grep -v amdgpu.ids | tee $O/shapes_code.log
cd /tmp
for sh in "224,4,4,sum+mean+max,0" "296,8,4,symnorm,1" "168,8,4,symnorm,1"; do
  tag=$(echo $sh | cut -d, -f1)
  w=molhiv; [ $tag = 168 ] && w=zinc
  EGC_SMALL_ONLY=$w EGC_STEP_SHAPE="$sh" rocprofv3 --kernel-trace --stats -d $O/kt_step_$tag -o kt --output-format csv -- python3 $R/tools/batch_train_step_time.py > $O/step_$tag.log 2>&1
  grep -v "amdgpu.ids\|rocprofv3\|Opened" $O/step_$tag.log | tail -n 3
done
find $O -name "*kernel_trace.csv" -delete
cd $R
python3 - <<'PY'
import csv, glob
for tag in ("224", "296", "168"):
    fs = glob.glob(f"gpurun_out/r06h/kt_step_{tag}/**/*kernel_stats.csv", recursive=True)
    if not fs: continue
    print("==", tag)
    for r in list(csv.DictReader(open(fs[0])))[:14]:
        print(f"  {r['Name'][:100]:100s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us {r['Percentage']}%")
PY
