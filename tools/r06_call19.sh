cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06q; mkdir -p $O
cd $R; timeout 2400 python -m pytest tests/test_backward_gpu.py tests/test_backward_golden.py tests/test_fuzz_gpu.py tests/test_nets_golden.py tests/test_train_golden.py tests/test_native_ext.py tests/test_determinism_gpu.py -x -q 2>&1 | tail -5; cd /tmp
for sh in "224,4,4,add+mean+max,1,lay" "296,8,4,symadd,1,lay" "168,8,4,symadd,1,lay"; do
  tag=$(echo $sh | cut -d, -f1); w=molhiv; [ $tag = 168 ] && w=zinc
  rm -rf $O/kt_$tag
  EGC_SMALL_ONLY=$w EGC_STEP_SHAPE="$sh" rocprofv3 --kernel-trace --stats -d $O/kt_$tag -o kt --output-format csv -- python3 $R/tools/batch_train_step_time.py > $O/step_$tag.log 2>&1
  EGC_SMALL_ONLY=$w EGC_STEP_SHAPE="$sh" python3 $R/tools/batch_train_step_time.py 2>&1 | grep -v amdgpu | head -2 | tee $O/step_${tag}_plain.log
  python3 -c "
import csv,glob
f=glob.glob('$O/kt_$tag/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'bwd_dst_fast' in r['Name'] or 'bwd_src' in r['Name']: print('$tag', r['Name'][:60], 'calls', r['Calls'], 'avg %.2f us' % (float(r['AverageNs'])/1e3))
"
done
find $O -name "*kernel_trace.csv" -delete
