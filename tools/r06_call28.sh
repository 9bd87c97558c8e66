# (round 6, closed: a row loop with look-ahead in bwd_src_kernel -- EGC_SRC_Q row groups per wavefront, the next group's pointers and first indices requested ahead, as agg_fast_kernel does -- measured 50.6 -> 57.1 / 59.3 / 68.3 us at 224 / H4 / B4 with 2 / 4 / 8 groups (records + LDS strip; 68 -> 89 registers) and 30.2 -> 27.2 us at 296 / H8 / B4: reverted)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06v; mkdir -p $O
cd $R; timeout 1800 python -m pytest tests/test_backward_gpu.py tests/test_backward_golden.py tests/test_fuzz_gpu.py tests/test_nets_golden.py tests/test_train_golden.py tests/test_native_ext.py tests/test_relational.py tests/test_determinism_gpu.py -x -q 2>&1 | tail -3; cd /tmp
export EGC_SMALL_ONLY=molhiv EGC_NO_NATIVE_TRAIN=1
for sh in "224,4,4,add+mean+max,1,lay" "296,8,4,symadd,1,lay"; do
for q in 1 2 4 8; do
  rm -rf $O/kt_v
  EGC_SRC_Q=$q EGC_STEP_SHAPE="$sh" rocprofv3 --kernel-trace --stats -d $O/kt_v -o kt --output-format csv -- python3 $R/tools/batch_train_step_time.py > /dev/null 2>&1
  f=$(find $O/kt_v -name "*kernel_stats.csv" | head -1)
  echo "$sh Q=$q $(python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'bwd_src' in r['Name']: print(r['Name'][10:50], 'calls', r['Calls'], 'avg %.2f us' % (float(r['AverageNs'])/1e3))
")"
done; done
unset EGC_NO_NATIVE_TRAIN
rm -rf $O/kt_v
rocprofv3 --kernel-trace --stats -d $O/kt_v -o kt --output-format csv -- python3 $R/tools/training_step_time.py > /dev/null 2>&1
python3 -c "
import csv,glob
f=glob.glob('$O/kt_v/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'bwd_src' in r['Name']: print('config2', r['Name'][10:50], 'calls', r['Calls'], 'avg %.2f us' % (float(r['AverageNs'])/1e3))
"
rm -rf $O/kt_v
