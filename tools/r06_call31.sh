# round 6: the training step of the reference's CIFAR nets (EGC-S 168 / H8 / B4 symadd, EGC-M 128 / H4 / B4 symadd, std, max) on the b2048 batch
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06x; rm -rf $O; mkdir -p $O
for sh in "168,8,4,symadd,1,lay" "128,4,4,symadd+std+max,1,lay"; do
  tag=$(echo $sh | cut -d, -f1)
  EGC_SMALL_ONLY=cifar EGC_STEP_SHAPE="$sh" python3 $R/tools/batch_train_step_time.py 2>&1 | grep -v amdgpu | head -3 | tee $O/cifar_step_$tag.log
  EGC_SMALL_ONLY=cifar EGC_STEP_SHAPE="$sh" rocprofv3 --kernel-trace --stats -d $O/kt_$tag -o kt --output-format csv -- python3 $R/tools/batch_train_step_time.py > /dev/null 2>&1
  python3 -c "
import csv,glob
f=glob.glob('$O/kt_$tag/**/*kernel_stats.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:12]: print('  ', r['Name'][:84], 'calls', r['Calls'], 'avg %.1f us' % (float(r['AverageNs'])/1e3))
"
done
find $O -name "*kernel_trace.csv" -delete
