cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06u; mkdir -p $O
export EGC_SMALL_ONLY=molhiv EGC_NO_NATIVE_TRAIN=1
for sh in "224,4,4,add+mean+max,1,lay" "296,8,4,symadd,1,lay"; do
for q in 1 2 4 8; do
  rm -rf $O/kt_v
  EGC_AGG_Q=$q EGC_STEP_SHAPE="$sh" rocprofv3 --kernel-trace --stats -d $O/kt_v -o kt --output-format csv -- python3 $R/tools/batch_train_step_time.py > /dev/null 2>&1
  f=$(find $O/kt_v -name "*kernel_stats.csv" | head -1)
  echo "$sh Q=$q $(python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'agg_fast' in r['Name']: print('calls', r['Calls'], 'avg %.2f us' % (float(r['AverageNs'])/1e3))
")"
done; done
# config 2 forward (headline kernel) and config 3 CSR path
for q in 1 2 4; do
  EGC_AGG_Q=$q python3 $R/bench.py --no-cpu-baseline --no-other-configs --steps 100 --warmup 10 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('config2 Q=$q', d['ms_per_step'], d['roofline']['launch_ms'])"
done
rm -rf $O/kt_v
