# round 6: where the extremum-gradient records stop paying -- ldb N / E (columns an entry receives on average) against time with and without
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06rec; rm -rf $O; mkdir -p $O
for norec in 0 1; do
  if [ $norec = 1 ]; then export EGC_BWD_NO_REC=1; else unset EGC_BWD_NO_REC; fi
  rm -rf $O/kt_v
  rocprofv3 --kernel-trace --stats -d $O/kt_v -o kt --output-format csv -- python3 $R/tools/arxiv_wide_step.py > $O/log 2>&1
  f=$(find $O/kt_v -name "*kernel_stats.csv" | head -1)
  echo "arxiv 136/184 no_rec=$norec $(grep 'block forward' $O/log | tr '\n' ' ') $(python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'bwd_dst' in r['Name'] or 'bwd_src' in r['Name']: print(r['Name'][10:36], '%.1f us;' % (float(r['AverageNs'])/1e3), end=' ')
")"
  rm -rf $O/kt_v
  rocprofv3 --kernel-trace --stats -d $O/kt_v -o kt --output-format csv -- python3 $R/tools/training_step_time.py > $O/log 2>&1
  f=$(find $O/kt_v -name "*kernel_stats.csv" | head -1)
  echo "config2 no_rec=$norec $(grep 'fwd+bwd' $O/log) $(python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'bwd_dst' in r['Name'] or 'bwd_src' in r['Name']: print(r['Name'][10:36], '%.1f us;' % (float(r['AverageNs'])/1e3), end=' ')
")"
done
unset EGC_BWD_NO_REC
for sh in "224,4,4,add+mean+max,1,lay" "296,8,4,symadd,1,lay"; do
  EGC_SMALL_ONLY=molhiv EGC_STEP_SHAPE="$sh" python3 $R/tools/batch_train_step_time.py 2>&1 | grep -v amdgpu | head -1
done
rm -rf $O/kt_v
