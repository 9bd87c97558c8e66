# round 6: the extremum-gradient records against the arg-byte path on low-degree batches (molhiv b2048: 2.1 entries per row, 224 columns
# -> every entry receives ~100 columns, every record overflows its 12 items)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06rec; rm -rf $O; mkdir -p $O
for cfg in "molhiv:224,4,4,add+mean+max,1,lay" "molhiv:128,8,4,symadd+max+mean,1,lay" "zinc:124,4,4,add+std+max,1,lay" "cifar:128,4,4,symadd+std+max,1,lay"; do
ds=${cfg%%:*}; sh=${cfg#*:}
for norec in 0 1; do
  rm -rf $O/kt_v
  if [ $norec = 1 ]; then export EGC_BWD_NO_REC=1; else unset EGC_BWD_NO_REC; fi
  EGC_NO_FUSED_BWD=1 EGC_SMALL_ONLY=$ds EGC_STEP_SHAPE="$sh" rocprofv3 --kernel-trace --stats -d $O/kt_v -o kt --output-format csv -- python3 $R/tools/batch_train_step_time.py > $O/log 2>&1
  f=$(find $O/kt_v -name "*kernel_stats.csv" | head -1)
  echo "$cfg no_rec=$norec $(grep 'COO' $O/log | head -1 | sed 's/.*hipGraph replay/replay/') $(python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'bwd_dst' in r['Name'] or 'bwd_src' in r['Name'] or 'bwd_records' in r['Name']: print(r['Name'][10:32], '%.1f us;' % (float(r['AverageNs'])/1e3), end=' ')
")"
done; done
unset EGC_BWD_NO_REC
rm -rf $O/kt_v
