cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06u; mkdir -p $O
for q in 1 2 4; do
  rm -rf $O/kt_v
  EGC_AGG_Q=$q rocprofv3 --kernel-trace --stats -d $O/kt_v -o kt --output-format csv -- python3 $R/tools/agg_rows_per_wave.py > $O/aggq_$q.log 2>&1
  f=$(find $O/kt_v -name "*kernel_stats.csv" | head -1)
  python3 -c "
import csv,sys,re
for r in sorted(csv.DictReader(open('$f')), key=lambda r: r['Name']):
    if 'agg_fast' in r['Name']:
        m=re.search(r'agg_fast_kernel<(\d+), (\d+), (\d+), egc::(\w+)<([^>]*)>', r['Name'])
        print('Q=$q', m.group(1,2,3,4), m.group(5)[:28], 'calls', r['Calls'], 'avg %.2f us' % (float(r['AverageNs'])/1e3))
"
done
rm -rf $O/kt_v
