# (round 6 ablation: needs the var_abl_* builds of egc_backward.hip with the temporary EGC_ABL_NO_DW / EGC_ABL_NO_REC hooks -- removed again after the measurement; numbers in DESIGN.md section 3.7)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06q; mkdir -p $O
for v in "" abl_norec; do
  lib=$R/egc_amd/lib/libegc_hip.so; [ -n "$v" ] && lib=$R/egc_amd/lib/var_$v.so
  rm -rf $O/kt_v
  EGC_NO_NATIVE_TRAIN=1 EGC_HIP_LIB=$lib rocprofv3 --kernel-trace --stats -d $O/kt_v -o kt --output-format csv -- python3 $R/tools/training_step_time.py > /dev/null 2>&1
  python3 -c "
import csv,glob
f=glob.glob('$O/kt_v/**/*kernel_stats.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:9]: print('[$v]', r['Name'][:70], 'calls', r['Calls'], 'avg %.2f us' % (float(r['AverageNs'])/1e3))
"
done
rm -rf $O/kt_v
