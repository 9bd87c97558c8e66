# (round 6: occupancy variants of bwd_src_kernel, tools/build_variant.sh srcw6 / srcw8 egc_backward "-DEGC_SRC_WAVES=6|8": no gain at the wide nets, 223 -> 264 / 519 us at config 2)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06t; mkdir -p $O
export EGC_SMALL_ONLY=molhiv EGC_NO_NATIVE_TRAIN=1
for sh in "224,4,4,add+mean+max,1,lay" "296,8,4,symadd,1,lay"; do
for v in "" srcw6 srcw8; do
  lib=$R/egc_amd/lib/libegc_hip.so; [ -n "$v" ] && lib=$R/egc_amd/lib/var_$v.so
  rm -rf $O/kt_v
  EGC_HIP_LIB=$lib EGC_STEP_SHAPE="$sh" rocprofv3 --kernel-trace --stats -d $O/kt_v -o kt --output-format csv -- python3 $R/tools/batch_train_step_time.py > /dev/null 2>&1
  f=$(find $O/kt_v -name "*kernel_stats.csv" | head -1)
  echo "$sh variant=[$v] $(python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'bwd_src' in r['Name']: print('calls', r['Calls'], 'avg %.2f us' % (float(r['AverageNs'])/1e3))
")"
done; done
# config 2 training step
for v in "" srcw6 srcw8; do
  lib=$R/egc_amd/lib/libegc_hip.so; [ -n "$v" ] && lib=$R/egc_amd/lib/var_$v.so
  rm -rf $O/kt_v
  EGC_HIP_LIB=$lib rocprofv3 --kernel-trace --stats -d $O/kt_v -o kt --output-format csv -- python3 $R/tools/training_step_time.py > /dev/null 2>&1
  f=$(find $O/kt_v -name "*kernel_stats.csv" | head -1)
  echo "config2 variant=[$v] $(python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'bwd_src' in r['Name']: print('calls', r['Calls'], 'avg %.2f us' % (float(r['AverageNs'])/1e3))
")"
done
rm -rf $O/kt_v
