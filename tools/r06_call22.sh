cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06r; mkdir -p $O
for sh in "224,4,4,add+mean+max,1,lay" "296,8,4,symadd,1,lay" "168,8,4,symadd,1,lay"; do
  tag=$(echo $sh | cut -d, -f1); w=molhiv; [ $tag = 168 ] && w=zinc
  rm -rf $O/kt_$tag
  EGC_SMALL_ONLY=$w EGC_STEP_SHAPE="$sh" rocprofv3 --kernel-trace --stats -d $O/kt_$tag -o kt --output-format csv -- python3 $R/tools/batch_train_step_time.py > $O/step_$tag.log 2>&1
  EGC_SMALL_ONLY=$w EGC_STEP_SHAPE="$sh" python3 $R/tools/batch_train_step_time.py 2>&1 | grep -v amdgpu | head -3 | tee $O/step_${tag}_plain.log
done
EGC_SMALL_ONLY=cifar EGC_STEP_SHAPE="168,8,4,symadd,1,lay" python3 $R/tools/batch_train_step_time.py 2>&1 | grep -v amdgpu | head -3 | tee $O/step_cifar168_plain.log
find $O -name "*kernel_trace.csv" -delete
