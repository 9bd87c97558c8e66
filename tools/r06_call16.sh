cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06o; mkdir -p $O
cd $R; timeout 2400 python -m pytest tests/test_gemm_gpu.py tests/test_callers.py tests/test_native_ext.py tests/test_nets_golden.py tests/test_train_golden.py tests/test_backward_golden.py tests/test_fused_bwd_gpu.py -x -q 2>&1 | tail -8; cd /tmp
for sh in "224,4,4,add+mean+max,1,lay" "296,8,4,symadd,1,lay"; do
  tag=$(echo $sh | cut -d, -f1)
  EGC_SMALL_ONLY=molhiv EGC_STEP_SHAPE="$sh" python3 $R/tools/batch_train_step_time.py 2>&1 | grep -v amdgpu | head -2 | tee $O/step_${tag}_plain.log
done
python3 $R/tools/batch_train_step_time.py 2>&1 | grep -v amdgpu | head -8 | tee $O/step_128.log
