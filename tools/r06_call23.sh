cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06s; mkdir -p $O
cd $R; timeout 2400 python -m pytest tests/test_gemm_gpu.py tests/test_parity_gpu.py tests/test_backward_gpu.py tests/test_native_ext.py tests/test_nets_golden.py -x -q 2>&1 | tail -5
EGC_SCAN_ROWS=13192,52771,211084 python3 tools/gemm_rows_scan.py 2>&1 | grep -v amdgpu | tee $O/gemm_rows_scan.log
for sh in "224,4,4,add+mean+max,1,lay" "296,8,4,symadd,1,lay"; do
  EGC_SMALL_ONLY=molhiv EGC_STEP_SHAPE="$sh" python3 $R/tools/batch_train_step_time.py 2>&1 | grep -v amdgpu | head -2
done
