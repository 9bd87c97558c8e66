cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06m; mkdir -p $O
cd $R
echo "== new"; python3 tools/xt_wide_time.py 2>&1 | grep -v amdgpu | tee $O/xt_wide_new.log
echo "== prev"; EGC_HIP_LIB=$R/egc_amd/lib/variants/libegc_hip_prev.so python3 tools/xt_wide_time.py 2>&1 | grep -v amdgpu | tee $O/xt_wide_prev.log
timeout 900 python -m pytest tests/test_backward_gpu.py tests/test_gemm_gpu.py tests/test_native_ext.py -x -q 2>&1 | tail -3
