cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R; timeout 1500 python -m pytest tests/test_gemm_gpu.py tests/test_backward_gpu.py tests/test_nets_golden.py tests/test_native_ext.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -3
EGC_SMALL_ONLY=cifar EGC_STEP_SHAPE="168,8,4,symadd,1,lay" python3 tools/batch_train_step_time.py 2>&1 | grep -v amdgpu | head -2
EGC_SMALL_ONLY=zinc EGC_STEP_SHAPE="168,8,4,symadd,1,lay" python3 tools/batch_train_step_time.py 2>&1 | grep -v amdgpu | head -2
