cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06q; mkdir -p $O
cd $R
EGC_XT_SWEEP=1 EGC_XT_SHAPES="52771,224,272;241600,168,116;52771,296,180;169343,136,184;16000,304,368" python3 tools/xt_wide_time.py 2>&1 | grep -v amdgpu | awk '/^N=/{c=0; print; next} {c++; if (c<=4) print}' | tee $O/xt_sweep2.log
timeout 600 python -m pytest tests/test_gemm_gpu.py -x -q -k "weight_grad" 2>&1 | tail -3
