cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06m; mkdir -p $O
cd $R
EGC_XT_SWEEP=1 python3 tools/xt_wide_time.py 2>&1 | grep -v amdgpu | tee $O/xt_sweep.log
