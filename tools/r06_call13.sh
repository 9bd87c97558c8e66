cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06m; mkdir -p $O
cd $R
EGC_XT_SHAPES="736389,352,208;1939743,352,208;169343,136,184;169343,256,320;100000,224,272" python3 tools/xt_wide_time.py 2>&1 | grep -v amdgpu | tee $O/xt_big.log
EGC_XT_SWEEP=1 EGC_XT_SHAPES="736389,352,208" python3 tools/xt_wide_time.py 2>&1 | grep -v amdgpu | head -8 | tee -a $O/xt_big.log
