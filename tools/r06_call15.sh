cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06n; mkdir -p $O
cd $R; python3 tools/gemm_rows_scan.py 2>&1 | grep -v amdgpu | tee $O/gemm_rows_scan.log
