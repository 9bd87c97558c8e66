#!/bin/bash
# round 5, GPU call 7: the reference's batched layer widths on the final binary (times + kernel stats), LDS atomic rates
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05j
mkdir -p $O
cd $R
timeout 600 python3 tools/batch_shapes_time.py > $O/batch_shapes.log 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt_shapes -o kt --output-format csv -- python3 $R/tools/batch_shapes_time.py > /dev/null 2>&1
find $O -name "*kernel_trace.csv" -delete
hipcc -O3 --offload-arch=gfx950 $R/tools/src/lds_atomic_bench.hip -o /tmp/lab 2>/dev/null && /tmp/lab > $O/lds_atomic_bench.log 2>&1
cd $R
grep -v amdgpu $O/batch_shapes.log | cut -c1-300
cat $O/lds_atomic_bench.log
