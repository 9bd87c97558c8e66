#!/bin/bash
# round 5, GPU call 1: the weightings round trip's cost (f3 closure), counters of the wide GEMM shapes and the config-5
# aggregate, the shader clock inside the long-k GEMM, the reference's batched shapes before this round's kernel work
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05a
mkdir -p $O
cd $R
python3 tools/f3_roundtrip_cost.py > $O/f3_default.log 2>&1
for v in gemm_nowstore gemm_now agg_wrow0; do EGC_HIP_LIB=$R/egc_amd/lib/var_$v.so python3 tools/f3_roundtrip_cost.py > $O/f3_$v.log 2>&1; done
python3 tools/batch_shapes_time.py > $O/batch_shapes_before.log 2>&1
EGC_HIP_LIB=$R/egc_amd/lib/var_gemmk_stamps.so python3 tools/gemm_time.py --mag > $O/clock_mag.log 2>&1
cd /tmp
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_wide_fetch -o pmc --output-format csv -- python3 $R/tools/gemm_time.py --wide > /dev/null 2> $O/pmc_wide_fetch.log
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_wide_write -o pmc --output-format csv -- python3 $R/tools/gemm_time.py --wide > /dev/null 2> $O/pmc_wide_write.log
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_mag_fetch -o pmc --output-format csv -- python3 $R/bench.py --workload mag --steps 10 --warmup 3 > /dev/null 2> $O/pmc_mag_fetch.log
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_mag_write -o pmc --output-format csv -- python3 $R/bench.py --workload mag --steps 10 --warmup 3 > /dev/null 2> $O/pmc_mag_write.log
find $O -name "*kernel_trace.csv" -delete
tail -n 20 $O/f3_default.log $O/f3_agg_wrow0.log $O/f3_gemm_now.log $O/f3_gemm_nowstore.log
cat $O/batch_shapes_before.log | cut -c1-400
tail -5 $O/clock_mag.log
du -sh $O
