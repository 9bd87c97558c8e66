#!/bin/bash
# Collects the rocprofv3 summaries kept under profiles/ (run on the GPU box through gpurun from the repository root:
#   gpurun -- 'bash tools/collect_profiles.sh r02'
# then `python tools/summarize_profiles.py r02` here copies / condenses them into profiles/).
# Counters are collected in their own passes (--pmc never combined with tracing), one counter set per pass.
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
B="python3 $R/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $O/kt2 -o kt --output-format csv -- $B --no-other-configs > $O/bench_config2_under_rocprof.json 2> $O/kt2.log
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- $B > $O/bench_under_rocprof.json 2> $O/kt.log
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc --output-format csv -- $B --no-other-configs --steps 20 --warmup 5 > /dev/null 2> $O/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o pmc --output-format csv -- $B --no-other-configs --steps 20 --warmup 5 > /dev/null 2> $O/pmc_write.log
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES -d $O/pmc_sq -o pmc --output-format csv -- $B --no-other-configs --steps 20 --warmup 5 > /dev/null 2> $O/pmc_sq.log
rocprofv3 --kernel-trace --stats -d $O/train -o kt --output-format csv -- python3 $R/tools/training_step_time.py > $O/train.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_train_fetch -o pmc --output-format csv -- python3 $R/tools/training_step_time.py > /dev/null 2> $O/pmc_train_fetch.log
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_train_write -o pmc --output-format csv -- python3 $R/tools/training_step_time.py > /dev/null 2> $O/pmc_train_write.log
rocprofv3 --kernel-trace --stats -d $O/block -o kt --output-format csv -- python3 $R/tools/block_step_time.py > $O/block.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/small -o kt --output-format csv -- python3 $R/tools/small_batch_step_time.py > $O/small.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/mag -o kt --output-format csv -- python3 $R/bench.py --workload mag --steps 20 --warmup 5 > $O/mag_bench.json 2> $O/mag.log
# round 3: the tile path of configs 3 / 4 (egc_amd.GraphBatch) beside the ordinary per-batch build, per workload
for w in molhiv cifar; do
  EGC_TILE_ONLY=$w rocprofv3 --kernel-trace --stats -d $O/tile_$w -o kt --output-format csv -- python3 $R/tools/batch_tile_time.py > $O/tile_$w.log 2>&1
done
EGC_TILE_ONLY=cifar rocprofv3 --pmc FETCH_SIZE -d $O/pmc_tile_fetch -o pmc --output-format csv -- python3 $R/tools/batch_tile_time.py > /dev/null 2> $O/pmc_tile_fetch.log
EGC_TILE_ONLY=cifar rocprofv3 --pmc WRITE_SIZE -d $O/pmc_tile_write -o pmc --output-format csv -- python3 $R/tools/batch_tile_time.py > /dev/null 2> $O/pmc_tile_write.log
# round 4: the one-launch layer for batches of whole graphs (egc_fused_tile.hip) beside the two-launch tile path and the ordinary path,
# per workload; its HBM-side traffic; the long-k GEMM's traffic at the ogbn-mag shape
for w in molhiv cifar zinc; do
  EGC_TILE_ONLY=$w rocprofv3 --kernel-trace --stats -d $O/fused_$w -o kt --output-format csv -- python3 $R/tools/fused_tile_time.py > $O/fused_$w.log 2>&1
done
for w in molhiv cifar; do
  EGC_TILE_ONLY=$w rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fused_${w}_fetch -o pmc --output-format csv -- python3 $R/tools/fused_tile_time.py > /dev/null 2> $O/pmc_fused_${w}_fetch.log
  EGC_TILE_ONLY=$w rocprofv3 --pmc WRITE_SIZE -d $O/pmc_fused_${w}_write -o pmc --output-format csv -- python3 $R/tools/fused_tile_time.py > /dev/null 2> $O/pmc_fused_${w}_write.log
done
EGC_TILE_ONLY=cifar rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES -d $O/pmc_fused_sq -o pmc --output-format csv -- python3 $R/tools/fused_tile_time.py > /dev/null 2> $O/pmc_fused_sq.log
EGC_TILE_ONLY=cifar rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $O/pmc_fused_lds -o pmc --output-format csv -- python3 $R/tools/fused_tile_time.py > /dev/null 2> $O/pmc_fused_lds.log
EGC_REPS=400 rocprofv3 --kernel-trace --stats -d $O/magk -o kt --output-format csv -- python3 $R/tools/gemm_time.py --mag > $O/magk.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_magk_fetch -o pmc --output-format csv -- python3 $R/tools/gemm_time.py --mag > /dev/null 2> $O/pmc_magk_fetch.log
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_magk_write -o pmc --output-format csv -- python3 $R/tools/gemm_time.py --mag > /dev/null 2> $O/pmc_magk_write.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $O/pmc_magk_sq -o pmc --output-format csv -- python3 $R/tools/gemm_time.py --mag > /dev/null 2> $O/pmc_magk_sq.log
cd $R
python3 tools/gemm_error.py > $O/gemm_error.log 2>&1
python3 tools/gemm_time.py --wide > $O/gemm_time_wide.log 2>&1
python3 tools/host_overhead_time.py > $O/host_overhead.log 2>&1
python3 tools/reference_shapes_time.py > $O/reference_shapes.log 2>&1   # every trained layer shape of the reference, arxiv-shaped graph
python3 tools/gemm_time.py > $O/gemm_time.log 2>&1
python3 bench.py --workload rmag --steps 10 --warmup 3 > $O/rmag_bench.json 2> $O/rmag.err
python3 bench.py > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err
# the raw per-dispatch traces are large: keep the statistics and the counter collections only
find $O -name "*kernel_trace.csv" -delete
du -sh $O
