#!/bin/bash
# round 5, GPU call 3: bench line, std layer at five / six wavefronts per SIMD, kernel times and counters of the WIDE one-launch form
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05c
mkdir -p $O
cd $R
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
EGC_HIP_LIB=$R/egc_amd/lib/var_agg6.so timeout 600 python3 bench.py --no-other-configs --no-cpu-baseline > $O/bench_agg6.json 2> $O/bench_agg6.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt_shapes -o kt --output-format csv -- python3 $R/tools/batch_shapes_time.py > $O/batch_shapes_under_rocprof.log 2>&1
for w in molhiv cifar; do
  EGC_TILE_ONLY=$w timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fused_${w}_fetch -o pmc --output-format csv -- python3 $R/tools/fused_tile_time.py > /dev/null 2> $O/pmc_fused_${w}_fetch.log
  EGC_TILE_ONLY=$w timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_fused_${w}_write -o pmc --output-format csv -- python3 $R/tools/fused_tile_time.py > /dev/null 2> $O/pmc_fused_${w}_write.log
done
EGC_SHAPES_ONLY="molhiv EGC-M" timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_ref224_fetch -o pmc --output-format csv -- python3 $R/tools/batch_shapes_time.py > /dev/null 2> $O/pmc_ref224_fetch.log
EGC_SHAPES_ONLY="molhiv EGC-M" timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_ref224_write -o pmc --output-format csv -- python3 $R/tools/batch_shapes_time.py > /dev/null 2> $O/pmc_ref224_write.log
EGC_SHAPES_ONLY="cifar EGC-S" timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES -d $O/pmc_wide168_sq -o pmc --output-format csv -- python3 $R/tools/batch_shapes_time.py > /dev/null 2> $O/pmc_wide168_sq.log
find $O -name "*kernel_trace.csv" -delete
cd $R
python3 tools/batch_shapes_time.py > $O/batch_shapes.log 2>&1
cut -c1-320 $O/batch_shapes.log
python3 - <<'PY'
import json
for f in ("bench.json","bench_agg6.json"):
    try:
        j=json.loads(open("gpurun_out/r05c/"+f).read().strip().splitlines()[-1])
        print(f, "value", j["value"], "ms", j["ms_per_step"], "roofline", j["roofline"]["frac"], "std", j.get("std_layer"))
        oc=j.get("other_configs",{})
        for k,v in oc.items():
            if isinstance(v,dict) and "layer_ms" in v: print("  ",k, "layer_ms %.4f"%v["layer_ms"], "frac", v.get("layer_frac"), v.get("path","")[:60], {p: round(v[p]["new_batch_every_call_ms"],4) for p in ("fused","fused_edge_ptr","tile") if p in v})
            elif isinstance(v,dict): print("  ",k, {kk:vv for kk,vv in v.items() if isinstance(vv,(int,float))})
    except Exception as e: print(f, "failed", e)
PY
du -sh $O
