#!/bin/bash
# round 5, GPU call 6: full GPU suite, the bench line, kernel stats of the bench command and of the batch training step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05i
mkdir -p $O
cd $R
timeout 2400 python3 -m pytest tests -q -m gpu --timeout 1200 -x > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 1200 python3 bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/kt_bench -o kt --output-format csv -- python3 $R/bench.py --no-other-configs --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt_step -o kt --output-format csv -- python3 $R/tools/batch_train_step_time.py > $O/step_under_rocprof.log 2>&1
find $O -name "*kernel_trace.csv" -delete
cd $R
python3 tools/batch_train_step_time.py > $O/batch_train_step.log 2>&1; grep -v amdgpu $O/batch_train_step.log | tail -6
for d in 0 2016 14336; do echo "EGC_FT_DBG=$d: $(EGC_FT_DBG=$d EGC_SMALL_ONLY=molhiv EGC_HIP_LIB=$R/egc_amd/lib/var_ft_stamps.so timeout 300 python3 tools/batch_train_step_time.py 2>&1 | grep 'bwd stamps' | tail -1)"; done > $O/bwd_stamps.log 2>&1
cat $O/bwd_stamps.log | cut -c1-420
python3 - <<'PY'
import json
j=json.loads(open("gpurun_out/r05i/bench.json").read().strip().splitlines()[-1])
print("value", j["value"], "ms", j["ms_per_step"], "roofline", j["roofline"]["frac"])
oc=j.get("other_configs",{})
for k,v in oc.items():
    if "training_step" in k: print(k, {a:b for a,b in v.items() if "ms" in a})
PY
