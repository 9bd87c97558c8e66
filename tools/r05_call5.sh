#!/bin/bash
# round 5, GPU call 5: kernel times of the 4-block training step on the molhiv batch of 2,048 (GraphBatch path and COO path)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05h
mkdir -p $O
EGC_SMALL_ONLY=molhiv timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/tools/batch_train_step_time.py > $O/step_under_rocprof.log 2>&1
find $O -name "*kernel_trace.csv" -delete
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r05h/kt/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:28]:
    print(f'{r["Name"][:110]:110s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:8.1f} us  total {float(r["TotalDurationNs"])/1e6:8.2f} ms')
PY
