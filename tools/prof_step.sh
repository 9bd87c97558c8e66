#!/bin/bash
# Kernel statistics of one script under rocprofv3 (run on the GPU box): tools/prof_step.sh <name> <script> [args]
# -> gpurun_out/<name>/..._kernel_stats.csv, first lines printed
N=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$N
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O -o kt --output-format csv -- python3 $GRAFT_REPO_ROOT/"$@" > $O/stdout.log 2> $O/stderr.log
cd $GRAFT_REPO_ROOT
f=$(find $O -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {r['Percentage']}")
PY
