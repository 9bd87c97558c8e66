#!/bin/bash
# round 5, GPU call 4: instruction-cache counters of the one-launch backward and forward (molhiv batch of 2,048)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05g
mkdir -p $O
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY"; do
  n=$(echo $set | cut -d' ' -f1)
  EGC_SMALL_ONLY=molhiv timeout 300 rocprofv3 --pmc $set -d $O/pmc_$n -o pmc --output-format csv -- python3 $R/tools/batch_train_step_time.py > /dev/null 2> $O/pmc_$n.log
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/r05g/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        if "fused_tile" not in k: continue
        k = ("bwd " if "Li0ELi1EE" in r["Kernel_Name"] else "fwd ") 
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == list(acc[k])[0]: cnt[k] += 1
    for k in acc:
        print(f.split("/")[2], k, {c: round(v / max(cnt[k],1)) for c, v in acc[k].items()})
PY
