#!/bin/bash
# round 5, GPU call 8: the batch training step on the final binary -- kernel stats per workload, counters of the two one-launch kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05k
mkdir -p $O
for w in molhiv zinc; do
  EGC_SMALL_ONLY=$w timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt_$w -o kt --output-format csv -- python3 $R/tools/batch_train_step_time.py > $O/step_${w}_under_rocprof.log 2>&1
done
find $O -name "*kernel_trace.csv" -delete
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY" "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $set | cut -d' ' -f1)
  EGC_SMALL_ONLY=molhiv timeout 300 rocprofv3 --pmc $set -d $O/pmc_$n -o pmc --output-format csv -- python3 $R/tools/batch_train_step_time.py > /dev/null 2> $O/pmc_$n.log
done
cd $R
python3 - <<'PY'
import csv, glob, collections
out = open("gpurun_out/r05k/pmc_fused_fwd_bwd.csv", "w"); w = csv.writer(out)
w.writerow(["kernel", "counter", "average per launch", "launches"])
for f in sorted(glob.glob("gpurun_out/r05k/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "fused_tile" not in kn: continue
        k = "fused_tile_kernel MODE 1 (backward)" if kn.split("(")[0].rstrip().endswith("0, 1>") else "fused_tile_kernel MODE 0 (forward)"
        acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for (k, c), v in sorted(acc.items()): w.writerow([k, c, round(v / cnt[(k, c)]), cnt[(k, c)]])
out.close()
print(open("gpurun_out/r05k/pmc_fused_fwd_bwd.csv").read())
for wl in ("molhiv", "zinc"):
    f = glob.glob(f"gpurun_out/r05k/kt_{wl}/**/*kernel_stats.csv", recursive=True)[0]
    print(wl)
    for r in list(csv.DictReader(open(f)))[:22]:
        print(f'  {r["Name"][:100]:100s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
