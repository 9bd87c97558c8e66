# round 6, call 3: tests of the ADVICE fixes; MFMA-pipe counters of the long-k GEMM at the ogbn-mag shape
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06e; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_fused_bwd_gpu.py tests/test_determinism_gpu.py tests/test_fused_tile_gpu.py -x -q 2>&1 | tail -15
cd /tmp
rocprofv3 -L > $O/counters.txt 2>&1
grep -o "SQ_[A-Z_]*MFMA[A-Z_]*\|SQ_VALU_MFMA[A-Z_]*\|GRBM_GUI_ACTIVE\|SQ_INSTS_VALU_MFMA[A-Z_0-9]*" $O/counters.txt | sort -u | head -40
EGC_REPS=100 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY -d $O/pmc_magk_mfma -o pmc --output-format csv -- python3 $R/tools/gemm_time.py --mag > $O/magk_mfma.log 2>&1
EGC_REPS=100 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $O/pmc_magk_grbm -o pmc --output-format csv -- python3 $R/tools/gemm_time.py --mag > $O/magk_grbm.log 2>&1
EGC_REPS=100 rocprofv3 --kernel-trace --stats -d $O/magk_kt -o kt --output-format csv -- python3 $R/tools/gemm_time.py --mag > $O/magk_kt.log 2>&1
find $O -name "*kernel_trace.csv" -delete
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in ("pmc_magk_mfma", "pmc_magk_grbm"):
    fs = glob.glob(f"gpurun_out/r06e/{d}/**/*counter_collection.csv", recursive=True)
    if not fs: print(d, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(d, k, {c: (len(v), sum(v) / len(v)) for c, v in cs.items()})
for f in glob.glob("gpurun_out/r06e/magk_kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3)
PY
tail -n 2 $O/magk_kt.log
