# round 6: the clock under the exact-fp32 x^T d kernel (GRBM_GUI_ACTIVE per XCD / kernel duration), ogbn-mag shape and 224 x 272
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06w; rm -rf $O; mkdir -p $O
export EGC_XT_SHAPES="736389,352,208;52771,224,272"
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/tools/xt_wide_time.py > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $O/pmc -o pmc --output-format csv -- python3 $R/tools/xt_wide_time.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $O/pmc2 -o pmc --output-format csv -- python3 $R/tools/xt_wide_time.py > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
st = glob.glob("gpurun_out/r06w/kt/**/*kernel_stats.csv", recursive=True)[0]
dur = {r["Name"][:60]: (float(r["AverageNs"]), int(r["Calls"])) for r in csv.DictReader(open(st)) if "xt_gemm_kernel" in r["Name"]}
for tag in ("pmc", "pmc2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/r06w/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "xt_gemm_kernel" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        avg = {c: sum(v) / len(v) for c, v in d.items()}
        ns = dur.get(k, (0, 0))[0]
        line = f"{k}: {ns / 1e3:.1f} us; " + ", ".join(f"{c} {v:.4g}" for c, v in avg.items())
        if "GRBM_GUI_ACTIVE" in avg and ns: line += f"; clock = GUI_ACTIVE / 8 XCDs / duration = {avg['GRBM_GUI_ACTIVE'] / 8 / ns:.2f} GHz"
        print(line)
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +5M -delete
