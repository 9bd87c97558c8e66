cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06p; mkdir -p $O
export EGC_SMALL_ONLY=molhiv EGC_STEP_SHAPE="224,4,4,add+mean+max,1,lay"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES -d $O/pmc_sq1 -o pmc --output-format csv -- python3 $R/tools/batch_train_step_time.py > /dev/null 2> $O/pmc_sq1.log
timeout 600 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_BRANCH -d $O/pmc_sq2 -o pmc --output-format csv -- python3 $R/tools/batch_train_step_time.py > /dev/null 2> $O/pmc_sq2.log
cd $R
python3 - <<'PY'
import csv, glob, collections
for tag in ("pmc_sq1", "pmc_sq2"):
    fs = glob.glob(f"gpurun_out/r06p/{tag}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in fs:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if any(s in k for s in ("bwd_dst_fast", "bwd_src_kernel", "agg_fast_kernel", "xt_gemm_kernel")):
                acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(tag, k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "dispatches", len(next(iter(d.values()))))
PY
find $O -name "*counter_collection.csv" -size +20M -delete
