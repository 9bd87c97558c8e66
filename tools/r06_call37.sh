# round 6: HBM-side traffic of the config-2 training step's kernels (FETCH_SIZE x 2 + WRITE_SIZE, separate passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06pmc2; rm -rf $O; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $O/$c -o pmc --output-format csv -- python3 $R/tools/training_step_time.py > /dev/null 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/r06pmc2/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "egc::" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:72]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, d in acc.items():
    f = sum(d.get("FETCH_SIZE", [0])) / max(1, len(d.get("FETCH_SIZE", [0])))
    w = sum(d.get("WRITE_SIZE", [0])) / max(1, len(d.get("WRITE_SIZE", [0])))
    rows.append((2 * f * 1024 + w * 1024, k, f, w, len(d.get("FETCH_SIZE", []))))
for tot, k, f, w, n in sorted(rows, reverse=True)[:12]:
    print(f"  {k:72s} launches {n:4d}  FETCH x2 {2 * f * 1024 / 1e6:8.1f} MB  WRITE {w * 1024 / 1e6:7.1f} MB")
PY
find $O -name "*counter_collection.csv" -size +5M -delete
