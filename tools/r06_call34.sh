# round 6: HBM-side traffic (FETCH_SIZE x 2 + WRITE_SIZE, separate passes; KiB units, gfx950 correction) of the kernels of the 224 / H4 / B4
# and 296 / H8 / B4 training steps on the final binary
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06pmc; rm -rf $O; mkdir -p $O
for sh in "224,4,4,add+mean+max,1,lay" "296,8,4,symadd,1,lay"; do
  tag=$(echo $sh | cut -d, -f1)
  for c in FETCH_SIZE WRITE_SIZE; do
    EGC_SMALL_ONLY=molhiv EGC_STEP_SHAPE="$sh" rocprofv3 --pmc $c -d $O/${tag}_$c -o pmc --output-format csv -- python3 $R/tools/batch_train_step_time.py > /dev/null 2>&1
  done
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for tag in ("224", "296"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(f"gpurun_out/r06pmc/{tag}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "egc::" in r["Kernel_Name"]:
                    acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows = []
    for k, d in acc.items():
        f = sum(d.get("FETCH_SIZE", [0])) / max(1, len(d.get("FETCH_SIZE", [0])))
        w = sum(d.get("WRITE_SIZE", [0])) / max(1, len(d.get("WRITE_SIZE", [0])))
        rows.append((2 * f * 1024 + w * 1024, k, f, w, len(d.get("FETCH_SIZE", []))))
    print("==", tag)
    for tot, k, f, w, n in sorted(rows, reverse=True)[:12]:
        print(f"  {k:70s} launches {n:5d}  FETCH x2 {2 * f * 1024 / 1e6:8.1f} MB  WRITE {w * 1024 / 1e6:7.1f} MB  total {tot / 1e6:8.1f} MB")
PY
find $O -name "*counter_collection.csv" -size +5M -delete
