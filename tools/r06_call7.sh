cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06i; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_determinism_gpu.py tests/test_fused_bwd_gpu.py tests/test_native_ext.py tests/test_nets_golden.py tests/test_train_golden.py tests/test_backward_golden.py -x -q 2>&1 | tail -5
cd /tmp
for lib in new r05; do
  if [ $lib = r05 ]; then export EGC_HIP_LIB=$R/egc_amd/lib/var_r05csr.so; else unset EGC_HIP_LIB; fi
  for w in molhiv zinc; do
    EGC_SMALL_ONLY=$w timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_${lib}_step_$w -o kt --output-format csv -- python3 $R/tools/batch_train_step_time.py > $O/step_${lib}_$w.log 2>&1
  done
done
unset EGC_HIP_LIB
EGC_SMALL_ONLY=molhiv rocprofv3 --pmc FETCH_SIZE -d $O/pmc_step_fetch -o pmc --output-format csv -- python3 $R/tools/batch_train_step_time.py > /dev/null 2>&1
EGC_SMALL_ONLY=molhiv rocprofv3 --pmc WRITE_SIZE -d $O/pmc_step_write -o pmc --output-format csv -- python3 $R/tools/batch_train_step_time.py > /dev/null 2>&1
find $O -name "*kernel_trace.csv" -delete
cd $R
python3 - <<'PY'
import csv, glob, collections
for lib in ("new", "r05"):
    for w in ("step_molhiv","step_zinc"):
        fs = glob.glob(f"gpurun_out/r06i/kt_{lib}_{w}/**/*kernel_stats.csv", recursive=True)
        for r in csv.DictReader(open(fs[0])):
            if "fused_tile_kernel" in r["Name"]:
                print(lib, w, r["Name"][-70:-30], "calls", r["Calls"], "avg %.2f us" % (float(r["AverageNs"])/1e3), "min %.2f" % (float(r["MinNs"])/1e3))
tot = collections.defaultdict(float)
for c, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    f = glob.glob(f"gpurun_out/r06i/pmc_step_{d}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "fused_tile_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c: acc[r["Kernel_Name"][-45:]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        b = sum(v) / len(v) * 1024 * (2 if c == "FETCH_SIZE" else 1); tot[k] += b
        print(c, k, len(v), "%.1f MB" % (b / 1e6))
print({k: "%.1f MB" % (v / 1e6) for k, v in tot.items()})
PY
