# round 6, call 1: the deterministic (input-order) CSR build of the one-launch kernels -- tests, then same-box A/B of kernel times
# against the round-5 build (egc_amd/lib/var_r05csr.so: round 5's egc_fused_tile*.o linked with today's other objects)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06d; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_determinism_gpu.py tests/test_fused_tile_gpu.py tests/test_fused_bwd_gpu.py -x -q 2>&1 | tail -40 > $O/tests.log
cat $O/tests.log
cd /tmp
for lib in new r05; do
  if [ $lib = r05 ]; then export EGC_HIP_LIB=$R/egc_amd/lib/var_r05csr.so; else unset EGC_HIP_LIB; fi
  for w in zinc molhiv cifar; do
    EGC_TILE_ONLY=$w timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_${lib}_$w -o kt --output-format csv -- python3 $R/tools/fused_tile_time.py > /dev/null 2>&1
  done
  for w in molhiv zinc; do
    EGC_SMALL_ONLY=$w timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_${lib}_step_$w -o kt --output-format csv -- python3 $R/tools/batch_train_step_time.py > $O/step_${lib}_$w.log 2>&1
  done
done
unset EGC_HIP_LIB
find $O -name "*kernel_trace.csv" -delete
cd $R
python3 - <<'PY'
import csv, glob
for lib in ("new", "r05"):
    for w in ("zinc","molhiv","cifar","step_molhiv","step_zinc"):
        fs = glob.glob(f"gpurun_out/r06d/kt_{lib}_{w}/**/*kernel_stats.csv", recursive=True)
        if not fs: print(lib, w, "no stats"); continue
        for r in csv.DictReader(open(fs[0])):
            if "fused_tile_kernel" in r["Name"]:
                print(lib, w, r["Name"][-70:-30], "calls", r["Calls"], "avg %.2f us" % (float(r["AverageNs"])/1e3), "min %.2f" % (float(r["MinNs"])/1e3))
PY
tail -n 3 $O/step_new_molhiv.log; tail -n 3 $O/step_r05_molhiv.log
