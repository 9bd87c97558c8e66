# round 6, call 4: the batch training block as one C++ autograd node -- tests, eager step time with and without it
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06f; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_native_ext.py tests/test_fused_bwd_gpu.py tests/test_hipgraph_gpu.py tests/test_callers.py tests/test_nets_golden.py tests/test_train_golden.py -x -q 2>&1 | tail -15
for m in native python; do
  if [ $m = python ]; then export EGC_NO_NATIVE_TRAIN=1; else unset EGC_NO_NATIVE_TRAIN; fi
  TOP=12 python3 tools/eager_batch_step_profile.py 2>&1 | grep -v amdgpu.ids | head -24 > $O/eager_zinc_$m.log
  head -1 $O/eager_zinc_$m.log
  python3 tools/batch_train_step_time.py > $O/step_$m.log 2>&1; grep -v amdgpu.ids $O/step_$m.log | tail -n 8
done
