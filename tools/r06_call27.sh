cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06u; mkdir -p $O
cd $R; timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_full_size_gpu.py tests/test_backward_gpu.py tests/test_fuzz_gpu.py tests/test_nets_golden.py tests/test_partition_gpu.py -x -q 2>&1 | tail -3; cd /tmp
for sh in "224,4,4,add+mean+max,1,lay" "296,8,4,symadd,1,lay"; do
  EGC_SMALL_ONLY=molhiv EGC_STEP_SHAPE="$sh" python3 $R/tools/batch_train_step_time.py 2>&1 | grep -v amdgpu | head -2
done
python3 $R/bench.py --workload mag --no-cpu-baseline --steps 30 --warmup 5 2>&1 | grep "\[mag\]" | tail -1
