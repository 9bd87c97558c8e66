# round 6: the profile collection kept under profiles/r06_* (run on the GPU box: gpurun -- 'bash tools/r06_final.sh')
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_final; rm -rf $O; mkdir -p $O
B="python3 $R/bench.py --no-cpu-baseline --no-other-configs --steps 200 --warmup 20"
# (1) the headline under the kernel trace: the dominant kernel's average duration (must agree with roofline.launch_ms of the bench line)
rocprofv3 --kernel-trace --stats -d $O/kt2 -o kt --output-format csv -- $B > $O/bench_config2_under_rocprof.json 2> $O/kt2.log
# (2) its HBM-side traffic: FETCH_SIZE and WRITE_SIZE in separate passes
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc --output-format csv -- $B --steps 20 --warmup 5 > /dev/null 2> $O/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o pmc --output-format csv -- $B --steps 20 --warmup 5 > /dev/null 2> $O/pmc_write.log
# (3) the one-launch batch kernels (forward: ZINC / molhiv / CIFAR; training step: both launches) -- kernel times and traffic
for w in zinc molhiv cifar; do
  EGC_TILE_ONLY=$w rocprofv3 --kernel-trace --stats -d $O/fused_$w -o kt --output-format csv -- python3 $R/tools/fused_tile_time.py > $O/fused_$w.log 2>&1
done
for w in molhiv cifar; do
  EGC_TILE_ONLY=$w rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fused_${w}_fetch -o pmc --output-format csv -- python3 $R/tools/fused_tile_time.py > /dev/null 2>&1
  EGC_TILE_ONLY=$w rocprofv3 --pmc WRITE_SIZE -d $O/pmc_fused_${w}_write -o pmc --output-format csv -- python3 $R/tools/fused_tile_time.py > /dev/null 2>&1
done
for w in molhiv zinc; do
  EGC_SMALL_ONLY=$w rocprofv3 --kernel-trace --stats -d $O/step_$w -o kt --output-format csv -- python3 $R/tools/batch_train_step_time.py > $O/step_$w.log 2>&1
done
EGC_SMALL_ONLY=molhiv rocprofv3 --pmc FETCH_SIZE -d $O/pmc_step_fetch -o pmc --output-format csv -- python3 $R/tools/batch_train_step_time.py > /dev/null 2>&1
EGC_SMALL_ONLY=molhiv rocprofv3 --pmc WRITE_SIZE -d $O/pmc_step_write -o pmc --output-format csv -- python3 $R/tools/batch_train_step_time.py > /dev/null 2>&1
# (3b) the reference's own wide batched nets: the CSR-path training step, kernel times + the un-profiled step time
for sh in "224,4,4,add+mean+max,1,lay" "296,8,4,symadd,1,lay" "168,8,4,symadd,1,lay"; do
  tag=$(echo $sh | cut -d, -f1); w=molhiv; [ $tag = 168 ] && w=zinc
  EGC_SMALL_ONLY=$w EGC_STEP_SHAPE="$sh" rocprofv3 --kernel-trace --stats -d $O/wide_$tag -o kt --output-format csv -- python3 $R/tools/batch_train_step_time.py > /dev/null 2>&1
  EGC_SMALL_ONLY=$w EGC_STEP_SHAPE="$sh" python3 $R/tools/batch_train_step_time.py 2>&1 | grep -v amdgpu > $O/wide_step_$tag.log
done
python3 $R/tools/gemm_rows_scan.py 2>&1 | grep -v amdgpu > $O/gemm_rows_scan.log
EGC_XT_SWEEP=1 python3 $R/tools/xt_wide_time.py 2>&1 | grep -v amdgpu > $O/xt_wide_sweep.log
find $O -name "*kernel_trace.csv" -delete
cd $R
# (4) un-profiled logs: eager / replayed training step, layer shapes per path, std / var modes, eager host profile
python3 tools/batch_train_step_time.py > $O/batch_train_step.log 2>&1
python3 tools/batch_shapes_time.py > $O/batch_shapes.log 2>&1
python3 tools/stdvar_modes.py > $O/stdvar_modes.log 2>&1
TOP=25 python3 tools/eager_batch_step_profile.py > $O/eager_step_native.log 2>&1
EGC_NO_NATIVE_TRAIN=1 TOP=25 python3 tools/eager_batch_step_profile.py > $O/eager_step_python.log 2>&1
python3 tools/eager_step_variants.py > $O/eager_step_variants.log 2>&1
hipcc -O2 --offload-arch=gfx950 tools/src/lds_atomic_order.hip -o /tmp/lao 2>/dev/null && /tmp/lao > $O/lds_atomic_order.log 2>&1
# (5) the bench in the driver's form, last (what BENCH_r06.json should look like)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; wc -c $O/bench_line.json; cp bench_detail.json $O/
du -sh $O
