#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b; mkdir -p $O
timeout 900 python3 -m pytest tests/test_fused_tile_gpu.py -x -q -m gpu --timeout 240 -k "messy" > $O/t_messy.log 2>&1; echo "messy rc=$?"; tail -15 $O/t_messy.log
timeout 900 python3 -m pytest tests/test_fused_tile_gpu.py -q -m gpu --timeout 400 -k "not messy" > $O/t_rest.log 2>&1; echo "rest rc=$?"; tail -15 $O/t_rest.log
timeout 600 python3 tools/batch_shapes_time.py > $O/batch_shapes.log 2>&1; cut -c1-330 $O/batch_shapes.log
