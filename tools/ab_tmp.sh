#!/bin/bash
mkdir -p gpurun_out
export MCL_C1F_DEPTH_LARGE=0
MCL_C1F_DEPTH=0 python -m pytest tests/test_backbone_gpu.py -x -q -m gpu -k "conv1x1" > gpurun_out/ab_tests.txt 2>&1
tail -n 1 gpurun_out/ab_tests.txt
for cfg in "MCL_C1F_DEPTH=4" "MCL_C1F_DEPTH=0" "MCL_C1F_DEPTH=0 MCL_C1F_WM_MIN_S=1000000000"; do echo "== $cfg"; env $cfg python tools/bench_dense_layer.py --only conv1x1_fwd 2>&1 | grep -v amdgpu; done
bash tools/ab_env_step.sh "MCL_C1F_DEPTH=4" "MCL_C1F_DEPTH=0" "MCL_C1F_DEPTH=0 MCL_C1F_WM_MIN_S=1000000000" "MCL_C1F_DEPTH=4" "MCL_C1F_DEPTH=0" "MCL_C1F_DEPTH=0 MCL_C1F_WM_MIN_S=1000000000"
