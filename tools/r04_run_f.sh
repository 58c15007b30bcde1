mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_layerwise_gpu.py -x -q -m gpu -s > gpurun_out/r04/layerwise_f.txt 2>&1
timeout 1200 python -m pytest tests/test_dist_gpu.py -q -m gpu -x 2>&1 | tail -30 > gpurun_out/r04/pytest_f.txt
for n in 2 4 1 2 1; do MCL_FORCE_DIST=1 MCL_DP_SEGMENTS=$n python bench.py --steps 60 --warmup 10 --no_cpu_baseline --profile_steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dp size-1 segments=$n', d['ms_per_step'], d['config'].get('dp_backward_segments'))"; done > gpurun_out/r04/ab_dp_segments.txt
python bench.py --steps 60 --warmup 10 --no_cpu_baseline --profile_steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('single graph', d['ms_per_step'])" >> gpurun_out/r04/ab_dp_segments.txt
grep -h "passed\|failed" gpurun_out/r04/layerwise_f.txt gpurun_out/r04/pytest_f.txt | tail -3; cat gpurun_out/r04/ab_dp_segments.txt
