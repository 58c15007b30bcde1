mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_layerwise_gpu.py -x -q -m gpu -s > gpurun_out/r04/layerwise_e.txt 2>&1
timeout 1200 python -m pytest tests/test_dist_gpu.py tests/test_engine_gpu.py tests/test_own_kernels_gpu.py -q -m gpu -x 2>&1 | tail -30 > gpurun_out/r04/pytest_e.txt
MCL_FORCE_DIST=1 python bench.py --steps 60 --warmup 10 --no_cpu_baseline --profile_steps 0 > gpurun_out/r04/bench_dp1_seg.json 2> gpurun_out/r04/bench_dp1_seg.err
MCL_FORCE_DIST=1 MCL_DP_SEGMENTS=0 python bench.py --steps 60 --warmup 10 --no_cpu_baseline --profile_steps 0 > gpurun_out/r04/bench_dp1_mono.json 2> gpurun_out/r04/bench_dp1_mono.err
python bench.py --steps 60 --warmup 10 --no_cpu_baseline --profile_steps 0 > gpurun_out/r04/bench_single.json 2>/dev/null
grep -h "passed\|failed" gpurun_out/r04/layerwise_e.txt gpurun_out/r04/pytest_e.txt | tail -3
for f in bench_dp1_seg bench_dp1_mono bench_single; do python -c "import json,sys; d=json.loads([l for l in open('gpurun_out/r04/$f.json') if l.startswith('{')][-1]); print('$f', d['ms_per_step'], d['config'].get('dp_backward_segments'))"; done
