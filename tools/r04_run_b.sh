mkdir -p gpurun_out/r04
python tools/list_step_kernels.py --batch 32 --image 64 --genes 171 > gpurun_out/r04/step_kernels_b.txt 2>&1
timeout 1500 python -m pytest tests/test_own_kernels_gpu.py tests/test_backbone_gpu.py tests/test_engine_gpu.py -q -m gpu 2>&1 | tail -30 > gpurun_out/r04/pytest_b.txt
python bench.py --no_cpu_baseline > gpurun_out/r04/bench_b.json 2> gpurun_out/r04/bench_b.err
for v in "1" "0" "1" "0"; do MCL_FOLD_BN1_FIX=$v python bench.py --steps 60 --warmup 10 --no_cpu_baseline --profile_steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fold=$v', d['ms_per_step'])"; done > gpurun_out/r04/ab_fold.txt
python tools/diag_cfg1_spot_noise.py > gpurun_out/r04/diag_cfg1_spot_noise.txt 2>&1
tail -4 gpurun_out/r04/pytest_b.txt; cat gpurun_out/r04/ab_fold.txt
