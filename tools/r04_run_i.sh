mkdir -p gpurun_out/r04
timeout 1800 python -m pytest tests/test_generic_conv_gpu.py -q -m gpu -s 2>&1 | grep -v "^  \|^$" | tail -60 > gpurun_out/r04/pytest_i.txt
for enc in "resnet50 2048" "res18 512"; do set -- $enc
  python bench.py --encoder $1 --image_dim $2 --steps 20 --warmup 5 --no_cpu_baseline --profile_steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 own kernels (bf16)', d['ms_per_step'], 'ms/step', d['value'], 'spots/s')"
  python bench.py --encoder $1 --image_dim $2 --steps 20 --warmup 5 --no_cpu_baseline --profile_steps 0 --unfused_backbone 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 stock PyTorch-ROCm modules (bf16 autocast)', d['ms_per_step'], 'ms/step', d['value'], 'spots/s')"
done > gpurun_out/r04/bench_resnets.txt 2>&1
python bench.py --backbone_dtype f32 --steps 10 --warmup 3 --no_cpu_baseline --profile_steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('densenet121 fp32 activations, own kernels', d['ms_per_step'], 'ms/step')" >> gpurun_out/r04/bench_resnets.txt 2>&1
tail -8 gpurun_out/r04/pytest_i.txt; cat gpurun_out/r04/bench_resnets.txt
