# bench lines of the BASELINE configs other than configs[1] (1 GPU): writes gpurun_out/other_configs.jsonl
mkdir -p gpurun_out; out=gpurun_out/other_configs.jsonl; : > $out
python bench.py --batch 8 --image 112 --genes 785 --steps 100 --warmup 20 --no_cpu_baseline --profile_steps 0 >> $out 2>/dev/null
python bench.py --batch 256 --image 256 --genes 3467 --infonce fp8 --steps 40 --warmup 10 --no_cpu_baseline --profile_steps 0 >> $out 2>/dev/null
python bench.py --batch 256 --image 256 --genes 3467 --infonce fused --steps 40 --warmup 10 --no_cpu_baseline --profile_steps 0 >> $out 2>/dev/null
for enc in vit vit_b16; do
  python bench.py --encoder $enc --image_dim 768 --batch 256 --steps 40 --warmup 10 --no_cpu_baseline --profile_steps 0 >> $out 2>/dev/null
  python bench.py --encoder $enc --image_dim 768 --batch 256 --steps 40 --warmup 10 --no_cpu_baseline --profile_steps 0 --unfused_backbone >> $out 2>/dev/null
done
python bench.py --unfused_backbone --steps 40 --warmup 10 --no_cpu_baseline --profile_steps 0 >> $out 2>/dev/null
cut -c1-520 $out
