mkdir -p gpurun_out
set -x
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/a_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/a_pytest.log
tail -5 gpurun_out/a_pytest.log
python bench.py --steps 100 --warmup 20 > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err
cat gpurun_out/a_bench.json
python tools/bench_infonce.py --unfused > gpurun_out/a_infonce.jsonl 2> gpurun_out/a_infonce.err
python tools/bench_dense_layer.py --json gpurun_out/a_dense.jsonl > gpurun_out/a_dense.log 2>&1
python tools/bench_gemm_bf16.py > gpurun_out/a_gemm.jsonl 2> gpurun_out/a_gemm.err
for enc in vit vit_b16; do
  python bench.py --encoder $enc --image_dim 768 --batch 256 --steps 30 --warmup 5 --no_cpu_baseline >> gpurun_out/a_vit.jsonl 2>> gpurun_out/a_vit.err
  python bench.py --encoder $enc --image_dim 768 --batch 256 --steps 30 --warmup 5 --no_cpu_baseline --unfused_backbone >> gpurun_out/a_vit.jsonl 2>> gpurun_out/a_vit.err
done
cat gpurun_out/a_vit.jsonl
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/a_smoke.log 2>&1; tail -2 gpurun_out/a_smoke.log
