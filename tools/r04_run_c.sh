mkdir -p gpurun_out/r04; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="python3 $R/bench.py --no_cpu_baseline --profile_steps 0"
cd /tmp
for v in 1 0; do
  export MCL_FOLD_BN1_FIX=$v
  rocprofv3 --kernel-trace -d $R/gpurun_out/hp/kt$v -o kt -- $B --steps 12 --warmup 4 > /dev/null 2>&1
done
unset MCL_FOLD_BN1_FIX
cd $R
for v in 1 0; do
  python tools/trace_timeline.py $(find gpurun_out/hp/kt$v -name "*.db" | head -1) gpurun_out/r04/timeline_fold$v.txt
  python tools/trace_gaps.py $(find gpurun_out/hp/kt$v -name "*.db" | head -1) --steady 6 > gpurun_out/r04/gaps_fold$v.txt
  python tools/rocpd_stats.py $(find gpurun_out/hp/kt$v -name "*.db" | head -1) gpurun_out/r04/kernel_stats_fold$v.csv --steady 8
done
rm -rf gpurun_out/hp
timeout 900 python -m pytest tests/test_layerwise_gpu.py -x -q -m gpu -s 2>&1 | tail -60 > gpurun_out/r04/layerwise_c.txt
tail -5 gpurun_out/r04/layerwise_c.txt
