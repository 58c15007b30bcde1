# Profiles of the shipped configuration at HEAD -> gpurun_out/head_* (copied into profiles/rNN_* afterwards)
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd /tmp
B="python3 $R/bench.py --no_cpu_baseline --profile_steps 0"
# 1. PMC passes (separate, counters only)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/hp/fetch -- $B --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/hp/write -- $B --steps 3 --warmup 1 > /dev/null 2>&1
# 2. kernel traces: shipped configuration, and every kernel alone (no side streams)
rocprofv3 --kernel-trace -d $R/gpurun_out/hp/kt -o kt -- $B --steps 12 --warmup 4 > /dev/null 2>&1
rocprofv3 --kernel-trace -d $R/gpurun_out/hp/ks -o ks -- $B --serial_lanes --steps 12 --warmup 4 > /dev/null 2>&1
cd $R
python tools/make_traffic_json.py gpurun_out/hp/fetch gpurun_out/hp/write gpurun_out/head_kernel_traffic.json gpurun_out/head_pmc_hbm_traffic.txt > /dev/null
cp gpurun_out/head_kernel_traffic.json profiles/kernel_traffic.json
python tools/rocpd_stats.py $(find gpurun_out/hp/kt -name "*.db" | head -1) gpurun_out/head_kernel_stats.csv --steady 8
python tools/rocpd_stats.py $(find gpurun_out/hp/ks -name "*.db" | head -1) gpurun_out/head_kernel_stats_serial.csv --steady 8
python tools/trace_gaps.py $(find gpurun_out/hp/kt -name "*.db" | head -1) --steady 6 > gpurun_out/head_gpu_idle_gaps.txt
mkdir -p gpurun_out/hp_keep_ks; cp $(find gpurun_out/hp/ks -name "*.db" | head -1) gpurun_out/hp_keep_ks/ks.db
rm -rf gpurun_out/hp
# 3. the bench line itself (with the traffic file just produced and the CPU baseline), and the no-overlap wall times
python bench.py --steps 100 --warmup 20 > gpurun_out/head_bench.json 2> gpurun_out/head_bench.err
cut -c1-400 gpurun_out/head_bench.json
python bench.py --serial_lanes --steps 60 --warmup 10 --no_cpu_baseline --profile_steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('serial lanes (no side streams)', d['ms_per_step'])" > gpurun_out/head_overlap_modes.txt
cat gpurun_out/head_overlap_modes.txt
# 4. micro-benchmarks
python tools/bench_infonce.py --unfused > gpurun_out/head_infonce.jsonl 2>/dev/null
python tools/bench_dense_layer.py --json gpurun_out/head_dense.jsonl > /dev/null 2>&1
python tools/bench_gemm_bf16.py > gpurun_out/head_gemm.jsonl 2>/dev/null
python tools/bench_spot_path.py > gpurun_out/head_spot.json 2>/dev/null
MCL_FUSED_HEAD=0 MCL_GROUP_LAYER_GRADS=0 python tools/bench_spot_path.py >> gpurun_out/head_spot.json 2>/dev/null
MCL_GEMM_SPLITK_MERGE=1 python tools/bench_spot_path.py >> gpurun_out/head_spot.json 2>/dev/null
python tools/bench_proj_head.py > gpurun_out/head_proj_head.jsonl 2>/dev/null
MCL_STAMPS=1 python tools/step_stamps.py --steps 30 > gpurun_out/head_step_stamps.txt 2>/dev/null
MCL_FORCE_DIST=1 python bench.py --steps 100 --warmup 20 --no_cpu_baseline --profile_steps 0 > gpurun_out/head_bench_dp_size1.json 2>/dev/null
MCL_FORCE_DIST=1 MCL_GRAD_WIRE=bf16 MCL_GRAD_BUCKETS=4 python bench.py --steps 60 --warmup 15 --no_cpu_baseline --profile_steps 0 > gpurun_out/head_bench_dp_size1_bf16wire.json 2>/dev/null
# per-layer kernel durations of one step (serial trace)
python tools/layer_times.py $(ls -d gpurun_out/hp_keep_ks/*.db | head -1) conv3x3_fwd conv3x3_bwd bn2_dz conv3x3_wrw conv1x1_fwd "bn1_bwd_kernel<1" "bn1_bwd_kernel<0" "bn1_bwd_kernel<2" bn1_fix "wrw_partial_kernel<1" "wrw_partial_kernel<2" > gpurun_out/head_layer_times_serial.txt 2>/dev/null
bash tools/run_other_configs.sh > /dev/null 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
