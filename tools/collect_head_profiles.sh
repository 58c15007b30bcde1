#!/bin/bash
# gpurun_out/head_* (written by tools/run_head_profiles.sh on the GPU box) -> profiles/rNN_*  :  tools/collect_head_profiles.sh r03
set -e
R=${1:-r03}; cd $(dirname $0)/..
cp gpurun_out/head_bench.json profiles/${R}_bench_dp1.json
cp gpurun_out/head_kernel_stats.csv profiles/${R}_bench_dp1_kernel_stats.csv
cp gpurun_out/head_kernel_stats_serial.csv profiles/${R}_bench_dp1_kernel_stats_serial.csv
cp gpurun_out/head_bench_dp_size1.json profiles/${R}_bench_dp_size1.json
cp gpurun_out/head_bench_dp_size1_bf16wire.json profiles/${R}_bench_dp_size1_bf16wire_4buckets.json
cp gpurun_out/other_configs.jsonl profiles/${R}_bench_other_configs.jsonl
cp gpurun_out/head_dense.jsonl profiles/${R}_dense_layer_microbench.jsonl
cp gpurun_out/head_gpu_idle_gaps.txt profiles/${R}_gpu_idle_gaps.txt
cp gpurun_out/head_infonce.jsonl profiles/${R}_infonce_microbench.jsonl
cp gpurun_out/head_layer_times_serial.txt profiles/${R}_layer_times_serial.txt
cp gpurun_out/head_overlap_modes.txt profiles/${R}_overlap_modes.txt
cp gpurun_out/head_pmc_hbm_traffic.txt profiles/${R}_pmc_hbm_traffic.txt
cp gpurun_out/head_spot.json profiles/${R}_spot_branch.json
cp gpurun_out/head_gemm.jsonl profiles/${R}_gemm_bf16_microbench.jsonl
cp gpurun_out/head_proj_head.jsonl profiles/${R}_proj_head_microbench.jsonl
cp gpurun_out/head_step_stamps.txt profiles/${R}_step_stamps.txt
cp gpurun_out/head_kernel_traffic.json profiles/kernel_traffic.json
ls -la profiles/${R}_* | wc -l
