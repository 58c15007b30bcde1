# SQ counters of the training step's kernels (separate --pmc passes, counters only) -> gpurun_out/step_sq_counters.txt
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --no_cpu_baseline --profile_steps 0 --steps 3 --warmup 1"
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/ssq/$tag -- $B > /dev/null 2>&1
done
cd $R
python tools/pmc_summary.py gpurun_out/ssq conv3x3_bwd_rows_kernel conv3x3_bwd_kernel bn1_bwd_kernel conv1x1_fwd_kernel conv3x3_fwd_rows_kernel conv3x3_fwd_kernel conv3x3_wrw_ky wrw_partial_kernel bn2_dz_kernel adam_table_kernel gemm_kernel > gpurun_out/step_sq_counters.txt
rm -rf gpurun_out/ssq
head -60 gpurun_out/step_sq_counters.txt
