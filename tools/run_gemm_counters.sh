# SQ counters of mcl_gemm_bf16 and of the hipBLASLt kernel on the same square problems (separate --pmc passes) -> gpurun_out/r04/gemm_counters.txt
mkdir -p gpurun_out/r04; cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
B="python3 $R/tools/bench_gemm_cube.py ${1:-4096} ${2:-4096} ${3:-4096}"
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/gq/$tag -- $B > /dev/null 2>&1
done
cd $R
python tools/pmc_summary.py gpurun_out/gq gemm_bf16_kernel Cijk > gpurun_out/r04/gemm_counters.txt
rm -rf gpurun_out/gq
cat gpurun_out/r04/gemm_counters.txt
