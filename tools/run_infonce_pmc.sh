# SQ counters of the fused InfoNCE kernels at B = 16384 (separate --pmc passes, no trace domains) -> gpurun_out/infonce_pmc.txt
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/ipmc/$tag -- python3 $R/tools/prof_infonce.py 16384 3 > /dev/null 2>&1
done
cd $R; python tools/pmc_summary.py gpurun_out/ipmc strip_kernel > gpurun_out/infonce_pmc.txt; cat gpurun_out/infonce_pmc.txt; rm -rf gpurun_out/ipmc
python bench.py --steps 20 --warmup 5 --no_cpu_baseline --profile_steps 0 2>/dev/null | wc -l
MCL_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --profile_steps 0 2>/dev/null | wc -l
