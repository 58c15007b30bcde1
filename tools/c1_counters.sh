#!/bin/bash
# memory-side counters of the conv1x1 forward kernels alone at S = 401408, C = 224 -> gpurun_out/c1_counters.txt
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for mode in 0; do
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
    tag=$(echo $set | cut -d' ' -f1)
    rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/c1c$mode/$tag -- python3 $R/tools/c1_single.py 401408 224 256 4 > /dev/null 2>&1
  done
done
cd $R
python tools/pmc_summary.py gpurun_out/c1c0 conv1x1_fwd_kernel > gpurun_out/c1_counters.txt
rm -rf gpurun_out/c1c0
cat gpurun_out/c1_counters.txt
