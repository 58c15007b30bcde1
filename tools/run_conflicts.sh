mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/cf -- python3 $R/tools/bench_dense_layer.py --only "$1" > /dev/null 2>&1
cd $R; python tools/pmc_summary.py gpurun_out/cf conv3x3 wrw_partial bn1_bwd; rm -rf gpurun_out/cf
