# r04 diagnosis: which of this round's changes moved the step (same box)
mkdir -p gpurun_out/r04; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="python3 $R/bench.py --no_cpu_baseline --profile_steps 0"
for v in "1" "0"; do MCL_FOLD_BN1_FIX=$v python bench.py --steps 60 --warmup 10 --no_cpu_baseline --profile_steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fold=$v', d['ms_per_step'])"; done > gpurun_out/r04/diag_a.txt
cd /tmp
MCL_SIDE_STREAM=0 MCL_OVERLAP_BRANCHES=0 rocprofv3 --kernel-trace -d $R/gpurun_out/hp/ks -o ks -- $B --steps 12 --warmup 4 > /dev/null 2>&1
cd $R
python tools/rocpd_stats.py $(find gpurun_out/hp/ks -name "*.db" | head -1) gpurun_out/r04/diag_a_kernel_stats_serial.csv --steady 8
rm -rf gpurun_out/hp
cat gpurun_out/r04/diag_a.txt
