mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/w3pmc/f -- python3 $R/tools/bench_dense_layer.py --only conv3x3_wrw_det > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/w3pmc/k -- python3 $R/tools/bench_dense_layer.py --only conv3x3_wrw_det > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/w3pmc/f/**/*counter_collection.csv', recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'conv3x3_wrw_ky' in r['Kernel_Name'] or 'wrw_merge' in r['Kernel_Name']:
            agg[(r['Kernel_Name'][:40], r['Grid_Size'])].append(float(r['Counter_Value']))
    for k, v in sorted(agg.items()):
        print(k, 'n=%d' % len(v), 'FETCH_SIZE avg %.0f KiB (x2 correction: %.1f MB)' % (sum(v)/len(v), sum(v)/len(v)*1024/2/1e6))
for f in glob.glob('gpurun_out/w3pmc/k/**/*kernel_trace.csv', recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'conv3x3_wrw_ky' in r['Kernel_Name'] or 'wrw_merge' in r['Kernel_Name']:
            agg[r['Kernel_Name'][:40]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    for k, v in sorted(agg.items()):
        print(k, 'n=%d' % len(v), 'durations us (launch order):', [round(x, 1) for x in v[:: max(1, len(v)//24)]])
PY
rm -rf gpurun_out/w3pmc
