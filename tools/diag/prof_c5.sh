#!/bin/bash
# per-kernel, per-grid durations of config 5 at batch 512 (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04/c5; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o c5 -- python3 $R/tools/time_kernelgan.py 512 > $OUT/run.log 2>&1
grep batch $OUT/run.log
python3 - $OUT <<'PY'
import csv, collections, sys
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1] + '/c5_kernel_trace.csv')):
    agg[(r['Kernel_Name'][:80], r.get('Grid_Size_X', r.get('Grid_Size', '')))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:9]:
    print(f"{sum(v)/len(v):9.1f} us x{len(v):3d}  grid {k[1]:>9s}  {k[0]}")
PY
