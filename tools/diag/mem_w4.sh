#!/bin/bash
# usage: tools/diag/mem_w4.sh   -- memory-path counters of the F(4x4) body layer (tools/time_wino4.py), one --pmc pass per set
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/mem_w4; rm -rf $OUT; mkdir -p $OUT
i=0
while read -r SET; do
  [ -z "$SET" ] && continue
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -- python3 tools/time_wino4.py winograd4 > /dev/null 2> $OUT/p$i.err || echo "pass $i failed: $SET"
done <<SETS
TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_TCC_WRITE_REQ TCP_TCC_WRITE_REQ_LATENCY
TCP_PENDING_STALL_CYCLES TCP_PERF_SEL_TOTAL_READ TCP_PERF_SEL_TOTAL_HIT_LRU_READ TCP_GATE_EN1
TCC_HIT TCC_MISS TCC_REQ TCC_READ TCC_WRITE
TA_BUSY TA_BUFFER_TOTAL_CYCLES TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES GRBM_GUI_ACTIVE
SETS
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/p*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'conv3x3_wino4_kernel' in r['Kernel_Name'] and 'pack' not in r['Kernel_Name']:
            agg[r['Kernel_Name'][48:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print('==', k)
    for c, v in sorted(d.items()):
        v = sorted(v); print(f"  {c:34s} median {v[len(v)//2]:.5g}  (n={len(v)})")
PY
