#!/bin/bash
# usage: tools/diag/clock_w4.sh <lib or ""> <name>   -- sustained shader clock of the F(4x4) body layer: GRBM_GUI_ACTIVE / 8 XCDs / kernel duration
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export KMSR_LIB_OVERRIDE=$1
N=$2; OUT=gpurun_out/clk_$N; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/time_wino4.py winograd4 > $OUT/trace.out 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -- python3 tools/time_wino4.py winograd4 > $OUT/pmc.out 2>&1
python3 - $OUT $N <<'PY'
import csv, glob, sys, collections
out, name = sys.argv[1], sys.argv[2]
dur = collections.defaultdict(list)
for f in glob.glob(out + '/trace/*/*_kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'conv3x3_wino4_kernel' in r['Kernel_Name']:
            dur[r['Kernel_Name'][48:60]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
cyc = collections.defaultdict(list)
for f in glob.glob(out + '/pmc/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'conv3x3_wino4_kernel' in r['Kernel_Name'] and r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            cyc[r['Kernel_Name'][48:60]].append(float(r['Counter_Value']))
for k in dur:
    d = sorted(dur[k])[len(dur[k]) // 2]; c = sorted(cyc[k])[len(cyc[k]) // 2] if cyc[k] else float('nan')
    print(f"{name:8s} {k}: median {d/1e3:7.1f} us, GRBM_GUI_ACTIVE {c:.4g} (/8 = {c/8:.4g} cycles) -> {c/8/d:.3f} GHz")
PY
