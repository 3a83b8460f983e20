#!/bin/bash
# config 5 (512 patches in one call): issue-side counters of the encoder kernels, one pass per counter set
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04/c5pmc; rm -rf $OUT; mkdir -p $OUT
i=0
for SET in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1)); echo "[pmc_c5] pass $i: $SET"
  timeout -k 10 200 rocprofv3 --pmc $SET --output-format csv -d $OUT/x$i -- python3 $R/tools/time_kernelgan.py 512 > $OUT/x$i.out 2> $OUT/x$i.err || echo "pass $i failed"
done
python3 - $OUT <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/x*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'kmsr' in r['Kernel_Name']:
            agg[(r['Kernel_Name'][28:75], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()):
        print(f"   {c:28s} {sum(x)/len(x):16.0f}  (n={len(x)})")
PY
