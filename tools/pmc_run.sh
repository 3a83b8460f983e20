#!/bin/bash
# usage: tools/pmc_run.sh <outdir-name> <kernel-substring> -- <command...>
# Runs one rocprofv3 --kernel-trace pass and several --pmc passes (each in its own run) and prints per-kernel means.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
NAME=$1; KSUB=$2; shift 3
OUT=gpurun_out/$NAME
rm -rf $OUT; mkdir -p $OUT
echo "[pmc_run] trace" ; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- "$@" > $OUT/trace.out 2> $OUT/trace.err
i=0
while read -r SET; do
  [ -z "$SET" ] && continue
  i=$((i+1)); echo "[pmc_run] pass $i: $SET"
  timeout -k 10 240 rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -- "$@" > /dev/null 2> $OUT/p$i.err || echo "[pmc_run] pass $i failed"
done <<SETS
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM
FETCH_SIZE
WRITE_SIZE
SETS
python3 - "$OUT" "$KSUB" <<'PY'
import csv, glob, collections, sys
out, ksub = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(f'{out}/trace/*/*_kernel_stats.csv')):
    for r in csv.DictReader(open(f)):
        print('STAT', r['Name'][:90], r['Calls'], 'avg_ns', r['AverageNs'], 'pct', r['Percentage'])
for d in sorted(glob.glob(f'{out}/p*/')):
    for f in glob.glob(d + '*/*_counter_collection.csv'):
        agg = collections.defaultdict(list)
        meta = None
        for r in csv.DictReader(open(f)):
            if ksub in r["Kernel_Name"] and "pack" not in r["Kernel_Name"]:
                agg[r['Counter_Name']].append(float(r['Counter_Value'])); meta = r
        for k, v in agg.items():
            print('PMC', k, 'n=%d' % len(v), 'mean=%.6g' % (sum(v) / len(v)))
        if meta: print('META', {k: meta[k] for k in ('VGPR_Count', 'Accum_VGPR_Count', 'SGPR_Count', 'LDS_Block_Size', 'Grid_Size', 'Workgroup_Size') if k in meta})
PY
