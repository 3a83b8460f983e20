# A/B in ONE gpurun call: product F(4x4,3x3) forward kernel vs the DPP-edge variant (tools/diag/libkmsr_w4_dpp.so, -DKMSR_W4_DPP_EDGES):
# timing (tools/time_wino4.py), then the LDS / VALU counters of both in their own --pmc passes.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04/ab_w4_dpp; rm -rf $OUT; mkdir -p $OUT
for V in product dpp; do
  if [ $V = dpp ]; then export KMSR_LIB_OVERRIDE=$PWD/tools/diag/libkmsr_w4_dpp.so; else unset KMSR_LIB_OVERRIDE; fi
  echo "== $V: timing"; python3 tools/time_wino4.py winograd4 2>&1 | grep winograd4
  echo "== $V: counters"
  timeout -k 10 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/$V -- python3 tools/time_wino4.py winograd4 > /dev/null 2> $OUT/$V.err || echo "pmc pass failed"
  python3 - "$OUT/$V" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'conv3x3_wino4_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg): print(f"PMC {k:24s} n={len(agg[k])} mean per launch = {sum(agg[k]) / len(agg[k]):.4g}")
PY
done
