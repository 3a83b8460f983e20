cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for mode in relu res; do
  for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    rm -rf gpurun_out/pf; rocprofv3 --pmc $C --output-format csv -d gpurun_out/pf -- python3 tools/one_vreg.py winograd_vreg 6 $mode > /dev/null 2>&1
    python3 - "$mode" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for f in glob.glob('gpurun_out/pf/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'winov_kernel' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items(): print(sys.argv[1], k, 'n=%d' % len(v), 'mean=%.6g' % (sum(v) / len(v)))
PY
  done
done
