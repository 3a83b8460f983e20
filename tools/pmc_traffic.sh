#!/bin/bash
# usage: tools/pmc_traffic.sh <outdir-name> -- <command...>
# HBM traffic per launch: one --kernel-trace --stats pass, then FETCH_SIZE and WRITE_SIZE in their own --pmc passes
# (MI355X_MICROARCH.md: the two do not fit one pass; FETCH_SIZE unit KB, x2 on gfx950 for wide coalesced reads).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
NAME=$1; shift 2
OUT=gpurun_out/$NAME
rm -rf $OUT; mkdir -p $OUT
echo "[pmc_traffic] trace"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- "$@" > $OUT/trace.out 2> $OUT/trace.err
for C in FETCH_SIZE WRITE_SIZE; do
  echo "[pmc_traffic] $C"
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- "$@" > /dev/null 2> $OUT/$C.err || echo "[pmc_traffic] $C failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys, json
out = sys.argv[1]
res = {"stats": [], "pmc": {}}
for f in sorted(glob.glob(f'{out}/trace/*/*_kernel_stats.csv')):
    for r in csv.DictReader(open(f)):
        res["stats"].append({"name": r['Name'][:120], "calls": int(r['Calls']), "avg_ns": float(r['AverageNs']), "pct": float(r['Percentage'])})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for f in glob.glob(f'{out}/{c}/*/*_counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name'][:120]].append(float(r['Counter_Value']))
    res["pmc"][c] = {k: {"n": len(v), "mean_kb": sum(v) / len(v)} for k, v in agg.items() if 'kmsr' in k}
json.dump(res, open(f'{out}/summary.json', 'w'), indent=1)
for s in res["stats"][:10]: print('STAT', s)
for c, d in res["pmc"].items():
    for k, v in d.items(): print('PMC', c, k[:70], v)
PY
