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
# optional extra counter sets (each its own pass), e.g. PMC_EXTRA="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum;TCC_HIT_sum TCC_MISS_sum"
i=0
IFS=';' read -ra SETS <<< "${PMC_EXTRA:-}"
for SET in "${SETS[@]}"; do
  [ -z "$SET" ] && continue
  i=$((i+1)); echo "[pmc_traffic] extra pass $i: $SET"
  timeout -k 10 300 rocprofv3 --pmc $SET --output-format csv -d $OUT/x$i -- "$@" > /dev/null 2> $OUT/x$i.err || echo "[pmc_traffic] extra pass $i failed"
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
    # the same counter per (kernel, grid size): one kernel name can serve launches of different shapes (the degradation kernel runs on
    # the bench's 48-plane batch AND on the north-star unit in one command)
    byg = collections.defaultdict(list)
    for f in glob.glob(f'{out}/{c}/*/*_counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if 'kmsr' in r['Kernel_Name']:
                byg[r['Kernel_Name'][:120] + '|grid=' + r.get('Grid_Size', '?')].append(float(r['Counter_Value']))
    res.setdefault("pmc_by_grid", {})[c] = {k: {"n": len(v), "mean_kb": sum(v) / len(v)} for k, v in byg.items()}
ex = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f'{out}/x*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'kmsr' in r['Kernel_Name']:
            ex[r['Kernel_Name'][:120] + '|grid=' + r.get('Grid_Size', '?')][r['Counter_Name']].append(float(r['Counter_Value']))
res["extra_by_grid"] = {k: {c: {"n": len(v), "mean": sum(v) / len(v)} for c, v in d.items()} for k, d in ex.items()}
json.dump(res, open(f'{out}/summary.json', 'w'), indent=1)
for k, d in res["extra_by_grid"].items():
    if 'degrade' in k or 'wino4_kernel<' in k: print('EXTRA', k[-110:], {c: '%.4g' % v["mean"] for c, v in d.items()})
for s in res["stats"][:10]: print('STAT', s)
for c, d in res["pmc"].items():
    for k, v in d.items(): print('PMC', c, k[:70], v)
PY
