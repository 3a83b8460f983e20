#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_mem
rm -rf $OUT; mkdir -p $OUT
CMD="python3 bench.py --workload degrade --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_REQ_sum --output-format csv -d $OUT/a -- $CMD > /dev/null 2> $OUT/a.err || echo "pass a failed"
rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum --output-format csv -d $OUT/b -- $CMD > /dev/null 2> $OUT/b.err || echo "pass b failed"
# TA_* counters: at most two per pass -- four in one --pmc set made rocprofv3 abort with "Request exceeds the capabilities of
# the hardware to collect" (round 1, gpurun_out/pmc_mem/c.err); the profiler's config check, not a kernel fault
rocprofv3 --pmc TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum --output-format csv -d $OUT/c -- $CMD > /dev/null 2> $OUT/c.err || echo "pass c failed"
rocprofv3 --pmc TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum --output-format csv -d $OUT/c2 -- $CMD > /dev/null 2> $OUT/c2.err || echo "pass c2 failed"
rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_GATE_EN1_sum TCP_TCR_TCP_STALL_CYCLES_sum --output-format csv -d $OUT/d -- $CMD > /dev/null 2> $OUT/d.err || echo "pass d failed"
rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/e -- $CMD > /dev/null 2> $OUT/e.err || echo "pass e failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -- $CMD > /dev/null 2> $OUT/f.err || echo "pass f failed"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/g -- $CMD > /dev/null 2> $OUT/g.err || echo "pass g failed"
python3 - <<'PY'
import csv, glob, collections
for d in ['a', 'b', 'c', 'c2', 'd', 'e', 'f', 'g']:
    fs = glob.glob(f'gpurun_out/pmc_mem/{d}/*/*_counter_collection.csv')
    if not fs: print(d, 'no output'); continue
    rows = list(csv.DictReader(open(fs[0])))
    agg = collections.defaultdict(list)
    for r in rows:
        if 'rowstream' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        print(d, k, 'n=',len(v), 'mean=%.5g'%(sum(v)/len(v)))
PY
