#!/bin/bash
# round-end records in ONE call: full GPU suite, the default bench line, rocprof kernel stats of the bench / train / config-5 commands
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04/final && O=$GRAFT_REPO_ROOT/gpurun_out/r04/final
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -5 $O/gpu_tests.log; exit 1; }
tail -1 $O/gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 600 python bench.py > $O/bench_line.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o b -- python3 $GRAFT_REPO_ROOT/bench.py --no-secondary --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o t -- python3 $GRAFT_REPO_ROOT/bench.py --workload train --steps 5 --warmup 2 --no-cpu-baseline > $O/train_prof.json 2> $O/train_prof.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -o c -- python3 $GRAFT_REPO_ROOT/tools/time_kernelgan.py 512 > $O/c5.log 2>&1
rm -f $O/*/*_kernel_trace.csv $O/*/*_agent_info.csv
ls $O $O/bench $O/train $O/c5
