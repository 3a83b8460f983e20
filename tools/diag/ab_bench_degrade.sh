# the bench's own degradation figures (config-2 launch back to back / in the step, north-star unit) by library build, in ONE call
for n in "$@"; do
  if [ $n = cur ]; then L=""; else L=$PWD/tools/diag/libkmsr_var_$n.so; fi
  KMSR_LIB_OVERRIDE=$L python bench.py --no-secondary --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['roofline_degrade']; b=d['roofline_degrade_northstar']
print('$n', 'step ms', d['ms_per_step'], '| config2 back-to-back us', a['kernel_ms']*1e3, 'in-step', a.get('kernel_ms_in_step_bracket',0)*1e3, 'frac', a['frac'], '| northstar us', b['kernel_ms']*1e3, 'frac', b['frac'])"
done
