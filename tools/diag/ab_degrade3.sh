# degradation kernel: product build against its timing-only ablations (no fold FMAs / no row DMA), config-2-like and north-star shapes
for n in "$@"; do
  if [ $n = cur ]; then L=""; else L=$PWD/tools/diag/libkmsr_var_$n.so; fi
  echo "== $n"
  KMSR_LIB_OVERRIDE=$L KMSR_TUNE_B=10 python tools/tune_degrade.py vb 4,8 2>&1 | grep rows
  KMSR_LIB_OVERRIDE=$L KMSR_TUNE_B=32 python tools/tune_degrade.py vb 16 2>&1 | grep rows
done
