for n in "$@"; do
  if [ $n = cur ]; then L=""; else L=$PWD/tools/diag/libkmsr_var_$n.so; fi
  echo "== $n"
  KMSR_LIB_OVERRIDE=$L KMSR_TUNE_B=10 python tools/tune_degrade.py vb 4,8,16 2>&1 | grep rows
  KMSR_LIB_OVERRIDE=$L KMSR_TUNE_B=32 python tools/tune_degrade.py vb 8,16,32 2>&1 | grep rows
  KMSR_LIB_OVERRIDE=$L KMSR_TUNE_B=32 python tools/tune_degrade.py va4 16 2>&1 | grep rows
  KMSR_LIB_OVERRIDE=$L KMSR_TUNE_B=32 python tools/tune_degrade.py va8 16 2>&1 | grep rows
done
