# config-2 shaped degradation (48 planes: B=16 C=3 needs a 3-band bank -> B=10 C=5 = 50 planes) and the north-star unit, by variant
for n in "$@"; do
  if [ $n = cur ]; then L=""; else L=$PWD/tools/diag/libkmsr_var_$n.so; fi
  echo "== $n"; KMSR_LIB_OVERRIDE=$L KMSR_TUNE_B=10 python tools/tune_degrade.py vb 4,8,16 2>&1 | grep rows; KMSR_LIB_OVERRIDE=$L KMSR_TUNE_B=32 python tools/tune_degrade.py vb 16 2>&1 | grep rows
done
