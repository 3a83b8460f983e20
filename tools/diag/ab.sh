# usage: ab.sh name1 name2 ...   (variants built by build_variant.sh; "cur" = the product library); two rounds to expose drift
for round in 1 2; do
for n in "$@"; do
  if [ $n = cur ]; then L=""; else L=$PWD/tools/diag/libkmsr_var_$n.so; fi
  echo "== $n (round $round)"; KMSR_LIB_OVERRIDE=$L python tools/time_wino4.py winograd4 2>&1 | grep "winograd4 .*us" | tail -2 | cut -c1-45
done; done
