for n in ps_soff ps_soff_nop; do
  L=$PWD/tools/diag/libkmsr_var_$n.so
  echo "== $n"; KMSR_LIB_OVERRIDE=$L python tools/diag/dbg_ps.py 2>&1 | cut -c1-200
done
