#!/bin/bash
# A/B in ONE call: non-temporal (aux = 2) policy on the F(4x4) forward kernel's A-operand loads and / or raw-row LDS-DMA
# (variants: -DKMSR_W4_A_AUX=2 / -DKMSR_W4_RAW_AUX=2 builds of conv3x3_wino4.hip with the two aux literals replaced by those macros)
cd $GRAFT_REPO_ROOT
for v in product w4nt_a w4nt_raw w4nt_both product; do
  echo "== $v"
  if [ $v = product ]; then timeout -k 10 120 python tools/time_wino4.py winograd4 2>&1 | grep -E "relu|residual"
  else KMSR_LIB_OVERRIDE=$PWD/tools/diag/libkmsr_var_$v.so timeout -k 10 120 python tools/time_wino4.py winograd4 2>&1 | grep -E "relu|residual"; fi
done
