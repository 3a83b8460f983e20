# A/B/C in ONE gpurun call: the F(4x4,3x3) forward kernel with its 3 raw-row DMA requests back to back (product) / 2 pairs apart / 4 pairs apart
for V in product 1 2 product 1 2; do
  if [ $V = product ]; then unset KMSR_LIB_OVERRIDE; else export KMSR_LIB_OVERRIDE=$PWD/tools/diag/libkmsr_w4_spread$V.so; fi
  echo "== $V"; python3 tools/time_wino4.py winograd4 2>&1 | grep winograd4
done
