python tools/time_wino4.py winograd4 2>&1 | grep "winograd4 "
for n in "$@"; do echo "ablation $n"; KMSR_LIB_OVERRIDE=$PWD/tools/diag/libkmsr_abl_w4_$n.so python tools/time_wino4.py winograd4 2>&1 | grep "winograd4 .*us"; done
