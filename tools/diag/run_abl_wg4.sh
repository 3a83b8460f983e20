# usage: bash tools/diag/run_abl_wg4.sh <layer prefix> <ablation names...>   (timing only; the ablated builds compute wrong results)
L=$1; shift
echo "product"; python tools/time_wgrad.py 32 $L 2>&1 | grep " ms "
for n in "$@"; do echo "ablation $n"; KMSR_LIB_OVERRIDE=$PWD/tools/diag/libkmsr_abl_wg4_$n.so python tools/time_wgrad.py 32 $L 2>&1 | grep " ms "; done
