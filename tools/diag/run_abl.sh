python tools/time_body_res.py 2>&1 | grep vreg
for n in nodma nostore nobarrier noepi; do echo "ablation $n"; KMSR_LIB_OVERRIDE=$PWD/tools/diag/libkmsr_abl_$n.so python tools/time_body_res.py 2>&1 | grep vreg; done
