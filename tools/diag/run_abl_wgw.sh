#!/bin/bash
# usage (on the GPU box): bash tools/diag/run_abl_wgw.sh [names...]
python tools/time_wgrad_body.py 32
for n in ${@:-novalu nostage nobar novalu_nostage noload nostore}; do echo "== $n"; KMSR_LIB_OVERRIDE=$PWD/tools/diag/libkmsr_wgw_$n.so python tools/time_wgrad_body.py 32; done
