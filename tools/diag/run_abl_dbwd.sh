#!/bin/bash
# usage (on the GPU box): bash tools/diag/run_abl_dbwd.sh [names...]
python tools/time_degrade_bwd.py
for n in ${@:-noload notaps nofma nolds nosigma}; do echo "== $n"; KMSR_LIB_OVERRIDE=$PWD/tools/diag/libkmsr_dbwd_$n.so python tools/time_degrade_bwd.py; done
