#!/bin/bash
# usage (on the GPU box): bash tools/diag/run_abl_bf16.sh [names...]
python tools/time_bf16_body.py
for n in ${@:-nomfma nodma nostore nores nodma_nostore_nores}; do KMSR_LIB_OVERRIDE=$PWD/tools/diag/libkmsr_bf16_$n.so python tools/time_bf16_body.py; done
