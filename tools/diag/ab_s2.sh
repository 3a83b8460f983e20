#!/bin/bash
# A/B of stride-2 direct conv variants in ONE call (boxes differ): product, then each tools/diag/libkmsr_var_s2_*.so
cd $GRAFT_REPO_ROOT
echo "== product"; timeout -k 10 120 python tools/time_s2.py
for v in tools/diag/libkmsr_var_s2_*.so; do echo "== $v"; KMSR_LIB_OVERRIDE=$PWD/$v timeout -k 10 120 python tools/time_s2.py; done
echo "== product again"; timeout -k 10 120 python tools/time_s2.py
