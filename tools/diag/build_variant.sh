#!/bin/bash
# usage: tools/diag/build_variant.sh <name> <path-to-conv3x3_wino4.hip variant>   -> tools/diag/libkmsr_var_<name>.so (diagnostic A/B builds)
set -e
R=/root/repo/kernel-modeling-super-resolution_amd/csrc
D=/tmp/var_$1; mkdir -p $D
cp $2 $D/conv3x3_wino4.hip
for h in kmsr_common.h kmsr_noise.h; do sed 's#"../../include/kmsr.h"#"/root/repo/include/kmsr.h"#' $R/$h > $D/$h; done
(cd $D && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize -c conv3x3_wino4.hip -o conv3x3_wino4.o)
OBJS=$(ls $R/*.o | grep -v conv3x3_wino4.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS $D/conv3x3_wino4.o -o /root/repo/tools/diag/libkmsr_var_$1.so
echo built $1
