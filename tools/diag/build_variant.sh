#!/bin/bash
# usage: tools/diag/build_variant.sh <name> <variant source> [csrc file it replaces = conv3x3_wino4.hip] [extra hipcc flags]
#   -> tools/diag/libkmsr_var_<name>.so (diagnostic A/B builds, loaded with KMSR_LIB_OVERRIDE; never part of the product)
set -e
R=/root/repo/kernel-modeling-super-resolution_amd/csrc
F=${3:-conv3x3_wino4.hip}
D=/tmp/var_$1; mkdir -p $D
cp $2 $D/$F
for h in kmsr_common.h kmsr_noise.h kmsr_wino4_xf.h; do sed 's#"../../include/kmsr.h"#"/root/repo/include/kmsr.h"#' $R/$h > $D/$h; done
(cd $D && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize $4 -c $F -o ${F%.hip}.o)
OBJS=$(ls $R/*.o | grep -v "/${F%.hip}.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS $D/${F%.hip}.o -o /root/repo/tools/diag/libkmsr_var_$1.so
echo built $1
