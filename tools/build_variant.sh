#!/bin/bash
# Build a variant of the HIP library for a same-box A/B: tools/build_variant.sh <name> "<extra -D flags>" <parts...>
# Recompiles only the listed parts of cmad_hip.hip (see its header comment) with the extra flags and links them with the in-tree
# objects of the other parts into ab_libs/<name>.so (ab_libs/ is git-ignored but travels to the GPU box).  The in-tree library
# is never touched: select the variant with CMAD_HIP_LIB=ab_libs/<name>.so (cmad_amd/_lib.py).
set -e
NAME=$1; FLAGS=$2; shift 2
CSRC=cmad_amd/csrc
mkdir -p ab_libs/obj_$NAME
OBJS=""
for k in $(seq 0 11); do
  if [[ " $* " == *" $k "* ]]; then
    /opt/rocm/bin/hipcc -O3 -std=c++20 --offload-arch=gfx950 -fPIC -DCM_PART=$k $FLAGS -c $CSRC/cmad_hip.hip -o ab_libs/obj_$NAME/part$k.o 2>/dev/null &
    OBJS="$OBJS ab_libs/obj_$NAME/part$k.o"
  else
    OBJS="$OBJS $CSRC/cmad_hip_part$k.o"
  fi
  OBJS="$OBJS $CSRC/cmad_hip_hnn_part$k.o"        # the HNN build's objects are linked as they are
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab_libs/$NAME.so $OBJS
rm -rf ab_libs/obj_$NAME
echo ab_libs/$NAME.so
