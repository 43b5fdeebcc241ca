#!/bin/bash
# Development build with phase stamps (-DGIM_TIMING, csrc/gim_common.h: GIM_TT) of the kernels that carry them, into gim_amd/lib/timing/libgimhip.so
# (the product library and its objects are untouched; every other object is re-used from gim_amd/lib/obj):
#   bash tools/build_timing.sh && GIM_LIB=gim_amd/lib/timing/libgimhip.so python tools/kernel_timing.py
set -e
cd "$(dirname "$0")/.."
python -m gim_amd.build > /dev/null
mkdir -p gim_amd/lib/timing
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -DGIM_TIMING"
objs=$(ls gim_amd/lib/obj/*.o)
for f in ${TIMING_FILES:-bneck_fused bneck_tail conv_igemm}; do
  /opt/rocm/bin/hipcc $F -c gim_amd/csrc/$f.hip -o gim_amd/lib/timing/$f.o &
  /opt/rocm/bin/hipcc $F -DGIM_HALF_KIND=1 -c gim_amd/csrc/$f.hip -o gim_amd/lib/timing/${f}_f16.o &
  objs=$(echo "$objs" | grep -v "/obj/$f.o" | grep -v "/obj/${f}_f16.o")
  objs="$objs gim_amd/lib/timing/$f.o gim_amd/lib/timing/${f}_f16.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gim_amd/lib/timing/libgimhip.so $objs
echo built gim_amd/lib/timing/libgimhip.so
