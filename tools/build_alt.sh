#!/bin/bash
# Baseline library for same-box A/Bs (tools/ab_forward.py lib=alt vs lib=new): gim_amd/lib/alt/libgimhip.so = the current objects with the
# listed source files taken from a git revision instead of the working tree.
#   bash tools/build_alt.sh <rev> file1.hip [file2.hip ...]       e.g.  bash tools/build_alt.sh HEAD bneck_fused.hip
set -e
cd "$(dirname "$0")/.."
rev=$1; shift
mkdir -p gim_amd/lib/alt /tmp/gim_alt/gim_amd/csrc /tmp/gim_alt/include
cp gim_amd/csrc/*.h /tmp/gim_alt/gim_amd/csrc/; cp include/gim_hip.h /tmp/gim_alt/include/
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment"
objs=$(ls gim_amd/lib/obj/*.o)
for f in "$@"; do
  b=${f%.hip}
  git show $rev:gim_amd/csrc/$f > /tmp/gim_alt/gim_amd/csrc/$f
  /opt/rocm/bin/hipcc $F -c /tmp/gim_alt/gim_amd/csrc/$f -o gim_amd/lib/alt/$b.o &
  if [ -f gim_amd/lib/obj/${b}_f16.o ]; then /opt/rocm/bin/hipcc $F -DGIM_HALF_KIND=1 -c /tmp/gim_alt/gim_amd/csrc/$f -o gim_amd/lib/alt/${b}_f16.o & fi
  objs=$(echo "$objs" | grep -v "/obj/$b.o" | grep -v "/obj/${b}_f16.o")
  objs="$objs gim_amd/lib/alt/$b.o"
  [ -f gim_amd/lib/obj/${b}_f16.o ] && objs="$objs gim_amd/lib/alt/${b}_f16.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gim_amd/lib/alt/libgimhip.so $objs
rm -f gim_amd/lib/alt/*.o
echo built gim_amd/lib/alt/libgimhip.so "($rev: $*)"
