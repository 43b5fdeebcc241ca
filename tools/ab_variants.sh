#!/bin/bash
# usage: tools/ab_variants.sh "<microbench args>" <dir> [<dir> ...]: runs the microbenchmark with gim_amd/lib/<dir>/libgimhip.so
args=$1; shift
cp gim_amd/lib/libgimhip.so /tmp/keep.so
for d in "$@"; do
  cp gim_amd/lib/$d/libgimhip.so gim_amd/lib/libgimhip.so
  echo "$d: $(python tools/microbench_conv.py $args 2>&1 | tail -1)"
done
cp /tmp/keep.so gim_amd/lib/libgimhip.so
