#!/bin/bash
cp gim_amd/lib/libgimhip.so /tmp/keep.so
for d in "$@"; do
  cp gim_amd/lib/$d/libgimhip.so gim_amd/lib/libgimhip.so
  echo "$d: $(python tools/microbench_cm.py --bf16 --planted 2>&1 | tail -1)"
done
cp /tmp/keep.so gim_amd/lib/libgimhip.so
