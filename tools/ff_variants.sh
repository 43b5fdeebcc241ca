#!/bin/bash
# builds fine_fused variants on the GPU box is not possible (hipcc exists there too, actually) -> build there
cd $GRAFT_REPO_ROOT
for v in "" "-DFF_NOPREFETCH" "-DFF_NOSYNC"; do
  touch gim_amd/csrc/fine_fused.hip
  GIM_HIPCC_EXTRA="$v" python -m gim_amd.build > /dev/null 2>&1
  echo "variant [$v]"; python tools/bench_fine.py 1500 2>&1 | grep "fused  "
done
