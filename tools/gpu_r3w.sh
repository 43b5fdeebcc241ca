#!/bin/bash
# round-3 GPU check W: experimental 192 x 128 / 4-wave tile (two workgroups per CU) against the 256 x 256 / 8-wave tile
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for rep in 1 2; do
for t in 0 1; do
  echo "== GIM_IGEMM_T192=$t"
  GIM_IGEMM_T192=$t python tools/microbench_conv.py --cin 196 --cout 196 --k 3 --H 240 --W 320 --B 16 --act leaky 2>&1 | tail -1
  GIM_IGEMM_T192=$t python tools/microbench_conv.py --cin 256 --cout 256 --k 3 --H 120 --W 160 --B 16 --act leaky 2>&1 | tail -1
  GIM_IGEMM_T192=$t python tools/microbench_conv.py --cin 256 --cout 256 --k 3 --H 240 --W 320 --B 16 --act relu 2>&1 | tail -1
  GIM_IGEMM_T192=$t GIM_CONV_HALO_MIN_TILES=100000000 python tools/microbench_conv.py --cin 196 --cout 128 --k 3 --H 240 --W 320 --B 16 --act none 2>&1 | tail -1
done
done
