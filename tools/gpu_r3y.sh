#!/bin/bash
# round-3 GPU check Y: ConvRefiner block at 24 channels as one launch (gim_dwconv5x5_pw32): tests, DKM / RoMa with and without it
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dkm.py tests/test_gpu_roma.py -m gpu -q --maxfail=5 --timeout=600 -p no:cacheprovider 2>&1 | tail -4
for f in 1 0; do
  echo "GIM_DWPW_FUSED=$f"
  GIM_DWPW_FUSED=$f python tools/bench_dkm.py --steps 3 2>&1 | tail -1 | cut -c1-260
  GIM_DWPW_FUSED=$f python tools/bench_roma.py --steps 3 2>&1 | tail -1 | cut -c1-260
done
