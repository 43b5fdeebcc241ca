#!/bin/bash
# round-3 GPU check X: 64-channel 3x3 layers back on the generic 256 x 64 tile (secondary workloads)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_conv_halo.py tests/test_gpu_lightglue.py -m gpu -q --maxfail=5 --timeout=600 -p no:cacheprovider 2>&1 | tail -3
python tools/bench_lightglue.py --steps 6 --warmup 2 2>&1 | tail -1 | cut -c1-400
python tools/bench_dkm.py --steps 3 2>&1 | tail -1 | cut -c1-400
python tools/bench_roma.py --steps 3 2>&1 | tail -1 | cut -c1-400
