#!/bin/bash
# round-3 GPU check M: full GPU suite, then the r03 profile set (headline kernel stats, HBM traffic, secondary workloads)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=900 -p no:cacheprovider > gpurun_out/r3m_tests.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/r3m_tests.log | cut -c1-300
bash tools/prof_bench.sh r03_m 10 | head -24 | cut -c1-200
bash tools/pmc_traffic.sh r03 | tail -30
bash tools/prof_dkm.sh r03_dkm 2 | head -16 | cut -c1-180
bash tools/prof_lightglue.sh r03_lightglue 4 | head -14 | cut -c1-180
bash tools/prof_tool.sh bench_roma.py r03_roma --steps 2 | head -16 | cut -c1-180
