#!/bin/bash
# round-3 GPU check B: whole GPU suite (fp64 GP parity mode of the dense matchers included) + smoke
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 --timeout=600 -p no:cacheprovider -s > gpurun_out/r3b_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3b_tests.log
grep -n "vs fp64\|vs the\|warp:\|passed\|failed\|Error\|rc=" gpurun_out/r3b_tests.log | tail -60
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3b_smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r3b_smoke.log
