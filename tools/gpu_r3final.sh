#!/bin/bash
# round-3 final GPU run: whole GPU suite, smoke(), the full bench line, the headline's rocprof summary
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=900 -p no:cacheprovider > gpurun_out/r3final_tests.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r3final_tests.log | cut -c1-300
timeout 600 python __graft_entry__.py smoke > gpurun_out/r3final_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r3final_smoke.log | cut -c1-400
timeout 1500 python bench.py > gpurun_out/r3final_full.json 2> gpurun_out/r3final_full.err; echo "full bench rc=$?"; cut -c1-300 gpurun_out/r3final_full.json
bash tools/prof_bench.sh r03_final 10 | head -30 | cut -c1-190
