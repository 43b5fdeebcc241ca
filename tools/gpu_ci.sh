#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprof.  Logs -> gpurun_out/ (merged back by gpurun).
# usage: tools/gpu_ci.sh [tests|bench|prof|all]
what=${1:-all}
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|gfx" | head -8 > gpurun_out/rocminfo.txt
nproc >> gpurun_out/rocminfo.txt
if [[ $what == tests || $what == all ]]; then
  timeout 1500 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
  tail -5 gpurun_out/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
  echo "smoke exit $?" >> gpurun_out/smoke.log; tail -3 gpurun_out/smoke.log
fi
if [[ $what == bench || $what == all ]]; then
  timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/bench.log 2>&1
  echo "bench exit $?" >> gpurun_out/bench.log; tail -3 gpurun_out/bench.log
fi
if [[ $what == prof || $what == all ]]; then
  rm -rf gpurun_out/prof
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o loftr -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline ) > gpurun_out/prof.log 2>&1
  echo "prof exit $?" >> gpurun_out/prof.log; tail -3 gpurun_out/prof.log
  find gpurun_out/prof -name "*kernel_stats*" | head
fi
