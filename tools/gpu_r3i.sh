#!/bin/bash
# round-3 GPU check I: row-panel coarse statistics kernel: exactness tests, microbenchmark, same-box A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -k "coarse" tests/test_gpu_loftr.py tests/test_gpu_loftr_fullsize.py -m gpu -q --maxfail=10 --timeout=600 -p no:cacheprovider > gpurun_out/r3i_tests.log 2>&1
echo "pytest rc=$?"; tail -30 gpurun_out/r3i_tests.log | cut -c1-300
B="GIM_BENCH_SKIP_DENSE=1 GIM_BENCH_SKIP_LIGHTGLUE=1 GIM_BENCH_SKIP_PARITY_MODE=1"
for i in 1 2; do
  for v in panel tile; do
    if [ $v = tile ]; then E="GIM_CM_PANEL=0"; else E=""; fi
    env $B $E timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3i_${v}_$i.json 2>gpurun_out/r3i_${v}_$i.err
    python -c "
import json
d=json.load(open('gpurun_out/r3i_${v}_$i.json')); print('$v $i', d['value'], d['ms_per_step'], d['config']['matches_per_pair'])" || tail -5 gpurun_out/r3i_${v}_$i.err
  done
done
for E in "GIM_CM_PANEL=1" "GIM_CM_PANEL=0"; do env $E timeout 120 python tools/microbench_cm.py --bf16 --planted --sigma 1.0 2>&1 | tail -4; done
