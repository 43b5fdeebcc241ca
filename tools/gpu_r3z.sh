#!/bin/bash
# round-3 GPU check Z: fine kernel launched with the device-side match count (no host round trip in front of it)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_fine_fused.py tests/test_gpu_loftr.py tests/test_gpu_loftr_fullsize.py tests/test_gpu_emit.py tests/test_gpu_zeb_e2e.py -m gpu -q --maxfail=10 --timeout=600 -p no:cacheprovider > gpurun_out/r3z_tests.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r3z_tests.log | cut -c1-300
B="GIM_BENCH_SKIP_DENSE=1 GIM_BENCH_SKIP_LIGHTGLUE=1 GIM_BENCH_SKIP_PARITY_MODE=1"
for i in 1 2 3; do
  for dc in 1 0; do
    env $B GIM_FINE_DEV_COUNT=$dc timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3z_${dc}_$i.json 2>gpurun_out/r3z_${dc}_$i.err
    python -c "
import json
d=json.load(open('gpurun_out/r3z_${dc}_$i.json')); r=d['roofline']; print('devcount=$dc $i', d['value'], d['ms_per_step'], d['config']['matches_per_pair'], 'fine', r['fused_kernels']['fine_fused'])" || tail -5 gpurun_out/r3z_${dc}_$i.err
  done
done
