#!/bin/bash
# round-3 GPU check AC: layer 1's first block with its downsample convolution inside the fused kernel
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bneck_fused.py tests/test_gpu_loftr.py tests/test_gpu_loftr_fullsize.py -m gpu -q --maxfail=10 --timeout=600 -p no:cacheprovider > gpurun_out/r3ac_tests.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r3ac_tests.log | cut -c1-300
B="GIM_BENCH_SKIP_DENSE=1 GIM_BENCH_SKIP_LIGHTGLUE=1 GIM_BENCH_SKIP_PARITY_MODE=1"
for i in 1 2; do
  for ds in 1 0; do
    env $B GIM_BNECK_DS=$ds timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3ac_${ds}_$i.json 2>gpurun_out/r3ac_${ds}_$i.err
    python -c "
import json
d=json.load(open('gpurun_out/r3ac_${ds}_$i.json')); r=d['roofline']; print('ds=$ds $i', d['value'], d['ms_per_step'], 'igemm', r['kernel_ms_per_step'], r['launches_per_step'], r['frac'], 'bneck64', r['fused_kernels']['bneck64_fused'], 'flip', d['config']['matches_per_pair'])" || tail -5 gpurun_out/r3ac_${ds}_$i.err
  done
done
