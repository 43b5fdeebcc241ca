#!/bin/bash
# round-3 GPU check U: bneck64 with the identity rows requested a conv3 ahead and W1n by asm LDS-DMA
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bneck_fused.py tests/test_gpu_loftr.py -m gpu -q --maxfail=10 --timeout=600 -p no:cacheprovider > gpurun_out/r3u_tests.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r3u_tests.log | cut -c1-300
B="GIM_BENCH_SKIP_DENSE=1 GIM_BENCH_SKIP_LIGHTGLUE=1 GIM_BENCH_SKIP_PARITY_MODE=1"
for i in 1 2; do
  for prec in fp16 bf16; do
    env $B timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --precision $prec > gpurun_out/r3u_${prec}_$i.json 2>gpurun_out/r3u_${prec}_$i.err
    python -c "
import json
d=json.load(open('gpurun_out/r3u_${prec}_$i.json')); r=d['roofline']; print('$prec $i', d['value'], d['ms_per_step'], 'igemm', r['kernel_ms_per_step'], r['frac'], 'fused', {k:(v['ms_per_step'], v['tflops']) for k,v in r['fused_kernels'].items()})" || tail -5 gpurun_out/r3u_${prec}_$i.err
  done
done
