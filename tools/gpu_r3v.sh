#!/bin/bash
# round-3 GPU check V: layer-3 Bottleneck tails as 4-wave workgroups (two per CU) vs 8-wave
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
for nw in 0 8; do
  GIM_BNECK_TAIL_NW=$nw timeout 900 python -m pytest tests/test_gpu_bneck_tail.py tests/test_gpu_loftr.py -m gpu -q --maxfail=10 --timeout=600 -p no:cacheprovider > gpurun_out/r3v_tests_$nw.log 2>&1
  echo "NW=$nw pytest rc=$?"; tail -3 gpurun_out/r3v_tests_$nw.log | cut -c1-300
done
B="GIM_BENCH_SKIP_DENSE=1 GIM_BENCH_SKIP_LIGHTGLUE=1 GIM_BENCH_SKIP_PARITY_MODE=1"
for i in 1 2; do
  for nw in 4 8; do
    env $B GIM_BNECK_TAIL_NW=$nw timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3v_${nw}_$i.json 2>gpurun_out/r3v_${nw}_$i.err
    python -c "
import json
d=json.load(open('gpurun_out/r3v_${nw}_$i.json')); r=d['roofline']; print('NW=$nw $i', d['value'], d['ms_per_step'], 'igemm', r['kernel_ms_per_step'], r['frac'], 'fused', {k:(v['ms_per_step'], v['tflops']) for k,v in r['fused_kernels'].items()})" || tail -5 gpurun_out/r3v_${nw}_$i.err
  done
done
