#!/bin/bash
# round-3 GPU check R: FPN upsample-add epilogue with the source pixels staged by LDS-DMA
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -k "ups" tests/test_gpu_loftr.py tests/test_gpu_loftr_fullsize.py -m gpu -q --maxfail=10 --timeout=600 -p no:cacheprovider > gpurun_out/r3r_tests.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r3r_tests.log | cut -c1-300
B="GIM_BENCH_SKIP_DENSE=1 GIM_BENCH_SKIP_LIGHTGLUE=1 GIM_BENCH_SKIP_PARITY_MODE=1"
for i in 1 2; do
  for prec in fp16 bf16; do
    env $B timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --precision $prec > gpurun_out/r3r_${prec}_$i.json 2>gpurun_out/r3r_${prec}_$i.err
    python -c "
import json
d=json.load(open('gpurun_out/r3r_${prec}_$i.json')); r=d['roofline']; print('$prec $i', d['value'], d['ms_per_step'], 'igemm', r['kernel_ms_per_step'], r['frac'], [x for x in r['top_layers_ms_tflops'] if 'ups' in x[0] or '196->196' in x[0]])" || tail -5 gpurun_out/r3r_${prec}_$i.err
  done
done
