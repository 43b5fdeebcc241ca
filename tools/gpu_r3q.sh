#!/bin/bash
# round-3 GPU check Q: fine_fused gather in one batch; then the full bench line (all sections) for the docs
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fine_fused.py tests/test_gpu_loftr.py -m gpu -q --maxfail=10 --timeout=600 -p no:cacheprovider > gpurun_out/r3q_tests.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r3q_tests.log | cut -c1-300
B="GIM_BENCH_SKIP_DENSE=1 GIM_BENCH_SKIP_LIGHTGLUE=1 GIM_BENCH_SKIP_PARITY_MODE=1"
for i in 1 2; do
    env $B timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3q_$i.json 2>gpurun_out/r3q_$i.err
    python -c "
import json
d=json.load(open('gpurun_out/r3q_$i.json')); r=d['roofline']; print('$i', d['value'], d['ms_per_step'], 'igemm', r['kernel_ms_per_step'], r['frac'], 'fused', {k:(v['ms_per_step'], v['tflops']) for k,v in r['fused_kernels'].items()})" || tail -5 gpurun_out/r3q_$i.err
done
timeout 1500 python bench.py > gpurun_out/r3q_full.json 2> gpurun_out/r3q_full.err; echo "full bench rc=$?"; cut -c1-1500 gpurun_out/r3q_full.json
