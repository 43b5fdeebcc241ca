#!/bin/bash
# round-3 GPU check L: token tails emitting the q/k/v projections (gim_token_mlp_emit), fp16 default mode
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_token_mlp.py tests/test_gpu_loftr.py tests/test_gpu_loftr_fullsize.py -m gpu -q --maxfail=10 --timeout=900 -p no:cacheprovider > gpurun_out/r3l_tests.log 2>&1
echo "pytest rc=$?"; tail -30 gpurun_out/r3l_tests.log | cut -c1-300
B="GIM_BENCH_SKIP_DENSE=1 GIM_BENCH_SKIP_LIGHTGLUE=1 GIM_BENCH_SKIP_PARITY_MODE=1"
for i in 1 2; do
  for v in emit gemm; do
    if [ $v = gemm ]; then E="GIM_TOKEN_EMIT=0"; else E="GIM_TOKEN_EMIT=1"; fi
    for prec in fp16 bf16; do
      env $B $E timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --precision $prec > gpurun_out/r3l_${v}_${prec}_$i.json 2>gpurun_out/r3l_${v}_${prec}_$i.err
      python -c "
import json
d=json.load(open('gpurun_out/r3l_${v}_${prec}_$i.json')); r=d['roofline']; print('$v $prec $i', d['value'], d['ms_per_step'], d['config']['matches_per_pair'], 'igemm', r['kernel_ms_per_step'], r['achieved'], r['frac'], 'tok', r['fused_kernels'].get('token_mlp'), 'cg', r.get('coarse_gemm'))" || tail -5 gpurun_out/r3l_${v}_${prec}_$i.err
    done
  done
done
