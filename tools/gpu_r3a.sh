#!/bin/bash
# round-3 GPU check A: fp16 flavour + hand-out kernels + bench line (one gpurun call)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 700 python -m pytest tests/test_gpu_emit.py tests/test_gpu_kernels.py tests/test_gpu_conv_halo.py tests/test_gpu_bneck_fused.py \
    tests/test_gpu_token_mlp.py tests/test_gpu_fine_fused.py tests/test_gpu_loftr.py tests/test_gpu_loftr_fullsize.py \
    -m gpu -q --maxfail=12 --timeout=300 -p no:cacheprovider -s > gpurun_out/r3a_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3a_tests.log
tail -40 gpurun_out/r3a_tests.log
GIM_BENCH_SKIP_DENSE=1 GIM_BENCH_SKIP_LIGHTGLUE=1 timeout 400 python bench.py --steps 20 --warmup 3 > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err
echo "bench rc=$?"
tail -c 3000 gpurun_out/r3a_bench.json
tail -5 gpurun_out/r3a_bench.err
