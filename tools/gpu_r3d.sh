#!/bin/bash
# round-3 GPU check D: pipelined upsample epilogue + merged cross-layer projections: tests, then same-box A/B of the step time
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_conv_halo.py tests/test_gpu_loftr.py tests/test_gpu_loftr_fullsize.py tests/test_gpu_token_mlp.py tests/test_gpu_emit.py \
    -m gpu -q --maxfail=10 --timeout=300 -p no:cacheprovider > gpurun_out/r3d_tests.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r3d_tests.log
B="GIM_BENCH_ALL_LAYERS=1 GIM_BENCH_SKIP_DENSE=1 GIM_BENCH_SKIP_LIGHTGLUE=1 GIM_BENCH_SKIP_PARITY_MODE=1"
for i in 1 2; do
  for v in new unfused; do
    if [ $v = unfused ]; then E="GIM_UPS_FUSED=0"; else E=""; fi
    env $B $E timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3d_${v}_$i.json 2>/dev/null
    python - <<PY
import json
d=json.load(open("gpurun_out/r3d_${v}_$i.json"))
L={l[0]:l for l in d["roofline"]["all_layers"]}
print("$v $i", d["value"], d["ms_per_step"], [ (k, L[k][2]) for k in L if "ups" in k or "256->196 k1" in k or "512->256 k1s1 M=307200" in k or "256->768" in k or "256->256 k1" in k])
PY
  done
done
