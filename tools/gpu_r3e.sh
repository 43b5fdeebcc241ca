#!/bin/bash
# round-3 GPU check E: layer-2 tail fusion kernel: tests, then same-box A/B of the step time
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_bneck_tail.py tests/test_gpu_bneck_fused.py tests/test_gpu_loftr.py tests/test_gpu_loftr_fullsize.py tests/test_gpu_token_mlp.py \
    -m gpu -q --maxfail=10 --timeout=300 -p no:cacheprovider > gpurun_out/r3e_tests.log 2>&1
echo "pytest rc=$?"; tail -25 gpurun_out/r3e_tests.log
B="GIM_BENCH_ALL_LAYERS=1 GIM_BENCH_SKIP_DENSE=1 GIM_BENCH_SKIP_LIGHTGLUE=1 GIM_BENCH_SKIP_PARITY_MODE=1"
for i in 1 2; do
  for v in tail notail; do
    if [ $v = notail ]; then E="GIM_BNECK_TAIL=0"; else E=""; fi
    env $B $E timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3e_${v}_$i.json 2>gpurun_out/r3e_${v}_$i.err
    python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r3e_${v}_$i.json"))
    L={l[0]:l for l in d["roofline"]["all_layers"]}
    print("$v $i", d["value"], d["ms_per_step"], d["roofline"]["fused_kernels"].get("bneck_tail"), [ (k, L[k][1], L[k][2]) for k in L if "128->512" in k or "512->128" in k], d["roofline"]["frac"])
except Exception as e:
    print("$v $i failed", e); print(open("gpurun_out/r3e_${v}_$i.err").read()[-1500:])
PY
  done
done
