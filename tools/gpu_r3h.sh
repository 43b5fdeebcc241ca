#!/bin/bash
# round-3: same-box sweep of the implicit-GEMM dispatch knobs (per-layer table of every variant)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
B="GIM_BENCH_ALL_LAYERS=1 GIM_BENCH_SKIP_DENSE=1 GIM_BENCH_SKIP_LIGHTGLUE=1 GIM_BENCH_SKIP_PARITY_MODE=1"
i=0
for E in "X=1" "GIM_CONV_HALO_MIN_TILES=2000" "GIM_CONV_HALO_MIN_TILES=100000000" "GIM_IGEMM_BIG_MIN_TILES=256" "GIM_IGEMM_BIG_MIN_TILES=1024" "GIM_IGEMM_BIG_MIN_NKT=8" "GIM_IGEMM_BIG=0" "X=2"; do
  i=$((i+1))
  env $B $E timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3h_$i.json 2>/dev/null
  python - "$E" $i <<'PY'
import json, sys
d = json.load(open("gpurun_out/r3h_%s.json" % sys.argv[2]))
L = d["roofline"]["all_layers"]
print(sys.argv[1], d["value"], d["ms_per_step"], "igemm ms", d["roofline"]["kernel_ms_per_step"])
print("   ", [(l[0].replace(" k", "k").replace(" M=", "@"), l[2]) for l in L[:14]])
PY
done
