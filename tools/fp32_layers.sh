#!/bin/bash
# per-layer table of the fp32 mode's implicit GEMMs (split products): [label, launches per step, ms per step, fp32-equivalent TFLOP/s]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
GIM_BENCH_ALL_LAYERS=1 GIM_FLAGS=fp32_split=${1:-1} GIM_BENCH_SKIP_DENSE=1 GIM_BENCH_SKIP_LIGHTGLUE=1 GIM_BENCH_SKIP_PARITY_MODE=1 timeout 900 python bench.py --precision fp32 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(d['value'], 'pairs/s', d['ms_per_step'], 'ms; igemm', d['roofline']['kernel_ms_per_step'], 'ms')
for l in d['roofline']['all_layers']: print(l)
print(d['roofline']['fused_kernels'])
"
