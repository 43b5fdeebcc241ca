#!/bin/bash
# fp32 mode of the benchmark workload with and without the split-operand products (gim_conv_args.split16): step time + parity block vs the CPU oracle
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for f in 0 1; do
  GIM_FLAGS=fp32_split=$f GIM_BENCH_SKIP_DENSE=1 GIM_BENCH_SKIP_LIGHTGLUE=1 GIM_BENCH_SKIP_PARITY_MODE=1 timeout 900 python bench.py --precision fp32 --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
p=d['parity']
print('fp32_split=$f', d['value'], 'pairs/s', d['ms_per_step'], 'ms | flips', p['flips'], 'flip_rate', p['flip_rate'], 'max|dmconf|', p['max_abs_dmconf'], 'max|dmkpts1|', p['max_abs_dmkpts1_px'], 'max|dexpec|', p['max_abs_dexpec_f'], '| igemm', d['roofline']['achieved'], 'TF/s', d['roofline']['kernel_ms_per_step'], 'ms')
"
done
