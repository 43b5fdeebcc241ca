#!/bin/bash
# rocprofv3 --kernel-trace --stats of the headline command (bench.py, gim_loftr only) -> gpurun_out/<tag>_kernel_stats.txt + bench line
#   tools/prof_bench.sh <tag> [steps]        (PROF_ARGS="--precision fp16" adds bench arguments)
tag=${1:-final}; steps=${2:-10}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; rm -rf $out; mkdir -p $out
( cd /tmp && TMPDIR=/tmp GIM_BENCH_SKIP_DENSE=1 GIM_BENCH_SKIP_LIGHTGLUE=1 GIM_BENCH_SKIP_PARITY_MODE=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- \
    python $GRAFT_REPO_ROOT/bench.py --steps $steps --warmup 2 --no-cpu-baseline $PROF_ARGS ) > $out/log.txt 2>&1
python - "$out" "$steps" <<'PY'
import csv, glob, sys
out, steps = sys.argv[1], int(sys.argv[2])
f = glob.glob(out + '/**/*kernel_stats.csv', recursive=True)
if not f:
    print('no stats csv'); print(open(out + '/log.txt').read()[-2000:]); sys.exit()
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
ig = [r for r in rows if 'igemm' in r['Name'] or 'conv3x3_halo' in r['Name']]  # every kernel behind gim_conv2d_bn_act
ig_calls = sum(int(r['Calls']) for r in ig); ig_ns = sum(float(r['TotalDurationNs']) for r in ig)
lines = [f"rocprofv3 --kernel-trace --stats -- python bench.py --steps {steps} --warmup 2 --no-cpu-baseline (GIM_BENCH_SKIP_DENSE=1 GIM_BENCH_SKIP_LIGHTGLUE=1 GIM_BENCH_SKIP_PARITY_MODE=1: the headline mode only)",
         f"total kernel time {tot/1e6:.3f} ms; igemm (gim_conv2d_bn_act) kernels: {ig_calls} launches, {ig_ns/1e6:.3f} ms, average {ig_ns/max(1,ig_calls)/1e3:.2f} us per launch"]
for r in rows[:26]:
    lines.append(f"{r['Name'][:100]:100s} calls {int(r['Calls']):6d} total_us {float(r['TotalDurationNs'])/1e3:11.1f} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {float(r['Percentage']):5.2f}")
open(out + '_kernel_stats.txt', 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines[:12]))
PY
grep -h '"metric"' $out/log.txt | tail -1 > ${out}_bench.json
