#!/bin/bash
# round-3 GPU check AA: two-wave potrf_diag (inverse of row panel rb overlaps the factorisation of panel rb + 1)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dkm.py tests/test_gpu_gp_pins.py tests/test_gpu_roma.py -m gpu -q --maxfail=5 --timeout=600 -p no:cacheprovider 2>&1 | tail -3
python tools/bench_dkm.py --steps 3 2>&1 | tail -1 | cut -c1-260
python tools/bench_roma.py --steps 3 2>&1 | tail -1 | cut -c1-260
out=$GRAFT_REPO_ROOT/gpurun_out/potrf; rm -rf $out; mkdir -p $out
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python $GRAFT_REPO_ROOT/tools/bench_dkm.py --steps 2 ) > $out/log.txt 2>&1
python - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)
for r in csv.DictReader(open(f[0])):
    if any(k in r['Name'] for k in ('potrf', 'gemm_sub', 'gemm_set', 'gemm_f64')):
        print(f"{r['Name'][:60]:60s} calls {int(r['Calls']):5d} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
