#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
for m in 1 0; do
  out=$GRAFT_REPO_ROOT/gpurun_out/cmprof_$m; rm -rf $out; mkdir -p $out
  ( cd /tmp && GIM_CM_PANEL=$m timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python $GRAFT_REPO_ROOT/tools/microbench_cm.py --bf16 --planted --sigma 1.0 ) > $out/log.txt 2>&1
  python - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:10]:
    print(f"{r['Name'][:70]:70s} calls {int(r['Calls']):5d} avg_us {float(r['AverageNs'])/1e3:9.1f}")
PY
done
