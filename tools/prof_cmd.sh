#!/bin/bash
# usage: tools/prof_cmd.sh <tag> <python script + args>   -> gpurun_out/<tag>_stats.txt (rocprofv3 --kernel-trace --stats)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; rm -rf $out; mkdir -p $out
( cd /tmp && TMPDIR=/tmp timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python $GRAFT_REPO_ROOT/"$@" ) > $out/log.txt 2>&1
python - "$out" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + '/**/*kernel_stats.csv', recursive=True)
if not f:
    print('no stats csv'); print(open(out + '/log.txt').read()[-2000:]); sys.exit()
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
lines = [f"total kernel time {tot/1e6:.3f} ms"]
for r in rows[:18]:
    lines.append(f"{r['Name'][:96]:96s} calls {int(r['Calls']):6d} total_us {float(r['TotalDurationNs'])/1e3:11.1f} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {float(r['Percentage']):5.2f}")
open(out + '_stats.txt', 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
PY
