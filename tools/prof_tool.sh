#!/bin/bash
# kernel-trace + stats of one bench tool -> gpurun_out/<tag>_stats.txt:  tools/prof_tool.sh <tool.py> <tag> [tool args...]
tool=$1; tag=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; rm -rf $out; mkdir -p $out
( cd /tmp && TMPDIR=/tmp timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python $GRAFT_REPO_ROOT/tools/$tool "$@" ) > $out/log.txt 2>&1
python - "$out" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + '/**/*kernel_stats.csv', recursive=True)
if not f:
    print('no stats csv'); print(open(out + '/log.txt').read()[-2000:]); sys.exit()
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
lines = [f"total kernel time {tot/1e6:.3f} ms"]
for r in rows[:32]:
    lines.append(f"{r['Name'][:100]:100s} calls {int(r['Calls']):6d} total_us {float(r['TotalDurationNs'])/1e3:11.1f} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {float(r['Percentage']):5.2f}")
open(out + '_stats.txt', 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
PY
grep -h '"metric"' $out/log.txt | tail -1 > ${out}_bench.json
