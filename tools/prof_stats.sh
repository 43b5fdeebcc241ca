#!/bin/bash
# kernel-trace + stats of tools/prof_forward.py; prints per-kernel totals (us) -> gpurun_out/<tag>_stats.txt
tag=${1:-prof}; n=${2:-6}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; rm -rf $out; mkdir -p $out
( cd /tmp && TMPDIR=/tmp timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python $GRAFT_REPO_ROOT/tools/prof_forward.py $n ) > $out/log.txt 2>&1
python - "$out" "$n" <<'PY'
import csv, glob, sys
out, n = sys.argv[1], int(sys.argv[2])
f = glob.glob(out + '/**/*kernel_stats.csv', recursive=True)
if not f:
    print('no stats csv'); print(open(out + '/log.txt').read()[-2000:]); sys.exit()
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
lines = [f"total kernel time {tot/1e6:.3f} ms over {n} forwards (1 eager warm-up + capture + replays) => ~{tot/1e6/n:.3f} ms per forward"]
for r in rows[:22]:
    lines.append(f"{r['Name'][:100]:100s} calls {int(r['Calls']):6d} total_us {float(r['TotalDurationNs'])/1e3:11.1f} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {float(r['Percentage']):5.2f}")
open(out + '_stats.txt', 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
PY
