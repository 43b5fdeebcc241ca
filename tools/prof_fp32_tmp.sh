cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pf32; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf32 -o p -- python $GRAFT_REPO_ROOT/tools/prof_forward.py 4 fp32 > /tmp/pf32.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/pf32/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms over 4 forwards")
for r in rows[:28]:
    print(f"{r['Name'][:110]:110s} calls {int(r['Calls']):5d} total_us {float(r['TotalDurationNs'])/1e3:10.1f} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {float(r['Percentage']):5.2f}")
PY
