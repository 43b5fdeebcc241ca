#!/bin/bash
# Where one replayed forward spends its time BETWEEN kernels: rocprofv3 --kernel-trace of tools/prof_forward.py, then per-dispatch
# start / end stamps of the last forwards -> span, sum of kernel durations, sum of the gaps and the gaps by the kernel they follow.
#   tools/prof_gaps.sh <tag> [forwards]        -> gpurun_out/<tag>_gaps.txt
tag=${1:-gaps}; n=${2:-8}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; rm -rf $out; mkdir -p $out
( cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out -o p -- python $GRAFT_REPO_ROOT/tools/prof_forward.py $n ) > $out/log.txt 2>&1
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + '/**/*kernel_trace.csv', recursive=True)
if not f:
    print('no kernel trace'); print(open(out + '/log.txt').read()[-2000:]); sys.exit()
rows = list(csv.DictReader(open(f[0])))
ks = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows))
# forwards start at the image layout kernel (two of them back to back per forward: keep the first of each pair)
starts = [i for i, k in enumerate(ks) if 'nchw_to_nhwc' in k[2] and (i == 0 or 'nchw_to_nhwc' not in ks[i - 1][2])]
lines = []
short = lambda s: s.replace('(anonymous namespace)::', '').replace('void ', '')[:70]
for fi in range(max(0, len(starts) - 3), len(starts)):
    a = starts[fi]
    b = starts[fi + 1] if fi + 1 < len(starts) else len(ks)
    fw = ks[a:b]
    span = fw[-1][1] - fw[0][0]
    dur = sum(e - s for s, e, _ in fw)
    gaps = [(fw[i + 1][0] - fw[i][1], fw[i][2], fw[i + 1][2]) for i in range(len(fw) - 1)]
    pos = sum(g for g, _, _ in gaps if g > 0)
    ovl = -sum(g for g, _, _ in gaps if g < 0)
    lines.append(f"forward {fi}: {len(fw)} dispatches, span {span / 1e3:.1f} us, sum of kernel durations {dur / 1e3:.1f} us, idle gaps {pos / 1e3:.1f} us, overlaps {ovl / 1e3:.1f} us")
    if fi == len(starts) - 2 or len(starts) == 1:
        by = collections.defaultdict(lambda: [0, 0.0])
        for g, p, nx in gaps:
            by[short(p)][0] += 1
            by[short(p)][1] += g
        lines.append("  gap after kernel (count, total us, mean us), largest totals first:")
        for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:24]:
            lines.append(f"    {k:70s} {c:4d} {t / 1e3:9.1f} {t / 1e3 / c:7.2f}")
        big = sorted(gaps, key=lambda g: -g[0])[:10]
        lines.append("  ten largest single gaps:")
        for g, p, nx in big:
            lines.append(f"    {g / 1e3:8.1f} us  after {short(p)[:48]:48s} before {short(nx)[:48]}")
open(out + '_gaps.txt', 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
PY
