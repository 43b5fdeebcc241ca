#!/bin/bash
# usage: tools/pmc_run.sh <tag> <pmc counters...> -- <microbench args>
# collects the counters in their own rocprofv3 pass (kernel-trace only) and prints per-kernel averages
tag=$1; shift
pmc=()
while [[ $1 != "--" ]]; do pmc+=("$1"); shift; done; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
( cd /tmp && TMPDIR=/tmp timeout 150 rocprofv3 --kernel-trace --pmc "${pmc[@]}" --output-format csv -d $out -o p -- python $GRAFT_REPO_ROOT/${PMC_SCRIPT:-tools/microbench_conv.py} "$@" ) > $out/log.txt 2>&1
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + '/**/*counter_collection.csv', recursive=True)
if not f: print('no counter csv', open(out+'/log.txt').read()[-1500:]); sys.exit()
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name'][:60]
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    if not any(t in k for t in ("igemm", "cm_stats", "sdpa", "la_", "fine_fused", "tok_")): continue
    print(k)
    for c, v in d.items(): print(f"   {c:28s} avg {sum(v)/len(v):14.1f}  (n={len(v)})")
PY
