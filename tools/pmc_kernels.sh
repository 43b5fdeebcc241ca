#!/bin/bash
# Any set of PMC counters per kernel family of the gim_loftr forward (eager launches), one rocprofv3 --pmc pass per counter group
# (kernel-trace only -- gpurun refuses --pmc beside the other trace domains).
#   tools/pmc_kernels.sh <tag> [precision] -- GROUP1_C1 GROUP1_C2 ... -- GROUP2_C1 ...      -> gpurun_out/<tag>_pmc_kernels.txt
# A group is one pass (SQ: 8 slots, TCC: 4, TCP/TA: 4, GRBM: 2).  Sums over the launches of 3 forwards.
tag=$1; shift
prec=bf16
if [[ $1 != "--" ]]; then prec=$1; shift; fi
shift
root=$GRAFT_REPO_ROOT
dst=$root/gpurun_out/${tag}_pmc_kernels.txt
: > $dst
g=0
grp=()
run_group() {
  [[ ${#grp[@]} -eq 0 ]] && return
  g=$((g + 1))
  out=$root/gpurun_out/pmck_$g; rm -rf $out; mkdir -p $out
  ( cd /tmp && TMPDIR=/tmp GIM_FLAGS=graph=0 timeout 300 rocprofv3 --kernel-trace --pmc "${grp[@]}" --output-format csv -d $out -o p -- python $root/tools/prof_forward.py 3 $prec ) > $out/log.txt 2>&1
  python - "$out" "$dst" "${grp[*]}" <<'PY'
import csv, glob, sys, collections
out, dst, names = sys.argv[1], sys.argv[2], sys.argv[3].split()
f = glob.glob(out + '/**/*counter_collection.csv', recursive=True)
with open(dst, 'a') as w:
    w.write(f"--- pass: rocprofv3 --kernel-trace --pmc {' '.join(names)} -- python tools/prof_forward.py 3 (GIM_FLAGS=graph=0); sums over all launches of a family\n")
    if not f:
        w.write('no counter csv: ' + open(out + '/log.txt').read()[-800:] + '\n'); sys.exit(0)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:58]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        key = (r.get('Dispatch_Id'), k)
        if key not in seen:
            seen.add(key); cnt[k] += 1
    keys = sorted(acc, key=lambda k: -max(acc[k].values()))[:30]
    for k in keys:
        w.write(f"{k:60s} n={cnt[k]:4d} " + ' '.join(f"{c}={acc[k].get(c, 0):.4g}" for c in names) + '\n')
PY
  grp=()
}
for a in "$@"; do
  if [[ $a == "--" ]]; then run_group; else grp+=("$a"); fi
done
run_group
head -c 6000 $dst
