#!/bin/bash
# MFMA-pipe occupancy of every kernel family of the gim_loftr forward (default mode, batch 8, 640x480, eager launches): one rocprofv3
# --pmc pass (kernel-trace only) of tools/prof_forward.py -> gpurun_out/<tag>_pmc_forward.txt
#   SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES) = fraction of the SIMD-cycles of busy CUs in which the MFMA pipe works
tag=${1:-r03}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/pmc_fwd; rm -rf $out; mkdir -p $out
( cd /tmp && TMPDIR=/tmp GIM_FLAGS=graph=0 timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES --output-format csv -d $out -o p -- python $root/tools/prof_forward.py 3 ) > $out/log.txt 2>&1
python - "$out" "$root/gpurun_out/${tag}_pmc_forward.txt" <<'PY'
import csv, glob, sys, collections
out, dst = sys.argv[1], sys.argv[2]
f = glob.glob(out + '/**/*counter_collection.csv', recursive=True)
if not f:
    print('no counter csv', open(out + '/log.txt').read()[-1500:]); sys.exit(1)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(f[0])):
    nm = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '')
    k = nm.split('(')[0][:64]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    key = (r.get('Dispatch_Id'), k)
    if key not in seen:
        seen.add(key); cnt[k] += 1
lines = ["rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES -- python tools/prof_forward.py 3   (GIM_FLAGS=graph=0)",
         "per kernel family over 3 forwards: launches, MFMA instructions, MFMA busy cycles, busy CU cycles, MFMA busy / (4 x busy CU cycles)"]
rows = []
for k, d in acc.items():
    mf, cu = d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0), d.get('SQ_BUSY_CU_CYCLES', 0.0)
    if mf <= 0: continue
    rows.append((mf, f"{k:66s} n={cnt[k]:4d}  insts_mfma {d.get('SQ_INSTS_MFMA', 0):14.0f}  mfma_busy {mf:16.0f}  busy_cu {cu:16.0f}  frac {mf / (4 * cu) if cu else 0:6.3f}"))
for _, l in sorted(rows, reverse=True): lines.append(l)
open(dst, 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines[:24]))
PY
