#!/bin/bash
# secondary engines under two builds of the library (gim_amd/lib/alt/libgimhip.so vs the tree's), one box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for rep in 1 2; do
for lib in gim_amd/lib/alt/libgimhip.so gim_amd/lib/libgimhip.so; do
  for t in "bench_dkm.py --steps 5" "bench_roma.py --steps 5" "bench_lightglue.py"; do
    GIM_LIB=$lib timeout 600 python tools/$t 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$lib'.split('/')[-2], '$t'.split()[0], {k:d[k] for k in d if k in ('pairs_per_s','match_ms','ms_per_step','ms_per_batch')})
"
  done
done
done
