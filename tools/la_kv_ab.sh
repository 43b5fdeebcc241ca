#!/bin/bash
# KV-state reduction under the three GIM_LA_KV2 settings (one process each: the choice is read once)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for m in ${LA_MODES:-0 1 3}; do GIM_LA_KV2=$m timeout 300 python tools/microbench_la_kv.py 10 2>&1 | tail -4; done
