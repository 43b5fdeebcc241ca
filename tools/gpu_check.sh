#!/bin/bash
# One GPU-box session, parameterised (replaces the one-shot tools/gpu_r3*.sh scripts of round 3):
#   gpurun --timeout T -- 'bash tools/gpu_check.sh <tag> <step> [<step> ...]'
# steps (run in the order given; every step logs to gpurun_out/<tag>_<step>.log and prints a short tail):
#   tests[=EXPR]   pytest -m gpu (-k EXPR)                 smoke          __graft_entry__.smoke()
#   bench[=ARGS]   python bench.py ARGS  (full line)       quick[=ARGS]   bench.py without secondary workloads / alt modes / CPU leg
#   prof           rocprofv3 --kernel-trace --stats of the headline command  -> <tag>_kernel_stats.txt      gaps   idle time between the kernels of a replayed forward (tools/prof_gaps.sh)
#   pmc            MFMA-pipe occupancy per kernel family (tools/pmc_forward.sh)   traffic   HBM bytes (tools/pmc_traffic.sh)
#   ab=V1;V2;...   same-box A/B of the step time over variants (tools/ab_forward.py: ENV=value[,ENV=value] or lib=alt / lib=new)
#   cmd=COMMAND    any shell command (micro-benchmarks)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
tag=$1; shift
mkdir -p gpurun_out; export TMPDIR=/tmp
o=gpurun_out/$tag
n=0
for step in "$@"; do
  n=$((n + 1)); name=${step%%=*}; arg=""; [[ $step == *=* ]] && arg=${step#*=}
  log=${o}_${n}_${name}.log
  case $name in
    tests) if [ -n "$arg" ]; then timeout 2400 python -m pytest tests -m gpu -q -x --timeout=900 -p no:cacheprovider -k "$arg" > $log 2>&1
           else timeout 2400 python -m pytest tests -m gpu -q -x --timeout=900 -p no:cacheprovider > $log 2>&1; fi
           echo "[$name $arg] rc=$?"; tail -6 $log | cut -c1-400 ;;
    smoke) timeout 900 python __graft_entry__.py smoke > $log 2>&1; echo "[smoke] rc=$?"; tail -2 $log | cut -c1-500 ;;
    bench) timeout 1800 python bench.py $arg > ${o}_${n}_bench.json 2> $log; echo "[bench $arg] rc=$?"; cut -c1-700 ${o}_${n}_bench.json; tail -3 $log | cut -c1-300 ;;
    quick) GIM_BENCH_SKIP_DENSE=1 GIM_BENCH_SKIP_LIGHTGLUE=1 GIM_BENCH_SKIP_PARITY_MODE=1 timeout 900 python bench.py --no-cpu-baseline $arg > ${o}_${n}_quick.json 2> $log
           echo "[quick $arg] rc=$?"; python - ${o}_${n}_quick.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d["roofline"]
    print(d["value"], "pairs/s", d["ms_per_step"], "ms | igemm", r["achieved"], "TF", r["kernel_ms_per_step"], "ms | coarse", r["coarse_gemm"], "| fused", r["fused_kernels"])
except Exception as e:
    print("no bench line:", e)
PY
           tail -3 $log | cut -c1-300 ;;
    prof)  bash tools/prof_bench.sh ${tag}_prof 10 2>&1 | head -34 | cut -c1-200 ;;
    gaps)  bash tools/prof_gaps.sh ${tag}_gaps ${arg:-8} 2>&1 | head -50 | cut -c1-200 ;;
    pmc)   bash tools/pmc_forward.sh ${tag} 2>&1 | head -26 | cut -c1-200 ;;
    traffic) bash tools/pmc_traffic.sh ${tag} 2>&1 | head -40 | cut -c1-200 ;;
    ab)    IFS=';' read -ra V <<< "$arg"; timeout 1500 python tools/ab_forward.py 2 "${V[@]}" > $log 2>&1; echo "[ab] rc=$?"; tail -12 $log | cut -c1-300 ;;
    cmd)   timeout 1500 bash -c "$arg" > $log 2>&1; echo "[cmd] rc=$?"; tail -25 $log | cut -c1-300 ;;
    *)     echo "unknown step $step" ;;
  esac
done
