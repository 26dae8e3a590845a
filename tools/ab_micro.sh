#!/bin/bash
# A/B of the decode step's execution forms on one box: NVLLM_MICRO (row-group streams) x NVLLM_GRAPH (hipGraph replay).
# usage (on the GPU box): bash tools/ab_micro.sh [extra bench.py args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for rep in 1 2; do
for cfg in "1 0" "1 1" "2 1" "4 1" "4 0" "3 1"; do
  set -- $cfg
  out=$(NVLLM_MICRO=$1 NVLLM_GRAPH=$2 timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --skip-tp-leg --profile-steps 0 --steps 64 --warmup 8 2>&1 | tail -1)
  echo "micro=$1 graph=$2 rep=$rep: $(echo "$out" | python3 -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); print("ms_per_step %.4f tok/s %.0f step_frac %.3f" % (d["ms_per_step"], d["value"], d["step_roofline"]["frac"]))
except Exception as e: print("FAILED", e)')"
done
done
