#!/bin/bash
R=${GRAFT_REPO_ROOT}
for tp in 8 4 2; do
  for mode in fused generic; do
    if [ $mode = generic ]; then export NVLLM_NO_FUSED=1; else unset NVLLM_NO_FUSED; fi
    echo "tp$tp $mode: $(timeout -k 10 280 python3 $R/bench.py --tp-proj-child $tp 2>&1 | grep TP_PROJ_RESULT | cut -c1-200)"
  done
done
