#!/bin/bash
# A/B two builds on the SAME GPU box, interleaved, with bench.py arguments:  bash tools/ab_bench_args.sh --model qwen3-8b ...
for v in new old new old new old; do
  if [ $v = old ]; then L=libnvllm_amd_old.so; else L=libnvllm_amd.so; fi
  echo $v $(NVLLM_LIB=$L timeout -k 10 300 python3 bench.py --no-cpu-baseline --skip-tp-leg --profile-steps 0 "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys;d=json.load(sys.stdin);print(d['ms_per_step'])")
done
