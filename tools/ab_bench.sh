#!/bin/bash
# A/B two builds of libnvllm_amd.so on the SAME GPU box, interleaved (cdna guide rule 24):
#   nano-vllm-candle_amd/libnvllm_amd.so (new) vs nano-vllm-candle_amd/libnvllm_amd_old.so (old)
cp nano-vllm-candle_amd/libnvllm_amd.so /tmp/cur.so
for v in new old new old new old; do
  if [ $v = old ]; then cp nano-vllm-candle_amd/libnvllm_amd_old.so nano-vllm-candle_amd/libnvllm_amd.so; else cp /tmp/cur.so nano-vllm-candle_amd/libnvllm_amd.so; fi
  echo $v $(timeout -k 10 300 python bench.py --steps 48 --warmup 4 --no-cpu-baseline --skip-tp-leg --profile-steps 0 2>/dev/null | tail -1 | python3 -c "import json,sys;d=json.load(sys.stdin);print(d['ms_per_step'])")
done
cp /tmp/cur.so nano-vllm-candle_amd/libnvllm_amd.so
