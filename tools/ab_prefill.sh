#!/bin/bash
# A/B two builds on the SAME GPU box, interleaved: warmed prefill of the bench prompts (ms)
#   nano-vllm-candle_amd/libnvllm_amd.so (new) vs nano-vllm-candle_amd/libnvllm_amd_old.so (old)
for v in new old new old new old; do
  if [ $v = old ]; then L=libnvllm_amd_old.so; else L=libnvllm_amd.so; fi
  echo $v $(NVLLM_LIB=$L timeout -k 10 200 python3 bench.py --no-cpu-baseline --skip-tp-leg --prefill-only --prefill-reps 3 2>/dev/null | python3 -c "import sys,json; [print(json.loads(l)['prefill']['ms']) for l in sys.stdin if l.startswith('{')]")
done
