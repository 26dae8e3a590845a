#!/bin/bash
# A/B one environment switch of libnvllm_amd.so on the SAME GPU box, interleaved: tools/ab_env.sh NVLLM_NO_ROWDIR
# prints ms/step with the variable unset ("base") and set to 1 ("VAR=1"), three rounds each.
VAR=${1:?env var name}
for i in 1 2 3; do
  echo "base  " $(timeout -k 10 300 python bench.py --steps 48 --warmup 4 --no-cpu-baseline --skip-tp-leg --profile-steps 0 2>/dev/null | tail -1 | python3 -c "import json,sys;d=json.load(sys.stdin);print(d['ms_per_step'])")
  echo "$VAR=1" $(env $VAR=1 timeout -k 10 300 python bench.py --steps 48 --warmup 4 --no-cpu-baseline --skip-tp-leg --profile-steps 0 2>/dev/null | tail -1 | python3 -c "import json,sys;d=json.load(sys.stdin);print(d['ms_per_step'])")
done
