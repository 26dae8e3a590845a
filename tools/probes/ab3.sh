#!/bin/bash
R=${GRAFT_REPO_ROOT}
run() { name=$1; shift
  out=$(env "$@" 2>&1 | tail -1)
  echo "$name: $(echo "$out" | python3 -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); print("ms_per_step %.4f tok/s %.0f frac %.3f" % (d["ms_per_step"], d["value"], d["step_roofline"]["frac"]))
except Exception as e: print("FAILED", e)')"
}
B="timeout -k 10 280 python3 $R/bench.py --no-cpu-baseline --skip-tp-leg --profile-steps 0"
run "8b b64 fused" $B --model qwen3-8b --steps 32 --warmup 4
run "8b b64 generic" NVLLM_NO_FUSED=1 $B --model qwen3-8b --steps 32 --warmup 4
run "32b b64 p128 fused" $B --model qwen3-32b --prompt-min 128 --prompt-max 128 --steps 24 --warmup 4
run "32b b64 p128 generic" NVLLM_NO_FUSED=1 $B --model qwen3-32b --prompt-min 128 --prompt-max 128 --steps 24 --warmup 4
run "0.6b b64" $B --steps 64 --warmup 8
