for v in 0 1 2 0; do
  NVLLM_ATTN_DBG=$v timeout -k 10 120 python3 bench.py --no-cpu-baseline --skip-tp-leg --prefill-only --prefill-reps 3 2>/dev/null | python3 -c "import sys,json; [print('attn dbg',$v, json.loads(l)['prefill']['ms']) for l in sys.stdin if l.startswith('{')]"
done
