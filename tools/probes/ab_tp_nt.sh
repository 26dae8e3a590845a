for tp in 8 4; do for v in 80 32 80 32; do
  echo tp$tp nt_min_mb=$v $(NVLLM_STREAM_NT_MB=$v timeout -k 10 200 python3 bench.py --tp-proj-child $tp --tp-model qwen3-32b --tp-batch 64 --tp-prompt 128 --tp-steps 32 2>/dev/null | grep TP_PROJ_RESULT | python3 -c "import sys,json; print(json.loads(sys.stdin.read().split(' ',1)[1])['per_rank_ms_per_step'])")
done; done
