"""One fuzz case under several option sets (which kernel family is at fault?).  python tools/dbg_fuzz_case.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nano_vllm_candle_amd as pkg  # noqa: E402
from oracle import oracle  # noqa: E402
from tests.util import oracle_config, random_calls  # noqa: E402

ctx = pkg.Context(0)
kw = dict(hidden_size=1536, head_dim=128, num_attention_heads=32, num_key_value_heads=32, intermediate_size=256, vocab_size=512, num_hidden_layers=3)
if len(sys.argv) > 1:
    kw.update(eval(sys.argv[1]))
cfg = pkg.Qwen3Config.tiny(**kw)
short = [1, 2, 3, 7, 15, 16, 17, 31, 32, 33, 48, 64, 65]
for opts in ({}, {"no_fused": 1}, {"no_rowpar": 1}, {"no_xpack": 1}, {"no_attn_prologue": 1}, {"tile_min_wgs": 0}, {"no_fused": 1, "no_rowpar": 1, "tile_min_wgs": 0}):
    for MS, mbt in ((6, 16), (28, 128)):
        seed = 500
        m = pkg.Qwen3ForCausalLM.from_synthetic(cfg, seed, ctx)
        for k, v in opts.items():
            m.set_option(k, v)
        m.kv_alloc(64, MS, mbt)
        om = oracle.Model(oracle_config(oracle, cfg)).fill_synthetic(seed)
        try:
            ops, worst = random_calls(m, om, cfg, seed, 60, 64, MS, max_new=4, lens_menu=short)
            print(f"ok   {opts} MS {MS} chunk {mbt}: {ops} calls, worst {worst:.2e}", flush=True)
        except Exception as e:  # noqa: BLE001
            print(f"FAIL {opts} MS {MS} chunk {mbt}: {type(e).__name__}: {str(e)[:200]}", flush=True)
        m.close()
