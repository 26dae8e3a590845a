"""Fuzz campaign at the coarse seam (GPU): the seeded random caller of tests/util.py (new requests, shuffled decode subsets,
resident / pipelined decode, multi-token continuation, re-prefill, free / id reuse) over several model shapes, pool sizes,
chunk sizes and options, every call checked against the CPU oracle.   python tools/fuzz_calls.py [seeds per case]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nano_vllm_candle_amd as pkg  # noqa: E402
from oracle import oracle  # noqa: E402  (a tool, not the product: the oracle is the checker here)
from tests.util import oracle_config, random_calls  # noqa: E402

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ctx = pkg.Context(0)
T = pkg.Qwen3Config.tiny
short = [1, 2, 3, 7, 15, 16, 17, 31, 32, 33, 48, 64, 65]
cases = [
    # name, config, pool blocks, slots, max_batched_tokens, max new per call, lens menu, options
    ("tiny, many short sequences (decode batches >= 16: packed planes)", T(), 64, 28, 48, 6, short, {}),
    ("tiny, chunks of 16 rows", T(), 24, 6, 16, 2, None, {}),
    ("tiny hd 128 kv 1", T(head_dim=128, num_attention_heads=8, num_key_value_heads=1), 40, 20, 64, 5, short + [255, 257], {}),
    ("0.6B layer shapes x 2 (register-direct decode GEMMs)", T(hidden_size=1024, head_dim=128, num_attention_heads=16, num_key_value_heads=8,
                                                              intermediate_size=3072, vocab_size=2048), 64, 28, 128, 6, short + [100, 255, 256, 257], {}),
    ("0.6B layer shapes x 2, 24-bit V", T(hidden_size=1024, head_dim=128, num_attention_heads=16, num_key_value_heads=8,
                                          intermediate_size=3072, vocab_size=2048), 64, 28, 128, 6, short + [100, 255, 256, 257], {"kv_v_bits": 24}),
    ("0.6B layer shapes x 2, tile GEMM on every prompt chunk", T(hidden_size=1024, head_dim=128, num_attention_heads=16, num_key_value_heads=8,
                                                                intermediate_size=3072, vocab_size=2048), 48, 12, 512, 3, [100, 255, 256, 257, 300, 390], {"tile_min_wgs": 1}),
    ("0.6B layer shapes x 2, no fused path", T(hidden_size=1024, head_dim=128, num_attention_heads=16, num_key_value_heads=8,
                                              intermediate_size=3072, vocab_size=2048), 64, 28, 128, 6, short + [255, 257], {"no_fused": 1}),
]
bad = 0
for name, cfg, NB, MS, mbt, max_new, menu, opts in cases:
    for seed in range(100, 100 + n_seeds):
        t0 = time.time()
        m = pkg.Qwen3ForCausalLM.from_synthetic(cfg, seed, ctx)
        for k, v in opts.items():
            m.set_option(k, v)
        m.kv_alloc(NB, MS, mbt)
        om = oracle.Model(oracle_config(oracle, cfg)).fill_synthetic(seed)
        try:
            ops, worst = random_calls(m, om, cfg, seed, 120, NB, MS, max_new=max_new, lens_menu=menu)
            print(f"ok   {name}: seed {seed}, {ops} calls, worst {worst:.2e}, {time.time() - t0:.1f} s", flush=True)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print(f"FAIL {name}: seed {seed}: {type(e).__name__}: {str(e)[:300]}", flush=True)
        m.close()
print("failures:", bad)
sys.exit(1 if bad else 0)
