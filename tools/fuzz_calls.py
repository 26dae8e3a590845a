"""Fuzz campaign at the coarse seam (GPU): the seeded random caller of tests/util.py (new requests, shuffled decode subsets,
resident / pipelined decode, multi-token continuation, re-prefill, free / id reuse) over several model shapes, pool sizes,
chunk sizes and options, every call checked against the CPU oracle.   python tools/fuzz_calls.py [seeds per case]
python tools/fuzz_calls.py --random-configs N [first seed]: N random model shapes instead (hidden 128..2560, head_dim 64 / 128,
GQA groups 1..16, intermediate 128..9728, odd vocabularies, 1-3 layers; the layer shapes of Qwen3-1.7B and 4B among them)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nano_vllm_candle_amd as pkg  # noqa: E402
from oracle import oracle  # noqa: E402  (a tool, not the product: the oracle is the checker here)
from tests.util import oracle_config, random_calls  # noqa: E402

import numpy as np  # noqa: E402

random_n = 0
if len(sys.argv) > 1 and sys.argv[1] == "--random-configs":
    random_n = int(sys.argv[2])
    first_seed = int(sys.argv[3]) if len(sys.argv) > 3 else 500
    n_seeds = 1
else:
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
if random_n:
    cases = []
    rng = np.random.default_rng(first_seed)
    named = [dict(hidden_size=2048, head_dim=128, num_attention_heads=16, num_key_value_heads=8, intermediate_size=6144),   # 1.7B layer
             dict(hidden_size=2560, head_dim=128, num_attention_heads=32, num_key_value_heads=8, intermediate_size=9728)]   # 4B layer
    for i in range(random_n):
        if i < len(named):
            kw = dict(named[i], num_hidden_layers=1, vocab_size=1024)
        else:
            hd = int(rng.choice([64, 128]))
            nh = int(rng.choice([2, 4, 6, 8, 10, 12, 16, 20, 32])) * (128 // hd if hd == 64 else 1) // (2 if hd == 64 else 1)
            nh = max(nh, 128 // hd)
            kvs = [k for k in range(1, nh + 1) if nh % k == 0 and nh // k <= 16]
            kw = dict(hidden_size=int(rng.choice([128, 256, 384, 512, 640, 768, 1024, 1280, 1536, 2048])), head_dim=hd,
                      num_attention_heads=nh, num_key_value_heads=int(rng.choice(kvs)),
                      intermediate_size=int(rng.choice([128, 256, 384, 640, 768, 1024, 1536, 3072, 4352])),
                      vocab_size=int(rng.choice([512, 1008, 2048, 4112])), num_hidden_layers=int(rng.integers(1, 4)))
        big = kw["hidden_size"] * (kw["intermediate_size"] * 3 + kw["num_attention_heads"] * kw["head_dim"] * 2) > 30e6
        menu = short if big else short + [100, 255, 256, 257]
        opts = {}
        if kw["head_dim"] == 128 and rng.random() < 0.3:
            opts["kv_v_bits"] = 24
        if rng.random() < 0.25:
            opts["tile_min_wgs"] = 1
        cases.append((f"random {kw} {opts}", T(**kw), 64, int(rng.choice([6, 20, 28])), int(rng.choice([16, 48, 128, 512])),
                      int(rng.integers(2, 7)), menu, opts))
bad = 0
for name, cfg, NB, MS, mbt, max_new, menu, opts in cases:
    for seed in (range(first_seed, first_seed + 1) if random_n else range(100, 100 + n_seeds)):
        t0 = time.time()
        m = pkg.Qwen3ForCausalLM.from_synthetic(cfg, seed, ctx)
        for k, v in opts.items():
            m.set_option(k, v)
        m.kv_alloc(NB, MS, mbt)
        om = oracle.Model(oracle_config(oracle, cfg)).fill_synthetic(seed)
        try:
            ops, worst = random_calls(m, om, cfg, seed, 60 if random_n else 120, NB, MS, max_new=max_new, lens_menu=menu)
            print(f"ok   {name}: seed {seed}, {ops} calls, worst {worst:.2e}, {time.time() - t0:.1f} s", flush=True)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print(f"FAIL {name}: seed {seed}: {type(e).__name__}: {str(e)[:300]}", flush=True)
        m.close()
print("failures:", bad)
sys.exit(1 if bad else 0)
