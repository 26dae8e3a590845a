"""How long does the HOST need to enqueue one decode step (0.6B, 64 sequences) vs how long the GPU needs to run it?"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nano_vllm_candle_amd as pkg

ctx = pkg.Context(0)
cfg = pkg.Qwen3Config.qwen3_0_6b()
m = pkg.Qwen3ForCausalLM.from_synthetic(cfg, 0, ctx)
m.kv_alloc(64 * 4, 64, 4096)
rng = np.random.default_rng(0)
seqs = [rng.integers(0, cfg.vocab_size, int(n)).tolist() for n in rng.integers(64, 513, 64)]
m.step(list(range(64)), seqs, True)
for _ in range(4):
    m.decode_next()
ctx.synchronize()
host = []
for rep in range(10):
    t0 = time.perf_counter()
    for _ in range(3):
        m.decode_enqueue()
    t1 = time.perf_counter()
    for _ in range(3):
        m.decode_collect()
    t2 = time.perf_counter()
    host.append(((t1 - t0) / 3 * 1e3, (t2 - t0) / 3 * 1e3))
print("host ms per enqueue (3 back-to-back) / total ms per step incl. GPU:", [(round(a, 3), round(b, 3)) for a, b in host])
