"""Stress for the prompt-chunk kernels (tile GEMM ring, prefill attention ring, 16-row norm) under memory load: several
contexts prefill the same prompts at once on one GPU, repeatedly; every result must equal the solo run bit for bit
(the kernels are deterministic; an LDS-DMA stage read before it was published shows up only under load).
Usage: python tools/dbg_conc_prefill.py [threads] [rounds]"""
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nano_vllm_candle_amd as pkg  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cfg = pkg.Qwen3Config.qwen3_0_6b()
cfg.num_hidden_layers = 6
rng = np.random.default_rng(5)
prompts = [rng.integers(0, cfg.vocab_size, int(n)).tolist() for n in (61, 293, 510, 515, 130, 77, 402, 333, 256, 199, 64, 480)]
N = len(prompts)


def make(ctx):
    m = pkg.Qwen3ForCausalLM.from_synthetic(cfg, 0, ctx)
    m.kv_alloc(num_blocks=3 * N + 2, max_seqs=N, max_batched_tokens=4096)
    return m


ctx0 = pkg.Context(0)
m0 = make(ctx0)
ref_ids, ref_lg = m0.step(list(range(N)), prompts, True, want_logits=True)
assert m0.counter("tile_gemm_launches") > 0, "the tile GEMM did not run: pick more rows"
print("solo run: rows", sum(len(p) for p in prompts), "tile launches", m0.counter("tile_gemm_launches"))
bad = []


def worker(i):
    ctx = pkg.Context(0)
    m = make(ctx)
    for r in range(ROUNDS):
        m.kv_alloc(num_blocks=3 * N + 2, max_seqs=N, max_batched_tokens=4096)
        ids, lg = m.step(list(range(N)), prompts, True, want_logits=True)
        if not (np.array_equal(ids, ref_ids) and np.array_equal(lg, ref_lg)):
            bad.append((i, r, float(np.abs(lg - ref_lg).max())))
    m.close()
    ctx.close()


ths = [threading.Thread(target=worker, args=(i,)) for i in range(T)]
for t in ths:
    t.start()
for t in ths:
    t.join()
print(f"{T} contexts x {ROUNDS} rounds:", "all bit-equal to the solo run" if not bad else f"MISMATCHES {bad[:8]}")
sys.exit(1 if bad else 0)
