"""CPU study of the KV-cache / P rounding choices at full model size: the oracle's attention core with
chosen operands rounded (oracle/qwen3_oracle.c oq3_set_study) against its own f32 result, on either synthetic profile.
    python tools/numerics_study.py [--profile 1] [--layers 28] [--tokens 300]
Prints the logits error (max|d| / max|ref| of the last row, the parity metric) per variant."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--profile", type=int, default=1)
ap.add_argument("--layers", type=int, default=28)
ap.add_argument("--tokens", type=int, default=300)
ap.add_argument("--vocab", type=int, default=8192)
ap.add_argument("--seeds", type=int, default=2)
a = ap.parse_args()
cfg = O.make_config(vocab_size=a.vocab, hidden_size=1024, head_dim=128, num_hidden_layers=a.layers, num_attention_heads=16,
                    num_key_value_heads=8, intermediate_size=3072)
VARIANTS = [("K f16", 1), ("V f16", 2), ("P f16", 4), ("K+V+P f16 (shipped round 2)", 7), ("K bf16", 8), ("V bf16", 16),
            ("K f16x2", 32), ("K f16x2 + V f16 + P f16", 32 | 2 | 4), ("K f16 + V f16 + P bf16x2", 1 | 2 | 128),
            ("K f16x2 + V f16 + P bf16x2", 32 | 2 | 128), ("K f16 + V f16x2 + P bf16x2", 1 | 64 | 128),
            ("V f16 + e5m2 lo", 256), ("K f16 + V f16+e5m2 + P f16", 1 | 256 | 4),
            ("K f16 + e5m2 lo", 512), ("K, V f16+e5m2 + P f16 hi+lo (shipped 24-bit K+V)", 512 | 256)]
for seed in range(a.seeds):
    m = O.Model(cfg).fill_synthetic(seed, a.profile)
    ids = np.random.default_rng(100 + seed).integers(0, a.vocab, (1, a.tokens))
    O.set_study(0)
    ref = m.compute_logits(m.forward(ids))[0]
    rows = [a.tokens // 4, a.tokens // 2, a.tokens - 1]
    print(f"profile {a.profile} seed {seed}: {a.layers} layers, {a.tokens} tokens; error at rows {rows} (worst)")
    for name, flags in VARIANTS:
        O.set_study(flags)
        lg = m.compute_logits(m.forward(ids))[0]
        err = max(float(np.abs(lg[r] - ref[r]).max() / np.abs(ref[r]).max()) for r in rows)
        print(f"  {name:34s} {err:.3e}")
    O.set_study(0)
