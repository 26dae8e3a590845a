"""Generates tests/golden/tiny_golden.npz from the CPU oracle (oracle/qwen3_oracle.c) on the tiny
synthetic config.  The reference itself cannot produce these (Rust, not buildable here; its own
full-model tests need local Qwen3 files): full-model parity is therefore pinned to the oracle, which
is pinned op-by-op by the reference's known answers (tests/test_oracle.py) and cross-checked against
transformers' Qwen3 (oracle/validate_vs_hf.py).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

TINY = dict(vocab_size=512, hidden_size=128, head_dim=64, num_hidden_layers=2, num_attention_heads=4,
            num_key_value_heads=2, intermediate_size=256, max_position_embeddings=4096, rms_norm_eps=1e-6,
            rope_theta=1e6, bos_token_id=1, eos_token_id=2)
SEED = 0
STEPS = 32


def main():
    rng = np.random.default_rng(1234)
    m = O.Model(O.make_config(**TINY)).fill_synthetic(SEED)
    out = {"seed": np.array(SEED), "steps": np.array(STEPS)}
    for k, v in TINY.items():
        out["cfg_" + k] = np.array(v)
    cases = {"b1": [rng.integers(0, 512, 16).tolist()],
             "b4": [rng.integers(0, 512, n).tolist() for n in (5, 23, 1, 40)]}
    for name, prompts in cases.items():
        seqs = [list(p) for p in prompts]
        logits, ids = [], []
        for _ in range(STEPS + 1):  # prefill + STEPS decode steps
            nxt, lg = m.run_greedy(seqs)
            logits.append(lg.copy())
            ids.append(nxt.copy())
            for s, t in zip(seqs, nxt):
                s.append(int(t))
        out[name + "_prompt_lens"] = np.array([len(p) for p in prompts])
        out[name + "_prompts"] = np.concatenate([np.array(p, np.uint32) for p in prompts])
        out[name + "_logits"] = np.stack(logits).astype(np.float32)  # [STEPS+1, B, V]
        out[name + "_ids"] = np.stack(ids).astype(np.uint32)         # [STEPS+1, B]
    # per-layer taps for the b1 prefill
    ids = np.array([cases["b1"][0]], np.uint32)
    hidden, th, tr = m.forward(ids, trace=True)
    out["b1_layer_h"] = th[:, 0]
    out["b1_layer_res"] = tr[:, 0]
    out["b1_hidden"] = hidden[0]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tiny_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
