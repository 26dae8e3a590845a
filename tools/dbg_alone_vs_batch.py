"""Why do a sequence's last-row logits differ between 'alone' (1-row chunk) and 'inside a 256-row decode step' although the
K/V cache holds the same bits?  8B layer shapes, few layers, context 4097; prints max|d|/max|ref| for option variants.
    python tools/dbg_alone_vs_batch.py [--layers 8] [--ctx 4096] [--batch 256]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nano_vllm_candle_amd as pkg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--layers", type=int, default=8)
ap.add_argument("--ctx", type=int, default=4096)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--vocab", type=int, default=8192)
a = ap.parse_args()
cfg = pkg.Qwen3Config.tiny(vocab_size=a.vocab, hidden_size=4096, head_dim=128, num_hidden_layers=a.layers, num_attention_heads=32,
                           num_key_value_heads=8, intermediate_size=12288, max_position_embeddings=8192)
ctx = pkg.Context(0)
rng = np.random.default_rng(8)
B, T = a.batch, a.ctx
prompts = [rng.integers(0, cfg.vocab_size, T if i < 2 else int(rng.integers(3, 200)), dtype=np.uint32).tolist() for i in range(B)]


def rel(x, y):
    return float(np.abs(x - y).max() / np.abs(y).max())


def run(opts_batch, opts_alone):
    m = pkg.Qwen3ForCausalLM.from_synthetic(cfg, 0, ctx)
    for k, v in opts_batch.items():
        m.set_option(k, v)
    blocks = 2 * (T // 256 + 2) + B + 40
    m.kv_alloc(num_blocks=blocks, max_seqs=B + 2, max_batched_tokens=4096)
    my = [list(p) for p in prompts]
    ids0, _ = m.step(list(range(B)), my, True)
    for p, t in zip(my, ids0):
        p.append(int(t))
    _, lg = m.step(list(range(B)), my, False, want_logits=True)
    for k, v in opts_alone.items():
        m.set_option(k, v)
    _, l1 = m.step([1000], [my[0]], True, want_logits=True)
    # the same sequence decoded alone on the cache the BATCH prefill wrote: only the last token's path differs
    _, l2 = m.step([0], [my[0]], False, want_logits=True)
    m.close()
    return rel(l1[0], lg[0]), rel(l2[0], lg[0]), rel(l1[0], l2[0])


for name, ob, oa in (("default", {}, {}), ("alone: no_fused", {}, {"no_fused": 1}), ("both: no_attn_prologue", {"no_attn_prologue": 1}, {}),
                     ("both: kv_v_bits 24", {"kv_v_bits": 24}, {})):
    r = run(ob, oa)
    print(f"{name:28s} fresh-prefill-alone vs batch {r[0]:.3e} | decode-alone-on-batch-cache vs batch {r[1]:.3e} | fresh vs decode-alone {r[2]:.3e}", flush=True)
