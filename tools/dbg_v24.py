"""decode parity of the 24-bit cache modes with and without the fused attention prologue (debug)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nano_vllm_candle_amd as pkg
from oracle import oracle
from tests.util import oracle_config, row_rel_err
ctx = pkg.Context(0)
cfg = pkg.Qwen3Config.tiny(hidden_size=256, head_dim=128, num_attention_heads=4, num_key_value_heads=2, intermediate_size=512, num_hidden_layers=2, vocab_size=1024)
om = oracle.Model(oracle_config(oracle, cfg)).fill_synthetic(3)
for opts in ({"kv_v_bits": 24}, {"kv_v_bits": 24, "no_attn_prologue": 1}, {"kv_v_bits": 24, "kv_k_bits": 24}, {"kv_v_bits": 24, "kv_k_bits": 24, "no_attn_prologue": 1}, {}):
    for B in (1, 3, 40):
        m = pkg.Qwen3ForCausalLM.from_synthetic(cfg, 3, ctx)
        for k, v in opts.items(): m.set_option(k, v)
        m.kv_alloc(B * 2 + 2, B, 256)
        rng = np.random.default_rng(1)
        seqs = [rng.integers(0, cfg.vocab_size, int(n)).tolist() for n in rng.integers(1, 300, B)]
        errs = []
        for step in range(3):
            ids, lg = m.step(list(range(B)), seqs, step == 0, want_logits=True)
            rid, rlg = om.run_greedy(seqs)
            errs.append(max(row_rel_err(lg[i:i+1], rlg[i:i+1]) for i in range(B)))
            for s, t in zip(seqs, rid): s.append(int(t))
        print(opts, "B", B, "errors per step", ["%.1e" % e for e in errs], flush=True)
        m.close()
