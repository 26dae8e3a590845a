"""Debug aid: a TP=1 model whose FULL shapes equal one rank's shard of Qwen3-32B at TP=8 (H 5120, 8 q heads, 1 kv head,
I 3200) runs exactly that rank's kernels (minus the all-reduce).  Prefill + decode steps vs the oracle, per step error."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nano_vllm_candle_amd as pkg  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests.util import oracle_config, rel_err  # noqa: E402

kw = dict(vocab_size=256, hidden_size=5120, head_dim=128, num_hidden_layers=1, num_attention_heads=8,
          num_key_value_heads=1, intermediate_size=3200)
for a in sys.argv[1:]:
    k, v = a.split("=")
    kw[k] = int(v)
cfg = pkg.Qwen3Config.tiny(**kw)
m = pkg.Qwen3ForCausalLM.from_synthetic(cfg, 0)
m.kv_alloc(8, 4, 64)
om = O.Model(oracle_config(O, cfg)).fill_synthetic(0)
rng = np.random.default_rng(7)
seqs = [rng.integers(0, cfg.vocab_size, int(n)).tolist() for n in (9, 31, 2, 17)]
for step in range(3):
    ids, lg = m.step([0, 1, 2, 3], seqs, step == 0, want_logits=True)
    rid, rlg = om.run_greedy(seqs)
    print("step", step, "errors", ["%.2e" % rel_err(g, r) for g, r in zip(lg, rlg)], flush=True)
    for s, t in zip(seqs, rid):
        s.append(int(t))
