"""Debug aid: N host threads, each with its OWN independent TP=1 model of the 32B/TP8 per-rank shapes, decode concurrently
on one GPU; every thread's logits vs the oracle.  Separates 'concurrent contexts' from 'tensor-parallel collectives'."""
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nano_vllm_candle_amd as pkg  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests.util import oracle_config, rel_err  # noqa: E402

n = int(sys.argv[1])
kw = dict(vocab_size=256, hidden_size=5120, head_dim=128, num_hidden_layers=1, num_attention_heads=8,
          num_key_value_heads=1, intermediate_size=3200)
for a in sys.argv[2:]:
    k, v = a.split("=")
    kw[k] = int(v)
cfg = pkg.Qwen3Config.tiny(**kw)
om = O.Model(oracle_config(O, cfg)).fill_synthetic(0)
rng = np.random.default_rng(7)
seqs = [rng.integers(0, cfg.vocab_size, int(x)).tolist() for x in (9, 31, 2, 17)]
steps = 4
res, errs = [None] * n, []
bar = threading.Barrier(n)


def worker(i):
    try:
        c = pkg.Context(0)
        mm = pkg.Qwen3ForCausalLM.from_synthetic(cfg, seed=0, ctx=c)
        mm.kv_alloc(8, 4, 64)
        my = [list(s) for s in seqs]
        out = []
        for step in range(steps):
            bar.wait()
            ids, lg = mm.step([0, 1, 2, 3], my, step == 0, want_logits=True)
            out.append(lg.copy())
            for s, t in zip(my, ids):
                s.append(int(t))
        res[i] = out
    except Exception as e:  # noqa: BLE001
        errs.append((i, repr(e)))
        bar.abort()


th = [threading.Thread(target=worker, args=(i,)) for i in range(n)]
[t.start() for t in th]
[t.join(timeout=600) for t in th]
print("errors:", errs)
ref = [list(s) for s in seqs]
for step in range(steps):
    rid, rlg = om.run_greedy(ref)
    print(f"step {step}:", ["%.1e" % max(rel_err(g, r) for g, r in zip(res[i][step], rlg)) for i in range(n)], flush=True)
    for s, t in zip(ref, rid):
        s.append(int(t))
