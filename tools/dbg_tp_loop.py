"""Debug aid: TP=N through the loopback communicator on one GPU with per-rank shapes of Qwen3-32B/TP8 (8 q heads, 1 kv
head, 3200 MLP columns per rank); prints every rank's logits error per step.  usage: dbg_tp_loop.py TP [key=value ...]"""
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nano_vllm_candle_amd as pkg  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests.util import oracle_config, rel_err  # noqa: E402

tp = int(sys.argv[1])
kw = dict(vocab_size=256 * tp, hidden_size=5120, head_dim=128, num_hidden_layers=1, num_attention_heads=8 * tp,
          num_key_value_heads=tp, intermediate_size=3200 * tp)
steps = 3
for a in sys.argv[2:]:
    k, v = a.split("=")
    if k == "steps":
        steps = int(v)
    else:
        kw[k] = int(v)
cfg = pkg.Qwen3Config.tiny(**kw)
om = O.Model(oracle_config(O, cfg)).fill_synthetic(0)
rng = np.random.default_rng(7)
seqs = [rng.integers(0, cfg.vocab_size, int(n)).tolist() for n in (9, 31, 2, 17)]
results, errors = [None] * tp, []


def worker(rank):
    try:
        c = pkg.Context(0, tp_rank=rank, tp_size=tp, loopback_group="dbg")
        mm = pkg.Qwen3ForCausalLM.from_synthetic(cfg, seed=0, ctx=c)
        mm.kv_alloc(8, 4, 64)
        mm.enable_taps(True)
        my = [list(s) for s in seqs]
        out = []
        for step in range(steps):
            ids, lg = mm.step(list(range(len(my))), my, step == 0, want_logits=True)
            taps = None
            if step > 0:  # decode: one row per sequence
                taps = [(mm.layer_tap(l, 0, len(my)), mm.layer_tap(l, 1, len(my))) for l in range(cfg.num_hidden_layers)]
            out.append((ids.copy(), lg.copy(), taps))
            for s, t in zip(my, ids):
                s.append(int(t))
        results[rank] = out
    except Exception as e:  # noqa: BLE001
        errors.append((rank, repr(e)))


th = [threading.Thread(target=worker, args=(r,)) for r in range(tp)]
[t.start() for t in th]
[t.join(timeout=600) for t in th]
print("errors:", errors)
ref = [list(s) for s in seqs]
for step in range(steps):
    rid, rlg = om.run_greedy(ref)
    for rank in range(tp):
        ids, lg, taps = results[rank][step]
        if taps is not None and rank == 0:
            for i, sq in enumerate(ref):
                hid, th, tr = om.forward(np.array([sq], np.uint32), trace=True)
                for l in range(cfg.num_hidden_layers):
                    print(f"   seq {i} layer {l}: h err {rel_err(taps[l][0][i], th[l, 0, -1]):.2e} res err {rel_err(taps[l][1][i], tr[l, 0, -1]):.2e}")
        print(f"step {step} rank {rank}: ids_ok {ids.tolist() == rid.tolist()} errs", ["%.2e" % rel_err(g, r) for g, r in zip(lg, rlg)],
              "same_as_rank0", bool(np.array_equal(lg, results[0][step][1])), flush=True)
    for s, t in zip(ref, results[0][step][0]):
        s.append(int(t))
