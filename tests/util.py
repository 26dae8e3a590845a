import numpy as np


def oracle_config(O, cfg):
    return O.make_config(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, head_dim=cfg.head_dim,
                         num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                         num_key_value_heads=cfg.num_key_value_heads, intermediate_size=cfg.intermediate_size,
                         max_position_embeddings=cfg.max_position_embeddings, rms_norm_eps=cfg.rms_norm_eps,
                         rope_theta=cfg.rope_theta, bos_token_id=cfg.bos_token_id, eos_token_id=cfg.eos_token_id)


def rel_err(got, ref):
    """the parity metric, fixed once: max|got-ref| / max|ref| (per tensor / per logits row)"""
    return float(np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64)).max() / max(np.abs(ref).max(), 1e-30))


def row_rel_err(got, ref):
    return max(rel_err(g, r) for g, r in zip(got, ref))


def bf16_round(a):
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.view(np.float32)


LOGITS_TOL = 1e-3  # BASELINE.json north_star: logits within 1e-3 relative of the reference CPU path


def random_calls(m, om, cfg, seed, iters, NB, MS, lock=None, max_new=2, lens_menu=None, tol=LOGITS_TOL):
    """Drive model m (pool of NB blocks, MS sequence slots) with a seeded random caller and check every call against the
    oracle model om (its dense re-forward of each sequence on its own); returns (calls made, worst logits error).
    lock: serialises the oracle calls when several rank threads drive the same sequence (tensor-parallel test)."""
    import contextlib

    import numpy as np

    guard = lock if lock is not None else contextlib.nullcontext()
    rng = np.random.default_rng(seed)
    live = {}           # seq_id -> tokens the GPU has been shown plus the one it has just produced
    next_new = 0
    lens_menu = lens_menu or [1, 2, 15, 16, 17, 31, 33, 64, 100, 255, 256, 257, 300]
    worst = 0.0
    ties = []  # accepted near-ties (relative gap of the two logits in the oracle)
    cur_op = [""]  # for the assertion messages

    def blocks_of(n):
        return (n + 255) // 256

    def check(ids, got_ids, got_lg):
        nonlocal worst
        with guard:
            rids, rlgs = om.run_greedy([live[sid] for sid in ids])  # one dense re-forward, every sequence on its own row
        for k, sid in enumerate(ids):
            rid, rlg = rids[k:k + 1], rlgs[k:k + 1]
            if got_lg is not None:
                e = row_rel_err(got_lg[k:k + 1], rlg)
                worst = max(worst, e)
                assert e < tol, (cur_op[0], len(ids), sid, len(live[sid]), e)
            gid = int(got_ids[k])
            if gid != int(rid[0]):
                # two logits closer than twice the tolerance may swap places: a tie, not an error ("ids exact where the
                # margin is clear"); the caller goes on with the GPU's choice, as the device-fed decode already has
                gap = float(rlg[0, int(rid[0])] - rlg[0, gid]) / float(np.abs(rlg).max())
                assert 0.0 <= gap <= 2 * tol, (cur_op[0], len(ids), sid, len(live[sid]), gid, int(rid[0]), gap)
                ties.append(gap)
            live[sid].append(gid)

    def pool_ok():
        # the token just produced is not in the cache yet: a sequence of n shown tokens holds blocks for n
        assert m.free_blocks() == NB - sum(blocks_of(len(t) - 1) for t in live.values())

    ops = 0
    for it in range(iters):
        op = rng.choice(["add", "decode", "resident", "pipelined", "grow", "reprefill", "free", "sample"],
                        p=[0.22, 0.24, 0.11, 0.07, 0.09, 0.07, 0.12, 0.08])
        cur_op[0] = f"call {it} {op}"
        if op == "add" or not live:
            n_new = int(rng.integers(1, max_new + 1))
            ids, ps = [], []
            for _ in range(n_new):
                if len(live) + len(ids) >= MS:
                    break
                n = int(rng.choice(lens_menu))
                used = sum(blocks_of(len(t) + 4) for t in live.values()) + sum(blocks_of(len(p) + 4) for p in ps)
                if used + blocks_of(n + 4) > NB:
                    continue
                # an id freed earlier may come back (llm_engine.rs never reuses ids, the ABI allows it)
                sid = next_new if rng.random() < 0.7 else int(rng.integers(0, next_new + 1))
                if sid in live or sid in ids:
                    sid = next_new
                next_new = max(next_new, sid + 1)
                ids.append(sid)
                ps.append(rng.integers(3, cfg.vocab_size, n).tolist())
            if not ids:
                continue
            for sid, p in zip(ids, ps):
                live[sid] = list(p)
            got, lg = m.step(ids, [live[s] for s in ids], True, want_logits=True)
            check(ids, got, lg)
        elif op == "decode":
            ids = [int(s) for s in rng.permutation(list(live))[: int(rng.integers(1, len(live) + 1))]]
            if any(blocks_of(len(live[s])) > blocks_of(len(live[s]) - 1) for s in ids) and m.free_blocks() < len(ids):
                continue
            got, lg = m.step(ids, [live[s] for s in ids], False, want_logits=True)
            check(ids, got, lg)
        elif op == "sample":
            # sample_token on the device (llm_engine.rs:97-133) over a shuffled subset: logits against the oracle, the drawn ids
            # against the host mirror of the draw applied to the GPU's own logits (engine.py: sample_token_host)
            from nano_vllm_candle_amd.engine import sample_key, sample_token_host

            ids = [int(s) for s in rng.permutation(list(live))[: int(rng.integers(1, len(live) + 1))]]
            if m.free_blocks() < len(ids):
                continue
            temps = [float(t) for t in rng.choice([0.0, 0.5, 1.0, 2.0], len(ids))]
            sseed = int(rng.integers(0, 2**62))
            got, lg = m.step_sample(ids, [live[s] for s in ids], False, temps, sseed, want_logits=True)
            with guard:
                _, rlgs = om.run_greedy([live[sid] for sid in ids])
            for k, sid in enumerate(ids):
                e = row_rel_err(lg[k:k + 1], rlgs[k:k + 1])
                worst = max(worst, e)
                assert e < tol, (cur_op[0], len(ids), sid, len(live[sid]), e)
                want = sample_token_host(lg[k], temps[k], sample_key(sseed, sid, len(live[sid])))
                assert int(got[k]) == int(want), (cur_op[0], len(ids), sid, temps[k], int(got[k]), int(want))
                live[sid].append(int(got[k]))
        elif op in ("resident", "pipelined"):
            ids = [int(s) for s in rng.permutation(list(live))[: int(rng.integers(1, len(live) + 1))]]
            k = int(rng.integers(1, 4))
            if m.free_blocks() < len(ids) * 1:
                continue
            got, lg = m.step(ids, [live[s] for s in ids], False, want_logits=True)
            check(ids, got, lg)
            if op == "resident":
                for _ in range(k):
                    check(ids, m.decode_next(), None)
            else:
                for _ in range(k):
                    m.decode_enqueue()
                for _ in range(k):
                    check(ids, m.decode_collect(), None)
        elif op == "grow":
            # the caller shows a sequence again with several more tokens than the cache holds (its own tokens: any ids do)
            sid = int(rng.choice(list(live)))
            extra = int(rng.integers(2, 40))
            if blocks_of(len(live[sid]) + extra) - blocks_of(len(live[sid]) - 1) > m.free_blocks():
                continue
            live[sid] += rng.integers(3, cfg.vocab_size, extra).tolist()
            got, lg = m.step([sid], [live[sid]], False, want_logits=True)
            check([sid], got, lg)
        elif op == "reprefill":
            sid = int(rng.choice(list(live)))
            if m.free_blocks() < 1:
                continue
            got, lg = m.step([sid], [live[sid]], True, want_logits=True)
            check([sid], got, lg)
        else:
            sid = int(rng.choice(list(live)))
            m.seq_free(sid)
            del live[sid]
        pool_ok()
        ops += 1
    for sid in list(live):
        m.seq_free(sid)
    assert m.free_blocks() == NB
    assert len(ties) <= max(3, ops // 20), ties  # ties are rare by construction
    return ops, worst
