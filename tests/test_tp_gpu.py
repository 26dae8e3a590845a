"""GPU: the tensor-parallel code path end to end on ONE GPU through the in-process loopback communicator
(one host thread per rank): sharded synthetic load, per-rank kernels on kv-head / MLP-column / vocab shards,
the two all-reduces per layer, vocab-parallel greedy ids and logits gather.  Oracle for TP=N is TP=1 (the
reference has no working TP): logits within tolerance (the order of the sum differs), ids equal."""
import threading

import numpy as np
import pytest

from tests.util import LOGITS_TOL, oracle_config, random_calls, row_rel_err

pytestmark = pytest.mark.gpu


def _run_tp(pkg, cfg, tp, seqs, steps, group, options=None, max_batched_tokens=40):
    results = [None] * tp
    errors = []

    def worker(rank):
        try:
            ctx = pkg.Context(0, tp_rank=rank, tp_size=tp, loopback_group=group)
            m = pkg.Qwen3ForCausalLM.from_synthetic(cfg, seed=4, ctx=ctx)
            for k, v in (options or {}).items():
                m.set_option(k, v)
            m.kv_alloc(8, 4, max_batched_tokens)
            my = [list(s) for s in seqs]
            out = []
            for step in range(steps):
                ids, lg = m.step(list(range(len(my))), my, step == 0, want_logits=True)
                out.append((ids.copy(), lg.copy()))
                for s, t in zip(my, ids):
                    s.append(int(t))
            nxt = m.decode_next()[:len(my)].copy()  # device-feedback path under TP
            results[rank] = (out, nxt, m.counter("tile_gemm_launches"))
            m.close()
            ctx.close()
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(tp)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert all(r is not None for r in results), "a rank hung"
    return results


@pytest.mark.parametrize("tp,kw", [(2, dict()),
                                   (4, dict(hidden_size=256, num_attention_heads=8, num_key_value_heads=4, head_dim=64,
                                            intermediate_size=512, num_hidden_layers=3, vocab_size=1024))])
def test_tp_equals_tp1(tp, kw, oracle):
    import nano_vllm_candle_amd as pkg

    cfg = pkg.Qwen3Config.tiny(**kw)
    rng = np.random.default_rng(8)
    seqs = [rng.integers(0, cfg.vocab_size, n).tolist() for n in (13, 50, 3)]  # 50 > 40: chunked prefill
    steps = 5
    om = oracle.Model(oracle_config(oracle, cfg)).fill_synthetic(4)
    res = _run_tp(pkg, cfg, tp, seqs, steps, f"g{tp}")
    ref_seqs = [list(s) for s in seqs]
    for step in range(steps):
        rid, rlg = om.run_greedy(ref_seqs)
        for rank in range(tp):
            ids, lg = res[rank][0][step]
            assert row_rel_err(lg, rlg) < LOGITS_TOL, (rank, step)
            assert ids.tolist() == rid.tolist(), (rank, step)
            assert np.array_equal(lg, res[0][0][step][1])  # every rank returns the same gathered logits
        for s, t in zip(ref_seqs, rid):
            s.append(int(t))
    rid, _ = om.run_greedy(ref_seqs)
    for rank in range(tp):
        assert res[rank][1].tolist() == rid.tolist()


def test_tp_random_call_sequences_match_the_oracle(oracle):
    # the seeded random caller of tests/test_model_gpu.py on a TP = 2 group: every rank makes the same calls (the reference's
    # one-process-per-rank model, tp.rs:21-31) and checks ids and gathered logits against the oracle by itself
    import nano_vllm_candle_amd as pkg

    cfg = pkg.Qwen3Config.tiny()
    tp, NB, MS, seed = 2, 20, 6, 21
    lock = threading.Lock()
    om = oracle.Model(oracle_config(oracle, cfg)).fill_synthetic(seed)
    results, errors = [None] * tp, []

    def worker(rank):
        try:
            ctx = pkg.Context(0, tp_rank=rank, tp_size=tp, loopback_group="grandom")
            m = pkg.Qwen3ForCausalLM.from_synthetic(cfg, seed=seed, ctx=ctx)
            m.kv_alloc(NB, MS, 96)
            results[rank] = random_calls(m, om, cfg, seed, 60, NB, MS, lock=lock)
            m.close()
            ctx.close()
        except BaseException as e:  # noqa: BLE001
            errors.append((rank, repr(e)))

    threads = [threading.Thread(target=worker, args=(r,), daemon=True) for r in range(tp)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=240)
    assert not errors, errors
    assert all(r is not None for r in results), "a rank hung"
    assert results[0] == results[1] and results[0][0] >= 30, results
    print(f"TP = 2 random call sequences: {results[0][0]} calls, worst logits error {results[0][1]:.2e}")


@pytest.mark.parametrize("fuse_qk", [1, 0])
def test_tp_prompt_chunk_through_the_tile_gemm(oracle, fuse_qk):
    # the driver's tensor-parallel leg prefills 4096-row chunks, i.e. the tile GEMM (tile_gemm.hip) on per-rank shard
    # shapes with the all-reduce behind o_proj / down_proj: forced here at TP = 2 on a model whose shards fit its blocks
    # (head_dim 128: the QKV epilogue variant too), 390 prompt rows in one chunk, then decode on the cache it filled
    import nano_vllm_candle_amd as pkg

    cfg = pkg.Qwen3Config.tiny(hidden_size=512, num_attention_heads=8, num_key_value_heads=4, head_dim=128,
                               intermediate_size=1024, num_hidden_layers=2, vocab_size=2048)
    rng = np.random.default_rng(11)
    seqs = [rng.integers(0, cfg.vocab_size, n).tolist() for n in (150, 97, 143)]
    steps, tp = 3, 2
    om = oracle.Model(oracle_config(oracle, cfg)).fill_synthetic(4)
    res = _run_tp(pkg, cfg, tp, seqs, steps, f"gtile{fuse_qk}", options={"tile_min_wgs": 1, "tile_fuse_qk": fuse_qk},
                  max_batched_tokens=512)
    assert all(r[2] == 4 * cfg.num_hidden_layers for r in res), [r[2] for r in res]  # all four projections of the prompt chunk
    ref_seqs = [list(s) for s in seqs]
    for step in range(steps):
        rid, rlg = om.run_greedy(ref_seqs)
        for rank in range(tp):
            ids, lg = res[rank][0][step]
            assert row_rel_err(lg, rlg) < LOGITS_TOL, (rank, step, row_rel_err(lg, rlg))
            assert ids.tolist() == rid.tolist(), (rank, step)
        for s, t in zip(ref_seqs, rid):
            s.append(int(t))


def test_oneshot_allreduce_gives_up_with_an_error_code_on_every_rank(oracle):
    # The give-up path of the one-shot all-reduce (csrc/oneshot.hip), run ONCE: rank 2 of a TP = 4 loopback group "forgets"
    # one push during a decode step; its peers' wait kernels (and its own: it polls its own flag too) must hit their spin
    # bound instead of hanging the stream, every rank must return NVLLM_ERCCL from that step -- the error words are gathered
    # so that nobody leaves early and strands the others in the (max, index) gather -- and the group must be usable again:
    # a following kv_alloc + prefill + decode on the loopback communicator's own all-reduce matches the oracle.
    import time

    import nano_vllm_candle_amd as pkg

    cfg = pkg.Qwen3Config.tiny(hidden_size=256, num_attention_heads=8, num_key_value_heads=4, head_dim=64,
                               intermediate_size=512, num_hidden_layers=2, vocab_size=1024)
    rng = np.random.default_rng(3)
    seqs = [rng.integers(0, cfg.vocab_size, n).tolist() for n in (9, 21, 5)]
    tp = 4
    om = oracle.Model(oracle_config(oracle, cfg)).fill_synthetic(4)
    results, errors = [None] * tp, []

    def worker(rank):
        try:
            ctx = pkg.Context(0, tp_rank=rank, tp_size=tp, loopback_group="giveup")
            m = pkg.Qwen3ForCausalLM.from_synthetic(cfg, seed=4, ctx=ctx)
            m.set_option("oneshot_allreduce", 1)
            m.set_option("oneshot_spins", 20000)  # ~ milliseconds instead of seconds
            m.kv_alloc(8, 4, 64)
            my = [list(s) for s in seqs]
            ids, _ = m.step([0, 1, 2], my, True)
            for s, t in zip(my, ids):
                s.append(int(t))
            assert m.counter("oneshot_calls") > 0
            if rank == 2:
                m.set_option("oneshot_skip_push", 1)
            t0 = time.perf_counter()
            code = None
            try:
                m.step([0, 1, 2], my, False)
            except pkg._lib.NvllmError as e:
                code = e.code
            dt = time.perf_counter() - t0
            # afterwards: fresh pool on the communicator's all-reduce, same sequences from scratch
            m.set_option("oneshot_allreduce", 0)
            m.kv_alloc(8, 4, 64)
            my = [list(s) for s in seqs]
            out = []
            for step in range(3):
                ids, lg = m.step([0, 1, 2], my, step == 0, want_logits=True)
                out.append((ids.copy(), lg.copy()))
                for s, t in zip(my, ids):
                    s.append(int(t))
            results[rank] = (code, dt, out)
            m.close()
            ctx.close()
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(tp)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    assert all(r is not None for r in results), "a rank hung"
    assert [r[0] for r in results] == [pkg._lib.ERCCL] * tp, [r[0] for r in results]
    assert max(r[1] for r in results) < 20.0, [r[1] for r in results]  # promptly: bounded spins, not a stuck stream
    ref = [list(s) for s in seqs]
    for step in range(3):
        rid, rlg = om.run_greedy(ref)
        for rank in range(tp):
            ids, lg = results[rank][2][step]
            assert row_rel_err(lg, rlg) < LOGITS_TOL and ids.tolist() == rid.tolist(), (rank, step)
        for s, t in zip(ref, rid):
            s.append(int(t))
