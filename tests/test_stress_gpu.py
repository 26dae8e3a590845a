"""GPU parity under Qwen3-LIKE weight statistics (generator profile 1, oracle/synth.h): full bf16 mantissas over five octaves,
layernorm weights in [2^-5, 2^5), q/k-norm weights in [2^-4, 2^4), four outlier hidden channels amplified x64 in
embed_tokens / o_proj / down_proj, so the residual stream carries massive activations (1e2..1e3 against a median near 1).
The reference's own real-weight vectors (tests/layer_test.rs:256-257, 274-275, 303-304, 342-343) need the author's Qwen3-0.6B
files, so this generator is the heaviest input the numerics decisions of DESIGN.md 5 (f16 K/V and P, bf16 hi+lo MFMA operands)
can be validated on here.  Same bar as everywhere: logits within 1e-3 of the f32 oracle (src/models/qwen3.rs:224-277,
src/layers/layernorm.rs:44-60), greedy ids equal where the oracle's top-2 margin is clear.  Every test prints its worst error
and the K/V pool's saturation count / largest stored magnitude (f16 clamps at 65504)."""
import threading

import numpy as np
import pytest

from tests.test_operating_point_gpu import check_rows, oracle_rows
from tests.util import oracle_config

pytestmark = pytest.mark.gpu

HEAVY = 1
# Measured (round 3, MI355X; tools/numerics_study.py reproduces the shares on the CPU): with the default 16-bit cache the heavy
# profile lands at 0.5e-3 .. 1.2e-3 -- the f16 rounding of V (not normalised, heavy-tailed through the layernorm weights) is the
# largest single term (6e-4 of ~7e-4), K 2..5e-4, P 1.5e-4.  The opt-in 24-bit V (kv_v_bits = 24: f16 + an e5m2 residual
# byte, V bytes x1.5) removes that term.  Bars: 24-bit V must hold the 1e-3 parity tolerance; the default must stay under
# DEFAULT_HEAVY_BOUND and is printed, so a drift is on record.
DEFAULT_HEAVY_BOUND = 2e-3
V_BITS = [48, 24, 16]  # 48: 24-bit K AND V (kv_k_bits = kv_v_bits = 24); 24: 24-bit V; 16: the default cache


def set_cache(m, bits):
    m.set_option("kv_v_bits", 24 if bits >= 24 else 16)
    m.set_option("kv_k_bits", 24 if bits == 48 else 16)


def tol_for(v_bits):
    from tests.util import LOGITS_TOL

    return LOGITS_TOL if v_bits >= 24 else DEFAULT_HEAVY_BOUND


@pytest.fixture(scope="module")
def pkg():
    import nano_vllm_candle_amd as p

    return p


@pytest.fixture(scope="module")
def ctx(pkg):
    from nano_vllm_candle_amd import layers

    return layers.default_context()


@pytest.fixture(scope="module")
def oracle_0_6b_heavy(oracle, pkg):
    cfg = pkg.Qwen3Config.qwen3_0_6b()
    return cfg, oracle.Model(oracle_config(oracle, cfg)).fill_synthetic(0, HEAVY)


def f16_bits_to_float(bits):
    return float(np.array([bits], np.uint16).view(np.float16)[0])


def kv_report(m, tag):
    sat = m.counter("kv_f16_saturated")
    mx = f16_bits_to_float(m.counter("kv_f16_absmax_bits"))
    print(f"[stress] {tag}: K/V elements at the f16 clamp: {sat}; largest |K|,|V| stored: {mx:.4g} (clamp 65504)")
    return sat, mx


def test_heavy_generator_matches_oracle_generator(pkg, ctx, oracle):
    # the two independent implementations of profile 1 (oracle/synth.h, csrc/synth_device.h), element for element
    import ctypes as C

    H = 1024
    cases = [("model.embed_tokens.weight", 0, 1, H), ("model.layers.3.mlp.down_proj.weight", 0, 2, 3072),
             ("model.layers.0.self_attn.o_proj.weight", 0, 2, 2048), ("model.layers.1.self_attn.q_proj.weight", 0, 0, 1),
             ("model.layers.2.input_layernorm.weight", 1, 1, H), ("model.norm.weight", 1, 1, H),
             ("model.layers.5.self_attn.k_norm.weight", 2, 0, 1)]
    for name, kind, axis, cols in cases:
        for seed in (0, 7):
            n = 1024 if kind else 40000
            first = 0 if kind else 3 * cols + 5
            want = oracle.synth_bf16(name, seed, kind, first, n, profile=HEAVY, axis=axis, cols=cols, hidden_size=H)
            got = np.empty(n, np.uint16)
            pkg._lib.check(pkg._lib.lib().nvllm_debug_synth_bf16_spec(ctx.h, name.encode(), seed, kind, HEAVY, axis, cols, H, first, n,
                                                                      got.ctypes.data_as(C.POINTER(C.c_uint16))), ctx.h)
            assert np.array_equal(got, want), (name, seed)
    # the statistics the profile promises (on the oracle's values): outlier channels exist and are x64
    w = oracle.synth_bf16("model.layers.3.mlp.down_proj.weight", 0, 0, 0, H * 3072, profile=HEAVY, axis=2, cols=3072, hidden_size=H)
    f = (w.astype(np.uint32) << 16).view(np.float32).reshape(H, 3072)
    row_max = np.abs(f).max(axis=1)
    assert 1 <= int((row_max > 1.0).sum()) <= 4 and np.abs(f).min() >= 2.0 ** -7


def test_heavy_profile_loaded_equals_generated(pkg, ctx, oracle):
    # whole-model form of the same check: a model generated in HBM == the oracle's tensors loaded through load_tensor
    cfg = pkg.Qwen3Config.tiny(vocab_size=512, hidden_size=256, head_dim=64, num_hidden_layers=2, num_attention_heads=4,
                               num_key_value_heads=2, intermediate_size=384)
    om = oracle.Model(oracle_config(oracle, cfg)).fill_synthetic(5, HEAVY)
    a = pkg.Qwen3ForCausalLM.from_synthetic(cfg, 5, ctx, profile=HEAVY)
    H, hd, nh, kv, I, V = cfg.hidden_size, cfg.head_dim, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.intermediate_size, cfg.vocab_size
    shapes = {"model.embed_tokens.weight": (V, H), "lm_head.weight": (V, H), "model.norm.weight": (H,)}
    for l in range(cfg.num_hidden_layers):
        p = f"model.layers.{l}."
        shapes.update({p + "self_attn.q_proj.weight": (nh * hd, H), p + "self_attn.k_proj.weight": (kv * hd, H),
                       p + "self_attn.v_proj.weight": (kv * hd, H), p + "self_attn.o_proj.weight": (H, nh * hd),
                       p + "mlp.gate_proj.weight": (I, H), p + "mlp.up_proj.weight": (I, H), p + "mlp.down_proj.weight": (H, I),
                       p + "input_layernorm.weight": (H,), p + "post_attention_layernorm.weight": (H,),
                       p + "self_attn.q_norm.weight": (hd,), p + "self_attn.k_norm.weight": (hd,)})
    b = pkg.Qwen3ForCausalLM.from_state_dict(cfg, {n: om.get_tensor(n, s) for n, s in shapes.items()}, ctx)
    seqs = [[5, 9, 200, 31, 77, 3, 18], [400, 3, 12]]
    for m in (a, b):
        m.kv_alloc(4, 2, 64)
    ia, la = a.step([0, 1], seqs, True, want_logits=True)
    ib, lb = b.step([0, 1], seqs, True, want_logits=True)
    assert np.array_equal(la, lb) and ia.tolist() == ib.tolist()
    rid, rlg = om.run_greedy(seqs)
    check_rows("tiny model, heavy profile", ia, la, rid, rlg)
    a.close()
    b.close()


@pytest.mark.parametrize("v_bits", V_BITS)
def test_0_6b_heavy_full_depth_long_contexts_vs_oracle(pkg, ctx, oracle_0_6b_heavy, v_bits):
    # tests/test_operating_point_gpu.py::test_0_6b_full_depth_long_contexts_vs_oracle on the heavy profile: 28 layers,
    # prompts {64, 292, 512, 511}, chunked prefill + 8 decode steps
    cfg, om = oracle_0_6b_heavy
    m = pkg.Qwen3ForCausalLM.from_synthetic(cfg, 0, ctx, profile=HEAVY)
    set_cache(m, v_bits)
    m.kv_alloc(num_blocks=12, max_seqs=4, max_batched_tokens=512)
    rng = np.random.default_rng(21)
    seqs = [rng.integers(0, cfg.vocab_size, n).tolist() for n in (64, 292, 512, 511)]
    worst = {}
    for step in range(9):
        ids, lg = m.step([0, 1, 2, 3], seqs, step == 0, want_logits=True)
        if step in (0, 1, 4, 8):
            rid, rlg = oracle_rows(om, seqs)
            worst[step] = check_rows(f"HEAVY 0.6B x28 layers, V {v_bits} bits, contexts {[len(s) for s in seqs]}, step {step}", ids, lg, rid, rlg,
                                     tol_for(v_bits))
        for s, t in zip(seqs, ids):
            s.append(int(t))
    print(f"[stress] 0.6B heavy full depth, V {v_bits} bits, worst error per checked step:", {k: f"{v:.2e}" for k, v in worst.items()})
    sat, _ = kv_report(m, "0.6B heavy, 4 sequences")
    assert sat == 0
    m.close()


@pytest.mark.parametrize("v_bits", V_BITS)
def test_0_6b_heavy_batch64_fused_decode_vs_oracle_sample(pkg, ctx, oracle_0_6b_heavy, v_bits):
    # the bench's own state (64 live sequences, prompts U[64,512] seed 0, fused batch-64 decode) on the heavy profile
    cfg, om = oracle_0_6b_heavy
    m = pkg.Qwen3ForCausalLM.from_synthetic(cfg, 0, ctx, profile=HEAVY)
    set_cache(m, v_bits)
    rng = np.random.default_rng(0)
    lens = rng.integers(64, 513, size=64)
    seqs = [rng.integers(0, cfg.vocab_size, size=int(n), dtype=np.uint32).tolist() for n in lens]
    m.kv_alloc(num_blocks=64 * 3, max_seqs=64, max_batched_tokens=4096)
    order = np.argsort(lens)
    sample = [int(order[-1]), int(order[0]), int(order[21]), int(order[42])]
    sids = list(range(64))
    ids, _ = m.step(sids, seqs, True)
    for s, t in zip(seqs, ids):
        s.append(int(t))
    for step in (1, 2, 3):
        ids, lg = m.step(sids, seqs, False, want_logits=True)
        if step in (1, 3):
            rid, rlg = oracle_rows(om, [seqs[i] for i in sample])
            check_rows(f"HEAVY 0.6B batch 64 fused decode, V {v_bits} bits, step {step}, sampled contexts {[len(seqs[i]) for i in sample]}",
                       ids[sample], lg[sample], rid, rlg, tol_for(v_bits))
        for s, t in zip(seqs, ids):
            s.append(int(t))
    sat, _ = kv_report(m, "0.6B heavy, batch 64")
    assert sat == 0
    m.close()


@pytest.mark.parametrize("v_bits", V_BITS)
def test_8b_layer_shapes_heavy_batch256_context4096_vs_oracle(pkg, ctx, oracle, v_bits):
    # configs[3] in miniature (tests/test_operating_point_gpu.py) on the heavy profile: Qwen3-8B layer shapes, 2 layers,
    # 256 live sequences, two of them at 4096 tokens of context
    cfg = pkg.Qwen3Config.tiny(vocab_size=4096, hidden_size=4096, head_dim=128, num_hidden_layers=2,
                               num_attention_heads=32, num_key_value_heads=8, intermediate_size=12288,
                               max_position_embeddings=8192)
    m = pkg.Qwen3ForCausalLM.from_synthetic(cfg, 0, ctx, profile=HEAVY)
    set_cache(m, v_bits)
    om = oracle.Model(oracle_config(oracle, cfg)).fill_synthetic(0, HEAVY)
    rng = np.random.default_rng(33)
    lens = [4096, 4000] + rng.integers(3, 200, 254).tolist()
    seqs = [rng.integers(0, cfg.vocab_size, int(n)).tolist() for n in lens]
    m.kv_alloc(num_blocks=2 * 17 + 254 + 4, max_seqs=256, max_batched_tokens=4096)
    sids = list(range(256))
    ids, _ = m.step(sids, seqs, True)
    for s, t in zip(seqs, ids):
        s.append(int(t))
    ids_a, lg_a = m.step([0, 1], seqs[:2], False, want_logits=True)
    rid, rlg = oracle_rows(om, seqs[:2])
    tol = tol_for(v_bits)
    check_rows(f"HEAVY 8B layer shapes, V {v_bits} bits, 2 rows at contexts 4097/4001 (split-KV + combine)", ids_a, lg_a, rid, rlg, tol)
    ids_b, lg_b = m.step(sids, seqs, False, want_logits=True)
    short = [2, 100, 255]
    rid_s, rlg_s = oracle_rows(om, [seqs[i] for i in short])
    check_rows(f"HEAVY 8B layer shapes, V {v_bits} bits, 256-row decode, short contexts", ids_b[short], lg_b[short], rid_s, rlg_s, tol)
    check_rows(f"HEAVY 8B layer shapes, V {v_bits} bits, 256-row decode, contexts 4097/4001", ids_b[:2], lg_b[:2], rid, rlg, tol)
    sat, _ = kv_report(m, "8B layer shapes heavy, batch 256")
    assert sat == 0 and np.isfinite(lg_b).all()
    m.close()


@pytest.mark.parametrize("oneshot,v_bits", [(0, 48), (0, 24), (1, 24), (0, 16)])
def test_32b_layer_shapes_heavy_tp8_shards_vs_oracle(pkg, oracle, oneshot, v_bits):
    # one layer at the Qwen3-32B shapes, TP = 8 through the loopback communicator, heavy profile: massive activations cross
    # the two all-reduces per layer as f32 partials (and, oneshot = 1, the device-side one-shot form)
    cfg = pkg.Qwen3Config.tiny(vocab_size=2048, hidden_size=5120, head_dim=128, num_hidden_layers=1,
                               num_attention_heads=64, num_key_value_heads=8, intermediate_size=25600)
    om = oracle.Model(oracle_config(oracle, cfg)).fill_synthetic(0, HEAVY)
    rng = np.random.default_rng(7)
    seqs = [rng.integers(0, cfg.vocab_size, int(n)).tolist() for n in rng.integers(2, 33, 20)]
    tp, steps = 8, 3
    results, errors = [None] * tp, []

    def worker(rank):
        try:
            c = pkg.Context(0, tp_rank=rank, tp_size=tp, loopback_group=f"g32b_heavy_{oneshot}_{v_bits}")
            mm = pkg.Qwen3ForCausalLM.from_synthetic(cfg, seed=0, ctx=c, profile=HEAVY)
            mm.set_option("oneshot_allreduce", oneshot)
            set_cache(mm, v_bits)
            mm.kv_alloc(len(seqs) + 4, len(seqs), 1024)
            my = [list(s) for s in seqs]
            out = []
            for step in range(steps):
                ids, lg = mm.step(list(range(len(my))), my, step == 0, want_logits=True)
                out.append((ids.copy(), lg.copy()))
                for s, t in zip(my, ids):
                    s.append(int(t))
            results[rank] = out
            mm.close()
            c.close()
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(tp)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors
    assert all(r is not None for r in results), "a rank hung"
    ref = [list(s) for s in seqs]
    for step in range(steps):
        rid, rlg = om.run_greedy(ref)
        for rank in (0, 5):
            ids, lg = results[rank][step]
            check_rows(f"HEAVY 32B layer shapes TP=8 (loopback, oneshot={oneshot}), V {v_bits} bits, rank {rank} step {step}", ids, lg, rid, rlg,
                       tol_for(v_bits))
        for s, t in zip(ref, rid):
            s.append(int(t))


def test_24_bit_v_cache_paths_on_a_small_model(pkg, ctx, oracle):
    # every writer and reader of the 24-bit V cache on a small head_dim-128 model, benign weights: a ragged batch whose prompts
    # go through the row kernel (qk_norm_rope_kvwrite) + attn_paged_kernel<.., 2, .., VLO>, decode through the fused
    # prologue's store_v24 + the VLO decode kernel, one sequence growing across a 256-token block boundary, a long one alone
    # (split-KV + combine), slots freed and reused, the device-feedback decode; then the pool goes back to 16 bits.
    # The 24-bit pool must never be WORSE than the 16-bit one beyond noise and must report its bytes; head_dim 64 refuses it.
    from tests.util import LOGITS_TOL, rel_err

    cfg = pkg.Qwen3Config.tiny(hidden_size=256, head_dim=128, num_attention_heads=4, num_key_value_heads=2, intermediate_size=512,
                               num_hidden_layers=3, vocab_size=1024)
    om = oracle.Model(oracle_config(oracle, cfg)).fill_synthetic(3)
    rng = np.random.default_rng(24)
    worst = {}
    for bits in (48, 24, 16):
        m = pkg.Qwen3ForCausalLM.from_synthetic(cfg, 3, ctx)
        set_cache(m, bits)
        m.kv_alloc(num_blocks=12, max_seqs=4, max_batched_tokens=128)
        assert m.kv_bytes_per_token == cfg.num_key_value_heads * cfg.head_dim * {48: 6, 24: 5, 16: 4}[bits] * cfg.num_hidden_layers
        rng = np.random.default_rng(24)
        seqs = [rng.integers(0, cfg.vocab_size, n).tolist() for n in (250, 37, 5)]  # 250 -> crosses 256 while decoding
        sids = [0, 1, 2]
        w = 0.0
        for step in range(10):
            ids, lg = m.step(sids, seqs, step == 0, want_logits=True)
            rid, rlg = om.run_greedy(seqs)
            w = max(w, max(rel_err(a, b) for a, b in zip(lg, rlg)))
            assert ids.tolist() == rid.tolist(), (bits, step)
            for s, t in zip(seqs, rid):
                s.append(int(t))
        nxt = m.decode_next()[:3]
        rid, _ = om.run_greedy(seqs)
        assert nxt.tolist() == rid.tolist()
        # free a slot, reuse it with a long prompt alone (chunked prefill at 128 rows, then split-KV decode)
        m.seq_free(1)
        long_seq = [rng.integers(0, cfg.vocab_size, 700).tolist()]
        for step in range(3):
            ids, lg = m.step([7], long_seq, step == 0, want_logits=True)
            rid, rlg = om.run_greedy(long_seq)
            w = max(w, rel_err(lg[0], rlg[0]))
            assert ids.tolist() == rid.tolist(), (bits, "long", step)
            long_seq[0].append(int(rid[0]))
        worst[bits] = w
        assert w < LOGITS_TOL, (bits, w)
        m.close()
    print(f"[stress] small model, worst logits error: 24-bit K+V {worst[48]:.3e}, 24-bit V {worst[24]:.3e}, 16-bit {worst[16]:.3e}")
    assert worst[24] < worst[16] * 1.2 and worst[48] < worst[24] * 1.2
    m64 = pkg.Qwen3ForCausalLM.from_synthetic(pkg.Qwen3Config.tiny(), 0, ctx)  # head_dim 64
    m64.set_option("kv_v_bits", 24)
    with pytest.raises(pkg._lib.NvllmError) as e:
        m64.kv_alloc(4, 2, 64)
    assert e.value.code == pkg._lib.EINVAL
    with pytest.raises(pkg._lib.NvllmError):
        m64.set_option("kv_v_bits", 20)
    m64.close()
    mk = pkg.Qwen3ForCausalLM.from_synthetic(cfg, 3, ctx)  # 24-bit K without 24-bit V: refused at kv_alloc
    mk.set_option("kv_k_bits", 24)
    with pytest.raises(pkg._lib.NvllmError) as e:
        mk.kv_alloc(4, 2, 64)
    assert e.value.code == pkg._lib.EINVAL
    mk.close()
