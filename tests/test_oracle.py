"""Pins the CPU oracle with every weight-free known answer the reference's own tests hold for this
path.  CPU only."""
import numpy as np


def test_silu_and_mul_known_answer(oracle):
    # reference: src/layers/activation.rs:26-36  [0,1,-1,2] -> [0, 1.4621172]
    out = oracle.silu_mul(np.array([[0.0, 1.0, -1.0, 2.0]], np.float32))
    assert abs(out[0, 0] - 0.0) < 1e-6
    assert abs(out[0, 1] - 1.4621172) < 1e-5


def test_rmsnorm_simple_known_answer(oracle):
    # reference: src/layers/layernorm.rs:68-89  eps 1e-5, ones weight
    x = np.array([[1, 2, 3, 4], [1, 1, 1, 1]], np.float32)
    y, res = oracle.rmsnorm(x, np.ones(4, np.float32), 1e-5)
    assert res is None
    assert y.shape == (2, 4)
    assert abs(y[0, 0] - 0.36515) < 1e-4
    assert abs(y[0, 3] - 1.46059) < 1e-4


def test_rmsnorm_with_residual_known_answer(oracle):
    # reference: src/layers/layernorm.rs:92-109
    x = np.full((1, 3), 0.5, np.float32)
    y, res = oracle.rmsnorm(x, np.ones(3, np.float32), 1e-6, residual=x.copy())
    assert res is not None
    assert res[0].tolist() == [1.0, 1.0, 1.0]
    assert abs(y[0, 0] - 1.0) < 1e-5


_W6 = np.array([[1, 0, -1, 2], [0, 1, 2, -1], [2, -1, 0, 1], [-2, 1, 1, 0], [1, 1, 1, 1], [3, 0, -2, 1]], np.float32)
_B6 = np.array([1, -2, 0, 3, -1, 2], np.float32)
_X = np.array([[1, 2, 3, 4], [-1, 0, 1, 2]], np.float32)


def test_replicated_linear_known_answer(oracle):
    # reference: src/layers/linear.rs:232-270 (exact)
    y = oracle.linear(_X, _W6, _B6)
    assert y.tolist() == [[7, 2, 4, 6, 9, 3], [3, -2, 0, 6, 1, -1]]


def test_column_parallel_two_rank_split_known_answer(oracle):
    # reference: src/layers/linear.rs:273-322: rank r holds rows [3r, 3r+3) of W and of the bias
    y0 = oracle.linear(_X, _W6[:3], _B6[:3])
    y1 = oracle.linear(_X, _W6[3:], _B6[3:])
    assert y0.tolist() == [[7, 2, 4], [3, -2, 0]]
    assert y1.tolist() == [[6, 9, 3], [6, 1, -1]]


def test_qkv_fused_equals_separate(oracle):
    # reference: src/layers/linear.rs:325-354
    wq = np.array([[1, 0, -1, 2], [0, 1, 2, -1]], np.float32)
    wk = np.array([[2, -1, 0, 1]], np.float32)
    wv = np.array([[-2, 1, 1, 0]], np.float32)
    x = np.array([[1, 2, 3, 4]], np.float32)
    out = oracle.linear(x, np.concatenate([wq, wk, wv], 0))
    assert out[:, 0:2].tolist() == (x @ wq.T).tolist()
    assert out[:, 2:3].tolist() == (x @ wk.T).tolist()
    assert out[:, 3:4].tolist() == (x @ wv.T).tolist()


def test_rope_norm_preserved_small(oracle):
    # reference: src/layers/rotary_embedding.rs:115-137  hd 8, base 1e4, std 0.01, diff < 1e-5
    rng = np.random.default_rng(0)
    q = (0.01 * rng.standard_normal((1, 2, 4, 8))).astype(np.float32)
    qr = oracle.rope_apply(q, 10000.0)
    assert np.abs((q ** 2).sum(-1) - (qr ** 2).sum(-1)).sum() < 1e-5


def test_rope_values_hd128(oracle):
    # reference: tests/layer_test.rs:440-503  hd 128, base 1e6: norms kept (<1e-3), pos 0 != pos 1
    rng = np.random.default_rng(1)
    q = rng.standard_normal((1, 2, 4, 128)).astype(np.float32)
    qr = oracle.rope_apply(q, 1000000.0)
    assert np.abs((q ** 2).sum(-1) - (qr ** 2).sum(-1)).sum() < 1e-3
    ones = np.ones((1, 2, 4, 128), np.float32)
    r = oracle.rope_apply(ones, 1000000.0)
    assert np.abs(r[0, :, 0] - r[0, :, 1]).sum() > 1.0
    # position 0 is the identity rotation
    assert np.array_equal(r[0, :, 0], ones[0, :, 0])


def test_rope_table_formula(oracle):
    # rotary_embedding.rs:56-80: inv_freq = 1/base^(2j/hd) in f32, angle = pos*inv_freq in f32
    cos, sin = oracle.rope_table(8, 10000.0, 5)
    j = np.arange(4, dtype=np.float32)
    inv = (np.float32(1.0) / np.power(np.float32(10000.0), (np.float32(2.0) * j) / np.float32(8.0))).astype(np.float32)
    ang = (np.arange(5, dtype=np.float32)[:, None] * inv[None]).astype(np.float32)
    assert np.allclose(cos, np.cos(ang), atol=1e-6) and np.allclose(sin, np.sin(ang), atol=1e-6)


def test_gqa_interleaved_expand(oracle):
    # reference: tests/debug_layer_test.rs:38,49-70: kv heads [h0,h1] serve q heads [h0,h0,h1,h1].
    # one token, softmax over a single key is 1 -> ctx of q-head h == v of kv-head h//rep
    v = np.array([1, 2, 3, 4, 5, 6], np.float32).reshape(1, 2, 1, 3)
    q = np.ones((1, 4, 1, 3), np.float32)
    k = np.ones((1, 2, 1, 3), np.float32)
    ctx = oracle.attention(q, k, v)
    assert ctx.reshape(-1).tolist() == [1, 2, 3, 1, 2, 3, 4, 5, 6, 4, 5, 6]


def test_tp_shard_math(oracle):
    # reference: src/tp.rs:94-98
    L = oracle.lib()
    assert L.oq3_tp_shard_size(100, 4) == 25
    assert L.oq3_tp_shard_offset(100, 4, 2) == 50


def test_argmax_last_max_wins(oracle):
    # llm_engine.rs:135-142: Iterator::max_by returns the LAST maximal element
    assert oracle.argmax_last(np.array([1, 5, 3, 5, 2], np.float32)) == 3
    assert oracle.argmax_last(np.array([7, 7, 7], np.float32)) == 2


def test_causal_invariance_of_position0_logits(oracle):
    # property from tests/layer_test.rs:164-202 on synthetic weights: pos-0 logits alone == in a 2-token seq
    m = oracle.Model(oracle.make_config()).fill_synthetic(0)
    a = m.compute_logits(m.forward(np.array([[7]], np.uint32)))[0, 0]
    b = m.compute_logits(m.forward(np.array([[7, 11]], np.uint32)))[0, 0]
    assert np.abs(a - b).mean() < 1e-6
    assert np.isfinite(a).all() and a.var() > 1e-6  # layer_test.rs:70,354-357 (no NaN/Inf, non-degenerate)


def test_right_padding_does_not_change_real_rows(oracle):
    # llm_engine.rs:80-90 pads with eos on the right; causal mask keeps real rows independent
    m = oracle.Model(oracle.make_config()).fill_synthetic(0)
    s0, s1 = [5, 9, 200, 31, 77], [400, 3]
    ids_b, lg_b = m.run_greedy([s0, s1])
    ids_0, lg_0 = m.run_greedy([s0])
    ids_1, lg_1 = m.run_greedy([s1])
    assert ids_b.tolist() == [ids_0[0], ids_1[0]]
    assert np.abs(lg_b[0] - lg_0[0]).max() < 1e-5 and np.abs(lg_b[1] - lg_1[0]).max() < 1e-5


def test_all_rows_mode_matches_last_row_mode(oracle):
    m = oracle.Model(oracle.make_config()).fill_synthetic(3)
    a = m.run_greedy([[1, 2, 3], [4, 5, 6, 7]], all_rows=True)
    b = m.run_greedy([[1, 2, 3], [4, 5, 6, 7]], all_rows=False)
    assert a[0].tolist() == b[0].tolist() and np.array_equal(a[1], b[1])


def test_synth_values_are_bf16_exact(oracle):
    m = oracle.Model(oracle.make_config()).fill_synthetic(0)
    w = m.get_tensor("model.layers.0.self_attn.q_proj.weight", (128, 64))
    bits = w.view(np.uint32)
    assert (bits & 0xFFFF == 0).all()
    assert np.abs(w).max() <= 128 * 2.0 ** -12 and w.std() > 0.01
    g = oracle.synth_bf16("model.layers.0.self_attn.q_proj.weight", 0, 0, 0, 128 * 64)
    assert np.array_equal(g, (bits >> 16).astype(np.uint16).reshape(-1))
    n = m.get_tensor("model.norm.weight", (64,))
    assert n.min() >= 0.875 and n.max() <= 1.125 and (n.view(np.uint32) & 0xFFFF == 0).all()
