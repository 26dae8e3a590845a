"""GPU parity tests of the fine seam (src/layers/ surface) through the C ABI, against the CPU oracle and the
reference's own known answers.  Each test reads like the reference unit test it mirrors."""
import numpy as np
import pytest

from tests.util import bf16_round, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Ly():
    from nano_vllm_candle_amd import layers

    return layers


@pytest.fixture(scope="module")
def rng():
    return np.random.default_rng(0)


# ---- activation.rs:26-36 -----------------------------------------------------------------------------
def test_silu_and_mul(Ly, oracle, rng):
    v = Ly.SiluAndMul()(np.array([[0.0, 1.0, -1.0, 2.0]], np.float32))
    assert abs(v[0][0] - 0.0) < 1e-6 and abs(v[0][1] - 1.4621172) < 1e-5
    x = rng.standard_normal((33, 6144)).astype(np.float32) * 3
    assert rel_err(Ly.SiluAndMul()(x), oracle.silu_mul(x)) < 1e-6
    x = rng.standard_normal((3, 10)).astype(np.float32)  # odd width -> generic kernel
    assert rel_err(Ly.SiluAndMul()(x), oracle.silu_mul(x)) < 1e-6


# ---- layernorm.rs:68-121 -----------------------------------------------------------------------------
def test_rms_norm_simple(Ly):
    y, res = Ly.RMSNorm(4, 1e-5)(np.array([[1, 2, 3, 4], [1, 1, 1, 1]], np.float32))
    assert res is None and y.shape == (2, 4)
    assert abs(y[0][0] - 0.36515) < 1e-4 and abs(y[0][3] - 1.46059) < 1e-4


def test_rms_norm_with_residual(Ly):
    x = np.full((1, 3), 0.5, np.float32)
    y, res = Ly.RMSNorm(3)(x, x.copy())
    assert res[0].tolist() == [1.0, 1.0, 1.0] and abs(y[0][0] - 1.0) < 1e-5


def test_rms_norm_dtype_consistency(Ly, rng):
    y, _ = Ly.RMSNorm(2)(rng.standard_normal((1, 2)).astype(np.float32))
    assert y.dtype == np.float32


@pytest.mark.parametrize("n", [64, 128, 1024, 4096, 5120, 8192, 100])
def test_rms_norm_vs_oracle(Ly, oracle, rng, n):
    x = rng.standard_normal((7, n)).astype(np.float32) * 2
    r = rng.standard_normal((7, n)).astype(np.float32)
    w = (1 + 0.1 * rng.standard_normal(n)).astype(np.float32)
    y, ro = Ly.RMSNorm.from_weight(w, 1e-6)(x, r)
    ry, rr = oracle.rmsnorm(x, w, 1e-6, r)
    assert np.array_equal(ro, rr)  # the residual sum is a single f32 add: bit-exact
    assert rel_err(y, ry) < 2e-6
    y, ro = Ly.RMSNorm.from_weight(w, 1e-6)(x)
    assert ro is None and rel_err(y, oracle.rmsnorm(x, w, 1e-6)[0]) < 2e-6


# ---- linear.rs:232-354 -------------------------------------------------------------------------------
_W6 = np.array([[1, 0, -1, 2], [0, 1, 2, -1], [2, -1, 0, 1], [-2, 1, 1, 0], [1, 1, 1, 1], [3, 0, -2, 1]], np.float32)
_B6 = np.array([1, -2, 0, 3, -1, 2], np.float32)
_X = np.array([[1, 2, 3, 4], [-1, 0, 1, 2]], np.float32)


def test_replicated_linear(Ly):
    layer = Ly.ReplicatedLinear(4, 6)
    layer.load_weights(_W6, _B6)
    assert layer(_X).tolist() == [[7, 2, 4, 6, 9, 3], [3, -2, 0, 6, 1, -1]]


def test_column_parallel_linear(Ly):
    from nano_vllm_candle_amd.tp import TPConfig

    l0 = Ly.ColumnParallelLinear(4, 6, tp=TPConfig(2, 0, 0))
    l1 = Ly.ColumnParallelLinear(4, 6, tp=TPConfig(2, 1, 0))
    l0.load_weights(_W6, _B6)  # full tensors in: each rank keeps its rows
    l1.load_weights(_W6, _B6)
    assert l0(_X).tolist() == [[7, 2, 4], [3, -2, 0]]
    assert l1(_X).tolist() == [[6, 9, 3], [6, 1, -1]]


def test_qkv_parallel_linear_split(Ly):
    wq = np.array([[1, 0, -1, 2], [0, 1, 2, -1]], np.float32)
    wk = np.array([[2, -1, 0, 1]], np.float32)
    wv = np.array([[-2, 1, 1, 0]], np.float32)
    qkv = Ly.QKVParallelLinear(4, 2, 1, 1)  # hidden 4, head 2, 1 q head... (2 q features), 1 kv head of size... see below
    qkv.linear.load_weights(np.concatenate([wq, wk, wv], 0))
    x = np.array([[1, 2, 3, 4]], np.float32)
    out = qkv(x)
    assert out[:, 0:2].tolist() == (x @ wq.T).tolist()
    assert out[:, 2:3].tolist() == (x @ wk.T).tolist()
    assert out[:, 3:4].tolist() == (x @ wv.T).tolist()


def test_row_parallel_bias_only_on_rank0(Ly):
    from nano_vllm_candle_amd.tp import TPConfig

    w = np.arange(8, dtype=np.float32).reshape(2, 4)
    b = np.array([10, 20], np.float32)
    r0 = Ly.RowParallelLinear(4, 2, tp=TPConfig(2, 0, 0))
    r1 = Ly.RowParallelLinear(4, 2, tp=TPConfig(2, 1, 0))
    r0.load_weights(w, b)
    r1.load_weights(w, b)
    x = np.array([[1, 1, 1, 1]], np.float32)
    # partial sums of the two ranks add up to x.W^T + b (the all-reduce is a no-op in this 1-process context)
    total = r0(x[:, :2]) + r1(x[:, 2:])
    assert total.tolist() == (x @ w.T + b).tolist()


@pytest.mark.parametrize("M,K,N", [(1, 1024, 4096), (7, 1024, 4096), (64, 1024, 4096), (64, 2048, 1024), (64, 3072, 1024),
                                   (64, 1024, 6144), (100, 1024, 6144), (256, 1024, 1024), (300, 128, 512), (17, 96, 48),
                                   (33, 5120, 256), (2, 1024, 151936)])
def test_linear_vs_oracle(Ly, oracle, rng, M, K, N):
    w = bf16_round(0.02 * rng.standard_normal((N, K)).astype(np.float32))  # checkpoints are bf16
    x = rng.standard_normal((M, K)).astype(np.float32)
    layer = Ly.ReplicatedLinear(K, N)
    layer.load_weights(w)
    assert rel_err(layer(x), oracle.linear(x, w)) < 1e-5


def test_linear_shape_mismatch_raises(Ly):
    with pytest.raises(ValueError):
        Ly.ReplicatedLinear(4, 6)(np.zeros((2, 5), np.float32))


# ---- rotary_embedding.rs:115-137, tests/layer_test.rs:440-503 ----------------------------------------
def test_rope_norm_preserved(Ly, rng):
    rope = Ly.RotaryEmbedding(8, 128, 10000.0)
    q = (0.01 * rng.standard_normal((1, 2, 4, 8))).astype(np.float32)
    k = (0.01 * rng.standard_normal((1, 2, 4, 8))).astype(np.float32)
    qr, kr = rope.apply(q, k)
    assert np.abs((q ** 2).sum(-1) - (qr ** 2).sum(-1)).sum() < 1e-5
    assert np.abs((k ** 2).sum(-1) - (kr ** 2).sum(-1)).sum() < 1e-5


def test_rope_values(Ly, oracle, rng):
    rope = Ly.RotaryEmbedding(128, 4096, 1000000.0)
    q = rng.standard_normal((1, 2, 4, 128)).astype(np.float32)
    k = rng.standard_normal((1, 2, 4, 128)).astype(np.float32)
    qr, kr = rope.apply(q, k)
    assert np.abs((q ** 2).sum(-1) - (qr ** 2).sum(-1)).sum() < 1e-3
    ones = np.ones((1, 2, 4, 128), np.float32)
    r, _ = rope.apply(ones, ones)
    assert np.abs(r[0, :, 0] - r[0, :, 1]).sum() > 1.0
    assert rel_err(qr, oracle.rope_apply(q, 1e6)) < 1e-6 and rel_err(kr, oracle.rope_apply(k, 1e6)) < 1e-6
    # long positions: table angles pos*inv_freq in f32 like the reference
    q = rng.standard_normal((1, 1, 700, 128)).astype(np.float32)
    qr, _ = rope.apply(q, q[:, :1])
    assert rel_err(qr, oracle.rope_apply(q, 1e6)) < 1e-6


# ---- attention (qwen3.rs:236-277), GQA interleave (tests/debug_layer_test.rs:38-70) --------------------
def test_gqa_interleaved_expand(Ly):
    hd = 64
    v = np.zeros((1, 2, 1, hd), np.float32)
    v[0, 0, 0, :3] = [1, 2, 3]
    v[0, 1, 0, :3] = [4, 5, 6]
    q = np.ones((1, 4, 1, hd), np.float32)
    k = np.ones((1, 2, 1, hd), np.float32)
    ctx = Ly.Attention(4, hd, hd ** -0.5)(q, k, v).reshape(4, hd)
    assert ctx[:, :3].tolist() == [[1, 2, 3], [1, 2, 3], [4, 5, 6], [4, 5, 6]]  # kv heads [h0,h0,h1,h1]


# Op-level bound, NOT the model tolerance: raw attention outputs of N(0,1) q/k/v (scores of sigma ~ sqrt(hd) before the
# scale, far hotter than a trained model's) with f16 K/V storage and f16 P.  The model-level tests hold
# the north-star 1e-3 on logits; this one pins the kernel's arithmetic, and prints what it measured.
ATTN_OP_TOL = 2e-3


@pytest.mark.parametrize("B,nh,kv,T,hd", [(1, 2, 1, 1, 64), (1, 2, 1, 5, 64), (2, 4, 2, 37, 128), (1, 16, 8, 300, 128),
                                          (2, 8, 2, 70, 64), (1, 8, 1, 33, 128), (1, 5, 1, 40, 128), (1, 16, 16, 31, 128),
                                          (1, 2, 2, 600, 128)])
def test_attention_vs_oracle(Ly, oracle, rng, B, nh, kv, T, hd):
    q = rng.standard_normal((B, nh, T, hd)).astype(np.float32)
    k = rng.standard_normal((B, kv, T, hd)).astype(np.float32)
    v = rng.standard_normal((B, kv, T, hd)).astype(np.float32)
    got = Ly.Attention(nh, hd, hd ** -0.5)(q, k, v)
    err = rel_err(got, oracle.attention(q, k, v))
    print(f"[parity] attention op B={B} nh={nh} kv={kv} T={T} hd={hd}: {err:.3e} (op bound {ATTN_OP_TOL:g})")
    assert err < ATTN_OP_TOL


def test_attention_online_softmax_rescale_branch(Ly, oracle, rng):
    # force the running max to jump at a chosen late tile (cdna guide rule 26): one key spikes against one query
    B, nh, kv, T, hd = 1, 2, 1, 200, 128
    q = rng.standard_normal((B, nh, T, hd)).astype(np.float32)
    k = rng.standard_normal((B, kv, T, hd)).astype(np.float32)
    v = rng.standard_normal((B, kv, T, hd)).astype(np.float32)
    k[0, 0, 150] = 6.0 * q[0, 0, 199] / np.linalg.norm(q[0, 0, 199]) * 3
    got = Ly.Attention(nh, hd, hd ** -0.5)(q, k, v)
    assert rel_err(got, oracle.attention(q, k, v)) < ATTN_OP_TOL


def test_causal_rows_ignore_later_tokens(Ly, rng):
    # row i of the context must not change when tokens after i change (qwen3.rs:260-271 mask)
    q = rng.standard_normal((1, 2, 20, 64)).astype(np.float32)
    k = rng.standard_normal((1, 1, 20, 64)).astype(np.float32)
    v = rng.standard_normal((1, 1, 20, 64)).astype(np.float32)
    a = Ly.Attention(2, 64, 0.125)(q, k, v)
    k2, v2 = k.copy(), v.copy()
    k2[:, :, 10:] += 5
    v2[:, :, 10:] -= 3
    b = Ly.Attention(2, 64, 0.125)(q, k2, v2)
    assert np.array_equal(a[:10], b[:10]) and not np.array_equal(a[10:], b[10:])


# ---- embedding / argmax ----------------------------------------------------------------------------------
def test_embedding_gather(Ly, rng):
    table = rng.standard_normal((50, 32)).astype(np.float32)
    ids = np.array([3, 49, 0, 3], np.uint32)
    assert np.array_equal(Ly.embedding(table, ids), table[ids])


def test_argmax_last_max_wins(Ly, oracle, rng):
    lg = rng.standard_normal((5, 151936)).astype(np.float32)
    lg[0, 5] = lg[0, 90000] = 50.0
    lg[1, :] = 1.0
    got = Ly.argmax_last(lg).tolist()
    assert got == [oracle.argmax_last(r) for r in lg]
    assert got[0] == 90000 and got[1] == 151935


# ---- synthetic generator: device == oracle, bit for bit ----------------------------------------------------
def test_device_generator_matches_oracle_generator(Ly, oracle):
    import ctypes as C

    import nano_vllm_candle_amd as pkg

    ctx = Ly.default_context()
    for name, kind in (("model.layers.3.mlp.up_proj.weight", 0), ("model.norm.weight", 1), ("lm_head.weight", 0)):
        out = np.empty(10000, np.uint16)
        pkg._lib.check(pkg._lib.lib().nvllm_op_synth_bf16(ctx.h, name.encode(), 7, kind, 12345, out.size,
                                                          out.ctypes.data_as(C.POINTER(C.c_uint16))), ctx.h)
        assert np.array_equal(out, oracle.synth_bf16(name, 7, kind, 12345, out.size))


@pytest.mark.parametrize("M,N,K,mode,packed", [
    (4096, 4096, 1024, 0, 0),    # Qwen3-0.6B QKV on a full prompt chunk: one workgroup per CU
    (4090, 6144, 1024, 2, 0),    # gate/up + SiLU*mul, ragged last row tile, 192-wide blocks, row-major planes out
    (1000, 6144, 1024, 2, 1),    # few row blocks (feature-major XCD map), planes out in fragment order
    (300, 1024, 3072, 0, 0),     # down_proj shape: narrow output -> 128-row blocks with K splits (slabs summed here)
    (4096, 1024, 2048, 0, 0),    # o_proj on a full chunk: 128 x 256 blocks, 2 K splits
    (513, 512, 128, 0, 0),       # shortest K: four stages
    (256, 6144, 4096, 0, 0),     # a few hundred decode rows (Qwen3-8B QKV at batch 256): one row block, K splits fill the chip
    (256, 6144, 1024, 2, 1),     # ... and the SwiGLU epilogue on 128-row blocks
])
def test_prefill_tile_gemm_equals_the_chunked_kernel(Ly, M, N, K, mode, packed):
    # tile_gemm.hip (256-row workgroup tiles, both operands in fragment order, three-stage LDS ring) against the chunked
    # kernel (which test_linear_vs_oracle pins to the oracle) on the same synthetic operands.  Both accumulate k-tiles in
    # ascending order, hi then lo, in f32, so unsplit results are equal to the last bit (K splits: to f32 rounding of the
    # slab sum); a published-too-early LDS stage or a wrong fragment shows as a large difference, not as rounding.
    import ctypes as C

    from nano_vllm_candle_amd import _lib
    from nano_vllm_candle_amd import layers

    ctx = layers.default_context()
    d, r, u0, u1 = C.c_float(), C.c_float(), C.c_float(), C.c_float()
    _lib.check(_lib.lib().nvllm_debug_gemm_tile_check(ctx.h, M, N, K, mode, packed, 2, C.byref(d), C.byref(r), C.byref(u0), C.byref(u1)), ctx.h)
    assert r.value > 0
    assert d.value <= 1e-5 * r.value, (d.value, r.value)
