"""First-contact GPU script: runs each fine-seam op against the oracle and prints errors (not a pytest)."""
import os, sys, traceback
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nano_vllm_candle_amd as pkg
from nano_vllm_candle_amd import layers as Ly
from oracle import oracle as O

rng = np.random.default_rng(0)
def bf16_round(a):
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
    return u.view(np.float32)
def rel(a, b): return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
def run(name, fn):
    try:
        print(f"[{name}]", fn(), flush=True)
    except Exception:
        print(f"[{name}] EXC"); traceback.print_exc(); sys.stdout.flush()

def t_synth():
    ctx = Ly.default_context()
    import ctypes as C
    out = np.empty(4096, np.uint16)
    pkg._lib.check(pkg._lib.lib().nvllm_op_synth_bf16(ctx.h, b"model.layers.1.mlp.up_proj.weight", 7, 0, 123, 4096, out.ctypes.data_as(C.POINTER(C.c_uint16))), ctx.h)
    return bool(np.array_equal(out, O.synth_bf16("model.layers.1.mlp.up_proj.weight", 7, 0, 123, 4096)))
def t_silu():
    x = rng.standard_normal((7, 64)).astype(np.float32)
    return rel(Ly.SiluAndMul()(x), O.silu_mul(x)), Ly.SiluAndMul()(np.array([[0, 1, -1, 2]], np.float32)).tolist()
def t_rms():
    out = []
    for n in (4, 3, 1024, 5120):
        x = rng.standard_normal((5, n)).astype(np.float32); r = rng.standard_normal((5, n)).astype(np.float32)
        w = (1 + 0.1 * rng.standard_normal(n)).astype(np.float32)
        y, ro = Ly.RMSNorm.from_weight(w, 1e-6)(x, r); ry, rr = O.rmsnorm(x, w, 1e-6, r)
        out.append((n, rel(y, ry), rel(ro, rr)))
    return out
def t_lin():
    out = []
    W6 = np.array([[1, 0, -1, 2], [0, 1, 2, -1], [2, -1, 0, 1], [-2, 1, 1, 0], [1, 1, 1, 1], [3, 0, -2, 1]], np.float32)
    B6 = np.array([1, -2, 0, 3, -1, 2], np.float32)
    X = np.array([[1, 2, 3, 4], [-1, 0, 1, 2]], np.float32)
    l = Ly.ReplicatedLinear(4, 6); l.load_weights(W6, B6)
    out.append(l(X).tolist())
    for (M, K, N) in [(1, 1024, 4096), (64, 1024, 4096), (64, 2048, 1024), (64, 3072, 1024), (100, 1024, 6144), (300, 128, 512), (17, 96, 48)]:
        w = bf16_round(0.02 * rng.standard_normal((N, K)).astype(np.float32)); x = rng.standard_normal((M, K)).astype(np.float32)
        l = Ly.ReplicatedLinear(K, N); l.load_weights(w)
        out.append(((M, K, N), rel(l(x), O.linear(x, w))))
    return out
def t_rope():
    q = rng.standard_normal((2, 4, 9, 128)).astype(np.float32); k = rng.standard_normal((2, 2, 9, 128)).astype(np.float32)
    qr, kr = Ly.RotaryEmbedding(128, 4096, 1e6).apply(q, k)
    return rel(qr, O.rope_apply(q, 1e6)), rel(kr, O.rope_apply(k, 1e6))
def t_attn():
    out = []
    for (B, nh, kv, T, hd) in [(1, 2, 1, 5, 64), (2, 4, 2, 37, 128), (1, 16, 8, 300, 128), (2, 8, 2, 70, 64), (1, 8, 1, 33, 128)]:
        q = rng.standard_normal((B, nh, T, hd)).astype(np.float32); k = rng.standard_normal((B, kv, T, hd)).astype(np.float32)
        v = rng.standard_normal((B, kv, T, hd)).astype(np.float32)
        got = Ly.Attention(nh, hd, hd ** -0.5)(q, k, v); ref = O.attention(q, k, v)
        out.append(((B, nh, kv, T, hd), rel(got, ref)))
    return out
def t_argmax():
    lg = rng.standard_normal((3, 1000)).astype(np.float32); lg[0, 5] = lg[0, 900] = 9.0
    return Ly.argmax_last(lg).tolist(), [O.argmax_last(r) for r in lg]
def t_model():
    import __graft_entry__ as g
    g.smoke(); return "ok"
for n, f in [("synth", t_synth), ("silu", t_silu), ("rms", t_rms), ("linear", t_lin), ("rope", t_rope), ("attn", t_attn), ("argmax", t_argmax), ("model", t_model)]:
    run(n, f)
