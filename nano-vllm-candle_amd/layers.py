"""Fine-seam mirror of the reference's layer surface (src/layers/*.rs) over the HIP ops.

Same names, argument meaning and error behaviour as the reference types; tensors are numpy arrays on
the way in and out (the reference's are candle Tensors), every forward runs a gfx950 kernel through
the C ABI -- there is no numpy arithmetic in this file.
"""
import ctypes as C

import numpy as np

from . import _lib
from .tp import get_tp

_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        from .context import Context

        _default_ctx = Context(0)
    return _default_ctx


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class _LinearBase:
    """LinearBase {tp_dim, tp_rank, tp_size, weight} (src/layers/linear.rs:4-20)"""

    def __init__(self, tp_dim, tp_rank, tp_size, input_size, output_size, bias=None, ctx=None):
        self.ctx = ctx or default_context()
        self.tp_dim, self.tp_rank, self.tp_size = tp_dim, tp_rank, tp_size
        self.input_size, self.output_size = input_size, output_size
        self.bias = None if bias is None else _f32(bias)
        self._w = None
        self.weight_shape = (output_size, input_size)
        self.load_weights(np.zeros((output_size, input_size), np.float32), bias)

    def _set_weight(self, w):
        w = _f32(w)
        if w.ndim != 2:
            raise ValueError("weight must be [out, in]")
        L = _lib.lib()
        if self._w is not None:
            L.nvllm_op_free_weight(self.ctx.h, self._w)
        h = C.c_void_p()
        _lib.check(L.nvllm_op_pack_weight(self.ctx.h, w.ctypes.data_as(C.c_void_p), _lib.DTYPE_F32, w.shape[0],
                                          w.shape[1], C.byref(h)), self.ctx.h)
        self._w = h
        self.weight_shape = w.shape

    def load_weights(self, weight, bias=None):
        """ParallelLinear::load_weights (linear.rs:22-24)"""
        self._set_weight(weight)
        self.bias = None if bias is None else _f32(bias)

    def _bias_for_forward(self):
        return self.bias

    def forward(self, x):
        x = _f32(x)
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        if x2.shape[1] != self.weight_shape[1]:
            raise ValueError(f"shape mismatch in linear: x {x.shape} vs weight {self.weight_shape}")
        ctx = self.ctx
        dx = ctx.to_device(x2)
        dy = ctx.empty((x2.shape[0], self.weight_shape[0]))
        b = self._bias_for_forward()
        db = ctx.to_device(b) if b is not None else None
        _lib.check(_lib.lib().nvllm_op_linear(ctx.h, dx.ptr, self._w, db.ptr if db else None, x2.shape[0], dy.ptr), ctx.h)
        return dy.numpy().reshape(*lead, self.weight_shape[0])

    __call__ = forward

    def __del__(self):
        try:
            if self._w is not None and self.ctx.h:
                _lib.lib().nvllm_op_free_weight(self.ctx.h, self._w)
        except Exception:
            pass


class ReplicatedLinear(_LinearBase):
    """src/layers/linear.rs:26-46"""

    def __init__(self, input_size, output_size, bias=None, ctx=None):
        super().__init__(0, 0, 1, input_size, output_size, bias, ctx)


class ColumnParallelLinear(_LinearBase):
    """src/layers/linear.rs:48-90: output features are sharded; rank r holds rows
    [r*out/size, (r+1)*out/size).  (The reference narrows by the FULL dim and only works for rank 0,
    SURVEY F7; this takes the shard the reference's own 2-rank test expects, linear.rs:273-322.)"""

    def __init__(self, input_size, output_size, bias=None, tp=None, ctx=None):
        tp = tp or get_tp()
        self._full_out = output_size
        super().__init__(0, tp.rank, tp.size, input_size, output_size // tp.size, None, ctx)
        if bias is not None:
            self.bias = self._shard_rows(_f32(bias))

    def _shard_rows(self, a):
        n = a.shape[0] // self.tp_size
        return np.ascontiguousarray(a[self.tp_rank * n:(self.tp_rank + 1) * n])

    def load_weights(self, weight, bias=None):
        weight = _f32(weight)
        if getattr(self, "_full_out", None) is not None and weight.shape[0] == self._full_out and self.tp_size > 1:
            weight = self._shard_rows(weight)
            bias = self._shard_rows(_f32(bias)) if bias is not None else None
        self._set_weight(weight)
        self.bias = None if bias is None else _f32(bias)


class QKVParallelLinear:
    """src/layers/linear.rs:121-175: fused [q;k;v] column-parallel projection"""

    def __init__(self, hidden_size, head_size, num_heads, num_kv_heads, bias=None, tp=None, ctx=None):
        self.head_size, self.num_heads, self.num_kv_heads = head_size, num_heads, num_kv_heads
        self.tp = tp or get_tp()
        out = (num_heads + 2 * num_kv_heads) * head_size
        self.linear = ColumnParallelLinear(hidden_size, out, bias, tp=TPSingle, ctx=ctx)

    def load_qkv(self, wq, wk, wv):
        """qwen3.rs:171: cat(q,k,v) along dim 0, each sharded by heads for this rank"""
        t = self.tp

        def sh(w, heads):
            w = _f32(w)
            n = heads // t.size * self.head_size
            return w[t.rank * n:(t.rank + 1) * n]

        self.linear.load_weights(np.concatenate([sh(wq, self.num_heads), sh(wk, self.num_kv_heads), sh(wv, self.num_kv_heads)], 0))

    def forward(self, x):
        return self.linear.forward(x)

    __call__ = forward


class RowParallelLinear(_LinearBase):
    """src/layers/linear.rs:177-223: input features are sharded (tp_dim = 1; the reference says 0,
    SURVEY F7); bias only on rank 0 (:188-192); forward all-reduces the partial sums across the TP group --
    the step the reference leaves out."""

    def __init__(self, input_size, output_size, bias=None, tp=None, ctx=None):
        tp = tp or get_tp()
        self._full_in = input_size
        super().__init__(1, tp.rank, tp.size, input_size // tp.size, output_size, bias, ctx)

    def load_weights(self, weight, bias=None):
        weight = _f32(weight)
        if getattr(self, "_full_in", None) is not None and weight.shape[1] == self._full_in and self.tp_size > 1:
            n = self._full_in // self.tp_size
            weight = np.ascontiguousarray(weight[:, self.tp_rank * n:(self.tp_rank + 1) * n])
        self._set_weight(weight)
        self.bias = None if bias is None else _f32(bias)

    def _bias_for_forward(self):
        return self.bias if self.tp_rank == 0 else None

    def forward(self, x):
        y = super().forward(x)
        if self.ctx.tp_size > 1:
            d = self.ctx.to_device(y)
            _lib.check(_lib.lib().nvllm_op_allreduce(self.ctx.h, d.ptr, y.size), self.ctx.h)
            y = d.numpy()
        return y

    __call__ = forward


class _TPSingle:
    size, rank, dim = 1, 0, 0


TPSingle = _TPSingle()


class RMSNorm:
    """src/layers/layernorm.rs:22-61.  forward(x, residual=None) -> (y, new_residual or None)"""

    def __init__(self, hidden_size, eps=None, ctx=None):
        self.weight = np.ones((1, hidden_size), np.float32)  # [1,H], layernorm.rs:29
        self.eps = 1e-6 if eps is None else eps
        self.ctx = ctx or default_context()

    @classmethod
    def from_weight(cls, weight, eps, ctx=None):
        w = _f32(weight).reshape(1, -1)
        o = cls(w.shape[1], eps, ctx)
        o.weight = w
        return o

    def forward(self, x, residual=None):
        x = _f32(x)
        if x.shape[-1] != self.weight.shape[1]:
            raise ValueError(f"shape mismatch in rmsnorm: x {x.shape} vs weight {self.weight.shape}")
        x2 = x.reshape(-1, x.shape[-1])
        ctx = self.ctx
        dx, dw, dy = ctx.to_device(x2), ctx.to_device(self.weight.reshape(-1)), ctx.empty(x2.shape)
        dr = dro = None
        if residual is not None:
            r2 = _f32(residual).reshape(x2.shape)
            dr, dro = ctx.to_device(r2), ctx.empty(x2.shape)
        _lib.check(_lib.lib().nvllm_op_rmsnorm(ctx.h, dx.ptr, dr.ptr if dr else None, dw.ptr, float(self.eps), x2.shape[0],
                                               x2.shape[1], dy.ptr, dro.ptr if dro else None), ctx.h)
        return dy.numpy().reshape(x.shape), (dro.numpy().reshape(x.shape) if dro else None)

    __call__ = forward


class RotaryEmbedding:
    """src/layers/rotary_embedding.rs:32-108.  apply(q [B,nh,T,hd], k [B,kv,T,hd]) -> (q_rot, k_rot)"""

    def __init__(self, head_size, max_position, base, ctx=None):
        self.head_size, self.max_position, self.base = head_size, max_position, float(base)
        self.ctx = ctx or default_context()

    def apply(self, q, k):
        q, k = _f32(q), _f32(k)
        B, nh, T, hd = q.shape
        if k.shape[0] != B or k.shape[2] != T or k.shape[3] != hd or hd != self.head_size:
            raise ValueError("shape mismatch in rope")
        ctx = self.ctx
        dq, dk = ctx.to_device(q), ctx.to_device(k)
        _lib.check(_lib.lib().nvllm_op_rope(ctx.h, dq.ptr, dk.ptr, B, nh, k.shape[1], T, hd, self.base), ctx.h)
        return dq.numpy(), dk.numpy()


class SiluAndMul:
    """src/layers/activation.rs:4-19"""

    def __init__(self, ctx=None):
        self.ctx = ctx or default_context()

    def forward(self, x):
        x = _f32(x)
        if x.shape[-1] % 2:
            raise ValueError("SiluAndMul needs an even last dim")
        x2 = x.reshape(-1, x.shape[-1])
        n = x2.shape[1] // 2
        ctx = self.ctx
        dx, dy = ctx.to_device(x2), ctx.empty((x2.shape[0], n))
        _lib.check(_lib.lib().nvllm_op_silu_mul(ctx.h, dx.ptr, x2.shape[0], n, dy.ptr), ctx.h)
        return dy.numpy().reshape(*x.shape[:-1], n)

    __call__ = forward


class Attention:
    """The attention `layers::Attention` names (src/layers/attention.rs:4-16 is a dead sdpa wrapper): causal GQA
    attention with the live path's math (src/models/qwen3.rs:236-277).  forward(q [B,nh,T,hd], k, v [B,kv,T,hd])
    -> ctx [B*T, nh*hd].  softcapping must be 0 (the live path has none)."""

    def __init__(self, num_heads, head_dim, scale, softcapping=0.0, ctx=None):
        if softcapping not in (0, 0.0, 1.0):
            raise ValueError("softcapping is not part of the reference's live attention path")
        self.num_heads, self.head_dim, self.scale, self.softcapping = num_heads, head_dim, float(scale), softcapping
        self.ctx = ctx or default_context()

    def forward(self, q, k, v):
        q, k, v = _f32(q), _f32(k), _f32(v)
        B, nh, T, hd = q.shape
        kv = k.shape[1]
        if nh != self.num_heads or hd != self.head_dim or k.shape != v.shape or k.shape[0] != B or k.shape[2] != T:
            raise ValueError("shape mismatch in attention")
        ctx = self.ctx
        dq, dk, dv, do = ctx.to_device(q), ctx.to_device(k), ctx.to_device(v), ctx.empty((B * T, nh * hd))
        _lib.check(_lib.lib().nvllm_op_attention(ctx.h, dq.ptr, dk.ptr, dv.ptr, B, nh, kv, T, hd, self.scale, do.ptr), ctx.h)
        return do.numpy()

    __call__ = forward


def embedding(table, ids, ctx=None):
    """Tensor::embedding (src/models/qwen3.rs:465-468)"""
    ctx = ctx or default_context()
    table = _f32(table)
    ids = np.ascontiguousarray(ids, dtype=np.uint32).reshape(-1)
    dt, di, dy = ctx.to_device(table), ctx.to_device(ids), ctx.empty((ids.size, table.shape[1]))
    _lib.check(_lib.lib().nvllm_op_embedding(ctx.h, dt.ptr, di.ptr, ids.size, table.shape[0], table.shape[1], dy.ptr), ctx.h)
    return dy.numpy()


def argmax_last(logits, ctx=None):
    """Qwen3ModelRunner::argmax (llm_engine.rs:135-142): LAST maximal element"""
    ctx = ctx or default_context()
    lg = _f32(logits)
    lg2 = lg.reshape(-1, lg.shape[-1])
    dl, di = ctx.to_device(lg2), ctx.empty((lg2.shape[0],), np.uint32)
    _lib.check(_lib.lib().nvllm_op_argmax(ctx.h, dl.ptr, lg2.shape[0], lg2.shape[1], di.ptr), ctx.h)
    return di.numpy()
