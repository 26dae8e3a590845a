"""Qwen3Config / Qwen3ForCausalLM mirror (src/models/qwen3.rs) over the C ABI.

The model object owns device weights (MFMA tile layout), the native KV block pool and the step
buffers; `step()` is the ModelRunner::run contract (src/engine/llm_engine.rs:145-189).
"""
import ctypes as C
import json
import os
import struct
from dataclasses import dataclass

import numpy as np

from . import _lib
from .context import Context


@dataclass
class Qwen3Config:
    """src/models/qwen3.rs:20-34"""
    vocab_size: int
    hidden_size: int
    head_dim: int
    num_hidden_layers: int
    num_attention_heads: int
    num_key_value_heads: int
    intermediate_size: int
    max_position_embeddings: int
    rms_norm_eps: float = 1e-6
    hidden_act: str = "silu"
    rope_theta: float = 1e6
    bos_token_id: int = 151643
    eos_token_id: int = 151645

    @classmethod
    def from_hf_dir(cls, model_dir):
        """qwen3.rs:77-101: head_dim optional -> hidden/heads; rope_theta default 1e6"""
        path = os.path.join(model_dir, "config.json")
        try:
            with open(path) as f:
                d = json.load(f)
        except OSError as e:
            raise RuntimeError(f"Failed to read HF config {path}: {e}") from e
        except json.JSONDecodeError as e:
            raise RuntimeError(f"Failed to parse HF config {path}: {e}") from e
        try:
            return cls(
                vocab_size=d["vocab_size"], hidden_size=d["hidden_size"],
                head_dim=d.get("head_dim") or d["hidden_size"] // d["num_attention_heads"],
                num_hidden_layers=d["num_hidden_layers"], num_attention_heads=d["num_attention_heads"],
                num_key_value_heads=d["num_key_value_heads"], intermediate_size=d["intermediate_size"],
                max_position_embeddings=d["max_position_embeddings"], rms_norm_eps=d["rms_norm_eps"],
                hidden_act=d["hidden_act"], rope_theta=d.get("rope_theta") or 1e6,
                bos_token_id=d["bos_token_id"], eos_token_id=d["eos_token_id"])
        except KeyError as e:
            raise RuntimeError(f"Failed to parse HF config {path}: missing field {e}") from e

    # public HF shapes, for synthetic-weight benchmarks (no checkpoint files exist offline)
    @classmethod
    def qwen3_0_6b(cls):
        return cls(151936, 1024, 128, 28, 16, 8, 3072, 40960)

    @classmethod
    def qwen3_8b(cls):
        return cls(151936, 4096, 128, 36, 32, 8, 12288, 40960)

    @classmethod
    def qwen3_32b(cls):
        return cls(151936, 5120, 128, 64, 64, 8, 25600, 40960)

    @classmethod
    def tiny(cls, **kw):
        d = dict(vocab_size=512, hidden_size=128, head_dim=64, num_hidden_layers=2, num_attention_heads=4,
                 num_key_value_heads=2, intermediate_size=256, max_position_embeddings=4096, bos_token_id=1, eos_token_id=2)
        d.update(kw)
        return cls(**d)

    def to_c(self):
        if self.hidden_act != "silu":
            raise ValueError(f"hidden_act {self.hidden_act!r} unsupported (silu only)")
        return _lib.Qwen3ConfigC(self.vocab_size, self.hidden_size, self.head_dim, self.num_hidden_layers,
                                 self.num_attention_heads, self.num_key_value_heads, self.intermediate_size,
                                 self.max_position_embeddings, self.rms_norm_eps, self.rope_theta, self.bos_token_id,
                                 self.eos_token_id)

    def hf_tensor_shapes(self):
        """HF tensor names the reference loads (qwen3.rs:150...526) -> full shapes"""
        c = self
        out = {"model.embed_tokens.weight": (c.vocab_size, c.hidden_size), "lm_head.weight": (c.vocab_size, c.hidden_size),
               "model.norm.weight": (c.hidden_size,)}
        for l in range(c.num_hidden_layers):
            p = f"model.layers.{l}."
            out[p + "self_attn.q_proj.weight"] = (c.num_attention_heads * c.head_dim, c.hidden_size)
            out[p + "self_attn.k_proj.weight"] = (c.num_key_value_heads * c.head_dim, c.hidden_size)
            out[p + "self_attn.v_proj.weight"] = (c.num_key_value_heads * c.head_dim, c.hidden_size)
            out[p + "self_attn.o_proj.weight"] = (c.hidden_size, c.num_attention_heads * c.head_dim)
            out[p + "mlp.gate_proj.weight"] = (c.intermediate_size, c.hidden_size)
            out[p + "mlp.up_proj.weight"] = (c.intermediate_size, c.hidden_size)
            out[p + "mlp.down_proj.weight"] = (c.hidden_size, c.intermediate_size)
            out[p + "input_layernorm.weight"] = (c.hidden_size,)
            out[p + "post_attention_layernorm.weight"] = (c.hidden_size,)
            out[p + "self_attn.q_norm.weight"] = (c.head_dim,)
            out[p + "self_attn.k_norm.weight"] = (c.head_dim,)
        return out


def read_safetensors(path):
    """Minimal safetensors reader (header JSON + raw little-endian data, memory-mapped): yields
    (name, np.ndarray view, dtype code).  Executes nothing from the file.  BF16 tensors come back as uint16."""
    with open(path, "rb") as f:
        n = struct.unpack("<Q", f.read(8))[0]
        header = json.loads(f.read(n))
    data = np.memmap(path, dtype=np.uint8, mode="r", offset=8 + n)
    for name, meta in header.items():
        if name == "__metadata__":
            continue
        b, e = meta["data_offsets"]
        dt = meta["dtype"]
        if dt == "BF16":
            arr, code = data[b:e].view(np.uint16), _lib.DTYPE_BF16
        elif dt == "F32":
            arr, code = data[b:e].view(np.float32), _lib.DTYPE_F32
        elif dt == "F16":
            arr, code = data[b:e].view(np.float16).astype(np.float32), _lib.DTYPE_F32
        else:
            raise ValueError(f"{path}: tensor {name} has unsupported dtype {dt}")
        yield name, arr.reshape(meta["shape"]), code


def checkpoint_files(model_dir):
    """safetensors files of an HF model directory: the shards named by model.safetensors.index.json (each once, in
    name order) or the single model.safetensors the reference expects (qwen3.rs:517-521; same error text)"""
    idx = os.path.join(model_dir, "model.safetensors.index.json")
    single = os.path.join(model_dir, "model.safetensors")
    if os.path.exists(idx):
        with open(idx) as f:
            names = sorted(set(json.load(f)["weight_map"].values()))
        return [os.path.join(model_dir, x) for x in names]
    if os.path.exists(single):
        return [single]
    raise RuntimeError(f"mmap {single}: no such file")


class Qwen3ForCausalLM:
    """src/models/qwen3.rs:503-551"""

    def __init__(self, cfg, ctx=None):
        self.cfg = cfg
        self.ctx = ctx or Context(0)
        h = C.c_void_p()
        cc = cfg.to_c()
        _lib.check(_lib.lib().nvllm_model_create(self.ctx.h, C.byref(cc), C.byref(h)), self.ctx.h)
        self.h = h
        self._kv = False

    # ---- construction ----
    @classmethod
    def from_hf_dir(cls, model_dir, ctx=None):
        """qwen3.rs:515-536 (+ sharded checkpoints and tied lm_head, which the reference cannot load: SURVEY F9)"""
        cfg = Qwen3Config.from_hf_dir(model_dir)
        m = cls(cfg, ctx)
        files = checkpoint_files(model_dir)
        for path in files:
            for name, arr, code in read_safetensors(path):
                m.load_tensor(name, arr, code)
        m.finalize()
        return m

    @classmethod
    def from_synthetic(cls, cfg, seed=0, ctx=None, profile=0):
        """weights from the deterministic generator, straight into HBM; profile 1 = heavy-tailed Qwen3-like statistics
        (include/nvllm_amd.h nvllm_model_fill_synthetic_profile)"""
        m = cls(cfg, ctx)
        _lib.check(_lib.lib().nvllm_model_fill_synthetic_profile(m.h, seed, profile), m.ctx.h)
        m.finalize()
        return m

    @classmethod
    def from_state_dict(cls, cfg, tensors, ctx=None):
        """tensors: name -> float32 ndarray (full HF shapes)"""
        m = cls(cfg, ctx)
        for name, arr in tensors.items():
            m.load_tensor(name, np.ascontiguousarray(arr, np.float32), _lib.DTYPE_F32)
        m.finalize()
        return m

    def load_tensor(self, name, arr, code=None):
        if code is None:
            code = _lib.DTYPE_BF16 if arr.dtype == np.uint16 else _lib.DTYPE_F32
            if code == _lib.DTYPE_F32:
                arr = np.ascontiguousarray(arr, np.float32)
        arr = np.ascontiguousarray(arr)
        shape = (C.c_int64 * arr.ndim)(*arr.shape)
        _lib.check(_lib.lib().nvllm_model_load_tensor(self.h, name.encode(), arr.ctypes.data_as(C.c_void_p), code, shape,
                                                      arr.ndim), self.ctx.h)

    def finalize(self):
        _lib.check(_lib.lib().nvllm_model_finalize(self.h), self.ctx.h)

    # ---- KV pool ----
    def kv_alloc(self, num_blocks, max_seqs, max_batched_tokens=4096, block_size=256):
        _lib.check(_lib.lib().nvllm_kv_alloc(self.h, num_blocks, block_size, max_seqs, max_batched_tokens), self.ctx.h)
        self._kv = True
        self.max_seqs = max_seqs

    def free_blocks(self):
        return _lib.lib().nvllm_kv_num_free_blocks(self.h)

    def seq_free(self, seq_id):
        _lib.check(_lib.lib().nvllm_seq_free(self.h, int(seq_id)), self.ctx.h)

    @property
    def weight_bytes(self):
        return _lib.lib().nvllm_model_weight_bytes(self.h)

    @property
    def kv_bytes_per_token(self):
        return _lib.lib().nvllm_kv_bytes_per_token(self.h)

    @property
    def last_step_bytes(self):
        return _lib.lib().nvllm_last_step_bytes(self.h)

    # ---- the hot path ----
    def step(self, seq_ids, token_lists, is_prefill, want_logits=False):
        """ModelRunner::run: -> (next_ids uint32[n], last_logits float32[n,V] or None)"""
        n = len(seq_ids)
        if n == 0:
            return np.empty(0, np.uint32), None
        arrs = [np.ascontiguousarray(t, dtype=np.uint32) for t in token_lists]
        u32p = C.POINTER(C.c_uint32)
        ptrs = (u32p * n)(*[a.ctypes.data_as(u32p) for a in arrs])
        lens = (C.c_int32 * n)(*[len(a) for a in arrs])
        ids = (C.c_int64 * n)(*[int(s) for s in seq_ids])
        nxt = np.empty(n, np.uint32)
        lg = np.empty((n, self.cfg.vocab_size), np.float32) if want_logits else None
        _lib.check(_lib.lib().nvllm_step(self.h, n, ids, ptrs, lens, int(bool(is_prefill)), nxt.ctypes.data_as(u32p),
                                         lg.ctypes.data_as(C.POINTER(C.c_float)) if lg is not None else None), self.ctx.h)
        return nxt, lg

    def step_sample(self, seq_ids, token_lists, is_prefill, temperatures, seed, want_logits=False):
        """ModelRunner::run with sample_token on the device (llm_engine.rs:97-133): -> (sampled ids, last_logits or None)"""
        n = len(seq_ids)
        if n == 0:
            return np.empty(0, np.uint32), None
        arrs = [np.ascontiguousarray(t, dtype=np.uint32) for t in token_lists]
        u32p = C.POINTER(C.c_uint32)
        ptrs = (u32p * n)(*[a.ctypes.data_as(u32p) for a in arrs])
        lens = (C.c_int32 * n)(*[len(a) for a in arrs])
        ids = (C.c_int64 * n)(*[int(s) for s in seq_ids])
        temps = np.ascontiguousarray(temperatures, np.float32)
        if temps.shape != (n,):
            raise ValueError("one temperature per sequence")
        nxt = np.empty(n, np.uint32)
        lg = np.empty((n, self.cfg.vocab_size), np.float32) if want_logits else None
        _lib.check(_lib.lib().nvllm_step_sample(self.h, n, ids, ptrs, lens, int(bool(is_prefill)),
                                                temps.ctypes.data_as(C.POINTER(C.c_float)), int(seed) & (2**64 - 1), nxt.ctypes.data_as(u32p),
                                                lg.ctypes.data_as(C.POINTER(C.c_float)) if lg is not None else None), self.ctx.h)
        return nxt, lg

    def decode_next(self, want_ids=True):
        n = getattr(self, "max_seqs", 0)
        buf = np.empty(max(n, 1), np.uint32)
        _lib.check(_lib.lib().nvllm_decode_next(self.h, buf.ctypes.data_as(C.POINTER(C.c_uint32)) if want_ids else None),
                   self.ctx.h)
        return buf

    def decode_enqueue(self):
        """put one more decode step on the stream without waiting (pipelined decode)"""
        _lib.check(_lib.lib().nvllm_decode_enqueue(self.h), self.ctx.h)

    def decode_collect(self):
        """ids of the oldest enqueued decode step"""
        buf = np.empty(max(getattr(self, "max_seqs", 1), 1), np.uint32)
        _lib.check(_lib.lib().nvllm_decode_collect(self.h, buf.ctypes.data_as(C.POINTER(C.c_uint32))), self.ctx.h)
        return buf

    PROF_KINDS = {"attn": 1, "gemm": 2, "norm": 3, "qk": 4, "silu": 5, "lm_head": 6, "empty": 7}

    def profile_kernel(self, kind):
        """HIP-event bracketing of one kernel class on the library stream (None/0 = off)"""
        k = self.PROF_KINDS.get(kind, kind) if kind else 0
        _lib.check(_lib.lib().nvllm_profile_kernel(self.h, int(k)), self.ctx.h)

    def profile_read(self):
        ms, n = C.c_double(), C.c_int64()
        _lib.check(_lib.lib().nvllm_profile_read(self.h, C.byref(ms), C.byref(n)), self.ctx.h)
        return ms.value, n.value

    def set_option(self, name, value):
        """tuning switch of the model (include/nvllm_amd_debug.h nvllm_debug_set_option)"""
        _lib.check(_lib.lib().nvllm_debug_set_option(self.h, name.encode(), int(value)), self.ctx.h)

    def counter(self, name):
        """debug counter of the model (include/nvllm_amd_debug.h nvllm_debug_get_counter)"""
        v = C.c_int64()
        _lib.check(_lib.lib().nvllm_debug_get_counter(self.h, name.encode(), C.byref(v)), self.ctx.h)
        return int(v.value)

    def enable_taps(self, on=True):
        _lib.check(_lib.lib().nvllm_debug_enable_taps(self.h, int(on)), self.ctx.h)

    def layer_tap(self, layer, what, rows):
        out = np.empty((rows, self.cfg.hidden_size), np.float32)
        _lib.check(_lib.lib().nvllm_debug_layer_tap(self.h, layer, what, out.ctypes.data_as(C.POINTER(C.c_float)), out.size),
                   self.ctx.h)
        return out

    def close(self):
        if getattr(self, "h", None):
            _lib.lib().nvllm_model_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
