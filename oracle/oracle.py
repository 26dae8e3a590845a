"""ctypes binding of the CPU oracle (oracle/qwen3_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libqwen3_oracle.so")


class Config(C.Structure):
    """mirrors oq3_config / Qwen3Config (src/models/qwen3.rs:20-34)"""

    _fields_ = [
        ("vocab_size", C.c_int32), ("hidden_size", C.c_int32), ("head_dim", C.c_int32),
        ("num_hidden_layers", C.c_int32), ("num_attention_heads", C.c_int32),
        ("num_key_value_heads", C.c_int32), ("intermediate_size", C.c_int32),
        ("max_position_embeddings", C.c_int32), ("rms_norm_eps", C.c_double), ("rope_theta", C.c_double),
        ("bos_token_id", C.c_int32), ("eos_token_id", C.c_int32),
    ]


def build(force=False):
    """compile the oracle if the .so is missing or older than its sources"""
    srcs = [os.path.join(_HERE, f) for f in ("qwen3_oracle.c", "synth.h", "Makefile")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None
threads = 0  # worker threads the oracle's loops use (set when the library is loaded)


def usable_cpus():
    """CPUs this process may really use: the scheduler affinity capped by the cgroup CPU quota.  A GPU box shows every
    host core (256) to a container that is allowed 16 CPUs' worth of time; one OpenMP thread per visible core then
    spends its time being throttled."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                quota, period = parts[0], int(parts[1])
            else:
                quota = parts[0]
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = int(f.read())
            if quota not in ("max", "-1") and period > 0:
                n = min(n, max(1, -(-int(quota) // period)))
            break
        except (OSError, ValueError, IndexError):
            continue
    env = os.environ.get("OMP_NUM_THREADS")
    if env and env.isdigit() and int(env) > 0:
        n = int(env)
    return n


def lib():
    global _lib, threads
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.oq3_set_threads.restype = C.c_int
        L.oq3_set_threads.argtypes = [C.c_int]
        threads = L.oq3_set_threads(usable_cpus())
        fp, u32p, i32p = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_int32)
        L.oq3_create.restype = C.c_void_p
        L.oq3_create.argtypes = [C.POINTER(Config)]
        L.oq3_destroy.argtypes = [C.c_void_p]
        L.oq3_set_tensor.argtypes = [C.c_void_p, C.c_char_p, fp, C.c_int64]
        L.oq3_get_tensor.argtypes = [C.c_void_p, C.c_char_p, fp, C.c_int64]
        L.oq3_fill_synthetic.argtypes = [C.c_void_p, C.c_uint64]
        L.oq3_fill_synthetic_profile.argtypes = [C.c_void_p, C.c_uint64, C.c_int]
        L.oq3_synth_bf16_spec.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int64,
                                          C.c_int64, C.POINTER(C.c_uint16)]
        L.oq3_synth_bf16.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_int64, C.c_int64, C.POINTER(C.c_uint16)]
        L.oq3_set_trace.argtypes = [C.c_void_p, fp, fp]
        L.oq3_forward.argtypes = [C.c_void_p, u32p, C.c_int, C.c_int, fp]
        L.oq3_compute_logits.argtypes = [C.c_void_p, fp, C.c_int, fp]
        L.oq3_run_greedy.argtypes = [C.c_void_p, C.c_int, C.POINTER(u32p), i32p, C.c_int, u32p, fp]
        L.oq3_linear.argtypes = [fp, fp, fp, C.c_int, C.c_int, C.c_int, fp]
        L.oq3_rmsnorm.argtypes = [fp, fp, fp, C.c_double, C.c_int, C.c_int, fp, fp]
        L.oq3_silu_mul.argtypes = [fp, C.c_int, C.c_int, fp]
        L.oq3_rope_table.argtypes = [C.c_int, C.c_float, C.c_int, fp, fp]
        L.oq3_rope_apply.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]
        L.oq3_attention.argtypes = [fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp]
        L.oq3_tp_shard_size.restype = C.c_int64
        L.oq3_tp_shard_size.argtypes = [C.c_int64, C.c_int]
        L.oq3_tp_shard_offset.restype = C.c_int64
        L.oq3_tp_shard_offset.argtypes = [C.c_int64, C.c_int, C.c_int]
        L.oq3_argmax_last.argtypes = [fp, C.c_int]
        L.oq3_set_study.argtypes = [C.c_int]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---- ops ---------------------------------------------------------------------------------------

def linear(x, w, bias=None):
    x, w = _f32(x), _f32(w)
    b = _f32(bias) if bias is not None else None
    M, K = x.shape
    N = w.shape[0]
    y = np.empty((M, N), np.float32)
    lib().oq3_linear(_fp(x), _fp(w), _fp(b), M, K, N, _fp(y))
    return y


def rmsnorm(x, weight, eps, residual=None):
    """RMSNorm::forward -> (y, new_residual or None)"""
    x, weight = _f32(x), _f32(weight).reshape(-1)
    rows, n = x.shape
    res = _f32(residual) if residual is not None else None
    y = np.empty_like(x)
    ro = np.empty_like(x) if res is not None else None
    lib().oq3_rmsnorm(_fp(x), _fp(res), _fp(weight), float(eps), rows, n, _fp(y), _fp(ro))
    return y, ro


def silu_mul(x):
    x = _f32(x)
    rows, two_n = x.shape
    y = np.empty((rows, two_n // 2), np.float32)
    lib().oq3_silu_mul(_fp(x), rows, two_n // 2, _fp(y))
    return y


def rope_table(hd, base, t):
    cos = np.empty((t, hd // 2), np.float32)
    sin = np.empty((t, hd // 2), np.float32)
    lib().oq3_rope_table(hd, float(base), t, _fp(cos), _fp(sin))
    return cos, sin


def rope_apply(x, base):
    """x [B,heads,T,hd] -> rotated copy"""
    x = _f32(x).copy()
    B, h, T, hd = x.shape
    lib().oq3_rope_apply(_fp(x), B, h, T, hd, float(base))
    return x


def attention(q, k, v):
    """q [B,nh,T,hd], k/v [B,kv,T,hd] -> ctx [B*T, nh*hd]"""
    q, k, v = _f32(q), _f32(k), _f32(v)
    B, nh, T, hd = q.shape
    kv = k.shape[1]
    ctx = np.empty((B * T, nh * hd), np.float32)
    lib().oq3_attention(_fp(q), _fp(k), _fp(v), B, nh, kv, T, hd, _fp(ctx))
    return ctx


def argmax_last(v):
    v = _f32(v).reshape(-1)
    return int(lib().oq3_argmax_last(_fp(v), v.size))


def set_study(flags):
    """numerics-study knob of the attention core (qwen3_oracle.c oq3_set_study; 0 = reference arithmetic, the default)"""
    lib().oq3_set_study(int(flags))


def synth_bf16(name, seed, kind, first, count, profile=0, axis=0, cols=1, hidden_size=1):
    """bf16 bits of elements [first, first+count) of tensor `name` (oracle/synth.h): kind 0 matrix / 1 norm / 2 q,k-norm;
    profile 1 also needs where the hidden channel sits (axis 1: idx % cols, 2: idx // cols) and hidden_size"""
    out = np.empty(count, np.uint16)
    if profile == 0:
        lib().oq3_synth_bf16(name.encode(), seed, kind, first, count, out.ctypes.data_as(C.POINTER(C.c_uint16)))
    else:
        lib().oq3_synth_bf16_spec(name.encode(), seed, kind, profile, axis, cols, hidden_size, first, count,
                                  out.ctypes.data_as(C.POINTER(C.c_uint16)))
    return out


# ---- model -------------------------------------------------------------------------------------

def make_config(**kw):
    d = dict(vocab_size=512, hidden_size=64, head_dim=32, num_hidden_layers=2, num_attention_heads=4,
             num_key_value_heads=2, intermediate_size=192, max_position_embeddings=4096, rms_norm_eps=1e-6,
             rope_theta=1e6, bos_token_id=1, eos_token_id=2)
    d.update(kw)
    return Config(**d)


class Model:
    def __init__(self, cfg):
        self.cfg = cfg
        self._h = lib().oq3_create(C.byref(cfg))
        self._trace = None

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oq3_destroy(self._h)
            self._h = None

    def fill_synthetic(self, seed=0, profile=0):
        """profile 0: the benign generator; 1: heavy-tailed Qwen3-like statistics (oracle/synth.h)"""
        lib().oq3_fill_synthetic_profile(self._h, seed, profile)
        return self

    def set_tensor(self, name, arr):
        arr = _f32(arr)
        rc = lib().oq3_set_tensor(self._h, name.encode(), _fp(arr), arr.size)
        if rc != 0:
            raise KeyError(f"oracle: unknown tensor or size mismatch: {name} ({arr.size})")

    def get_tensor(self, name, shape):
        out = np.empty(shape, np.float32)
        rc = lib().oq3_get_tensor(self._h, name.encode(), _fp(out), out.size)
        if rc != 0:
            raise KeyError(f"oracle: unknown tensor or size mismatch: {name}")
        return out

    def forward(self, ids, trace=False):
        """ids [B,T] -> hidden [B,T,H]; with trace also per-layer (h, residual) [L,B,T,H]"""
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        B, T = ids.shape
        H, L = self.cfg.hidden_size, self.cfg.num_hidden_layers
        hidden = np.empty((B, T, H), np.float32)
        th = tr = None
        if trace:
            th = np.empty((L, B, T, H), np.float32)
            tr = np.empty((L, B, T, H), np.float32)
        lib().oq3_set_trace(self._h, _fp(th), _fp(tr))
        rc = lib().oq3_forward(self._h, ids.ctypes.data_as(C.POINTER(C.c_uint32)), B, T, _fp(hidden))
        lib().oq3_set_trace(self._h, None, None)
        if rc != 0:
            raise ValueError("oracle: token id out of range")
        return (hidden, th, tr) if trace else hidden

    def compute_logits(self, hidden):
        hidden = _f32(hidden)
        shp = hidden.shape
        h2 = hidden.reshape(-1, shp[-1])
        out = np.empty((h2.shape[0], self.cfg.vocab_size), np.float32)
        lib().oq3_compute_logits(self._h, _fp(h2), h2.shape[0], _fp(out))
        return out.reshape(*shp[:-1], self.cfg.vocab_size)

    def run_greedy(self, seqs, all_rows=False, want_logits=True):
        """Qwen3ModelRunner::run with the argmax path: list of token lists -> (next_ids, last_logits)"""
        n = len(seqs)
        arrs = [np.ascontiguousarray(s, dtype=np.uint32) for s in seqs]
        u32p = C.POINTER(C.c_uint32)
        ptrs = (u32p * n)(*[a.ctypes.data_as(u32p) for a in arrs])
        lens = np.array([len(a) for a in arrs], np.int32)
        nxt = np.empty(n, np.uint32)
        lg = np.empty((n, self.cfg.vocab_size), np.float32) if want_logits else None
        rc = lib().oq3_run_greedy(self._h, n, ptrs, lens.ctypes.data_as(C.POINTER(C.c_int32)), int(all_rows),
                                  nxt.ctypes.data_as(u32p), _fp(lg))
        if rc != 0:
            raise ValueError("oracle: token id out of range")
        return nxt, lg
