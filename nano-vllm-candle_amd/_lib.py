"""ctypes binding of libnvllm_amd.so (C ABI: include/nvllm_amd.h).

The product path has NO CPU fallback: if the HIP library is missing or fails to load, importing the
package succeeds but the first call raises NvllmLibraryMissing loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NVLLM_LIB selects another build of the same library in this directory (the stamped diagnostic build)
LIB_PATH = os.path.join(_HERE, os.path.basename(os.environ.get("NVLLM_LIB", "libnvllm_amd.so")))

OK, EINVAL, EHIP, ENOMEM, ESTATE, ERCCL = 0, -1, -2, -3, -4, -5
DTYPE_F32, DTYPE_BF16 = 0, 1
RCCL_ID_BYTES = 128


class NvllmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"nvllm error {code}: {msg}")
        self.code = code


class NvllmLibraryMissing(ImportError):
    pass


class Qwen3ConfigC(C.Structure):
    """nvllm_qwen3_config == Qwen3Config (src/models/qwen3.rs:20-34)"""

    _fields_ = [
        ("vocab_size", C.c_int32), ("hidden_size", C.c_int32), ("head_dim", C.c_int32),
        ("num_hidden_layers", C.c_int32), ("num_attention_heads", C.c_int32),
        ("num_key_value_heads", C.c_int32), ("intermediate_size", C.c_int32),
        ("max_position_embeddings", C.c_int32), ("rms_norm_eps", C.c_double), ("rope_theta", C.c_double),
        ("bos_token_id", C.c_int32), ("eos_token_id", C.c_int32),
    ]


_vp, _fp = C.c_void_p, C.c_void_p  # device pointers travel as integers
_SIGS = {
    "nvllm_last_error": (C.c_char_p, [_vp]),
    "nvllm_rccl_unique_id": (C.c_int, [C.c_void_p]),
    "nvllm_ctx_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(_vp)]),
    "nvllm_ctx_create_loopback": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_char_p, C.POINTER(_vp)]),
    "nvllm_ctx_create_null_comm": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "nvllm_ctx_destroy": (C.c_int, [_vp]),
    "nvllm_ctx_synchronize": (C.c_int, [_vp]),
    "nvllm_ctx_stream": (C.c_void_p, [_vp]),
    "nvllm_ctx_tp_rank": (C.c_int, [_vp]),
    "nvllm_ctx_tp_size": (C.c_int, [_vp]),
    "nvllm_timer_start": (C.c_int, [_vp]),
    "nvllm_timer_stop": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "nvllm_model_create": (C.c_int, [_vp, C.POINTER(Qwen3ConfigC), C.POINTER(_vp)]),
    "nvllm_model_destroy": (C.c_int, [_vp]),
    "nvllm_model_load_tensor": (C.c_int, [_vp, C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    "nvllm_model_fill_synthetic": (C.c_int, [_vp, C.c_uint64]),
    "nvllm_model_fill_synthetic_profile": (C.c_int, [_vp, C.c_uint64, C.c_int]),
    "nvllm_model_finalize": (C.c_int, [_vp]),
    "nvllm_model_weight_bytes": (C.c_int64, [_vp]),
    "nvllm_tp_shard": (C.c_int, [C.POINTER(Qwen3ConfigC), C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_int64)]),
    "nvllm_kv_alloc": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "nvllm_kv_num_free_blocks": (C.c_int, [_vp]),
    "nvllm_kv_bytes_per_token": (C.c_int64, [_vp]),
    "nvllm_seq_free": (C.c_int, [_vp, C.c_int64]),
    "nvllm_step": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.POINTER(C.c_uint32)),
                             C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_float)]),
    "nvllm_step_sample": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_int32), C.c_int,
                                    C.POINTER(C.c_float), C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_float)]),
    "nvllm_decode_next": (C.c_int, [_vp, C.POINTER(C.c_uint32)]),
    "nvllm_decode_enqueue": (C.c_int, [_vp]),
    "nvllm_decode_collect": (C.c_int, [_vp, C.POINTER(C.c_uint32)]),
    "nvllm_last_step_bytes": (C.c_int64, [_vp]),
    "nvllm_debug_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "nvllm_debug_stamps": (C.c_int, [_vp, C.c_int]),
    "nvllm_debug_stamps_read": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_uint64), C.c_int64]),
    "nvllm_profile_kernel": (C.c_int, [_vp, C.c_int]),
    "nvllm_profile_read": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "nvllm_debug_layer_tap": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int64]),
    "nvllm_debug_enable_taps": (C.c_int, [_vp, C.c_int]),
    "nvllm_op_pack_weight": (C.c_int, [_vp, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "nvllm_op_free_weight": (C.c_int, [_vp, _vp]),
    "nvllm_op_linear": (C.c_int, [_vp, _fp, _vp, _fp, C.c_int, _fp]),
    "nvllm_op_rmsnorm": (C.c_int, [_vp, _fp, _fp, _fp, C.c_double, C.c_int, C.c_int, _fp, _fp]),
    "nvllm_op_silu_mul": (C.c_int, [_vp, _fp, C.c_int, C.c_int, _fp]),
    "nvllm_op_rope": (C.c_int, [_vp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]),
    "nvllm_op_attention": (C.c_int, [_vp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _fp]),
    "nvllm_op_embedding": (C.c_int, [_vp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp]),
    "nvllm_op_argmax": (C.c_int, [_vp, _fp, C.c_int, C.c_int, _fp]),
    "nvllm_op_allreduce": (C.c_int, [_vp, _fp, C.c_int64]),
    "nvllm_op_synth_bf16": (C.c_int, [_vp, C.c_char_p, C.c_uint64, C.c_int, C.c_int64, C.c_int64, C.POINTER(C.c_uint16)]),
    "nvllm_debug_synth_bf16_spec": (C.c_int, [_vp, C.c_char_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int64, C.c_int64,
                                              C.POINTER(C.c_uint16)]),
    "nvllm_debug_gemm_bench": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "nvllm_debug_gemm_bench2": (C.c_int, [_vp] + [C.c_int] * 10 + [C.POINTER(C.c_float)]),
    "nvllm_debug_get_counter": (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_int64)]),
    "nvllm_debug_gemm_tile_check": (C.c_int, [_vp] + [C.c_int] * 6 + [C.POINTER(C.c_float)] * 4),
    "nvllm_debug_xcc_map": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32)]),
    "nvllm_debug_attn_bench": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "nvllm_dev_alloc": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "nvllm_dev_free": (C.c_int, [_vp, _vp]),
    "nvllm_dev_upload": (C.c_int, [_vp, _vp, C.c_void_p, C.c_size_t]),
    "nvllm_dev_download": (C.c_int, [_vp, C.c_void_p, _vp, C.c_size_t]),
}
EXPORTED_SYMBOLS = tuple(_SIGS)
_DEBUG_EXPORTS_ADDED_LATER = ("nvllm_debug_stamps", "nvllm_debug_stamps_read", "nvllm_ctx_create_null_comm", "nvllm_debug_set_option",
                              "nvllm_debug_gemm_tile_check", "nvllm_debug_get_counter")

_lib = None


def lib():
    """load the library (once); raise loudly when it is not there"""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NvllmLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C nano-vllm-candle_amd/csrc).  There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            if name in _DEBUG_EXPORTS_ADDED_LATER and not hasattr(L, name):
                continue  # an older build swapped in by tools/ab_bench.sh may lack a newer debug export
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(rc, ctx_handle=None):
    if rc != OK:
        msg = lib().nvllm_last_error(ctx_handle)
        raise NvllmError(rc, msg.decode() if msg else "?")
