"""Device context + small device-buffer helper over the C ABI."""
import ctypes as C

import numpy as np

from . import _lib


class Context:
    """nvllm_ctx: one GPU, one stream, optional RCCL communicator (TP group)."""

    def __init__(self, device=0, tp_rank=0, tp_size=1, rccl_id=None, loopback_group=None, null_comm=False):
        L = _lib.lib()
        h = C.c_void_p()
        if null_comm:  # projection aid: the rank's shard shapes, collectives skipped (results meaningless)
            _lib.check(L.nvllm_ctx_create_null_comm(device, tp_rank, tp_size, C.byref(h)), None)
            self.h = h
            self.device = device
            return
        if loopback_group is not None:  # test-only in-process communicator (one host thread per rank)
            _lib.check(L.nvllm_ctx_create_loopback(device, tp_rank, tp_size, loopback_group.encode(), C.byref(h)), None)
            self.h = h
            self.device = device
            return
        idbuf = None
        if rccl_id is not None:
            idbuf = C.create_string_buffer(bytes(rccl_id), _lib.RCCL_ID_BYTES)
        _lib.check(L.nvllm_ctx_create(device, tp_rank, tp_size, idbuf, C.byref(h)), None)
        self.h = h
        self.device = device

    @staticmethod
    def make_rccl_id():
        buf = C.create_string_buffer(_lib.RCCL_ID_BYTES)
        _lib.check(_lib.lib().nvllm_rccl_unique_id(buf), None)
        return bytes(buf.raw)

    @property
    def tp_rank(self):
        return _lib.lib().nvllm_ctx_tp_rank(self.h)

    @property
    def tp_size(self):
        return _lib.lib().nvllm_ctx_tp_size(self.h)

    @property
    def stream(self):
        return _lib.lib().nvllm_ctx_stream(self.h)

    def synchronize(self):
        _lib.check(_lib.lib().nvllm_ctx_synchronize(self.h), self.h)

    def timer_start(self):
        _lib.check(_lib.lib().nvllm_timer_start(self.h), self.h)

    def timer_stop(self):
        ms = C.c_float()
        _lib.check(_lib.lib().nvllm_timer_stop(self.h, C.byref(ms)), self.h)
        return ms.value

    def close(self):
        if getattr(self, "h", None):
            _lib.lib().nvllm_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- device buffers ----
    def to_device(self, arr):
        return DeviceArray.from_host(self, arr)

    def empty(self, shape, dtype=np.float32):
        return DeviceArray(self, shape, dtype)


class DeviceArray:
    """dense row-major device buffer described by (shape, dtype)"""

    def __init__(self, ctx, shape, dtype=np.float32):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        _lib.check(_lib.lib().nvllm_dev_alloc(ctx.h, self.nbytes, C.byref(p)), ctx.h)
        self.ptr = p

    @classmethod
    def from_host(cls, ctx, arr):
        arr = np.ascontiguousarray(arr)
        d = cls(ctx, arr.shape, arr.dtype)
        if d.nbytes:
            _lib.check(_lib.lib().nvllm_dev_upload(ctx.h, d.ptr, arr.ctypes.data_as(C.c_void_p), d.nbytes), ctx.h)
        return d

    def numpy(self):
        out = np.empty(self.shape, self.dtype)
        if self.nbytes:
            _lib.check(_lib.lib().nvllm_dev_download(self.ctx.h, out.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes),
                       self.ctx.h)
        return out

    def free(self):
        if getattr(self, "ptr", None) and self.ctx.h:
            _lib.lib().nvllm_dev_free(self.ctx.h, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
