"""Row-parallel decode GEMMs with weights HBM-cold / Infinity-Cache-warm / L2-warm, and the workgroup -> XCD map."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nano_vllm_candle_amd as pkg

ctx = pkg.Context(0)
L = pkg._lib.lib()
for g in [(64, 1, 4, 1024), (96, 1, 4, 1024), (32, 1, 4, 512), (64, 8, 1, 256), (256, 1, 1, 64)]:
    n = g[0] * g[1] * g[2]
    out = np.zeros(n, np.int32)
    rc = L.nvllm_debug_xcc_map(ctx.h, g[0], g[1], g[2], g[3], out.ctypes.data_as(C.POINTER(C.c_int32)))
    ok = bool((out == (np.arange(n) % 8)).all())
    print(f"grid {g}: rc={rc} first 24 = {out[:24].tolist()} id%8 rule holds: {ok}", flush=True)

M = 64
shapes = {"qkv": (4096, 1024, 12), "o": (1024, 2048, 10), "gate_up": (6144, 1024, 11), "down": (1024, 3072, 10)}
for name, (N, K, mode) in shapes.items():
    mb = N * K * 2 / 1e6
    for label, rot in (("L2-warm", 1), ("MALL-warm", max(2, int(120 / mb))), ("HBM-cold", max(2, int(640 / mb)))):
        us = C.c_float()
        rc = L.nvllm_debug_gemm_bench2(ctx.h, M, N, K, 0, 0, 0, 0, mode, rot, 200, C.byref(us))
        print(f"{name:8s} N={N} K={K} {mb:5.1f} MB {label:9s} rot={rot:3d}: rc={rc} {us.value:6.2f} us", flush=True)
