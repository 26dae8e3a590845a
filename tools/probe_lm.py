"""LM-head kernels alone (cold weights): streaming kernel (mode 20) vs chunked kernel (mode 21)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nano_vllm_candle_amd as pkg

ctx = pkg.Context(0)
L = pkg._lib.lib()
for (M, N, K) in [(64, 151936, 1024), (16, 151936, 1024), (64, 24576, 4096), (64, 151936 // 8, 5120)]:
    if K % 256:
        continue
    mb = N * K * 2 / 1e6
    for mode in (20, 21):
        us = C.c_float()
        rc = L.nvllm_debug_gemm_bench2(ctx.h, M, N, K, 0, 0, 0, 0, mode, max(2, int(700 / mb)), 30, C.byref(us))
        print(f"M={M} N={N} K={K} {mb:6.1f} MB mode={mode}: rc={rc} {us.value:7.2f} us {mb / max(us.value, 1e-3) * 1e3 / 1e3:6.2f} TB/s", flush=True)
