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

# big layer matrices (Qwen3-8B / 32B shapes), 64 rows: streaming kernel (22) vs phase-stepped row-parallel kernel (12)
for name, (N, K) in {"8B qkv": (6144, 4096), "8B o": (4096, 4096), "8B gate_up": (24576, 4096), "8B down": (4096, 12288),
                     "32B qkv": (10240, 5120), "32B down": (5120, 25600)}.items():
    mb = N * K * 2 / 1e6
    for mode in (23, 22, 12):
        us = C.c_float()
        rc = L.nvllm_debug_gemm_bench2(ctx.h, 64, N, K, 0, 0, 0, 0, mode, max(2, int(700 / mb)), 20, C.byref(us))
        print(f"{name:12s} N={N} K={K} {mb:6.1f} MB mode={mode}: rc={rc} {us.value:7.2f} us {mb / max(us.value, 1e-3):6.2f} TB/s", flush=True)
