"""What bounds the streaming decode GEMM (csrc/stream_gemm.hip)?  Diagnostic build only:
    make -C nano-vllm-candle_amd/csrc stamps && NVLLM_LIB=libnvllm_amd_stamps.so python tools/ablate_stream.py
Times the kernel alone on cold weights (64 rows, packed planes) as it is and with one part removed: 1 no MFMAs / LDS fragment
reads, 2 no x staging, 3 no weight loads, 4 no slab stores.  Shapes whose launch uses one n-tile per wave (the ablation
variants are instantiated for those): Qwen3-32B QKV and o_proj, Qwen3-8B o_proj and down."""
import ctypes as C
import os
import sys

os.environ.setdefault("NVLLM_LIB", "libnvllm_amd_stamps.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nano_vllm_candle_amd as pkg  # noqa: E402

ctx = pkg.Context(0)
L = pkg._lib.lib()
names = {0: "as shipped", 1: "no MFMA / LDS reads", 2: "no x staging", 3: "no weight loads", 4: "no slab stores"}
for name, (N, K) in {"32B qkv": (10240, 5120), "32B o": (5120, 8192), "8B o": (4096, 4096), "8B down": (4096, 12288)}.items():
    mb = N * K * 2 / 1e6
    for abl in (0, 1, 2, 3, 4, 0):
        us = C.c_float()
        rc = L.nvllm_debug_gemm_bench2(ctx.h, 64, N, K, 0, 0, 0, 0, 23 + 100 * abl, max(2, int(700 / mb)), 20, C.byref(us))
        print(f"{name:8s} {mb:6.1f} MB  {names[abl]:20s}: rc={rc} {us.value:7.2f} us {mb / max(us.value, 1e-3):6.2f} TB/s", flush=True)
