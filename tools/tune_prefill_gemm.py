"""Prefill GEMM shapes (Qwen3-0.6B, M rows per chunk): microseconds per launch and share of the dense bf16 MFMA peak
(algorithmic = 2MNK; the pipe issues twice that: hi + lo activation planes).
Usage: python tools/tune_prefill_gemm.py [M ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nano_vllm_candle_amd as pkg

Ms = [int(a) for a in sys.argv[1:]] or [4096]
ctx = pkg.Context(0)
L = pkg._lib.lib()
shapes = {"qkv": (4096, 1024, 0), "o": (1024, 2048, 0), "gate_up": (6144, 1024, 2), "down": (1024, 3072, 0),
          "8b_qkv": (6144, 4096, 0), "8b_gate_up": (24576, 4096, 2), "8b_down": (4096, 12288, 0)}
for M in Ms:
    tot_us = tot_fl = 0.0
    for name, (N, K, mode) in shapes.items():
        us = C.c_float()
        rc = L.nvllm_debug_gemm_bench2(ctx.h, M, N, K, 0, 0, 0, 0, mode, 1, 20, C.byref(us))
        if rc != 0:
            print(name, "ERR", L.nvllm_last_error(ctx.h))
            continue
        fl = 2.0 * M * N * K
        print(f"M={M:6d} {name:11s} N={N:6d} K={K:6d}: {us.value:9.1f} us  {fl/us.value/1e6:7.1f} TFLOP/s algorithmic = {100*fl/us.value/1e6/2500:5.1f} % of peak "
              f"(pipe {200*fl/us.value/1e6/2500:5.1f} %)", flush=True)
        if not name.startswith("8b"):
            tot_us += us.value; tot_fl += fl
    print(f"M={M}: 0.6B layer GEMMs {tot_us:.1f} us, {100*tot_fl/tot_us/1e6/2500:.1f} % of peak")
