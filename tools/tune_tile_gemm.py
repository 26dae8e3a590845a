"""Prefill tile GEMM (tile_gemm.hip) against the chunked kernel: max |difference| and microseconds per launch.
Usage: python tools/tune_tile_gemm.py [M ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nano_vllm_candle_amd as pkg

Ms = [int(a) for a in sys.argv[1:]] or [4096]
ctx = pkg.Context(0)
L = pkg._lib.lib()
shapes = {"qkv": (4096, 1024, 0), "gate_up": (6144, 1024, 2), "o": (1024, 2048, 0), "down": (1024, 3072, 0),
          "8b_qkv": (6144, 4096, 0), "8b_o": (4096, 4096, 0), "8b_gate_up": (24576, 4096, 2), "8b_down": (4096, 12288, 0)}
for M in Ms:
    for name, (N, K, mode) in shapes.items():
        for packed in ([0, 1] if mode == 2 else [0]):
            d, r, u0, u1 = C.c_float(), C.c_float(), C.c_float(), C.c_float()
            rc = L.nvllm_debug_gemm_tile_check(ctx.h, M, N, K, mode, packed, 10, C.byref(d), C.byref(r), C.byref(u0), C.byref(u1))
            if rc != 0:
                print(name, "ERR", L.nvllm_last_error(ctx.h).decode())
                continue
            fl = 2.0 * M * N * K
            print(f"M={M:6d} {name:11s} N={N:6d} K={K:6d} packed_out={packed}: diff {d.value:.3e} (ref max {r.value:.3e})  chunked {u0.value:8.1f} us "
                  f"({200*fl/u0.value/1e6/2500:5.1f} % pipe)  tile {u1.value:8.1f} us ({200*fl/u1.value/1e6/2500:5.1f} % pipe)", flush=True)
