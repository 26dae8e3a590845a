"""Sweep GEMM decompositions (n-tiles/wave, waves/WG, K splits) for the decode shapes on the GPU.
Usage: python tools/tune_gemm.py [M]   -> prints microseconds per launch and achieved weight GB/s."""
import ctypes as C
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nano_vllm_candle_amd as pkg

M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx = pkg.Context(0)
L = pkg._lib.lib()
shapes = {"qkv": (4096, 1024), "o": (1024, 2048), "gate_up": (6144, 1024), "down": (1024, 3072), "lm_head": (151936, 1024)}
for name, (N, K) in shapes.items():
    res = []
    splits = [1] if name == "lm_head" else [1, 2, 4, 8, 16]
    for nt, nw, ns in itertools.product([1, 2], [2, 4, 8], splits):
        if (K // 32) // ns < 4:
            continue
        us = C.c_float()
        rc = L.nvllm_debug_gemm_bench(ctx.h, M, N, K, nt, nw, ns, 50, C.byref(us))
        if rc != 0:
            print(name, nt, nw, ns, "ERR", L.nvllm_last_error(ctx.h))
            continue
        res.append((us.value, nt, nw, ns))
    res.sort()
    print(f"{name} M={M} N={N} K={K} weight={N*K*2/1e6:.1f}MB")
    for us, nt, nw, ns in res[:6]:
        print(f"   nt={nt} nw={nw} split={ns}: {us:7.2f} us  {N*K*2/us/1e3:7.1f} GB/s")
    us, nt, nw, ns = res[-1]
    print(f"   worst nt={nt} nw={nw} split={ns}: {us:7.2f} us")
