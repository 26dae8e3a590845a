"""Cold-weight GEMM sweep (weights rotated through > Infinity Cache worth of copies) incl. the SwiGLU mode."""
import ctypes as C
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nano_vllm_candle_amd as pkg

M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx = pkg.Context(0)
L = pkg._lib.lib()


def run(N, K, mt, nt, nw, ns, mode, rot):
    us = C.c_float()
    rc = L.nvllm_debug_gemm_bench2(ctx.h, M, N, K, mt, nt, nw, ns, mode, rot, 60, C.byref(us))
    return us.value if rc == 0 else None


shapes = {"qkv": (4096, 1024, 0), "o": (1024, 2048, 0), "down": (1024, 3072, 0), "gate_up slabs": (6144, 1024, 0),
          "gate_up swiglu": (6144, 1024, 2)}
for name, (N, K, mode) in shapes.items():
    rot = max(2, int(320e6 / (N * K * 2)))
    res = []
    if mode == 2:
        grid = itertools.product([1, 2, 4], [2], [1, 2, 4, 8], [1])
    else:
        grid = itertools.product([4], [1, 2], [2, 4, 8], [1, 2, 4, 8, 16])
    for mt, nt, nw, ns in grid:
        if (K // 32) // ns < 4 or nw == 1:
            continue
        us = run(N, K, mt, nt, nw, ns, mode, rot)
        if us:
            res.append((us, mt, nt, nw, ns))
    res.sort()
    print(f"{name} M={M} N={N} K={K} weight={N*K*2/1e6:.1f}MB rot={rot}", flush=True)
    for us, mt, nt, nw, ns in res[:6]:
        print(f"   mt={mt} nt={nt} nw={nw} split={ns}: {us:7.2f} us  {N*K*2/us/1e3:7.1f} GB/s", flush=True)
